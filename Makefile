# Builds the MI355X (gfx950) library and the CPU oracle.  `python -c "import __graft_entry__ as g; g.build()"`
# runs the same commands.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := ccfindr_amd/csrc
LIB := ccfindr_amd/lib/libvbnmf_hip.so
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -Wall -Wno-unused-function
SRCS := $(CSRC)/host.cpp $(CSRC)/mtx.cpp $(CSRC)/engine.hip
HDRS := $(CSRC)/common.h $(CSRC)/kernels.h $(CSRC)/mlnmf.h $(CSRC)/special.h $(CSRC)/init.h $(CSRC)/comm.h include/vbnmf.h

all: $(LIB) oracle

$(LIB): $(SRCS) $(HDRS)
	mkdir -p $(dir $(LIB))
	$(HIPCC) $(HIPFLAGS) -shared -o $@ -x hip $(SRCS) -pthread

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf ccfindr_amd/lib oracle/_build
.PHONY: all oracle clean
