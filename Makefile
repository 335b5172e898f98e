# Builds the MI355X (gfx950) library and the CPU oracle.  `python -c "import __graft_entry__ as g; g.build()"`
# runs the same commands.  One object per source (build/, git-ignored): a change to the host side does not recompile
# the kernels' 24 rank instantiations.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := ccfindr_amd/csrc
LIB := ccfindr_amd/lib/libvbnmf_hip.so
OBJ := build/obj
CXXFLAGS ?= -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -pthread
HIPFLAGS ?= $(CXXFLAGS) --offload-arch=$(ARCH)
HOST_HDRS := $(CSRC)/common.h include/vbnmf.h
DEV_HDRS := $(HOST_HDRS) $(CSRC)/kernels.h $(CSRC)/mlnmf.h $(CSRC)/special.h $(CSRC)/init.h $(CSRC)/comm.h

all: $(LIB) oracle testlibs

$(OBJ)/host.o: $(CSRC)/host.cpp $(HOST_HDRS)
	mkdir -p $(OBJ)
	$(HIPCC) $(CXXFLAGS) -c -o $@ $<

$(OBJ)/mtx.o: $(CSRC)/mtx.cpp $(HOST_HDRS)
	mkdir -p $(OBJ)
	$(HIPCC) $(CXXFLAGS) -c -o $@ $<

$(OBJ)/order.o: $(CSRC)/order.cpp $(HOST_HDRS)
	mkdir -p $(OBJ)
	$(HIPCC) $(CXXFLAGS) -c -o $@ $<

$(OBJ)/engine.o: $(CSRC)/engine.hip $(DEV_HDRS)
	mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c -o $@ -x hip $<

$(LIB): $(OBJ)/host.o $(OBJ)/mtx.o $(OBJ)/order.o $(OBJ)/engine.o
	mkdir -p $(dir $(LIB))
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -pthread

oracle:
	$(MAKE) -C oracle

# test infrastructure: the stand-in for librccl that lets two processes on ONE GPU run the library's RCCL protocol
testlibs:
	$(MAKE) -C tests/fake_rccl

clean:
	rm -rf ccfindr_amd/lib oracle/_build build tests/fake_rccl/_build
.PHONY: all oracle testlibs clean
