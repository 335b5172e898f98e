#!/usr/bin/env python3
"""profiles/ubench/kernel_resources.py [OBJ] [PATTERN] -- VGPRs, spills, scratch and LDS of the kernels in the device code
of build/obj/engine.o (from the code object's metadata notes): the check that a change to a kernel has not started to spill."""
import re, subprocess, sys, tempfile, os
obj = sys.argv[1] if len(sys.argv) > 1 else "build/obj/engine.o"
pat = sys.argv[2] if len(sys.argv) > 2 else "k_update"
llvm = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([llvm + "llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fat.bin", obj, f"{d}/copy.o"])
    subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/dev.co"])
    notes = subprocess.run([llvm + "llvm-readelf", "--notes", f"{d}/dev.co"], capture_output=True, text=True).stdout
blocks = notes.split("  - .agpr_count:")
for b in blocks[1:]:
    name = re.search(r"\.name:\s+(\S+)", b)
    if not name or pat not in name.group(1):
        continue
    m = re.match(r"_ZN5vbnmf\d+([a-z_0-9]+?)ILi(\d+)", name.group(1))
    demangled = f"{m.group(1)}<{m.group(2)}>" if m else name.group(1)[:40]
    def g(k):
        m = re.search(k + r":\s+(\d+)", b)
        return int(m.group(1)) if m else -1
    print(f"{demangled:40s} vgpr {g(r'.vgpr_count'):4d} spill {g(r'.vgpr_spill_count'):3d} sgpr {g(r'.sgpr_count'):4d} scratch {g(r'.private_segment_fixed_size'):5d} lds {g(r'.group_segment_fixed_size'):6d}")
