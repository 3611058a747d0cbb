#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a built library (default: the in-tree one):
    python tools/kernel_regs.py [lib.so] [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "saugns_amd", "libsaugns_amd.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
llvm = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as t:
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, t + "/fat.bin"])
    subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + t + "/fat.bin",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + t + "/k.co"])
    txt = subprocess.run([llvm + "llvm-readelf", "--notes", t + "/k.co"], capture_output=True, text=True).stdout
rows = []
for blk in txt.split("- .agpr_count")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt in name:
        rows.append("%-62s vgpr %4s sgpr %4s spill v%s s%s scratch %5s maxwg %s" % (
            name.replace("sauhip::", "").replace("void ", ""), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
            g("sgpr_spill_count"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
print("\n".join(sorted(rows)))
