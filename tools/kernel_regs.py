#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel of the built library, from the code objects' metadata
(llvm-readelf --notes of slam_amd/lib/obj/*.o as the build left them; no GPU needed):

    python tools/kernel_regs.py [regex]        e.g.  python tools/kernel_regs.py 'pair|fused|raycast_tiled|spread'
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
keys = ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")
print("%-14s %-70s %5s %6s %5s %6s %8s %8s" % ("object", "kernel", "vgpr", "vspill", "sgpr", "sspill", "scratchB", "ldsB"))
with tempfile.TemporaryDirectory() as tmp:
    for o in sorted(glob.glob(os.path.join(ROOT, "slam_amd", "lib", "obj", "*.o"))):
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        if subprocess.call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fat], stderr=subprocess.DEVNULL):
            continue
        if subprocess.call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL):
            continue
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
        for blk in notes.split("- .agpr_count:")[1:]:
            m = re.search(r"^\s+\.name:\s+(\S+)", blk, re.M)
            if not m:
                continue
            name = subprocess.check_output(["c++filt", m.group(1)], text=True).strip()
            name = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
            name = re.sub(r"\(.*", "", name)
            if not pat.search(name):
                continue
            v = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in keys}
            print("%-14s %-70s %5d %6d %5d %6d %8d %8d" % (os.path.basename(o)[:-2], name[:70], v["vgpr_count"], v["vgpr_spill_count"],
                                                          v["sgpr_count"], v["sgpr_spill_count"], v["private_segment_fixed_size"],
                                                          v["group_segment_fixed_size"]))
