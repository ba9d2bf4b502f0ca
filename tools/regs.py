#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of one csrc file: python tools/regs.py conv.hip [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mccnn_amd import build as B
src = os.path.join(B.CSRC, sys.argv[1])
cmd = [B._hipcc()] + B.FLAGS + B.FILE_FLAGS.get(sys.argv[1], []) + sys.argv[2:] + ["-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC, "-c", src, "-o", "/tmp/regs.o",
                                 "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("mccnn::", "").replace("void ", "")}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-60s %5s %5s %5s %7s %5s %4s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "spill", "occ", "lds"))
for r in rows:
    print("%-60s %5s %5s %5s %7s %5s %4s %6s" % (r["name"][:60], r.get("vgpr"), r.get("agpr"), r.get("sgpr"), r.get("scratch"), r.get("spill"), r.get("occ"), r.get("lds")))
