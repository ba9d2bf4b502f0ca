#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: python tools/spills.py conv_rows.hip 'dw_bwd_rowsILi2ELb0E' [extra hipcc flags]
Prints every scratch_* instruction of the (mangled-name substring) kernel with the loop labels around it and the
number of MFMAs seen so far (inner loops are where a spill costs)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mccnn_amd import build as B
src = os.path.join(B.CSRC, sys.argv[1])
cmd = [B._hipcc()] + B.FLAGS + B.FILE_FLAGS.get(sys.argv[1], []) + sys.argv[3:] + ["-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC, "--cuda-device-only", "-S", src, "-o", "/tmp/spills.s"]
subprocess.run(cmd, capture_output=True, text=True)
lines = open("/tmp/spills.s").read().split("\n")
start = [i for i, l in enumerate(lines) if sys.argv[2] in l and l.rstrip().endswith(sys.argv[2].join(["", ""])) is not None and re.match(r"^_Z\w+:", l) and sys.argv[2] in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
mf = 0
for i, l in enumerate(lines[start:end]):
    if "v_mfma" in l:
        mf += 1
    if "scratch_" in l or re.match(r"^\.LBB", l) and "Loop" in l:
        print(i, l.strip()[:110], "[mfma %d]" % mf)
