"""Instruction histogram of the hottest basic block (most MFMAs) of a kernel in a hipcc -S listing.
usage: isa_hist.py conv.s <substring of mangled kernel name>"""
import collections
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append((name, cur)); cur = []; name = t
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
blocks.sort(key=lambda b: -sum(1 for i in b[1] if i.startswith("v_mfma")))
for name, ins in blocks[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    h = collections.Counter(ins)
    groups = collections.Counter()
    for k, v in h.items():
        g = ("mfma" if k.startswith("v_mfma") else "dpp/mov" if k in ("v_mov_b32_dpp", "v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32")
             else "pk" if k.startswith("v_pk") else "valu" if k.startswith("v_") else "lds" if k.startswith("ds_")
             else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "salu" if k.startswith("s_") else "other")
        groups[g] += v
    print(name, "total", len(ins), dict(groups))
    print("  ", ", ".join(f"{k}:{v}" for k, v in h.most_common(40)))
