"""Writes tests/golden/reference_constants.json by PARSING the reference's sources in the build container
(/root/reference is read-only and does not travel to the GPU box; only the parsed numbers -- data, no source text -- are
committed). Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_constants.py

What is extracted, and from where:
  cell_offsets         the 27 x 3 neighbour-cell table            tf_ops/find_neighbors.cu  (cellOffsetsCPU)
  cell_offsets_pool    the 27 x 3 Poisson colour-phase table      tf_ops/poisson_sampling.cu (cellOffsetsPoolCPU)
  gauss_norm           the Gaussian normalisation literal         tf_ops/compute_pdf.cu     (0.39894228)
  block_mlp_size       default of --MLPSize                       tf_ops/genCompileScript.py
  num_cells_known      numCells(scale_inv) for the radii the reference's models use, evaluated with the FORMULA TEXT parsed
                       out of determineNumCells (tf_ops/sort_gpu.cu: `(int)(1.0f/pCellSize)`, `== 0 ? 1`) in NumPy float32
  model_radii          the radii collected from models/*.py that feed that formula
tests/test_oracle_cpu.py compares the oracle (and through it the HIP kernels' compiled-in tables) against this file."""
import glob
import json
import os
import re

import numpy as np

REF = os.environ.get("MCCNN_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_constants.json")


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def table(src, name):
    m = re.search(r"int\s+%s\s*\[27\]\s*\[3\]\s*=\s*\{(.*?)\};" % re.escape(name), src, re.S)
    assert m, name
    rows = re.findall(r"\{\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*\}", m.group(1))
    assert len(rows) == 27, (name, len(rows))
    return [[int(v) for v in r] for r in rows], src[:m.start()].count("\n") + 1


def main():
    out = {"_generated_by": "tests/golden/make_reference_constants.py (parsed from the reference sources, no source text kept)"}
    src = read("tf_ops/find_neighbors.cu")
    out["cell_offsets"], line = table(src, "cellOffsetsCPU")
    out["cell_offsets_line"] = "tf_ops/find_neighbors.cu:%d" % line
    src = read("tf_ops/poisson_sampling.cu")
    out["cell_offsets_pool"], line = table(src, "cellOffsetsPoolCPU")
    out["cell_offsets_pool_line"] = "tf_ops/poisson_sampling.cu:%d" % line

    src = read("tf_ops/compute_pdf.cu")
    lits = sorted(set(re.findall(r"\(\s*(0\.3989\d+)\s*\)\s*\*\s*exp", src)))
    assert len(lits) == 1, lits
    out["gauss_norm"] = lits[0]                       # kept as the literal's digits: the oracle must use the same decimal
    out["gauss_norm_uses"] = len(re.findall(r"0\.3989\d+", src))

    src = read("tf_ops/genCompileScript.py")
    m = re.search(r"--MLPSize'\s*,\s*default\s*=\s*(\d+)", src)
    assert m
    out["block_mlp_size"] = int(m.group(1))

    # determineNumCells: the scale-invariant branch, as text -> evaluated in float32
    src = read("tf_ops/sort_gpu.cu")
    m = re.search(r"int\s+determineNumCells\s*\(.*?\{(.*?)\n\}", src, re.S)
    assert m
    body = m.group(1)
    f = re.search(r"int\s+numCellsCPU\s*=\s*\(int\)\s*\(\s*1\.0f\s*/\s*pCellSize\s*\)\s*;", body)
    g = re.search(r"numCellsCPU\s*=\s*\(\s*numCellsCPU\s*==\s*0\s*\)\s*\?\s*1\s*:\s*numCellsCPU\s*;", body)
    assert f and g, "determineNumCells no longer reads (int)(1.0f/pCellSize), 0 -> 1"
    out["num_cells_formula"] = "max(1, (int)(1.0f / cellSize)) in float32"

    def num_cells(r):
        v = int(np.float32(1.0) / np.float32(r))      # C's (int) truncates toward zero; the quotient is positive
        return 1 if v == 0 else v

    # radii the reference's own models pass (relative-radius networks): literals of the radius lists in models/*.py
    radii = set()

    def literal(tok):
        tok = tok.strip().replace(" ", "")
        try:
            return float(tok)
        except ValueError:
            mm = re.fullmatch(r"math\.sqrt\(3(?:\.0)?\)(?:\+([0-9.]+))?", tok)
            return (float(np.sqrt(3.0)) + (float(mm.group(1)) if mm.group(1) else 0.0)) if mm else None

    for p in sorted(glob.glob(os.path.join(REF, "models", "*.py"))):
        txt = open(p).read()
        toks = re.findall(r"convRadius\s*=\s*([^,\n)]+(?:\([^)]*\))?[^,\n)]*)", txt)
        for lst in re.findall(r"PointHierarchy\([^\[\n]*\[([^\]]*)\]", txt):
            toks += lst.split(",")
        for tok in toks:
            v = literal(tok)
            if v is not None:
                radii.add(v)
    survey = [0.1, 0.2, 0.4, 0.8, 0.05, 0.025, 0.03, 0.15, float(np.sqrt(3.0)) + 0.1]   # SURVEY 8(a)'s list
    allr = sorted(set(survey) | {r for r in radii if 0 < r < 10})
    out["model_radii"] = sorted(radii)
    out["num_cells_known"] = [[r, num_cells(r)] for r in allr]

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, "radii from models:", sorted(radii))


if __name__ == "__main__":
    main()
