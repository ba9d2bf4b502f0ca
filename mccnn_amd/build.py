"""Builds libmccnn_hip.so (hand-written HIP kernels + C-ABI, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU
box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, os.environ.get("MCCNN_LIB_NAME", "libmccnn_hip.so"))  # override only for A/B experiments
SOURCES = ["api_misc.hip", "scan.hip", "grid.hip", "neighbors.hip", "poisson.hip", "conv.hip", "conv_f1.hip", "conv_rows.hip", "exec.hip"]
HEADERS = ["common.h", "chain.h", "batch.h", "conv_mfma.h", "debug_opts.h"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # bit-exact geometry: no FMA contraction, correctly rounded f32 divide / sqrt (see csrc/common.h)
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
    # MFMA results straight into VGPRs (no v_accvgpr_read per value): gfx950 has a unified register file
    "-mllvm", "-amdgpu-mfma-vgpr-form=1",
]


# Per-file flags. The edge-streaming convolution kernels (conv.hip, conv_f1.hip) are compiled WITHOUT SLP vectorisation: the
# vectoriser pairs their accumulating FMAs into v_pk_fma_f32 (the same ~26 FMA per cycle and SIMD as v_fma_f32 on gfx950)
# at the price of register-pairing moves and, in the 176-sum sweep of the small-Fin combin layers, spills -- the 3 -> 8
# backward sweep runs 224 instead of 252 us without it, the Fin = 1 backward 334 instead of 338, the streaming forward 117
# instead of 121 (round 5, tools/ktab.sh through the ctypes binding). The row kernels (conv_rows.hip) keep it: dw_bwd_rows is
# 0.5 % faster with the packed form. Same arithmetic either way (one IEEE fma per element): results do not change.
# (A/B: MCCNN_EXTRA_FLAGS=-fslp-vectorize comes later on the command line and switches it back on.)
FILE_FLAGS = {
    "conv.hip": ["-fno-slp-vectorize"],
    "conv_f1.hip": ["-fno-slp-vectorize"],
}

TORCH_EXT = os.path.join(LIB_DIR, "_mccnn_torch.so")   # the PyTorch-ROCm extension over the C-ABI (csrc/torch_ext.cpp)
TORCH_EXT_SRC = os.path.join(CSRC, "torch_ext.cpp")


def torch_ext_needs_build():
    if not os.path.exists(TORCH_EXT):
        return True
    t = os.path.getmtime(TORCH_EXT)
    return any(os.path.getmtime(d) > t for d in (TORCH_EXT_SRC, os.path.join(ROOT, "include", "mccnn.h"), os.path.join(CSRC, "debug_opts.h")))


def build_torch_ext(force=False, verbose=False):
    """csrc/torch_ext.cpp -> lib/_mccnn_torch.so: host-only C++ (autograd node, buffers, streams) against the torch
    headers of THIS interpreter, linked to libmccnn_hip.so next to it. g++ -- there is no device code in it."""
    if not force and not torch_ext_needs_build():
        return TORCH_EXT
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include", os.path.join(ROOT, "include"), CSRC]
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-deprecated-declarations", TORCH_EXT_SRC, "-o", TORCH_EXT + ".tmp",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_mccnn_torch",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-DTORCH_API_INCLUDE_EXTENSION_H"]
    cmd += ["-I" + i for i in inc]
    cmd += ["-L" + tlib, "-L" + LIB_DIR, "-l:" + os.path.basename(LIB), "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
            "-ltorch_python", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(TORCH_EXT + ".tmp", TORCH_EXT)
    return TORCH_EXT


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES]


def needs_build():
    """The HIP library is older than one of its sources (the torch extension is checked on its own: ensure_torch_ext)."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "mccnn.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def ensure_torch_ext(force=False, verbose=False):
    """Builds lib/_mccnn_torch.so when it is missing or stale. A failure (no g++, other torch headers, ABI) is reported
    and leaves the ctypes binding of the HIP library in charge (mccnn_amd.native falls back to it) -- it never fails the
    build of the library itself."""
    if os.environ.get("MCCNN_LIB_NAME") is not None:  # (A/B builds of the kernels keep the extension of the default library)
        return None
    if not force and not torch_ext_needs_build():
        return TORCH_EXT
    try:
        return build_torch_ext(force=True, verbose=verbose)
    except Exception as err:  # noqa: BLE001 -- anything the host toolchain can throw
        print("mccnn_amd.build: torch extension not built (%s); mccnn_amd.native will use the ctypes binding" % err, file=sys.stderr)
        return None


def build(force=False, verbose=False):
    if not force and not needs_build():
        ensure_torch_ext(verbose=verbose)
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(LIB) + "." + os.path.basename(src) + ".o")
        per_file = FILE_FLAGS.get(os.path.basename(src), [])
        cmd = [hipcc] + FLAGS + per_file + os.environ.get("MCCNN_EXTRA_FLAGS", "").split() + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # hipcc leaves its unbundled link inputs (<lib>.<n>.hipv4-..., <lib>.<n>.host-...) next to the output
    for f in os.listdir(LIB_DIR):
        if f.startswith(os.path.basename(LIB) + ".") and ("hipv4-" in f or ".host-" in f):
            os.remove(os.path.join(LIB_DIR, f))
    ensure_torch_ext(force=True, verbose=verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
