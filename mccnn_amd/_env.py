"""Run-time switches of the package.

Four environment variables select a code path and are part of the interface (README, "Switches"):

    MCCNN_NATIVE=0        layers go op by op through the Python op surface instead of the native step executor
    MCCNN_TORCH_EXT=0     the ctypes binding of the C-ABI instead of lib/_mccnn_torch.so
    MCCNN_ROW_KERNELS=0   depth-wise layers on the edge-streaming kernels instead of the row-per-lane ones
    MCCNN_GEO_PREFETCH=0  no learned prefetch of the next layers' geometry

Everything else -- A/B switches of single kernels, tracing, fault injection for the soak tests -- is ONE list, shared with
the library (csrc/debug_opts.h names the library's keys):

    MCCNN_DEBUG="key=value,key,..."      (a bare key means key=1; read once per process)

Python-side keys (default): fuse_sort (1), native_prefetch (1), plan_prefetch (1), plan_prefetch_max_e (1e12),
geo_prefetch_min (5), mailbox_copy (0), count_mailbox (1), ecap_scale (1), hier_pmode (1),
rows_min_degree (16), unsorted_max_points (32768), geo_trace (0).
(MCCNN_LIB_NAME / MCCNN_EXTRA_FLAGS belong to mccnn_amd.build: A/B builds of the library.)"""
import os


def _parse():
    out = {}
    for item in os.environ.get("MCCNN_DEBUG", "").split(","):
        item = item.strip()
        if not item:
            continue
        k, _, v = item.partition("=")
        out[k.strip()] = v.strip() if _ else "1"
    return out


_OPTS = _parse()


def flag(name, default=True):
    """One of the four documented path switches: on unless MCCNN_<NAME>=0 (off unless =1 when default is False)."""
    v = os.environ.get("MCCNN_" + name)
    if v is None:
        return default
    return v != "0"


def debug(key, default, cast=None):
    """Value of `key` in the MCCNN_DEBUG list, converted like `default` (or by `cast`)."""
    v = _OPTS.get(key)
    if v is None:
        return default
    if cast is not None:
        return cast(v)
    if isinstance(default, bool):
        return v not in ("0", "", "false")
    if isinstance(default, int):
        return int(float(v))
    if isinstance(default, float):
        return float(v)
    return v
