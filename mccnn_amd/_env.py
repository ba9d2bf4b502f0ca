"""Run-time switches of the package.

Four environment variables select a code path and are part of the interface (README, "Switches"):

    MCCNN_NATIVE=0        layers go op by op through the Python op surface instead of the native step executor
    MCCNN_TORCH_EXT=0     the ctypes binding of the C-ABI instead of lib/_mccnn_torch.so
    MCCNN_ROW_KERNELS=0   depth-wise layers on the edge-streaming kernels instead of the row-per-lane ones
    MCCNN_GEO_PREFETCH=0  no learned prefetch of the next layers' geometry

Everything else -- A/B switches of single kernels, tracing, fault injection for the soak tests -- is ONE list, shared with
the library (csrc/debug_opts.h names the library's keys):

    MCCNN_DEBUG="key=value,key,..."      (a bare key means key=1; read once per process)

THE table of keys is KNOWN_KEYS below -- the same table as kDebugKeys of csrc/debug_opts.h (tests/test_capi_cpu.py checks
that they are equal and that every key any source file queries is in it); a key of MCCNN_DEBUG that is not in it is
reported once on stderr instead of being silently ignored. Python-side keys (default): fuse_sort (1), native_prefetch (1),
plan_prefetch (1), plan_prefetch_max_e (1e12), geo_prefetch_min (5), mailbox_copy (0), count_mailbox (1), ecap_scale (1),
hier_pmode (1), rows_min_degree (16), unsorted_max_points (32768), geo_trace (0), nw_no_order (0).
(MCCNN_LIB_NAME / MCCNN_EXTRA_FLAGS belong to mccnn_amd.build: A/B builds of the library.)"""
import os
import sys

KNOWN_KEYS = (
    # library (csrc/debug_opts.h)
    "small_off", "plan_small_off", "plan_small", "plan_small_max_l", "plan_mid_l", "plan_min_l", "rows_force",
    "rows_min_degree", "unsorted_max_points", "force_valu", "no_f1", "f1_x4_min_e", "f1_x4_waves_per_cu", "nw_lean",
    "nw_group", "nw_group_fill", "nw_lds_pad", "scan_bg_tiles", "issue_thread", "issue_inline", "job_delay_us",
    "hier_trace", "geo_own_pool", "trace_terminate", "nw_fused", "geo_batch", "plan_batch_all", "aabb_one_max", "plan_large_batch", "plan_batch_sync", "caller_join_off", "bwd_min_chunks", "geo_arena",
    # Python side
    "fuse_sort", "native_prefetch", "plan_prefetch", "plan_prefetch_max_e", "geo_prefetch_min", "mailbox_copy",
    "count_mailbox", "ecap_scale", "hier_pmode", "geo_trace", "nw_no_order", "aabb_ext")


def _parse():
    out = {}
    for item in os.environ.get("MCCNN_DEBUG", "").split(","):
        item = item.strip()
        if not item:
            continue
        k, _, v = item.partition("=")
        out[k.strip()] = v.strip() if _ else "1"
        if k.strip() not in KNOWN_KEYS:
            sys.stderr.write("mccnn: MCCNN_DEBUG key %r is not known (mccnn_amd/_env.py KNOWN_KEYS): ignored\n" % k.strip())
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
