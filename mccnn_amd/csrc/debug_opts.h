// Run-time switches of the library and of the torch extension.
//
// FOUR environment variables select a code path and are part of the interface (README, "Switches"):
//     MCCNN_NATIVE=0        layers go op by op through the Python op surface instead of the native step executor
//     MCCNN_TORCH_EXT=0     the ctypes binding of the C-ABI instead of lib/_mccnn_torch.so
//     MCCNN_ROW_KERNELS=0   depth-wise layers on the edge-streaming kernels instead of the row-per-lane ones
//     MCCNN_GEO_PREFETCH=0  no learned prefetch of the next layers' geometry
// Everything else -- A/B switches of single kernels, tracing, fault injection for the soak tests -- is ONE list:
//     MCCNN_DEBUG="key=value,key,..."      (a bare key means key=1; read once per process)
// THE table of keys is kDebugKeys below (library keys first, then the Python side's: mccnn_amd/_env.py carries the same
// table and tests/test_capi_cpu.py checks that the two are equal and that every key any source file queries is in it). A
// key of MCCNN_DEBUG that is in neither is reported ONCE on stderr -- a misspelt switch must not silently time the default
// on both sides of an A/B. Library keys (default): small_off (0) plan_small_off (0) plan_small (4096: capacity of the
// single-workgroup plan layout, clamped to [1024, MCCNN_PLAN_SMALL]; the plan_small=8192 experiment of NOTES needed a
// rebuild with a larger MCCNN_PLAN_SMALL) plan_small_max_l (16) plan_mid_l (16) plan_min_l (4) rows_force (0)
// rows_min_degree (16) unsorted_max_points (32768) force_valu (0) no_f1 (0) f1_x4_min_e (2000000) f1_x4_waves_per_cu (0)
// nw_lean (-1) nw_group (0) nw_group_fill (0) nw_lds_pad (-1) scan_bg_tiles (8) issue_thread (1) issue_inline (0)
// job_delay_us (0) hier_trace (0) geo_own_pool (1) trace_terminate (0) nw_fused (1: lists of <= 2048 centres scan their
// counts inside the fill pass) geo_batch (1: the geometries prefetch_step starts go out as ONE batch, one launch per kernel kind)
// plan_batch_all (1: the pieces of all of them as one batch as well; 0 = only geometries with a small plan)
// aabb_one_max (8192: compute_aabb in one launch up to this many points) plan_large_batch (1: large plans join the batch)
// plan_batch_sync (0: debugging -- a synchronisation and a stderr line per launch of the plan batch) bwd_min_chunks (2: 64-edge chunks per wave
// of conv_bwd_mfma at least) caller_join_off (0: fault
// injection -- a geometry nobody joined does not order the caller's stream behind its events: the round-6 lifetime bug).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace mccnn {

#define MCCNN_DEBUG_KEYS                                                                                                   \
    "small_off", "plan_small_off", "plan_small", "plan_small_max_l", "plan_mid_l", "plan_min_l", "rows_force",            \
    "rows_min_degree", "unsorted_max_points", "force_valu", "no_f1", "f1_x4_min_e", "f1_x4_waves_per_cu", "nw_lean",      \
    "nw_group", "nw_group_fill", "nw_lds_pad", "scan_bg_tiles", "issue_thread", "issue_inline", "job_delay_us",           \
    "hier_trace", "geo_own_pool", "trace_terminate", "nw_fused", "geo_batch", "plan_batch_all", "aabb_one_max", "plan_large_batch", "plan_batch_sync", "caller_join_off", "bwd_min_chunks", "geo_arena",            \
    /* Python side (mccnn_amd/_env.py) */                                                                                \
    "fuse_sort", "native_prefetch", "plan_prefetch", "plan_prefetch_max_e", "geo_prefetch_min", "mailbox_copy",           \
    "count_mailbox", "ecap_scale", "hier_pmode", "geo_trace", "nw_no_order", "aabb_ext"
static const char* const kDebugKeys[] = {MCCNN_DEBUG_KEYS};

// Parses MCCNN_DEBUG once; items whose key is not in kDebugKeys are reported on stderr (once per process and library).
inline const std::string& debug_list() {
    static const std::string list = [] {
        const char* e = getenv("MCCNN_DEBUG");
        std::string l(e ? e : "");
        size_t p = 0;
        while (p < l.size()) {
            size_t q = l.find(',', p);
            if (q == std::string::npos) q = l.size();
            size_t a = p, b = q;
            while (a < b && l[a] == ' ') ++a;
            while (b > a && l[b - 1] == ' ') --b;
            size_t eq = l.find('=', a);
            if (eq == std::string::npos || eq > b) eq = b;
            size_t kb = eq;
            while (kb > a && l[kb - 1] == ' ') --kb;
            const std::string key = l.substr(a, kb - a);
            if (!key.empty()) {
                bool known = false;
                for (const char* k : kDebugKeys) known = known || key == k;
                if (!known) fprintf(stderr, "mccnn: MCCNN_DEBUG key '%s' is not known (csrc/debug_opts.h kDebugKeys): ignored\n", key.c_str());
            }
            p = q + 1;
        }
        return l;
    }();
    return list;
}

// value of `key` in MCCNN_DEBUG ("" for a bare key), or nullptr
inline const char* debug_opt(const char* key) {
    const std::string& list = debug_list();
    static thread_local std::string val;
    const size_t kl = strlen(key);
    size_t p = 0;
    while (p < list.size()) {
        size_t q = list.find(',', p);
        if (q == std::string::npos) q = list.size();
        const size_t next = q + 1;
        while (p < q && list[p] == ' ') ++p;                 // (blanks around an item are ignored, as on the Python side)
        while (q > p && list[q - 1] == ' ') --q;
        if (q - p >= kl && list.compare(p, kl, key) == 0 && (q - p == kl || list[p + kl] == '=')) {
            val = (q - p == kl) ? std::string("1") : list.substr(p + kl + 1, q - p - kl - 1);
            return val.c_str();
        }
        p = next;
    }
    return nullptr;
}
inline int debug_int(const char* key, int dflt) { const char* v = debug_opt(key); return v ? atoi(v) : dflt; }
inline double debug_float(const char* key, double dflt) { const char* v = debug_opt(key); return v ? atof(v) : dflt; }

}  // namespace mccnn
