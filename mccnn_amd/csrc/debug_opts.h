// Run-time switches of the library and of the torch extension.
//
// FOUR environment variables select a code path and are part of the interface (README, "Switches"):
//     MCCNN_NATIVE=0        layers go op by op through the Python op surface instead of the native step executor
//     MCCNN_TORCH_EXT=0     the ctypes binding of the C-ABI instead of lib/_mccnn_torch.so
//     MCCNN_ROW_KERNELS=0   depth-wise layers on the edge-streaming kernels instead of the row-per-lane ones
//     MCCNN_GEO_PREFETCH=0  no learned prefetch of the next layers' geometry
// Everything else -- A/B switches of single kernels, tracing, fault injection for the soak tests -- is ONE list:
//     MCCNN_DEBUG="key=value,key,..."      (a bare key means key=1; read once per process)
// Keys (default): small_off (0) plan_small_off (0) plan_min_l (4) rows_force (0) rows_min_degree (16)
// unsorted_max_points (32768) force_valu (0) no_f1 (0) f1_x4_min_e (2000000) f1_x4_waves_per_cu (0) nw_lean (-1)
// nw_group (0) nw_group_fill (0) nw_lds_pad (-1) scan_bg_tiles (8) issue_thread (1) issue_inline (0) job_delay_us (0)
// hier_trace (0); the Python side (mccnn_amd/_env.py) reads the same list for its own keys.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

namespace mccnn {

// value of `key` in MCCNN_DEBUG ("" for a bare key), or nullptr
inline const char* debug_opt(const char* key) {
    static const std::string list = [] { const char* e = getenv("MCCNN_DEBUG"); return std::string(e ? e : ""); }();
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
