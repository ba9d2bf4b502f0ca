// Version / diagnostics entry points of libmccnn_hip.
#include "common.h"
#include <cstdlib>

namespace mccnn {
std::atomic<long long> g_launches{0};
thread_local int g_background = 0;
std::atomic<int> g_small_off{debug_int("small_off", 0) ? 1 : 0};
std::atomic<int> g_f1_x4_min_edges{debug_int("f1_x4_min_e", 2000000)};

// The head of a call chain whose first kernel itself needs zeros (histogram counters of a grid build, ...): ONE launch
// for everything the chain wants cleared up front; every later need is met by a kernel of the chain (clear_span_dev).
__global__ __launch_bounds__(256) void clear_spans(ClearSpan a, ClearSpan b, ClearSpan c) {
    clear_span_dev(a);
    clear_span_dev(b);
    clear_span_dev(c);
}
// caller-owned buffers of any 4-byte-aligned size (a zero-filled scatter target, a one-word counter): 4-byte granular
__global__ __launch_bounds__(256) void zero_words(unsigned* __restrict__ p, unsigned long long words) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < words; k += stride) p[k] = 0u;
}
__global__ __launch_bounds__(256) void fill_words(unsigned* __restrict__ p, unsigned long long words, unsigned v) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < words; k += stride) p[k] = v;
}
int launch_fill_words(void* p, size_t words, unsigned v, hipStream_t s) {
    if (words == 0) return 0;
    long long blocks = (long long)((words + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    fill_words<<<(int)blocks, 256, 0, s>>>(reinterpret_cast<unsigned*>(p), (unsigned long long)words, v);
    MCCNN_LAUNCHED();
    return 0;
}
int launch_zero_words(void* p, size_t words, hipStream_t s) {
    if (words == 0) return 0;
    if ((((uintptr_t)p) & 15) == 0 && (words & 3) == 0) return launch_clear_spans(clear_span(p, words * 4), no_span(), no_span(), s);
    long long blocks = (long long)((words + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    zero_words<<<(int)blocks, 256, 0, s>>>(reinterpret_cast<unsigned*>(p), (unsigned long long)words);
    MCCNN_LAUNCHED();
    return 0;
}
// ... of a whole batch of geometries (mccnn_geometry_build_batch): every span in one launch
__global__ __launch_bounds__(256) void clear_spans_batch(SpanBatch sb) {
    for (int k = 0; k < sb.count; ++k) clear_span_dev(sb.sp[k]);
}
int launch_clear_batch(const SpanBatch& sb, hipStream_t s) {
    unsigned long long most = 0;
    for (int k = 0; k < sb.count; ++k) most = sb.sp[k].n16 > most ? sb.sp[k].n16 : most;
    if (most == 0) return 0;
    long long blocks = (long long)((most + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    clear_spans_batch<<<(int)blocks, 256, 0, s>>>(sb);
    MCCNN_LAUNCHED();
    return 0;
}
int launch_clear_spans(ClearSpan a, ClearSpan b, ClearSpan c, hipStream_t s) {
    const unsigned long long most = a.n16 > b.n16 ? (a.n16 > c.n16 ? a.n16 : c.n16) : (b.n16 > c.n16 ? b.n16 : c.n16);
    if (most == 0) return 0;
    long long blocks = (long long)((most + 255) / 256);
    if (blocks > 4096) blocks = 4096;   // 16 B per thread and trip: 16 MB per sweep of the grid
    clear_spans<<<(int)blocks, 256, 0, s>>>(a, b, c);
    MCCNN_LAUNCHED();
    return 0;
}
}

extern "C" {

int mccnn_block_size(void) { return MCCNN_MLP; }
int mccnn_abi_version(void) { return 10; }  // 2: optional forward state of spatial_conv; 3: bf16 rows, device-side counts; 4: compute_pdf_dn; 5: row plans, aabb_extent; 6: rowplan_build, build_grid; 7: feat_index of the row kernels, hierarchy_level, find_neighbors_count2; 8: background_launches, debug_f1_x4_min_edges; 9: native step executor (mccnn_geometry_*, mccnn_conv_*); 10: mccnn_geometry_build_batch, mccnn_geometry_prebuild_batch
const char* mccnn_arch(void) { return "gfx950"; }
int mccnn_background_launches(int on) { const int prev = mccnn::g_background; mccnn::g_background = on ? 1 : 0; return prev; }
int mccnn_debug_f1_x4_min_edges(int edges) { return mccnn::g_f1_x4_min_edges.exchange(edges < 0 ? 0 : edges); }
int mccnn_debug_small_kernels(int on) { return mccnn::g_small_off.exchange(on ? 0 : 1) == 0 ? 1 : 0; }
long long mccnn_debug_launch_count(void) { return mccnn::g_launches.load(std::memory_order_relaxed); }

const char* mccnn_error_string(int code) {
    switch (code) {
        case MCCNN_OK: return "ok";
        case MCCNN_E_BADARG: return "invalid argument (null pointer or non-positive size/attribute)";
        case MCCNN_E_BATCHID: return "batch id outside [0, batch_size)";
        case MCCNN_E_TOOLARGE: return "problem does not fit 32-bit indexing (B*nc^3, E or LDS tile)";
        case MCCNN_E_WORKSPACE: return "workspace missing or smaller than the *_workspace_bytes query";
        case MCCNN_E_SHAPE: return "kernel-MLP shape rule violated (spatial_conv.cc:258-300)";
        case MCCNN_E_CAPACITY: return "neighbour list longer than the capacity the geometry was built with";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown mccnn error";
}

}  // extern "C"
