// Shared host/device helpers of libmccnn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cfloat>
#include <cstdint>
#include "mccnn.h"
#include "debug_opts.h"

#define MCCNN_MLP 8        // BLOCK_MLP_SIZE (genCompileScript.py:20)
#define MCCNN_WAVE 64      // CDNA wavefront

#define MCCNN_HIP(call)                         \
    do {                                        \
        hipError_t e__ = (call);                \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)
#define MCCNN_MEMSET(call)                      \
    do {                                        \
        ::mccnn::g_launches.fetch_add(1, std::memory_order_relaxed); \
        MCCNN_HIP(call);                        \
    } while (0)
#define MCCNN_LAUNCHED()                        \
    do {                                        \
        ::mccnn::g_launches.fetch_add(1, std::memory_order_relaxed); \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

namespace mccnn {

// kernel launches issued by the library so far (diagnostics: mccnn_debug_launch_count; the launch-bound small-batch
// regime is measured in launches per step). Defined in api_misc.hip.
extern std::atomic<long long> g_launches;
// Test / A-B switch (mccnn_debug_small_kernels, MCCNN_SMALL_OFF in the environment): non-zero = small problems take
// the multi-launch kernels of the large ones instead of their single-workgroup forms. Defined in api_misc.hip.
extern std::atomic<int> g_small_off;
inline bool small_kernels_on() { return g_small_off.load(std::memory_order_relaxed) == 0; }
// Lists of at least this many edges take the four-edges-per-lane forward pass of the Fin = 1 layers (conv_f1.hip;
// mccnn_debug_f1_x4_min_edges, MCCNN_F1_X4_MIN_E in the environment)
extern std::atomic<int> g_f1_x4_min_edges;
// mccnn_background_launches: the calling THREAD's launches run beside more important kernels of another queue (the
// geometry of the next batch under the convolutions of the current one). Kernels that would otherwise fill every wave
// slot then hold back (see neighbors.hip). Defined in api_misc.hip.
extern thread_local int g_background;

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
    char* base;
    size_t cap, off;
    Arena(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = align_up(count * sizeof(T));
        if (off + bytes > cap) return nullptr;
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
};

// ---------------------------------------------------------------------------------------
// Clearing without launches of its own. Counters, status words of the single-pass scans, selection bytes ... have to be
// zero when the kernel that uses them starts. A hipMemsetAsync is a launch (~4 us on both sides of the queue, 13 per
// step of BASELINE cfg1, 44 of cfg4 in round 5); instead a kernel that runs EARLIER in the same chain clears what a later
// one needs, grid-stride, on its way (clear_span_dev at the top of the kernel), and only the head of a call chain -- when
// its first kernel itself needs zeros -- launches clear_spans. Spans are 16-byte granular: every arena piece is 256-byte
// aligned and padded (Arena::take / align_up).
struct ClearSpan {
    void* p;
    unsigned long long n16;  // 16-byte units
};
inline ClearSpan clear_span(void* p, size_t bytes) { return ClearSpan{p, (unsigned long long)((bytes + 15) / 16)}; }
inline ClearSpan no_span() { return ClearSpan{nullptr, 0ull}; }
__device__ __forceinline__ void clear_span_dev(const ClearSpan& s) {
    if (s.n16 == 0) return;
    uint4* q = reinterpret_cast<uint4*>(s.p);
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < s.n16; k += stride)
        q[k] = make_uint4(0u, 0u, 0u, 0u);
}
// the head of a chain: up to three spans in one launch (defined in api_misc.hip)
int launch_clear_spans(ClearSpan a, ClearSpan b, ClearSpan c, hipStream_t s);
// caller-owned buffers (any 4-byte aligned address and size): a zero-filled scatter target, a one-word counter
int launch_zero_words(void* p, size_t words, hipStream_t s);
int launch_fill_words(void* p, size_t words, unsigned v, hipStream_t s);   // (usePDF = False: a tensor of ones)

// ---------------------------------------------------------------------------------------
// One launch per kernel KIND over all geometries of a step (exec.hip: mccnn_geometry_build_batch). The per-geometry
// arguments of a kind travel BY VALUE in the kernel arguments (<= MCCNN_BATCH_MAX items, < 4 KB), a workgroup finds its
// item from the prefix of the items' workgroup counts.
#define MCCNN_BATCH_MAX 16
struct BatchBlocks {
    int first[MCCNN_BATCH_MAX + 1];   // first[k] = first workgroup of item k, first[count] = grid size
    int count;
};
__device__ __forceinline__ int batch_item(const BatchBlocks& b, int blk, int& local, int& blocks) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < MCCNN_BATCH_MAX; ++i) k += (i < b.count && blk >= b.first[i]) ? 1 : 0;
    local = blk - b.first[k];
    blocks = b.first[k + 1] - b.first[k];
    return k;
}
// generic single-pass prefix sum item (scan.hip: scan_chained_batch)
struct ScanItem {
    const int* in;
    int* out;
    unsigned long long* status;   // tiles + 1 words, zero on entry (the head clear of the batch)
    int* total;
    int* total2;
    int n, tiles;
};
struct ScanBatch { ScanItem it[MCCNN_BATCH_MAX]; };
int launch_scan_batch(const ScanBatch& sb, int count, hipStream_t s);   // scan.hip
struct SpanBatch { ClearSpan sp[3 * MCCNN_BATCH_MAX]; int count; };
int launch_clear_batch(const SpanBatch& sb, hipStream_t s);             // api_misc.hip

// Exclusive prefix sum of n int32 (out may alias in). If total != nullptr the grand
// total is written there. ws must hold scan_workspace_bytes(n). Defined in scan.hip.
// Up to 2 M elements the scan is ONE launch (decoupled look-back); its status words are the first
// scan_status_bytes(n) bytes of ws and have to be zero on entry: pass status_zeroed = true when the caller has cleared
// them together with something it clears anyway, otherwise the scan issues the memset itself.
size_t scan_workspace_bytes(int n);
size_t scan_status_bytes(int n);
int exclusive_scan_i32(const int* in, int* out, int n, int* total, void* ws, hipStream_t s, bool status_zeroed = false,
                       int* total2 = nullptr /* a second destination of the total, e.g. a pinned host word */);

// ---------------------------------------------------------------------------------------
// Geometry helpers. The library is compiled with -ffp-contract=off, so each expression
// below rounds exactly like the oracle's (and the reference source's) f32 expression:
// correctly rounded divide / sqrt (-fhip-fp32-correctly-rounded-divide-sqrt), no FMA.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float max_extent(const float* __restrict__ mn, const float* __restrict__ mx, int b) {
    // sort_gpu.cu:50-52
    return fmaxf(fmaxf(mx[b * 3] - mn[b * 3], mx[b * 3 + 1] - mn[b * 3 + 1]), mx[b * 3 + 2] - mn[b * 3 + 2]);
}
__device__ __forceinline__ int cell_coord(float p, float mn, float cs, int nc) {
    // sort_gpu.cu:55
    return max(min((int)floorf((p - mn) / cs), nc - 1), 0);
}
__device__ __forceinline__ float point_dist(float ax, float ay, float az, float cx, float cy, float cz) {
    // find_neighbors.cu:95-96
    float dx = ax - cx, dy = ay - cy, dz = az - cz;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

// dx*dx + dy*dy + dz*dz exactly as the reference spells it (left to right, no FMA: -ffp-contract=off)
__device__ __forceinline__ float point_dist2(float ax, float ay, float az, float cx, float cy, float cz) {
    float dx = ax - cx, dy = ay - cy, dz = az - cz;
    return dx * dx + dy * dy + dz * dz;
}
// The reference tests sqrt(d2) < R (find_neighbors.cu:96-97). sqrt is monotone and correctly rounded, so the test is
// EXACTLY equivalent to d2 < T with T = the smallest float whose rounded square root reaches R. Computing T once per
// centre removes a correctly-rounded sqrt (~10 instructions) from every candidate test.
__device__ __forceinline__ float sqrt_threshold(float R) {
    float t = R * R;
    // walk to the boundary: at most a couple of ulps away from R*R
    for (int it = 0; it < 8 && t > 0.0f && sqrtf(__uint_as_float(__float_as_uint(t) - 1)) >= R; ++it)
        t = __uint_as_float(__float_as_uint(t) - 1);
    for (int it = 0; it < 8 && sqrtf(t) < R; ++it) t = __uint_as_float(__float_as_uint(t) + 1);
    return t;
}

// find_neighbors.cu:282-291 : entry o = (1 - o%3, 1 - (o/3)%3, 1 - o/9)
__device__ __forceinline__ void neigh_offset(int o, int& dx, int& dy, int& dz) {
    dx = 1 - (o % 3);
    dy = 1 - ((o / 3) % 3);
    dz = 1 - (o / 9);
}

// poisson_sampling.cu:192-196 packed as (dx+1) | (dy+1)<<2 | (dz+1)<<4
__device__ __constant__ const unsigned char kPoolOffsets[27] = {
    2 | (2 << 2) | (0 << 4), 1 | (0 << 2) | (2 << 4), 1 | (2 << 2) | (2 << 4), 1 | (2 << 2) | (1 << 4),
    1 | (1 << 2) | (2 << 4), 1 | (0 << 2) | (1 << 4), 0 | (2 << 2) | (0 << 4), 1 | (0 << 2) | (0 << 4),
    2 | (1 << 2) | (1 << 4), 2 | (0 << 2) | (2 << 4), 2 | (1 << 2) | (2 << 4), 0 | (2 << 2) | (2 << 4),
    0 | (1 << 2) | (1 << 4), 2 | (0 << 2) | (0 << 4), 1 | (2 << 2) | (0 << 4), 0 | (0 << 2) | (1 << 4),
    0 | (2 << 2) | (1 << 4), 1 | (1 << 2) | (1 << 4), 1 | (1 << 2) | (0 << 4), 2 | (2 << 2) | (1 << 4),
    2 | (1 << 2) | (0 << 4), 2 | (0 << 2) | (1 << 4), 0 | (1 << 2) | (2 << 4), 2 | (2 << 2) | (2 << 4),
    0 | (1 << 2) | (0 << 4), 0 | (0 << 2) | (0 << 4), 0 | (0 << 2) | (2 << 4)};
__device__ __forceinline__ void pool_offset(int o, int& dx, int& dy, int& dz) {
    unsigned v = kPoolOffsets[o];
    dx = (int)(v & 3) - 1;
    dy = (int)((v >> 2) & 3) - 1;
    dz = (int)((v >> 4) & 3) - 1;
}

// Workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only): hand every XCD one CONTIGUOUS run of
// tiles. Tiles follow the cell-coherent visiting order, so an XCD then works on one region of space and its private L2
// holds that region's points and cell ranges once -- with the default interleaving every one of the 8 L2s pulled the
// whole point set through the fabric (22 + 28 MB fetched for 3 MB of inputs on the 100k room). Bijective for any n.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// every kernel clamps a batch id before it indexes a per-cloud table (mccnn_check_batch_ids reports invalid ids)
__device__ __forceinline__ int clamp_batch(int b, int B) { return max(0, min(b, B - 1)); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// wave64 inclusive scan (int)
__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a 1024-thread workgroup; wsum: 17 ints of LDS
__device__ __forceinline__ int block1024_excl_scan(int v, int& total, int* wsum) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 16; ++k) { const int x = wsum[k]; wsum[k] = run; run += x; }
        wsum[16] = run;
    }
    __syncthreads();
    total = wsum[16];
    const int ex = wsum[wave] + incl - v;
    __syncthreads();  // wsum may be reused right away
    return ex;
}


}  // namespace mccnn
