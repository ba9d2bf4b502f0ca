// Device-wide exclusive prefix sum (int32): reduce-then-scan, wave64 shuffles inside a
// workgroup, recursion over the per-tile sums. Replaces the reference's two hand-rolled
// 3-level Hillis-Steele scans (sort_gpu.cu:87-146, find_neighbors.cu:122-176), which cap
// the input at 512^3 / 256^3 elements.
#include "common.h"

namespace mccnn {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

// Block-wide exclusive scan of one value per thread; returns exclusive prefix, total via ref.
__device__ __forceinline__ int block_excl_scan(int v, int& total, int* lds /*>=4 ints*/) {
    int incl = wave_incl_scan(v);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        int s = lds[w];
        if (w < wave) woff += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return woff + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums(const int* __restrict__ in, int n,
                                                               int* __restrict__ sums) {
    __shared__ int lds[4];
    long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    int tot;
    block_excl_scan(s, tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// Scans one tile per block. offsets == nullptr -> single tile (n <= SCAN_TILE). rawSums: `offsets` holds the
// UNSCANNED tile sums and every block adds up the sums of the tiles before it itself (<= SCAN_TILE tiles): saves the
// middle launch of the three-level scheme, which at 100k elements costs as much as the scan proper.
// `out` may alias `in` (callers scan in place): neither is __restrict__; every thread loads its items before the
// first barrier of block_excl_scan and stores after the last one.
__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles(const int* in, int* out,
                                                           int n, const int* offsets,
                                                           int* __restrict__ total, int rawSums) {
    __shared__ int lds[4];
    int off = 0;
    if (offsets) {
        if (rawSums) {
            int part = 0;
            for (int t = threadIdx.x; t < (int)blockIdx.x; t += SCAN_THREADS) part += offsets[t];
            block_excl_scan(part, off, lds);
        } else {
            off = offsets[blockIdx.x];
        }
    }
    long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int tot;
    int ex = block_excl_scan(s, tot, lds);
    int run = ex + off;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) *total = run;
}

size_t scan_workspace_bytes(int n) {
    size_t bytes = 0;
    long long m = n;
    while (m > SCAN_TILE) {
        m = (m + SCAN_TILE - 1) / SCAN_TILE;
        bytes += align_up((size_t)m * sizeof(int));
    }
    return bytes + 256;
}

int exclusive_scan_i32(const int* in, int* out, int n, int* total, void* ws, hipStream_t s) {
    if (n <= 0) {
        if (total) MCCNN_HIP(hipMemsetAsync(total, 0, sizeof(int), s));
        return 0;
    }
    int tiles = ceil_div(n, SCAN_TILE);
    if (tiles == 1) {
        scan_tiles<<<1, SCAN_THREADS, 0, s>>>(in, out, n, nullptr, total, 0);
        MCCNN_LAUNCHED();
        return 0;
    }
    int* sums = (int*)ws;
    void* rest = (char*)ws + align_up((size_t)tiles * sizeof(int));
    scan_tile_sums<<<tiles, SCAN_THREADS, 0, s>>>(in, n, sums);
    MCCNN_LAUNCHED();
    const int raw = tiles <= SCAN_TILE ? 1 : 0;
    if (!raw) {
        int rc = exclusive_scan_i32(sums, sums, tiles, nullptr, rest, s);
        if (rc) return rc;
    }
    scan_tiles<<<tiles, SCAN_THREADS, 0, s>>>(in, out, n, sums, total, raw);
    MCCNN_LAUNCHED();
    return 0;
}

}  // namespace mccnn
