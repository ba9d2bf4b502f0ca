// Device-wide exclusive prefix sum (int32): reduce-then-scan, wave64 shuffles inside a
// workgroup, recursion over the per-tile sums. Replaces the reference's two hand-rolled
// 3-level Hillis-Steele scans (sort_gpu.cu:87-146, find_neighbors.cu:122-176), which cap
// the input at 512^3 / 256^3 elements.
#include "chain.h"

namespace mccnn {

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
                                                           int* __restrict__ total, int rawSums, int* __restrict__ total2) {
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
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) {
        if (total) *total = run;
        if (total2) *total2 = run;
    }
}

// ------------------------------------------------------------------ single pass (decoupled look-back)
// One launch instead of two for up to SCAN_CHAIN_TILES tiles (2 M elements: every use at the benchmark sizes). Tile b
// publishes its aggregate, then wave 0 looks back over the status words of its predecessors -- 64 at a time -- until it
// meets an inclusive prefix, and publishes its own. A status word is ONE aligned 8-byte agent-scope store: bits 63..62 =
// state (0 empty, 1 aggregate, 2 inclusive prefix), low 32 bits = value -- the data is the flag, no fences needed.
// Forward progress: a workgroup takes its tile from an atomic TICKET (the word after the last status word), not from
// blockIdx -- a tile's predecessors are then held by workgroups that started before it and are running, whatever
// order the dispatcher hands workgroups out in and however few of them fit beside other kernels (these scans also run
// on a side stream under VGPR-bound convolution kernels, where residency of the whole grid is not guaranteed). The
// words must be zero on entry (the callers fold that into a memset / kernel they run anyway).
constexpr int SCAN_CHAIN_TILES = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void scan_chained(const int* in, int* out, int n,
                                                             unsigned long long* status, int* total, int* total2) {
    __shared__ int lds[4];
    __shared__ int sOff;
    __shared__ int sTile;
    if (threadIdx.x == 0)
        sTile = (int)__hip_atomic_fetch_add(status + gridDim.x, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int tile = sTile;
    const long long base = (long long)tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int tot;
    const int ex = block_excl_scan(s, tot, lds);
    if (threadIdx.x < 64) {
        const int excl = chain_lookback(status, tile, tot, (int)threadIdx.x);   // chain.h
        if (threadIdx.x == 0) sOff = excl;
    }
    __syncthreads();
    int run = ex + sOff;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (tile == (int)gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) {
        if (total) *total = run;
        if (total2) *total2 = run;
    }
}

// The same for up to MCCNN_BATCH_MAX independent arrays in ONE launch (one geometry each): an item's tiles are chained
// among themselves only (own status words, own ticket behind them).
__global__ __launch_bounds__(SCAN_THREADS) void scan_chained_batch(ScanBatch sb, BatchBlocks bb) {
    __shared__ int lds[4];
    __shared__ int sOff;
    __shared__ int sTile;
    int local, blocks;
    const ScanItem& it = sb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    if (threadIdx.x == 0)
        sTile = (int)__hip_atomic_fetch_add(it.status + blocks, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int tile = sTile;
    const int n = it.n;
    const long long base = (long long)tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? it.in[base + k] : 0;
        s += v[k];
    }
    int tot;
    const int ex = block_excl_scan(s, tot, lds);
    if (threadIdx.x < 64) {
        const int excl = chain_lookback(it.status, tile, tot, (int)threadIdx.x);
        if (threadIdx.x == 0) sOff = excl;
    }
    __syncthreads();
    int run = ex + sOff;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) it.out[base + k] = run;
        run += v[k];
    }
    if (tile == blocks - 1 && threadIdx.x == SCAN_THREADS - 1) {
        if (it.total) *it.total = run;
        if (it.total2) *it.total2 = run;
    }
}
int launch_scan_batch(const ScanBatch& sb, int count, hipStream_t s) {
    BatchBlocks bb;
    bb.count = count;
    int run = 0;
    for (int k = 0; k < count; ++k) { bb.first[k] = run; run += sb.it[k].tiles; }
    for (int k = count; k <= MCCNN_BATCH_MAX; ++k) bb.first[k] = run;
    if (run == 0) return 0;
    scan_chained_batch<<<run, SCAN_THREADS, 0, s>>>(sb, bb);
    MCCNN_LAUNCHED();
    return 0;
}

size_t scan_status_bytes(int n) {
    const long long tiles = ((long long)n + SCAN_TILE - 1) / SCAN_TILE;
    // one status word per tile + the ticket counter
    return (tiles > 1 && tiles <= SCAN_CHAIN_TILES) ? align_up((size_t)(tiles + 1) * sizeof(unsigned long long)) : 0;
}

size_t scan_workspace_bytes(int n) {
    size_t bytes = scan_status_bytes(n);
    long long m = n;
    while (m > SCAN_TILE) {
        m = (m + SCAN_TILE - 1) / SCAN_TILE;
        bytes += align_up((size_t)m * sizeof(int));
    }
    return bytes + 256;
}

int exclusive_scan_i32(const int* in, int* out, int n, int* total, void* ws, hipStream_t s, bool status_zeroed, int* total2) {
    if (n <= 0) {
        int rc = total ? launch_zero_words(total, 1, s) : 0;
        if (!rc && total2) rc = launch_zero_words(total2, 1, s);
        return rc;
    }
    int tiles = ceil_div(n, SCAN_TILE);
    if (tiles == 1) {
        scan_tiles<<<1, SCAN_THREADS, 0, s>>>(in, out, n, nullptr, total, 0, total2);
        MCCNN_LAUNCHED();
        return 0;
    }
    size_t sb = scan_status_bytes(n);
    // Background launches (the geometry of the next batch beside the convolution kernels of this one) take the two-launch
    // form: there the look-back's polls travel through an L2 the convolution kernels keep busy, and the 106-tile scan of
    // the room's cell table took 112 us of the side queue's chain instead of 6 (profiles/r03_pipeline_overlap.txt).
    static const int bgTiles = debug_int("scan_bg_tiles", 8);
    if (g_background && bgTiles > 0 && tiles >= bgTiles) sb = 0;
    if (sb) {  // single pass; the status words sit at the start of the workspace
        if (!status_zeroed) {  // (every caller inside the library has an earlier kernel of its chain clear the words)
            int rc = launch_clear_spans(clear_span(ws, sb), no_span(), no_span(), s);
            if (rc) return rc;
        }
        scan_chained<<<tiles, SCAN_THREADS, 0, s>>>(in, out, n, (unsigned long long*)ws, total, total2);
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
    scan_tiles<<<tiles, SCAN_THREADS, 0, s>>>(in, out, n, sums, total, raw, total2);
    MCCNN_LAUNCHED();
    return 0;
}

}  // namespace mccnn
