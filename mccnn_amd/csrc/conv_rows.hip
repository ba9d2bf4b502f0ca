// "Row per lane" convolution kernels over SELL-64-sigma layouts of the neighbour list.
//
// The streaming kernels of conv.hip map lanes to 64 CONSECUTIVE EDGES of the CSR list; the sum over a centre's edges is
// then a segmented wave scan (48 fused-DPP multiply-adds per MLP block on the pipe the MFMAs share), and the feature
// gradient of a depth-wise layer needs a second full evaluation of the kernel MLP over the transposed list. Here a lane
// owns a ROW of the list -- a centre i for the forward pass, a neighbour point j for the backward pass -- and walks the
// row's edges one per iteration: the sum over the row is a plain per-lane accumulator (no scan, no carry, no atomics,
// every output row written exactly once by one lane), the 4x4x1 MFMA form still evaluates the kernel MLP for 64 edges at
// once (it never needed the 64 edges to be neighbours in the list), and in the backward pass the SAME sweep that feeds the
// 176 weight-gradient sums also finishes the feature-gradient row of its point: the MLP is evaluated once, not twice.
//
// Layout (built once per neighbour list by mccnn_rowplan_layout / _fill, shared by every layer over the list): rows are
// grouped into windows of SELL_SIGMA rows of the cell-coherent visiting order, sorted by descending edge count inside a
// window, and cut into slices of 64; slice s stores len_s = (largest count in the slice) x 64 slots, slot (it, lane) =
// edge #it of the lane's row: one 16-byte record (delta, 1 / (pdf K)) and the index of the row at the other end of the
// edge, padded with zero records (1/(pdf K) = 0 zeroes every term a padding lane feeds). Loads are perfectly coalesced;
// padding is ~4 % on the 100k-point room at sigma = 1024 (31 % unsorted).
#include "conv_mfma.h"
#include "chain.h"
#include "batch.h"
#include <cstdlib>
#include <type_traits>

namespace mccnn {

constexpr int SELL_SIGMA = 1024;
// Rows longer than ROWS_L edges (pooling / up-sampling layers: hundreds of neighbours per row) are cut into VIRTUAL rows of
// at most ROWS_L edges, each with a lane of its own: (a) a list with few, long rows still fills the chip, and (b) the
// number of slots has a bound the host knows -- windows are sorted by length, so a window's slices hold at most
// 64 ROWS_L + (its edges) slots -- hence plans are laid out and filled WITHOUT any size read-back. The pieces of a cut
// row leave their partial sums in a scratch row each; a small pass adds them in order (deterministic).
constexpr int ROWS_L = 128;  // longest virtual row; lists with few rows use shorter ones (plan_sizes) to fill the chip

// (Workgroup ids are handed to the 8 XCDs round-robin, id % 8: xcd_contiguous() of common.h gives the logical index with
// every XCD owning one contiguous run of [0, grid) -- a bijection for any grid size; the launchers here still pad the
// grid to a multiple of 8 and surplus indices exit.)

struct RowPlan {
    const int* vrow;      // [64 S] row id of (slice, lane), -1 = padding lane
    const int* vcode;     // [64 S] virtual row id v of (slice, lane); ~v when the row is NOT cut (its sums go straight out)
    const int* sliceOff;  // [S + 1] first slot of every slice; sliceOff[S] = total slots
    const int* vposRow;   // [rows] first virtual row id of every row
    const float4* rec;    // [slots] (delta0, delta1, delta2, 1 / (pdf K))
    const int* other;     // [slots] the row at the other end of the edge (forward plan: neighbour j; transposed: centre i)
    int numRows, S;
};

#define MCCNN_PLAN_SMALL 4096  // capacity of the single-workgroup layout (rows and virtual rows): 49 KB of LDS. (8 192, measured in
                               // round 5: the same step times with 12 launches fewer, but 99 KB of LDS make every plan_small launch 8 -> 12 us)
struct PlanSizes {
    bool small;       // single-workgroup layout (plan_small)
    int L;            // longest virtual row of this plan
    long long vcap;   // bound on the number of virtual rows
    int windows, S;   // windows of SELL_SIGMA virtual rows, slices of 64
    long long slots;  // bound on the number of slots
};
static int plan_small_limit() {
    static const int v = max(1024, min(MCCNN_PLAN_SMALL, debug_int("plan_small", 4096)));
    return v;
}
static PlanSizes plan_sizes(int rows, int e) {
    PlanSizes z;
    // A lane walks its virtual row serially (~1 us per edge of latency), so a list with few rows is cut finer: about
    // 64 k virtual rows (1 k slices) when the edges allow it, pieces of 16 .. 128 edges. Large lists keep 128: a row
    // of the usual 30 - 100 edges then stays in one piece.
    z.L = ROWS_L;
    static const int minL = debug_int("plan_min_l", 4);  // A/B switch, read once
    const int lim = plan_small_limit();
    // (round 5: pieces of the small form are at most 16 edges long. Without the cap a list of 1 273 rows x 32 edges got
    // pieces of 32 -- 192 workgroups walking 32 serial iterations: 36 us forward / 49 us backward where the same list
    // in pieces of 16 takes 29 / 40; cfg4's convolution chain 1.455 -> 1.365 ms. A larger single-workgroup layout
    // (plan_small=8192) with the same cap: the same times, fewer launches, 99 KB of LDS; kept at 4 096. The floor of
    // 16 for mid-size lists below: 8 and 32 both measured slower.)
    static const int maxL = debug_int("plan_small_max_l", 16);
    int Ls = minL;
    if (rows <= lim / 4 * 3 && minL < 16)
        while (Ls < ROWS_L && rows + e / Ls > lim / 8 * 7) Ls <<= 1;
    if (rows <= lim / 4 * 3 && minL < 16 && Ls <= maxL) {
        // the coarse levels of a hierarchy: the chip is empty, every iteration of a lane is exposed latency -- pieces as
        // short as the single-workgroup layout (plan_small_limit() virtual rows) allows
        z.L = Ls;
    } else if (rows < 16384) {
        static const int midL = debug_int("plan_mid_l", 16);
        long long want = e / 65536;
        int L = midL;
        while (L < want && L < ROWS_L) L <<= 1;
        z.L = L;
    }
    z.vcap = (long long)rows + e / z.L;
    z.windows = (int)((z.vcap + SELL_SIGMA - 1) / SELL_SIGMA);
    z.S = z.windows * (SELL_SIGMA / 64);
    z.slots = (long long)e + 64LL * z.L * z.windows;
    // Small lists (the coarse levels of a hierarchy: a few hundred rows) are launch-bound: six launches of layout, 25 us
    // of it a bitonic sort whose only purpose is less padding. One workgroup does the whole layout for them, in row
    // order; unsorted, a slice holds at most 64 L slots.
    static const bool allowSmall = !debug_int("plan_small_off", 0);  // A/B switch, read once
    z.small = allowSmall && rows <= lim && z.vcap <= lim;
    if (z.small) z.slots = (long long)z.L * (z.vcap + 64);
    return z;
}

// ------------------------------------------------------------------------------------------------ layout
// pieces per row position of the visiting order
__device__ __forceinline__ void vr_count_body(int p, const int* __restrict__ rowStart, int rows, int e, const int* __restrict__ order,
                                              int* __restrict__ vcnt, int L) {
    if (p >= rows) return;
    int r = order ? order[p] : p;
    r = max(0, min(r, rows - 1));
    const int deg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
    vcnt[p] = max(1, (deg + L - 1) / L);
}
__global__ __launch_bounds__(256) void vr_count(const int* __restrict__ rowStart, int rows, int e, const int* __restrict__ order,
                                                int* __restrict__ vcnt, int L, ClearSpan x1, ClearSpan x2) {
    clear_span_dev(x1);  // the status words of the two prefix sums that follow in this chain (common.h)
    clear_span_dev(x2);
    vr_count_body(blockIdx.x * blockDim.x + threadIdx.x, rowStart, rows, e, order, vcnt, L);
}
// (batch form: the status words are cleared by the head launch of the batch)
__global__ __launch_bounds__(256) void vr_count_batch(LargeBatch lb, BatchBlocks bb) {
    int local, blocks;
    const LargeItem& t = lb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    vr_count_body(local * 256 + (int)threadIdx.x, t.rowStart, t.rows, t.e, t.order, t.vcnt, t.L);
}
// virtual row id -> row; row -> its first virtual row id
__global__ __launch_bounds__(256) void vr_expand(const int* __restrict__ rowStart, int rows, int e, const int* __restrict__ order,
                                                 const int* __restrict__ vposP, int* __restrict__ vposRow,
                                                 int* __restrict__ vlistRow, int L) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rows) return;
    int r = order ? order[p] : p;
    r = max(0, min(r, rows - 1));
    const int deg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
    const int vc = max(1, (deg + L - 1) / L), v0 = vposP[p];
    vposRow[r] = v0;
    for (int k = 0; k < vc; ++k) vlistRow[v0 + k] = r;
}
// vr_expand with the prefix sum of the pieces inside (single pass, decoupled look-back: chain.h) -- the launch between
// vr_count and vr_expand is gone. Tile = 2048 consecutive row positions of the visiting order; status words (tiles + the
// ticket) cleared by vr_count, which runs before. vTotal receives the number of virtual rows (sell_sort reads it).
__device__ __forceinline__ void vr_scan_expand_body(const int nblk, const int* __restrict__ rowStart, int rows, int e,
                                                    const int* __restrict__ order, const int* __restrict__ vcnt,
                                                    unsigned long long* status, int* __restrict__ vposRow,
                                                    int* __restrict__ vlistRow, int* __restrict__ vTotal) {
    __shared__ int lds[4];
    __shared__ int sOff;
    __shared__ int sTile;
    if (threadIdx.x == 0)
        sTile = (int)__hip_atomic_fetch_add(status + nblk, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int tile = sTile;
    const int base = tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < rows) ? vcnt[base + k] : 0;
        s += v[k];
    }
    int tot;
    const int ex = block_excl_scan(s, tot, lds);
    if (threadIdx.x < 64) {
        const int excl = chain_lookback(status, tile, tot, (int)threadIdx.x);
        if (threadIdx.x == 0) sOff = excl;
    }
    __syncthreads();
    int run = ex + sOff;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int p = base + k;
        if (p < rows) {
            int r = order ? order[p] : p;
            r = max(0, min(r, rows - 1));
            vposRow[r] = run;
            for (int j = 0; j < v[k]; ++j) vlistRow[run + j] = r;
        }
        run += v[k];
    }
    if (tile == nblk - 1 && threadIdx.x == SCAN_THREADS - 1) *vTotal = run;
}
__global__ __launch_bounds__(SCAN_THREADS) void vr_scan_expand(const int* __restrict__ rowStart, int rows, int e,
                                                               const int* __restrict__ order, const int* __restrict__ vcnt,
                                                               unsigned long long* status, int* __restrict__ vposRow,
                                                               int* __restrict__ vlistRow, int* __restrict__ vTotal) {
    vr_scan_expand_body((int)gridDim.x, rowStart, rows, e, order, vcnt, status, vposRow, vlistRow, vTotal);
}
__global__ __launch_bounds__(SCAN_THREADS) void vr_scan_expand_batch(LargeBatch lb, BatchBlocks bb) {
    int local, blocks;
    const LargeItem& t = lb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    vr_scan_expand_body(blocks, t.rowStart, t.rows, t.e, t.order, t.vcnt, t.st1, t.vposRow, t.vlistRow, t.vTotal);
}

// The whole layout of a small list in one workgroup of 1024 threads (rows, virtual rows <= MCCNN_PLAN_SMALL): pieces per
// row and their prefix sum (4 row positions per thread), then one thread per VIRTUAL row -- its row found by a binary
// search over the rows' first virtual-row ids -- so that a list of 73 rows with 150 pieces each is expanded by 1024
// threads, not by 73 (the per-row loop of the first version took 52 us on such a list); slice lengths by one wave per
// slice, offsets by a serial pass over <= 65 slices. No sort: the slices keep the visiting order.
__device__ __forceinline__ void plan_small_body(const int* __restrict__ rowStart, int rows, int e, const int* __restrict__ order,
                                                int L, int S, int* __restrict__ vrow, int* __restrict__ vcode,
                                                int* __restrict__ sliceOff, int* __restrict__ vposRow) {
    __shared__ int posV0[MCCNN_PLAN_SMALL + 1];  // row position -> first virtual row id (exclusive prefix of the pieces)
    __shared__ int posRow[MCCNN_PLAN_SMALL];     // row position -> row id
    __shared__ int lens[MCCNN_PLAN_SMALL + 64];  // virtual row -> length
    __shared__ int slen[MCCNN_PLAN_SMALL / 64 + 2];
    __shared__ int wsum[17];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int PER = MCCNN_PLAN_SMALL / 1024;  // 4 consecutive row positions per thread
    int vc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int p = t * PER + k;
        vc[k] = 0;
        if (p < rows) {
            int r = order ? order[p] : p;
            r = max(0, min(r, rows - 1));
            const int deg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
            posRow[p] = r;
            vc[k] = max(1, (deg + L - 1) / L);
        }
        sum += vc[k];
    }
    int Vt;
    int v0 = block1024_excl_scan(sum, Vt, wsum);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int p = t * PER + k;
        if (p < rows) {
            posV0[p] = v0;
            vposRow[posRow[p]] = v0;
            v0 += vc[k];
        }
    }
    if (t == 0) posV0[rows] = Vt;
    __syncthreads();
    for (int v = t; v < S * 64; v += 1024) {
        if (v < Vt && v < MCCNN_PLAN_SMALL) {
            int lo = 0, hi = rows - 1;  // largest position p with posV0[p] <= v
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (posV0[mid] <= v) lo = mid; else hi = mid - 1;
            }
            const int r = posRow[lo];
            const int deg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
            const int piece = v - posV0[lo];
            vrow[v] = r;
            vcode[v] = (deg > L) ? v : ~v;
            lens[v] = max(0, min(L, deg - piece * L));
        } else {
            vrow[v] = -1;
            vcode[v] = -1;
            if (v < MCCNN_PLAN_SMALL + 64) lens[v] = 0;
        }
    }
    __syncthreads();
    for (int sl = wave; sl < S; sl += 16) {
        int d = lens[min(sl * 64 + lane, MCCNN_PLAN_SMALL + 63)];
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) d = max(d, __shfl_xor(d, s2, 64));
        if (lane == 0) slen[sl] = d * 64;
    }
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int sl = 0; sl < S; ++sl) { sliceOff[sl] = run; run += slen[sl]; }
        sliceOff[S] = run;
    }
}
__global__ __launch_bounds__(1024) void plan_small(const int* __restrict__ rowStart, int rows, int e, const int* __restrict__ order,
                                                   int L, int S, int* __restrict__ vrow, int* __restrict__ vcode,
                                                   int* __restrict__ sliceOff, int* __restrict__ vposRow) {
    plan_small_body(rowStart, rows, e, order, L, S, vrow, vcode, sliceOff, vposRow);
}
// ... of a BATCH of small lists: one workgroup per plan, one launch
__global__ __launch_bounds__(1024) void plan_small_batch(PlanSmallBatch pb) {
    const PlanSmallItem& t = pb.it[blockIdx.x];
    plan_small_body(t.rowStart, t.rows, t.e, t.order, t.L, t.S, t.vrow, t.vcode, t.sliceOff, t.vposRow);
}

// One workgroup per window of SELL_SIGMA virtual rows: a STABLE sort by descending length (ties keep the visiting order) ->
// the layout is a deterministic function of the list, so gradients are bit-reproducible run to run. Lengths are at most
// ROWS_L: the key (L - len, padding last) has 8 bits and the sort is two 4-bit passes of a least-significant-digit radix
// sort -- per pass 16 ballots per element slot, one 256-entry scan, two barriers -- instead of the 55 barrier-separated
// stages of a bitonic network over 1024 keys (27 us per launch whatever the list, 14-16 launches per step of a
// segmentation network; ~8 us now).
__device__ __forceinline__ void sell_radix_pass(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int shift,
                                                int* __restrict__ cnt /* [16 digits][16 = slot * 4 + wave] */,
                                                int* __restrict__ wsum /* 5 */) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned key[4];
    int rank[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        key[i] = src[i * 256 + t];
        const int d = (int)((key[i] >> (10 + shift)) & 15u);
        rank[i] = 0;
#pragma unroll
        for (int dd = 0; dd < 16; ++dd) {
            const unsigned long long m = __ballot(d == dd);
            if (d == dd) rank[i] = __popcll(m & lt);
            if (lane == 0) cnt[dd * 16 + i * 4 + wave] = __popcll(m);
        }
    }
    __syncthreads();
    {   // exclusive scan of the 256 counters in (digit, slot, wave) order
        const int v = cnt[t];
        const int incl = wave_incl_scan(v);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < wave; ++k) base += wsum[k];
        __syncthreads();
        cnt[t] = base + incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = (int)((key[i] >> (10 + shift)) & 15u);
        dst[cnt[d * 16 + i * 4 + wave] + rank[i]] = key[i];
    }
    __syncthreads();
}

// The slices' first slots (sliceOff = exclusive prefix of the slice lengths, sliceOff[S] = total) come out of the same
// launch: the windows are chained by a decoupled look-back over their slot totals (chain.h; windows taken from a ticket;
// status words cleared by vr_count at the head of the layout) -- no prefix-sum launch behind the sort.
__device__ __forceinline__ void sell_sort_body(const int nblk, const int* __restrict__ rowStart, int rows, int e,
                                               const int* __restrict__ vlistRow, const int* __restrict__ vposRow,
                                               const int* __restrict__ vTotal, int* __restrict__ vrow,
                                               int* __restrict__ vcode, int* __restrict__ sliceOff, int L,
                                               unsigned long long* status) {
    __shared__ unsigned key[SELL_SIGMA];
    __shared__ unsigned key2[SELL_SIGMA];
    __shared__ int rowOf[SELL_SIGMA];
    __shared__ int lenOf[SELL_SIGMA];
    __shared__ int cutOf[SELL_SIGMA];
    __shared__ int cnt[256];
    __shared__ int wsum[5];
    __shared__ int sWin, sBase, sLen[SELL_SIGMA / 64];
    if (threadIdx.x == 0)
        sWin = (int)__hip_atomic_fetch_add(status + nblk, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int win = sWin;
    const int V = *vTotal;
    const int w0 = win * SELL_SIGMA;
    for (int k = threadIdx.x; k < SELL_SIGMA; k += 256) {
        const int v = w0 + k;
        int r = -1, len = 0, cut = 0;
        if (v < V) {
            r = vlistRow[v];
            const int deg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
            const int piece = v - vposRow[r];
            len = max(0, min(L, deg - piece * L));
            cut = deg > L;
        }
        rowOf[k] = r;
        lenOf[k] = len;
        cutOf[k] = cut;
        key[k] = ((v < V ? (unsigned)(L - len) : (unsigned)(L + 1)) << 10) | (unsigned)k;  // padding sorts last
    }
    __syncthreads();
    sell_radix_pass(key, key2, 0, cnt, wsum);
    sell_radix_pass(key2, key, 4, cnt, wsum);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int g = wave; g < SELL_SIGMA / 64; g += 4) {
        const int slice = w0 / 64 + g;
        const int kk = (int)(key[g * 64 + lane] & 1023u);
        const int r = rowOf[kk];
        vrow[slice * 64 + lane] = r;
        vcode[slice * 64 + lane] = (r >= 0) ? (cutOf[kk] ? (w0 + kk) : ~(w0 + kk)) : -1;
        int d = lenOf[kk];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) d = max(d, __shfl_xor(d, s, 64));
        if (lane == 0) sLen[g] = d * 64;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane2 = threadIdx.x;
        int tot = 0;
#pragma unroll
        for (int g = 0; g < SELL_SIGMA / 64; ++g) tot += sLen[g];
        const int excl = chain_lookback(status, win, tot, lane2);
        if (lane2 == 0) sBase = excl;
    }
    __syncthreads();
    if (threadIdx.x < SELL_SIGMA / 64) {
        int run = sBase;
        for (int g = 0; g < (int)threadIdx.x; ++g) run += sLen[g];
        sliceOff[w0 / 64 + threadIdx.x] = run;
        if (win == nblk - 1 && threadIdx.x == SELL_SIGMA / 64 - 1) sliceOff[w0 / 64 + SELL_SIGMA / 64] = run + sLen[threadIdx.x];
    }
}
__global__ __launch_bounds__(256) void sell_sort(const int* __restrict__ rowStart, int rows, int e,
                                                 const int* __restrict__ vlistRow, const int* __restrict__ vposRow,
                                                 const int* __restrict__ vTotal, int* __restrict__ vrow,
                                                 int* __restrict__ vcode, int* __restrict__ sliceOff, int L,
                                                 unsigned long long* status) {
    sell_sort_body((int)gridDim.x, rowStart, rows, e, vlistRow, vposRow, vTotal, vrow, vcode, sliceOff, L, status);
}
__global__ __launch_bounds__(256) void sell_sort_batch(LargeBatch lb, BatchBlocks bb) {
    int local, blocks;
    const LargeItem& t = lb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    sell_sort_body(blocks, t.rowStart, t.rows, t.e, t.vlistRow, t.vposRow, t.vTotal, t.vrow, t.vcode, t.sliceOff, t.L, t.st2);
}

// The records of a slice in (iteration, lane) order: a pure PERMUTATION of the per-edge records (delta, 1 / (pdf K)) that
// mccnn_edge_records writes once per list in edge order (coalesced reads, the gathers of points / centres / row lengths
// paid once for both plans). Workgroup (slice, chunk of 16 iterations), wave w takes 4 consecutive iterations: in the
// forward plan a lane's 4 edges are 64 contiguous bytes of records and 32 of pairs. (One wave per slice walking all its
// iterations with the set-up arithmetic inline was 260 us per plan on the 100k room.)
// The lane id read from the hardware again (two VALU instructions) instead of a value kept live -- or spilled -- across a
// sweep that needs every register.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
#define MCCNN_FILL_CHUNK 16
// INL: the records are computed here from the geometry instead of permuted from the edge-order array -- small lists, where
// the extra launch and buffer of mccnn_edge_records cost more than evaluating every record twice (once per plan).
template <bool TR, bool INL>
__device__ __forceinline__ void sell_fill_body(const int slice, const int by, const float4* __restrict__ recE, const int2* __restrict__ packed, int e,
                                               const int* __restrict__ rowStart, int rows, const int* __restrict__ permT,
                                               const RowPlan& p, long long cap, float4* __restrict__ rec, int* __restrict__ oth,
                                               int L, const ConvArgs& a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int off = p.sliceOff[slice];
    const int len = (p.sliceOff[slice + 1] - off) >> 6;
    const int it0 = by * MCCNN_FILL_CHUNK + wave * (MCCNN_FILL_CHUNK / 4);
    if (it0 >= len || (long long)off + (long long)len * 64 > cap) return;  // (the bound of plan_sizes makes the latter impossible)
    const int r = p.vrow[slice * 64 + lane];
    int base = 0, deg = 0;
    if (r >= 0) {
        const int code = p.vcode[slice * 64 + lane];
        const int piece = (code < 0 ? ~code : code) - p.vposRow[r];
        const int rb = rowStart[r];
        const int rdeg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rb;
        base = rb + piece * L;
        deg = max(0, min(L, rdeg - piece * L));
    }
    constexpr int K = MCCNN_FILL_CHUNK / 4;
    // all loads of the 4 iterations first (clamped indices: no branch between them)
    int eid[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = base + min(it0 + k, max(deg - 1, 0));
        eid[k] = TR ? permT[min(t, e - 1)] : min(t, e - 1);
    }
    const int2 p0 = packed[TR ? permT[min(base, e - 1)] : min(base, e - 1)];
    const int pad = (deg > 0) ? (TR ? p0.y : p0.x) : 0;  // padding slots: the row's first neighbour (valid, cached; 1/(pdf K) = 0)
    float4 rc[K];
    int2 pr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { rc[k] = INL ? edge_record(a, eid[k]) : recE[eid[k]]; pr[k] = packed[eid[k]]; }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int it = it0 + k;
        if (it < len) {
            const bool real = it < deg;
            const size_t slot = (size_t)off + (size_t)it * 64 + lane;
            rec[slot] = real ? rc[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            oth[slot] = real ? (TR ? pr[k].y : pr[k].x) : pad;
        }
    }
}
template <bool TR, bool INL>
__global__ __launch_bounds__(256) void sell_fill(const float4* __restrict__ recE, const int2* __restrict__ packed, int e,
                                                 const int* __restrict__ rowStart, int rows, const int* __restrict__ permT,
                                                 RowPlan p, long long cap, float4* __restrict__ rec, int* __restrict__ oth,
                                                 int L, ConvArgs a) {
    sell_fill_body<TR, INL>((int)blockIdx.x, (int)blockIdx.y, recE, packed, e, rowStart, rows, permT, p, cap, rec, oth, L, a);
}
// ... of a BATCH of small plans (records evaluated inline): one launch for all forward plans, one for all transposed ones
template <bool TR>
__global__ __launch_bounds__(256) void sell_fill_batch(SellFillBatch fb, BatchBlocks bb) {
    int local, blocks;
    const SellFillItem& t = fb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    ConvArgs a = {};
    a.pts = t.pts; a.bids = t.bids; a.pdfs = t.pdfs; a.samples = t.samples; a.start = t.start; a.packed = t.packed; a.mn = t.mn; a.mx = t.mx;
    a.n = t.n; a.m = t.m; a.e = t.e; a.radius = t.radius; a.invRadius = 1.0f / t.radius; a.scaleInv = t.scaleInv; a.avg = t.avg; a.B = t.B;
    const RowPlan p = {t.vrow, t.vcode, t.sliceOff, t.vposRow, nullptr, nullptr, t.rows, t.S};
    sell_fill_body<TR, true>(local % t.S, local / t.S, nullptr, t.packed, t.e, t.rowStart, t.rows, t.permT, p, t.cap, t.rec, t.oth, t.L, a);
}

// ------------------------------------------------------------------------------------------------ scatter fill (transposed plan)
// Large lists (every plan that is not laid out by plan_small). sell_fill above GATHERS: a lane owns a row, so the three
// reads of a slot (perm_t, record, pair) all touch a line of their own -- the transposed plan of the 100k room fetched
// 1.01 GB to write 94 MB in 168 us, behind 54 us of tr_rank. plan_scatter_tr turns it round: a thread owns a POSITION of
// the row-grouped list, looks up where its row's piece lies (vinfo: first slot and slice length per virtual row, written
// by plan_bases) and stores one record and one index. The stores of one row are 1 KB apart, but the eight rows that share
// a 128-byte line are rows of one window of SELL_SIGMA points, i.e. of one run of positions: with the workgroups in
// XCD-contiguous order every line fills up inside ONE L2. RANK = true is the last phase of the transposition itself
// (tr_rank of conv.hip: ascending edge id inside a row) with the plan written in the same pass -- 120 us for both on the
// room (ablations: no rank scan 94, no record gather 112, linear instead of scattered stores 92); RANK = false starts from
// a finished perm_t.
// Padding: the thread that owns the LAST edge of a virtual row fills the row's remaining slots of its slice (zero
// records, its own index: a line the sweep has just used); virtual rows without any edge and the padding lanes behind the
// last row are filled by plan_bases (index 0).
// (Measured and dropped for the FORWARD plan: the same scatter with threads in edge order -- 189 us against 30 + 89 us:
// centres are listed in the caller's order, the windows in the cell-coherent visiting order, so the eight rows of a line
// are far apart in edge order. The forward plan takes plan_fill_tiles below. And for the transposed plan: tr_rank +
// plan_fill_tiles over perm_t, 54 + 118 us.)
__device__ __forceinline__ int row_len(const int* __restrict__ rowStart, int rows, int e, int r) {
    return ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
}
__device__ __forceinline__ void plan_bases_body(const int pos, const RowPlan& p, const int* __restrict__ rowStart, int rows, int e, int L,
                                                long long cap, int2* __restrict__ vinfo, float4* __restrict__ rec, int* __restrict__ oth) {
    if (pos >= p.S * 64) return;
    const int slice = pos >> 6, lane = pos & 63;
    const int off = p.sliceOff[slice];
    int len = (p.sliceOff[slice + 1] - off) >> 6;
    if ((long long)off + (long long)len * 64 > cap) len = 0;  // (the bound of plan_sizes makes this impossible)
    const int r = p.vrow[pos];
    int deg = 0;
    if (r >= 0) {
        const int code = p.vcode[pos];
        const int vid = code < 0 ? ~code : code;
        vinfo[vid] = make_int2(off + lane, len);
        deg = max(0, min(L, row_len(rowStart, rows, e, r) - (vid - p.vposRow[r]) * L));
    }
    if (deg == 0) {
        for (int it = 0; it < len; ++it) {
            const size_t slot = (size_t)off + (size_t)it * 64 + lane;
            rec[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
            oth[slot] = 0;
        }
    }
}
__global__ __launch_bounds__(256) void plan_bases(RowPlan p, const int* __restrict__ rowStart, int rows, int e, int L, long long cap,
                                                  int2* __restrict__ vinfo, float4* __restrict__ rec, int* __restrict__ oth) {
    plan_bases_body(blockIdx.x * blockDim.x + threadIdx.x, p, rowStart, rows, e, L, cap, vinfo, rec, oth);
}
__global__ __launch_bounds__(256) void plan_bases_batch(LargeBatch lb, BatchBlocks bb) {
    int local, blocks;
    const LargeItem& t = lb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const RowPlan p = {t.vrow, t.vcode, t.sliceOff, t.vposRow, nullptr, nullptr, t.rows, t.S};
    plan_bases_body(local * 256 + (int)threadIdx.x, p, t.rowStart, t.rows, t.e, t.L, t.cap, t.vinfo, t.rec, t.oth);
}
// the slot of edge #r of row `row` (deg edges), and the padding behind the row's last edge
__device__ __forceinline__ void plan_put(int row, int r, int deg, int L, const int* __restrict__ vposRow,
                                         const int2* __restrict__ vinfo, float4* __restrict__ rec, int* __restrict__ oth,
                                         const float4& rc, int other) {
    const int pc = r / L, it = r - pc * L;
    const int2 vi = vinfo[vposRow[row] + pc];
    const int plen = min(L, deg - pc * L);
    if (it >= vi.y) return;  // (only a layout cut off by its capacity)
    const size_t slot = (size_t)vi.x + (size_t)it * 64;
    rec[slot] = rc;
    oth[slot] = other;
    if (it == plen - 1) {
        for (int k = plen; k < vi.y; ++k) {
            const size_t sp = (size_t)vi.x + (size_t)k * 64;
            rec[sp] = make_float4(0.f, 0.f, 0.f, 0.f);
            oth[sp] = other;
        }
    }
}
template <bool RANK>
__device__ __forceinline__ void plan_scatter_tr_body(const int p, const float4* __restrict__ recE, const int2* __restrict__ packed, int e, int n,
                                                     const int* __restrict__ startT, const int* __restrict__ tmp,
                                                     int* __restrict__ permT, const int* __restrict__ vposRow,
                                                     const int2* __restrict__ vinfo, int L, float4* __restrict__ rec,
                                                     int* __restrict__ oth) {
    if (p >= e) return;
    const int v = RANK ? tmp[p] : permT[p];
    const int2 pr = packed[v];
    const int j = max(0, min(pr.x, n - 1));
    const int s0 = startT[j], s1 = startT[j + 1];
    int r = p - s0;
    if (RANK) {  // stable order inside a row: ascending edge id (see tr_rank)
        r = 0;
        int q = s0;
        for (; q + 4 <= s1; q += 4) {
            const int a0 = tmp[q], a1 = tmp[q + 1], a2 = tmp[q + 2], a3 = tmp[q + 3];
            r += (a0 < v) + (a1 < v) + (a2 < v) + (a3 < v);
        }
        for (; q < s1; ++q) r += (tmp[q] < v) ? 1 : 0;
        permT[s0 + r] = v;
    }
    plan_put(j, r, s1 - s0, L, vposRow, vinfo, rec, oth, recE[v], pr.y);
}
template <bool RANK>
__global__ __launch_bounds__(256) void plan_scatter_tr(const float4* __restrict__ recE, const int2* __restrict__ packed, int e, int n,
                                                       const int* __restrict__ startT, const int* __restrict__ tmp,
                                                       int* __restrict__ permT, const int* __restrict__ vposRow,
                                                       const int2* __restrict__ vinfo, int L, float4* __restrict__ rec,
                                                       int* __restrict__ oth) {
    plan_scatter_tr_body<RANK>(xcd_contiguous(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x, recE, packed, e, n, startT, tmp, permT,
                               vposRow, vinfo, L, rec, oth);
}
__global__ __launch_bounds__(256) void plan_scatter_tr_batch(LargeBatch lb, BatchBlocks bb) {
    int local, blocks;
    const LargeItem& t = lb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const int p = xcd_contiguous(local, blocks) * 256 + (int)threadIdx.x;
    if (t.rank) plan_scatter_tr_body<true>(p, t.recE, t.packed, t.e, t.n, t.startT, t.tmp, t.permT, t.vposRow, t.vinfo, t.L, t.rec, t.oth);
    else plan_scatter_tr_body<false>(p, t.recE, t.packed, t.e, t.n, t.startT, nullptr, t.permT, t.vposRow, t.vinfo, t.L, t.rec, t.oth);
}

// ------------------------------------------------------------------------------------------------ tile fill (forward plan)
// The permutation of sell_fill with BOTH sides coalesced: a workgroup owns (slice, 16 iterations) as in sell_fill, but reads the
// tile row by row -- 16 consecutive lanes take 16 consecutive edges of one row (128 contiguous bytes of pairs; the record
// is evaluated on the spot, EVAL, and its edge-order copy written as 256 contiguous bytes, or read from the edge-order
// array) -- parks it in LDS and writes it out iteration by iteration: 1 KB per wave store.
// LDS tile [it][row] with a row stride of 65 float4: the 8 lanes one ds_write_b128 group serves hold 8 iterations of a
// row = 8 x 4 distinct banks.
template <bool EVAL>
__device__ __forceinline__ void plan_fill_tiles_body(const int lin, const ConvArgs& a, const float4* __restrict__ recIn, float4* __restrict__ recOut,
                                                     const int* __restrict__ rowStart, int rows, const RowPlan& p, long long cap,
                                                     float4* __restrict__ rec, int* __restrict__ oth, int L,
                                                     float4 (*tRec)[65], int (*tOth)[65]) {
    // workgroups in XCD-contiguous order, the chunks of a slice side by side: the slices of a window -- rows of one region of
    // space, whose gathered lines they share -- stay on one XCD
    const int chunks = (L + MCCNN_FILL_CHUNK - 1) / MCCNN_FILL_CHUNK;
    const int slice = lin / chunks;
    if (slice >= p.S) return;
    const int off = p.sliceOff[slice];
    const int len = (p.sliceOff[slice + 1] - off) >> 6;
    const int it0 = (lin - slice * chunks) * MCCNN_FILL_CHUNK;
    if (it0 >= len || (long long)off + (long long)len * 64 > cap) return;
    const int e = a.e;
    const int itl = threadIdx.x & 15, it = it0 + itl;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int rl = pass * 16 + (threadIdx.x >> 4);
        const int r = p.vrow[slice * 64 + rl];
        int base = 0, deg = 0;
        if (r >= 0) {
            const int code = p.vcode[slice * 64 + rl];
            const int piece = (code < 0 ? ~code : code) - p.vposRow[r];
            const int rb = rowStart[r];
            const int rdeg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rb;
            base = rb + piece * L;
            deg = max(0, min(L, rdeg - piece * L));
        }
        float4 rc = make_float4(0.f, 0.f, 0.f, 0.f);
        int other = 0;
        if (deg > 0) {
            const bool real = it < deg;
            const int t = min(base + (real ? it : 0), e - 1);  // padding slots: the row's first neighbour (valid, cached)
            const int eid = t;
            const int2 pr = a.packed[eid];
            other = pr.x;
            if (real) {
                if (EVAL) {
                    rc = edge_record(a, eid);
                    if (recOut) recOut[eid] = rc;
                } else {
                    rc = recIn[eid];
                }
            }
        }
        tRec[itl][rl] = rc;
        tOth[itl][rl] = other;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MCCNN_FILL_CHUNK / 4; ++k) {
        const int il = k * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (it0 + il < len) {
            const size_t slot = (size_t)off + (size_t)(it0 + il) * 64 + lane;
            rec[slot] = tRec[il][lane];
            oth[slot] = tOth[il][lane];
        }
    }
}
template <bool EVAL>
__global__ __launch_bounds__(256) void plan_fill_tiles(ConvArgs a, const float4* __restrict__ recIn, float4* __restrict__ recOut,
                                                       const int* __restrict__ rowStart, int rows, RowPlan p, long long cap, float4* __restrict__ rec, int* __restrict__ oth,
                                                       int L) {
    __shared__ float4 tRec[MCCNN_FILL_CHUNK][65];
    __shared__ int tOth[MCCNN_FILL_CHUNK][65];
    plan_fill_tiles_body<EVAL>(xcd_contiguous(blockIdx.x, gridDim.x), a, recIn, recOut, rowStart, rows, p, cap, rec, oth, L, tRec, tOth);
}
__device__ __forceinline__ ConvArgs large_conv_args(const LargeItem& t) {
    ConvArgs a = {};
    a.pts = t.pts; a.bids = t.bids; a.pdfs = t.pdfs; a.samples = t.samples; a.start = t.start; a.packed = t.packed; a.mn = t.mn; a.mx = t.mx;
    a.n = t.n; a.m = t.m; a.e = t.e; a.radius = t.radius; a.invRadius = t.radius > 0.0f ? 1.0f / t.radius : 0.0f; a.scaleInv = t.scaleInv;
    a.avg = t.avg; a.B = t.B;
    return a;
}
__global__ __launch_bounds__(256) void plan_fill_tiles_batch(LargeBatch lb, BatchBlocks bb) {
    __shared__ float4 tRec[MCCNN_FILL_CHUNK][65];
    __shared__ int tOth[MCCNN_FILL_CHUNK][65];
    int local, blocks;
    const LargeItem& t = lb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const ConvArgs a = large_conv_args(t);
    const RowPlan p = {t.vrow, t.vcode, t.sliceOff, t.vposRow, nullptr, nullptr, t.rows, t.S};
    const int lin = xcd_contiguous(local, blocks);
    if (t.eval) plan_fill_tiles_body<true>(lin, a, nullptr, t.recE, t.rowStart, t.rows, p, t.cap, t.rec, t.oth, t.L, tRec, tOth);
    else plan_fill_tiles_body<false>(lin, a, t.recE, nullptr, t.rowStart, t.rows, p, t.cap, t.rec, t.oth, t.L, tRec, tOth);
}
// the per-edge records of the transposed plans of a batch whose forward plan did not leave them (conv.hip edge_records)
__global__ __launch_bounds__(256) void edge_records_batch(LargeBatch lb, BatchBlocks bb) {
    int local, blocks;
    const LargeItem& t = lb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const int k = local * 256 + (int)threadIdx.x;
    if (k >= t.e) return;
    const ConvArgs a = large_conv_args(t);
    t.recE[k] = edge_record(a, k);
}

// The pieces of cut rows left their sums in scratch rows (one per virtual row id, `cols` 32-bit words wide -- f32 rows,
// or bf16 rows whose pieces are kept in f32): out[r] = sum over the row's pieces, in order. One wave per 64 rows; cut rows
// are rare (none at all on a list whose rows hold <= ROWS_L edges), a wave without one returns after two loads.
// Rows without any edge are written here as well (zeros).
template <bool BF>
__device__ __forceinline__ void rows_combine_body(int bid, int tid, const int* __restrict__ rowStart, int rows, int e,
                                                  const int* __restrict__ vposRow, const float* __restrict__ scratch,
                                                  int cols, void* __restrict__ out, int L, const int* __restrict__ outIdx) {
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = (bid * 4 + wave) * 64;
    const int r = r0 + lane;
    int deg = 0;
    if (r < rows) deg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
    // ... and rows WITHOUT an edge: zero pieces, a zero row. The row kernels store such a row only when its slice holds
    // an edge of some other row; a slice made of empty rows alone (they sort to the end of their window) has length 0
    // like the padding slices behind the list and is skipped there.
    unsigned long long cut = __ballot(deg > L || (r < rows && deg == 0));
    while (cut) {
        const int l = (int)__builtin_ctzll(cut);
        cut &= cut - 1;
        const int rr = r0 + l;
        const int pieces = (__builtin_amdgcn_readlane(deg, l) + L - 1) / L;
        const int v0 = vposRow[rr];
        const int ro = outIdx ? outIdx[rr] : rr;
        for (int c = lane; c < cols; c += 64) {
            float acc = 0.f;
            for (int k = 0; k < pieces; ++k) acc += scratch[(size_t)(v0 + k) * cols + c];
            if (BF) reinterpret_cast<__bf16*>(out)[(size_t)ro * cols + c] = (__bf16)acc;
            else reinterpret_cast<float*>(out)[(size_t)ro * cols + c] = acc;
        }
    }
}

template <bool BF>
__global__ __launch_bounds__(256) void rows_combine(const int* __restrict__ rowStart, int rows, int e,
                                                    const int* __restrict__ vposRow, const float* __restrict__ scratch,
                                                    int cols, void* __restrict__ out, int L, const int* __restrict__ outIdx) {
    rows_combine_body<BF>(blockIdx.x, threadIdx.x, rowStart, rows, e, vposRow, scratch, cols, out, L, outIdx);
}

// The same for lists with few rows, whose plans use short pieces (most rows are cut): one thread per (row, 4 columns).
template <bool BF>
__device__ __forceinline__ void rows_combine_par_body(long long t, const int* __restrict__ rowStart, int rows, int e,
                                                      const int* __restrict__ vposRow, const float* __restrict__ scratch,
                                                      int cols, void* __restrict__ out, int L, const int* __restrict__ outIdx) {
    const int c4 = cols >> 2;
    if (t >= (long long)rows * c4) return;
    const int r = (int)(t / c4), c = (int)(t - (long long)r * c4) * 4;
    const int deg = ((r + 1 < rows) ? rowStart[r + 1] : e) - rowStart[r];
    if (deg <= L && deg != 0) return;  // (a row without an edge: zero pieces, a zero row -- see rows_combine)
    const int pieces = (deg + L - 1) / L, v0 = vposRow[r];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < pieces; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(scratch + (size_t)(v0 + k) * cols + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int ro = outIdx ? outIdx[r] : r;
    if (BF) {
        unsigned* o = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(out) + (size_t)ro * cols + c);
        o[0] = f32x2_to_bf16(acc.x, acc.y);
        o[1] = f32x2_to_bf16(acc.z, acc.w);
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)ro * cols + c) = acc;
    }
}
template <bool BF>
__global__ __launch_bounds__(256) void rows_combine_par(const int* __restrict__ rowStart, int rows, int e,
                                                        const int* __restrict__ vposRow, const float* __restrict__ scratch,
                                                        int cols, void* __restrict__ out, int L, const int* __restrict__ outIdx) {
    rows_combine_par_body<BF>((long long)blockIdx.x * blockDim.x + threadIdx.x, rowStart, rows, e, vposRow, scratch, cols, out, L, outIdx);
}

// The two small kernels that follow the backward row kernel in ONE launch (workgroups of 1024 threads): the first `nRed`
// sum the weight-gradient partials (reduce_partials_body), the rest combine the cut / empty rows of the feature gradient,
// four 256-thread pieces per workgroup. A depth-wise layer's backward pass is 2 launches instead of 3 -- on the coarse
// levels of a network every launch is ~5 us on both sides of the queue.
template <bool BF>
__global__ __launch_bounds__(1024) void rows_combine_reduce(const int* __restrict__ rowStart, int rows, int e,
                                                            const int* __restrict__ vposRow, const float* __restrict__ scratch,
                                                            int cols, void* __restrict__ out, int L, const int* __restrict__ outIdx,
                                                            int par, int nRed, const float* __restrict__ partials, int numWaves, int nb,
                                                            float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                            float* __restrict__ db2, float* __restrict__ dw3, float* __restrict__ db3) {
    if ((int)blockIdx.x < nRed) {
        reduce_partials_body(blockIdx.x, partials, numWaves, nb, dw1, db1, dw2, db2, dw3, db3);
        return;
    }
    const int vb = ((int)blockIdx.x - nRed) * 4 + (threadIdx.x >> 8), vt = threadIdx.x & 255;
    if (par) rows_combine_par_body<BF>((long long)vb * 256 + vt, rowStart, rows, e, vposRow, scratch, cols, out, L, outIdx);
    else rows_combine_body<BF>(vb, vt, rowStart, rows, e, vposRow, scratch, cols, out, L, outIdx);
}

template <bool BF>
static void launch_combine_reduce(const int* rowStart, int rows, int e, const int* vposRow, const float* scratch, int cols, void* out,
                                  int L, hipStream_t s, const int* outIdx, const float* partials, int numWaves, int nb, float* dw1,
                                  float* db1, float* dw2, float* db2, float* dw3, float* db3) {
    const int par = rows < 16384 ? 1 : 0;
    const long long units = par ? ceil_div((long long)rows * (cols >> 2), 256) : ceil_div(rows, 256);   // 256-thread pieces
    const int nRed = (int)ceil_div((long long)nb * 176, 16);
    rows_combine_reduce<BF><<<(int)(nRed + ceil_div(units, 4)), 1024, 0, s>>>(rowStart, rows, e, vposRow, scratch, cols, out, L, outIdx,
                                                                           par, nRed, partials, numWaves, nb, dw1, db1, dw2, db2, dw3, db3);
}

template <bool BF>
static void launch_combine(const int* rowStart, int rows, int e, const int* vposRow, const float* scratch, int cols, void* out,
                           int L, hipStream_t s, const int* outIdx = nullptr) {
    if (rows < 16384)
        rows_combine_par<BF><<<ceil_div((long long)rows * (cols >> 2), 256), 256, 0, s>>>(rowStart, rows, e, vposRow, scratch, cols, out, L, outIdx);
    else
        rows_combine<BF><<<ceil_div(rows, 256), 256, 0, s>>>(rowStart, rows, e, vposRow, scratch, cols, out, L, outIdx);
}

// ------------------------------------------------------------------------------------------------ staged row gather
// The four waves of a workgroup work on four consecutive MLP blocks of the same 64 rows, so per iteration they need the
// four 32-byte pieces of ONE 128-byte line of each of the 64 gathered rows. Gathered piece by piece (every wave its own
// 32 bytes per lane) the L1 issues a request per (wave, lane) and keeps nothing: 4x the L1<->L2 requests the bytes
// need, and that request rate -- not HBM, not the matrix pipe -- bounded the wide depth-wise kernels (dw256 forward on
// the room: 1.03 ms; gather alone 1.13 ms; without it 0.66 ms). Staged: wave kp fetches the FULL lines of rows
// 16 kp .. 16 kp + 15 (lane l: row 16 kp + (l & 15), 32-byte piece l >> 4 -- four lanes cover one line, one request),
// parks them in LDS one iteration ahead, and after the barrier every wave reads its own piece of all 64 rows.
// LDS image per buffer: plane h (first / second 16 bytes of a piece) x [producer kp][piece c][row r] x 16 bytes, so a
// reader's 16 lanes of one producer group touch 256 contiguous bytes (conflict-free).
// bf16 rows: a line of 4 blocks is 64 bytes, pieces are 16 bytes, one plane.
struct RowStage {
    f32x4* buf;  // [2 buffers][2 planes][256]
    __device__ __forceinline__ f32x4* plane(int b, int h) const { return buf + (b * 2 + h) * 256; }
};
#define MCCNN_STAGE_FLOATS (2 * 2 * 256 * 4)

// producer side: the two (f32) / one (bf16) 16-byte loads of this lane's piece; cols = row length in elements
template <bool BF>
__device__ __forceinline__ void stage_load(const void* __restrict__ base, int row, int cols, int col0 /* first column of the tile */,
                                           int lane, f32x4& v0, f32x4& v1) {
    const int c = lane >> 4;  // piece 0..3 = block col0/8 + c
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (BF) {
        const unsigned short* rp = reinterpret_cast<const unsigned short*>(base) + (size_t)row * cols;
        v0 = (col0 + c * 8 < cols) ? *reinterpret_cast<const f32x4*>(rp + col0 + c * 8) : z;
        v1 = z;
    } else {
        const float* rp = reinterpret_cast<const float*>(base) + (size_t)row * cols;
        const bool ok = col0 + c * 8 < cols;
        v0 = ok ? *reinterpret_cast<const f32x4*>(rp + col0 + c * 8) : z;
        v1 = ok ? *reinterpret_cast<const f32x4*>(rp + col0 + c * 8 + 4) : z;
    }
}
template <bool BF>
__device__ __forceinline__ void stage_store(const RowStage& st, int b, int wave, int lane, const f32x4& v0, const f32x4& v1) {
    st.plane(b, 0)[wave * 64 + lane] = v0;
    if (!BF) st.plane(b, 1)[wave * 64 + lane] = v1;
}
// reader side: the 8 values of this wave's block for the lane's row
template <bool BF>
__device__ __forceinline__ void stage_read(const RowStage& st, int b, int wave, int lane, float* f) {
    const int slot = (lane >> 4) * 64 + wave * 16 + (lane & 15);
    if (BF) {
        const f32x4 u = st.plane(b, 0)[slot];
        bf16x8_to_f32(make_uint4(__float_as_uint(u.x), __float_as_uint(u.y), __float_as_uint(u.z), __float_as_uint(u.w)), f);
    } else {
        const f32x4 u0 = st.plane(b, 0)[slot], u1 = st.plane(b, 1)[slot];
        f[0] = u0.x; f[1] = u0.y; f[2] = u0.z; f[3] = u0.w; f[4] = u1.x; f[5] = u1.y; f[6] = u1.z; f[7] = u1.w;
    }
}

// ------------------------------------------------------------------------------------------------ depth-wise forward
// Work item = (slice, MLP block q): the block's weights stay in 44 VGPRs for the whole slice -- the loop reads no
// weights -- and the lane's 8 sums are stored once at the end. The four waves of a workgroup take four consecutive
// blocks of one slice and gather the feature rows together (RowStage). Workgroups are dispatched in slice order =
// windows of descending row length: fine-grained items, no tail. FEAT: 2 = f32 rows, 4 = bf16 rows.
template <int FEAT>
__global__ __launch_bounds__(256) void dw_fwd_rows(ConvArgs a, RowPlan p, float* __restrict__ out, float* __restrict__ scratch,
                                                   int qTiles, int spw, int groups, const int* __restrict__ featIdx) {
    __shared__ float stageMem[MCCNN_STAGE_FLOATS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i4 = lane & 3;
    constexpr bool BF = FEAT == 4;
    unsigned short* out16 = reinterpret_cast<unsigned short*>(out);
    // Workgroup -> (block tile, group of spw consecutive slices), tile-major, every XCD a contiguous range: the
    // workgroups resident on one XCD work on ONE 128-byte column of the feature rows for neighbouring slices, whose
    // reuse of the lines its L2 serves. spw > 1 on lists of short rows: a slice of 9-edge rows is nine iterations, the
    // weight loads and the prologue of a workgroup want more than that to pay for.
    const int L = xcd_contiguous(blockIdx.x, gridDim.x);
    if (L >= groups * qTiles) return;
    const int qt = L / groups, g = L - qt * groups;
    const int q = qt * 4 + wave;
    const bool mine = q < a.nb;  // a wave beyond the last block still gathers and meets the barriers
    const RowStage st = {reinterpret_cast<f32x4*>(stageMem)};
    BlockWeights w;
    load_block_weights(a, mine ? q : 0, i4, w);
    const int prow = wave * 16 + (lane & 15);  // the row of the slice this lane fetches as a producer
    const int sEnd = min((g + 1) * spw, p.S);
    for (int slice = g * spw; slice < sEnd; ++slice) {
        const int off = p.sliceOff[slice];
        const int len = (p.sliceOff[slice + 1] - off) >> 6;
        if (len == 0) continue;  // slices beyond the list's last virtual row (the layout is sized by a bound)
        const int r = p.vrow[slice * 64 + lane];
        float acc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = 0.f;
        float4 rcN = p.rec[(size_t)off + lane];
        int jP = 0;  // producer: neighbour index of row `prow`, two iterations ahead of the consumer
        f32x4 s0, s1;
        // featIdx (optional): the feature rows lie in the order of the UNSORTED points, row j of the sorted list is
        // featIdx[j] there -- the layer reads them where they are instead of from a sorted copy (small levels only: the
        // gathers lose their spatial order)
        {
            const int j0 = p.other[(size_t)off + prow];
            stage_load<BF>(a.feats, featIdx ? featIdx[j0] : j0, a.Fin, qt * 32, lane, s0, s1);
        }
        if (len > 1) { jP = p.other[(size_t)off + 64 + prow]; if (featIdx) jP = featIdx[jP]; }
        stage_store<BF>(st, 0, wave, lane, s0, s1);
        __syncthreads();
        for (int it = 0; it < len; ++it) {
            const float4 rc = rcN;
            const bool more = it + 1 < len;
            if (more) {  // iteration it + 1: its lines are requested now and parked after this iteration's arithmetic
                stage_load<BF>(a.feats, jP, a.Fin, qt * 32, lane, s0, s1);
                rcN = p.rec[(size_t)off + (size_t)(it + 1) * 64 + lane];
                if (it + 2 < len) { jP = p.other[(size_t)off + (size_t)(it + 2) * 64 + prow]; if (featIdx) jP = featIdx[jP]; }
            }
            float f[8];
            stage_read<BF>(st, it & 1, wave, lane, f);
            float a1[8], a2[8], o[8];
            MCCNN_PHASE();
            mlp_block_regs(w, rc.x, rc.y, rc.z, a1, a2, o);
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_fmaf(f[n] * rc.w, o[n], acc[n]);
            if (more) stage_store<BF>(st, (it + 1) & 1, wave, lane, s0, s1);
            __syncthreads();  // also orders the next slice's first store behind this slice's last read
        }
        if (r >= 0 && mine) {
            const int code = p.vcode[slice * 64 + lane];
            if (code >= 0) {  // a piece of a cut row: its partial sums wait in the scratch row for rows_combine
                float4* dst = reinterpret_cast<float4*>(scratch + (size_t)code * a.outF + q * 8);
                dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            } else if (BF) {
                reinterpret_cast<uint4*>(out16 + (size_t)r * a.outF)[q] = f32x8_to_bf16(acc);
            } else {
                float4* dst = reinterpret_cast<float4*>(out + (size_t)r * a.outF + q * 8);
                dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        }
    }
}

// (A row-per-lane edge pass for combin layers with ONE input feature -- conv_f1.hip's A_i = sum_e s_e a2_e with layers 1
// and 2 in 26 VGPRs, the scalar s of an iteration handed to the four block waves through LDS -- was built and measured
// on the 100k room, commit bd91ab3: 1to64 forward 0.223 ms against 0.227 ms for the streaming kernels, 1to16 0.142 against
// 0.083 ms. With nb = 8 a slice yields only two workgroups: 3 126 workgroups of ~47 barrier-coupled iterations are 1.7
// rounds of the 1 792 resident ones, and an iteration is ~1 us of latency. Dropped; these layers keep conv_f1.hip.)
// ------------------------------------------------------------------------------------------------ depth-wise backward
// Transposed plan (rows = neighbour points j). A workgroup takes a group of `spw` consecutive slices and 4 consecutive MLP
// blocks (wave k: block q0 + k, so the four 32-byte pieces of a gathered out-gradient row are one 128-byte line shared by
// the workgroup). Per (slice, block) sweep: the block's 176 weight-gradient sums stay in VGPRs across all slices of the
// group; the 8 feature-gradient sums of the lane's own point are finished at the end of the slice and stored once.
// Math: spatial_conv.cu:563-680 (see conv_bwd_mfma in conv.hip for the same steps in the edge-major form).
// IDX (featIdx given) is a template parameter: compiled in, the extra row look-ups cost the sweep three more spilled
// registers even when the index is absent (2.39 -> 2.78 ms for dw256 on the room); only small levels use it.
template <int FEAT, bool IDX>
__global__ __launch_bounds__(256, 2) void dw_bwd_rows(ConvArgs a, RowPlan p, const float* __restrict__ outGrad,
                                                      float* __restrict__ featGrad, float* __restrict__ scratch,
                                                      float* __restrict__ partials, int spw, int groups,
                                                      const int* __restrict__ featIdx) {
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i4 = lane & 3;
    const int qTiles = (a.nb + 3) >> 2;
    // tile-major, XCD-contiguous (see dw_fwd_rows): the out-gradient rows a workgroup gathers are shared with the
    // neighbouring slice groups that run beside it on the same XCD
    const int L = xcd_contiguous(blockIdx.x, gridDim.x);
    if (L >= groups * qTiles) return;
    const int qt = L / groups, g = L - qt * groups;
    {   // this workgroup's four blocks only
        ConvArgs t = a;
        const int q0 = qt * 4;
        t.w1 = a.w1 + (size_t)q0 * 24; t.b1 = a.b1 + (size_t)q0 * 8;
        t.w2 = a.w2 + (size_t)q0 * 64; t.b2 = a.b2 + (size_t)q0 * 8;
        t.w3 = a.w3 + (size_t)q0 * 64; t.b3 = a.b3 + (size_t)q0 * 8;
        t.nb = min(4, a.nb - q0);
        stage_weights<MCCNN_WQ_BWD>(t, lds);
    }
    __syncthreads();
    const int q = qt * 4 + wave;
    if (q >= a.nb) return;
    const int qq = q;
    constexpr bool BF = FEAT == 4;
    const unsigned short* feats16 = reinterpret_cast<const unsigned short*>(a.feats);
    const unsigned short* og16 = reinterpret_cast<const unsigned short*>(outGrad);
    unsigned short* fg16 = reinterpret_cast<unsigned short*>(featGrad);

    float gw3[64], gb3[8], gw2[64], gb2[8], gw1[24], gb1[8];
#pragma unroll
    for (int k = 0; k < 64; ++k) { gw3[k] = 0.f; gw2[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { gb3[k] = 0.f; gb2[k] = 0.f; gb1[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 24; ++k) gw1[k] = 0.f;

    const int sEnd = min((g + 1) * spw, p.S);
    for (int slice = g * spw; slice < sEnd; ++slice) {
        const int off = p.sliceOff[slice];
        const int len = (p.sliceOff[slice + 1] - off) >> 6;
        if (len == 0) continue;  // slices beyond the list's last virtual row
        // (the row id is looked up AGAIN after the sweep: kept live across it, it -- and the addresses derived from it --
        // were spilled to scratch around every slice; likewise the lane id is read afresh per slice, so that nothing
        // derived from it has to survive from one slice to the next)
        const int lane = fresh_lane();
        int jr = max(p.vrow[slice * 64 + lane], 0);
        if (IDX) jr = featIdx[jr];  // features and their gradient in the order of the unsorted points (see dw_fwd_rows)
        // (jr dies with the feature load below: the row index of the gradient is looked up AGAIN at the end of the slice --
        // kept live across the sweep it cost five more spilled registers and 16 % of the kernel's time)
        // the lane's own feature row piece is constant over the slice: parked in LDS (two conflict-free float4 planes
        // per wave) instead of 8 VGPRs -- the 176 sums leave no room for it
        f32x4* fpark = reinterpret_cast<f32x4*>(lds + 4 * MCCNN_WQ_BWD) + wave * 128;
        {
            float ff0[8];
            if (BF) {
                const uint4 fu = reinterpret_cast<const uint4*>(feats16 + (size_t)jr * a.Fin)[qq];
                bf16x8_to_f32(fu, ff0);
            } else {
                const float4* fp = reinterpret_cast<const float4*>(a.feats + (size_t)jr * a.Fin + qq * 8);
                const float4 fa = fp[0], fb = fp[1];
                ff0[0] = fa.x; ff0[1] = fa.y; ff0[2] = fa.z; ff0[3] = fa.w; ff0[4] = fb.x; ff0[5] = fb.y; ff0[6] = fb.z; ff0[7] = fb.w;
            }
            fpark[lane] = (f32x4){ff0[0], ff0[1], ff0[2], ff0[3]};
            fpark[64 + lane] = (f32x4){ff0[4], ff0[5], ff0[6], ff0[7]};
        }
        float dF[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) dF[n] = 0.f;
        // (Staging the out-gradient rows through LDS like the forward's feature rows was measured and dropped: parked
        // right after the load the wait is exposed at the top of every iteration, parked late the 8 extra VGPRs spill
        // the sums -- 3.38 ms against 2.40 ms for dw256 on the room. Each wave gathers its own 32-byte piece. Also
        // measured and dropped: WAVE SPECIALISATION -- a producer wave per block with all five 8x8 products' weights in
        // 76 VGPRs (MLP, feature gradient, u / t3 / t4) handing 11 float4 per lane and iteration through LDS to a
        // consumer wave that only feeds the 176 sums, one barrier per iteration, double-buffered: parity-green, 199
        // VGPRs, no spills, but 3.40 ms -- the producer's chain of 38 dependent MFMA steps per iteration is a latency the
        // barrier makes every wave of the workgroup wait for, and two waves per SIMD leave nothing to fill it with.)
        float4 rcN = make_float4(0.f, 0.f, 0.f, 0.f);
        int iN = 0;
        if (len > 0) { rcN = p.rec[(size_t)off + lane]; iN = p.other[(size_t)off + lane]; }
        for (int it = 0; it < len; ++it) {
            const float4 rc = rcN;
            const int ci = iN;
            const float inv = rc.w;
            float g8[8];
#ifdef MCCNN_ROWS_BWD_NOGATHER  // timing ablation (wrong results): the out-gradient rows come from registers
            {
                const float gv = __int_as_float(0x3f000000 | (ci & 0xffff));
#pragma unroll
                for (int n = 0; n < 8; ++n) g8[n] = gv;
            }
#else
            if (BF) {
                const uint4 gu = reinterpret_cast<const uint4*>(og16 + (size_t)ci * a.outF)[q];
                bf16x8_to_f32(gu, g8);
            } else {
                const float4* gp = reinterpret_cast<const float4*>(outGrad + (size_t)ci * a.outF + q * 8);
                const float4 ga = gp[0], gb = gp[1];
                g8[0] = ga.x; g8[1] = ga.y; g8[2] = ga.z; g8[3] = ga.w; g8[4] = gb.x; g8[5] = gb.y; g8[6] = gb.z; g8[7] = gb.w;
            }
#endif
            if (it + 1 < len) {  // after this iteration's gather: vmcnt retires in order
                const size_t sn = (size_t)off + (size_t)(it + 1) * 64 + lane;
                rcN = p.rec[sn];
                iN = p.other[sn];
            }
            float a1[8], a2[8], o[8];
            bool p1[8], p2[8];
            int woff = wave * MCCNN_WQ_BWD;
            asm volatile("" : "+s"(woff));
            const float* wq = lds + woff;
            const f32x4* w4 = reinterpret_cast<const f32x4*>(wq);
            {
                float pre1[8], pre2[8];
                MCCNN_PHASE();
                mlp_block_mfma(wq, i4, rc.x, rc.y, rc.z, pre1, a1, pre2, a2, o);
#pragma unroll
                for (int k = 0; k < 8; ++k) { p1[k] = pre1[k] >= 0.0f; p2[k] = pre2[k] >= 0.0f; }
            }
            // feature gradient of the lane's point: og * o / (pdf K)      (spatial_conv.cu:400)
#pragma unroll
            for (int n = 0; n < 8; ++n) dF[n] = __builtin_fmaf(g8[n] * inv, o[n], dF[n]);
            float gf[8];
            {
                const f32x4 f0 = fpark[lane], f1 = fpark[64 + lane];
#pragma unroll
                for (int n = 0; n < 8; ++n) gf[n] = g8[n] * (n < 4 ? f0[n & 3] : f1[n & 3]);
            }
            // dW3 += u a2^T, db3 += u, u = g f / (pdf K)          (spatial_conv.cu:383-399)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const float u = gf[n] * inv;
#pragma unroll
                for (int k = 0; k < 8; ++k) gw3[n * 8 + k] = fmaf(u, a2[k], gw3[n * 8 + k]);
                gb3[n] += u;
            }
            // t3 = 1[pre2 >= 0] * W3^T (g f) / (pdf K)             (:403-414)
            float t3[8];
            MCCNN_PHASE();
            layer8<false>(w4 + 62, nullptr, i4, gf, t3);
            MCCNN_PHASE();
#pragma unroll
            for (int k = 0; k < 8; ++k) t3[k] = p2[k] ? t3[k] * inv : 0.f;
            // dW2 += t3 a1^T, db2 += t3                            (:419-425)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int l = 0; l < 8; ++l) gw2[k * 8 + l] = fmaf(t3[k], a1[l], gw2[k * 8 + l]);
                gb2[k] += t3[k];
            }
            // t4 = 1[pre1 >= 0] * W2^T t3                          (:428-434)
            float t4[8];
            size_t so = (size_t)off + (size_t)it * 64 + lane;
            asm volatile("" : "+v"(so));
            const float4 dl = p.rec[so];
            MCCNN_PHASE();
            layer8<false>(w4 + 46, nullptr, i4, t3, t4);
            MCCNN_PHASE();
            // dW1 += t4 delta^T, db1 += t4                         (:439-444)
            // (delta is read again here -- an L1 hit, requested before the t4 chain -- instead of staying live in three
            // VGPRs across the whole iteration; the address goes through an opaque register so the load is not merged
            // with the one at the top)
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const float v = p1[l] ? t4[l] : 0.f;
                gw1[l * 3] = fmaf(v, dl.x, gw1[l * 3]);
                gw1[l * 3 + 1] = fmaf(v, dl.y, gw1[l * 3 + 1]);
                gw1[l * 3 + 2] = fmaf(v, dl.z, gw1[l * 3 + 2]);
                gb1[l] += v;
            }
        }
        const int slot = slice * 64 + fresh_lane();  // (opaque: not merged with the look-up above the sweep)
        const int r = p.vrow[slot];
        const int code = p.vcode[slot];
        if (r >= 0 && code >= 0) {  // a piece of a cut row
            float4* dst = reinterpret_cast<float4*>(scratch + (size_t)code * a.Fin + q * 8);
            dst[0] = make_float4(dF[0], dF[1], dF[2], dF[3]);
            dst[1] = make_float4(dF[4], dF[5], dF[6], dF[7]);
        } else if (r >= 0) {
            const int ro = IDX ? featIdx[r] : r;
            if (BF) {
                reinterpret_cast<uint4*>(fg16 + (size_t)ro * a.Fin)[q] = f32x8_to_bf16(dF);
            } else {
                float4* dst = reinterpret_cast<float4*>(featGrad + (size_t)ro * a.Fin + q * 8);
                dst[0] = make_float4(dF[0], dF[1], dF[2], dF[3]);
                dst[1] = make_float4(dF[4], dF[5], dF[6], dF[7]);
            }
        }
    }
    // transposing wave reduction; partial row layout: w1[24] b1[8] w2[64] b2[8] w3[64] b3[8]
    {
        const int lane = fresh_lane();
        float r2 = wave_reduce64(gw2, lane);
        float r3 = wave_reduce64(gw3, lane);
        float misc[64];
#pragma unroll
        for (int k = 0; k < 24; ++k) misc[k] = gw1[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { misc[24 + k] = gb1[k]; misc[32 + k] = gb2[k]; misc[40 + k] = gb3[k]; }
#pragma unroll
        for (int k = 48; k < 64; ++k) misc[k] = 0.f;
        float rm = wave_reduce64(misc, lane);
        float* pq = partials + ((size_t)g * a.nb + q) * 176;
        pq[32 + lane] = r2;
        pq[104 + lane] = r3;
        if (lane < 32) pq[lane] = rm;                 // w1, b1
        else if (lane < 40) pq[96 + lane - 32] = rm;  // b2
        else if (lane < 48) pq[168 + lane - 40] = rm; // b3
    }
}

// (Measured and dropped, commits 69cbed3 and the one before this text: the same sweep at ONE wave per SIMD with the block's
// 76 weight operands in registers instead of LDS -- with two chunks of 64 edges per iteration, their MFMA chains and VALU
// phases issued side by side (256 VGPRs + 166 AGPRs): 3.90 ms; with one chunk per iteration (256 + 107): 3.31 ms; against
// 2.43-2.49 ms for dw256 on the room. A wave alone takes ~3 250 cycles per (64 edges, block) even without a single LDS
// read -- VALU instructions that consume MFMA results wait for them (12 instead of 8 cycles per instruction in
// tools/issue_probe) -- and only a second wave fills those waits; its LDS weight reads are the cheaper evil.)
// defined in conv.hip
void launch_reduce_partials(const float* partials, int rows, int nb, float* dw1, float* db1, float* dw2, float* db2,
                            float* dw3, float* db3, hipStream_t s);
int launch_edge_records(const ConvArgs& a, float4* rec, hipStream_t s);
int conv_fill_args(ConvArgs& a, const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                   const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                   const float* aabb_min, const float* aabb_max, const float* w1, const float* b1, const float* w2,
                   const float* b2, const float* w3, const float* b3, int n, int m, int e, int Fin, int Fout, int combin,
                   int batch_size, float radius, int scale_inv, int avg);

// (the same rule as plan_inline_records below)
static bool plan_inline_records_fwd(int rows, int e) { return plan_sizes(rows, e).small && e <= 262144; }
int launch_plan_small_batch(const PlanSmallBatch& pb, int count, hipStream_t s) {
    if (count <= 0) return 0;
    plan_small_batch<<<count, 1024, 0, s>>>(pb);
    MCCNN_LAUNCHED();
    return 0;
}
int launch_sell_fill_batch(const SellFillBatch& fb, int count, int transposed, hipStream_t s) {
    BatchBlocks bb;
    bb.count = count;
    int run = 0;
    for (int k = 0; k < count; ++k) { bb.first[k] = run; run += fb.it[k].S * fb.it[k].chunks; }
    for (int k = count; k <= MCCNN_BATCH_MAX; ++k) bb.first[k] = run;
    if (run == 0) return 0;
    if (transposed) sell_fill_batch<true><<<run, 256, 0, s>>>(fb, bb);
    else sell_fill_batch<false><<<run, 256, 0, s>>>(fb, bb);
    MCCNN_LAUNCHED();
    return 0;
}
// whether the plan of (rows, e) -- transposed over n points when `transposed` -- is one of the small ones the batch form takes:
// single-workgroup layout, records evaluated inline, single-workgroup transposition
bool transpose_small(int e, int n);
bool plan_batchable(int rows, int e, int transposed) {
    if (rows <= 0 || e <= 0) return false;
    const PlanSizes z = plan_sizes(rows, e);
    (void)transposed;   // (the transposition of a list too long for one workgroup is a batch chain of its own: conv.hip tr_chain_item)
    return z.small && plan_inline_records_fwd(rows, e);
}
// the items of one small plan into the three batches; `tws`: cnt [n] | slot [e] | tmp [e] of the transposition (transposed only)
size_t plan_batch_tr_ws_bytes(int n, int e) { return align_up((size_t)(n > 0 ? n : 1) * 4) + 2 * align_up((size_t)(e > 0 ? e : 1) * 4); }
int plan_batch_items(int transposed, const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs, const float* samples,
                     const int* start_idx, const int* packed, const float* aabb_min, const float* aabb_max, int n, int m, int e,
                     int batch_size, float radius, int scale_inv, int avg, const int* order, int* start_t, int* perm_t,
                     int tlist_ready, void* plan_buffer, void* tws, size_t tws_bytes, TrSmallItem* tr /* may stay unused */,
                     bool* use_tr, PlanSmallItem& lay, SellFillItem& fill) {
    const int rows = transposed ? n : m;
    if (!plan_batchable(rows, e, transposed)) return MCCNN_E_BADARG;
    long long off[6], total, cap, srows;
    int S;
    int rc = mccnn_rowplan_buffer(rows, e, off, &total, &S, &cap, &srows);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(plan_buffer);
    int* vrow = reinterpret_cast<int*>(base + off[0]);
    int* vcode = reinterpret_cast<int*>(base + off[1]);
    int* sliceOff = reinterpret_cast<int*>(base + off[2]);
    int* vposRow = reinterpret_cast<int*>(base + off[3]);
    int* other = reinterpret_cast<int*>(base + off[4]);
    float4* rec = reinterpret_cast<float4*>(base + off[5]);
    const PlanSizes z = plan_sizes(rows, e);
    *use_tr = false;
    if (transposed && !tlist_ready && tr) {   // (tr == nullptr: the caller transposes the list by a chain of its own, ahead of the layout)
        if (!tws || tws_bytes < plan_batch_tr_ws_bytes(n, e) || !start_t || !perm_t) return MCCNN_E_WORKSPACE;
        char* w = reinterpret_cast<char*>(tws);
        int* cnt = reinterpret_cast<int*>(w);
        int* slot = reinterpret_cast<int*>(w + align_up((size_t)n * 4));
        int* tmp = reinterpret_cast<int*>(w + align_up((size_t)n * 4) + align_up((size_t)e * 4));
        *tr = TrSmallItem{reinterpret_cast<const int2*>(packed), cnt, slot, tmp, start_t, perm_t, e, n};
        *use_tr = true;
    }
    const int* rowStart = transposed ? start_t : start_idx;
    lay = PlanSmallItem{rowStart, transposed ? nullptr : order, vrow, vcode, sliceOff, vposRow, rows, e, z.L, z.S};
    fill = SellFillItem{sorted_pts, sorted_batch_ids, pdfs, samples, start_idx, reinterpret_cast<const int2*>(packed), aabb_min, aabb_max,
                        rowStart, transposed ? perm_t : nullptr, vrow, vcode, sliceOff, vposRow, rec, other, z.slots,
                        n, m, e, batch_size, scale_inv, avg, rows, z.S, z.L, ceil_div(z.L, MCCNN_FILL_CHUNK), radius};
    return 0;
}

static int bwd_rows_spw(int S, int nb, int rows, int e) {
    // workgroups are dispatched as slots free up: >= ~12 rounds of the 2048 resident waves keep the tail below one
    // workgroup in twelve, while every extra slice per wave amortises the 176-value reduction at its end -- and on
    // lists of short rows (a slice of 9-edge rows is nine iterations) the set-up of a workgroup
    long long spw = ((long long)S * nb) / 24576;
    const long long avg = rows > 0 ? (e / rows > 0 ? e / rows : 1) : 1;
    if (spw > 8) spw = 8;
    if (spw * avg < 48 && (long long)S * nb >= 16384) spw = (48 + avg - 1) / avg;  // only when the list fills the chip anyway
    if (spw > 16) spw = 16;
    if (spw < 1) spw = 1;
    return (int)spw;
}
static int fwd_rows_spw(int S, int nb, int rows, int e) {
    const long long avg = rows > 0 ? (e / rows > 0 ? e / rows : 1) : 1;
    long long spw = 1;
    if (avg < 32 && (long long)S * nb >= 16384) spw = (32 + avg - 1) / avg;
    if (spw > 8) spw = 8;
    return (int)spw;
}

}  // namespace mccnn

using namespace mccnn;

extern "C" {

int mccnn_rowplan_sizes(int rows, int e, int* num_slices, long long* slot_capacity, long long* scratch_rows) {
    if (rows < 0 || e < 0 || !num_slices || !slot_capacity || !scratch_rows) return MCCNN_E_BADARG;
    const PlanSizes z = plan_sizes(rows, e);
    if (z.slots > 0x7fffffffLL || z.vcap > 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    *num_slices = z.S;
    *slot_capacity = z.slots;
    *scratch_rows = z.vcap;
    return 0;
}

// status words of the pieces' chain (vr_scan_expand): one per tile of 2 048 row positions + the ticket; never less than
// what a stand-alone prefix sum over the rows would take (the bound of earlier rounds)
static size_t layout_scan1_bytes(int rows) {
    const size_t chain = align_up(((size_t)ceil_div(rows > 0 ? rows : 1, 2048) + 1) * 8) + 256;
    const size_t scan = scan_workspace_bytes(rows > 0 ? rows : 1);
    return chain > scan ? chain : scan;
}
size_t mccnn_rowplan_workspace_bytes(int rows, int e) {
    if (rows <= 0) return 256;
    const PlanSizes z = plan_sizes(rows, e);
    return align_up((size_t)(rows + 1) * sizeof(int)) * 2 + align_up((size_t)z.vcap * sizeof(int)) +
           align_up((size_t)z.S * sizeof(int)) + layout_scan1_bytes(rows) + scan_workspace_bytes(z.S) + 512;
}

int mccnn_rowplan_layout(const int* row_start, int rows, int e, const int* order, int* plan_vrow, int* plan_vcode,
                         int* slice_off, int* vpos_row, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (rows < 0 || e < 0 || !slice_off) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) return launch_zero_words(slice_off, 1, s);
    if (!row_start || !plan_vrow || !plan_vcode || !vpos_row) return MCCNN_E_BADARG;
    if (!ws || ws_bytes < mccnn_rowplan_workspace_bytes(rows, e)) return MCCNN_E_WORKSPACE;
    const PlanSizes z = plan_sizes(rows, e);
    if (z.slots > 0x7fffffffLL || z.vcap > 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    if (z.small) {
        plan_small<<<1, 1024, 0, s>>>(row_start, rows, e, order, z.L, z.S, plan_vrow, plan_vcode, slice_off, vpos_row);
        MCCNN_LAUNCHED();
        return 0;
    }
    Arena ar(ws, ws_bytes);
    int* vcnt = ar.take<int>((size_t)rows + 1);
    int* vposP = ar.take<int>((size_t)rows + 1);
    int* vlistRow = ar.take<int>((size_t)z.vcap);
    int* sliceSlots = ar.take<int>((size_t)z.S);
    void* scan1 = ar.take<char>(layout_scan1_bytes(rows));
    void* scan2 = ar.take<char>(scan_workspace_bytes(z.S));
    if (!vcnt || !vposP || !vlistRow || !sliceSlots || !scan1 || !scan2) return MCCNN_E_WORKSPACE;
    // Three launches (round 5: six with a memset): vr_count (+ the status words of both chains), vr_scan_expand (prefix sum
    // of the pieces + expansion), sell_sort (+ the slices' offsets). The status areas are the scan workspaces' first bytes:
    // tiles + 1 words for the pieces' chain (2 048 positions per tile), windows + 1 for the slices' chain.
    const int tiles = ceil_div(rows, SCAN_TILE);
    if ((size_t)(tiles + 1) * 8 > layout_scan1_bytes(rows) || (size_t)(z.windows + 1) * 8 > align_up((size_t)z.S * 4)) return MCCNN_E_WORKSPACE;
    unsigned long long* st1 = reinterpret_cast<unsigned long long*>(scan1);
    // (the slices' chain needs windows + 1 words: S / 16 + 1 -- they fit the unused slice-length array + its scan workspace)
    unsigned long long* st2 = reinterpret_cast<unsigned long long*>(sliceSlots);
    vr_count<<<ceil_div(rows, 256), 256, 0, s>>>(row_start, rows, e, order, vcnt, z.L, clear_span(st1, (size_t)(tiles + 1) * 8),
                                                  clear_span(st2, (size_t)(z.windows + 1) * 8));
    MCCNN_LAUNCHED();
    vr_scan_expand<<<tiles, SCAN_THREADS, 0, s>>>(row_start, rows, e, order, vcnt, st1, vpos_row, vlistRow, vposP + rows);
    MCCNN_LAUNCHED();
    sell_sort<<<z.windows, 256, 0, s>>>(row_start, rows, e, vlistRow, vpos_row, vposP + rows, plan_vrow, plan_vcode, slice_off, z.L, st2);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_edge_records(const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs, const float* samples,
                       const int* start_idx, const int* packed, const float* aabb_min, const float* aabb_max, int n, int m,
                       int e, int batch_size, float radius, int scale_inv, int avg, void* rec_edges, mccnn_stream_t stream) {
    if (n < 0 || m < 0 || e < 0 || batch_size <= 0 || !(radius > 0.0f)) return MCCNN_E_BADARG;
    if (e == 0) return 0;
    if (!sorted_pts || !sorted_batch_ids || !pdfs || !samples || !start_idx || !packed || !aabb_min || !aabb_max || !rec_edges)
        return MCCNN_E_BADARG;
    ConvArgs a = {};
    a.pts = sorted_pts; a.bids = sorted_batch_ids; a.pdfs = pdfs; a.samples = samples; a.start = start_idx;
    a.packed = reinterpret_cast<const int2*>(packed); a.mn = aabb_min; a.mx = aabb_max;
    a.n = n; a.m = m; a.e = e; a.radius = radius; a.invRadius = 1.0f / radius; a.scaleInv = scale_inv; a.avg = avg;
    a.B = batch_size;
    return launch_edge_records(a, reinterpret_cast<float4*>(rec_edges), (hipStream_t)stream);
}

int mccnn_rowplan_fill(int transposed, const void* rec_edges, const int* packed, int rows, int e, const int* row_start,
                       const int* perm_t, const int* plan_vrow, const int* plan_vcode, const int* slice_off,
                       const int* vpos_row, void* rec, int* other, mccnn_stream_t stream) {
    if (rows < 0 || e < 0) return MCCNN_E_BADARG;
    if (rows == 0 || e == 0) return 0;
    if (!rec_edges || !packed || !row_start || !plan_vrow || !plan_vcode || !slice_off || !vpos_row || !rec || !other ||
        (transposed && !perm_t))
        return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const PlanSizes z = plan_sizes(rows, e);
    RowPlan p = {plan_vrow, plan_vcode, slice_off, vpos_row, nullptr, nullptr, rows, z.S};
    const dim3 grid(z.S, ceil_div(z.L, MCCNN_FILL_CHUNK));
    const ConvArgs none = {};
    if (transposed)
        sell_fill<true, false><<<grid, 256, 0, s>>>(reinterpret_cast<const float4*>(rec_edges), reinterpret_cast<const int2*>(packed), e,
                                                    row_start, rows, perm_t, p, z.slots, reinterpret_cast<float4*>(rec), other, z.L, none);
    else
        sell_fill<false, false><<<grid, 256, 0, s>>>(reinterpret_cast<const float4*>(rec_edges), reinterpret_cast<const int2*>(packed), e,
                                                     row_start, rows, nullptr, p, z.slots, reinterpret_cast<float4*>(rec), other, z.L, none);
    MCCNN_LAUNCHED();
    return 0;
}

// Plans of small lists evaluate their records inside the fill (no edge-order record array, one launch less per list).
static bool plan_inline_records(int rows, int e) { return plan_sizes(rows, e).small && e <= 262144; }
int mccnn_rowplan_inline_records(int rows, int e) { return (rows > 0 && e > 0 && plan_inline_records(rows, e)) ? 1 : 0; }

static size_t plan_al(size_t ints) { return (ints + 63) / 64 * 64; }  // 256-byte aligned pieces

int mccnn_rowplan_buffer(int rows, int e, long long offsets[6], long long* total_bytes, int* num_slices,
                         long long* slot_capacity, long long* scratch_rows) {
    if (rows < 0 || e < 0 || !offsets || !total_bytes || !num_slices || !slot_capacity || !scratch_rows) return MCCNN_E_BADARG;
    const PlanSizes z = plan_sizes(rows, e);
    if (z.slots > 0x7fffffffLL || z.vcap > 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    *num_slices = z.S;
    *slot_capacity = z.slots;
    *scratch_rows = z.vcap;
    size_t o = 0;
    offsets[0] = (long long)(o * 4); o += plan_al((size_t)64 * z.S);            // plan_vrow
    offsets[1] = (long long)(o * 4); o += plan_al((size_t)64 * z.S);            // plan_vcode
    offsets[2] = (long long)(o * 4); o += plan_al((size_t)z.S + 1);             // slice_off
    offsets[3] = (long long)(o * 4); o += plan_al((size_t)(rows > 0 ? rows : 1));  // vpos_row
    offsets[4] = (long long)(o * 4); o += plan_al((size_t)(z.slots > 0 ? z.slots : 1));  // plan_other
    offsets[5] = (long long)(o * 4); o += plan_al((size_t)(z.slots > 0 ? z.slots : 1) * 4);  // plan_rec (16 B per slot)
    *total_bytes = (long long)(o * 4);
    return 0;
}

// Sizes that hold for EVERY edge count up to e_cap (the capacity a neighbour list was guessed at): a caller that allocates
// a plan before the list's true size is known -- ahead of the build, on another thread -- takes these. The longest
// virtual row L is a step function of (rows, e) and the layout's arrays are monotone in e for a fixed L: the maximum over
// the possible L at e_cap bounds them all.
int mccnn_rowplan_bound(int rows, int e_cap, int transposed, long long* buffer_bytes, long long* ws_bytes) {
    if (rows < 0 || e_cap < 0 || !buffer_bytes || !ws_bytes) return MCCNN_E_BADARG;
    size_t buf = 0, ws = 256;
    for (int L = 1; L <= ROWS_L; L <<= 1) {
        const long long vcap = (long long)rows + e_cap / L;
        const long long windows = (vcap + SELL_SIGMA - 1) / SELL_SIGMA;
        const long long S = windows * (SELL_SIGMA / 64);
        long long slots = (long long)e_cap + 64LL * L * windows;
        const long long small = (long long)L * (vcap + 64);
        if (rows <= plan_small_limit() && small > slots) slots = small;
        if (slots > 0x7fffffffLL || vcap > 0x7fffffffLL) return MCCNN_E_TOOLARGE;
        size_t o = 0;
        o += plan_al((size_t)64 * S) * 2 + plan_al((size_t)S + 1) + plan_al((size_t)(rows > 0 ? rows : 1));
        o += plan_al((size_t)(slots > 0 ? slots : 1)) + plan_al((size_t)(slots > 0 ? slots : 1) * 4);
        if (o * 4 > buf) buf = o * 4;
        const size_t w = align_up((size_t)(rows + 1) * sizeof(int)) * 2 + align_up((size_t)vcap * sizeof(int)) +
                         align_up((size_t)S * sizeof(int)) + layout_scan1_bytes(rows) + scan_workspace_bytes((int)S) + 512 +
                         align_up((size_t)vcap * sizeof(int2));
        if (w > ws) ws = w;
    }
    if (transposed) ws += mccnn_transpose_neighbors_workspace_bytes(rows, e_cap);  // (see build_ws_bytes)
    *buffer_bytes = (long long)buf;
    *ws_bytes = (long long)ws;
    return 0;
}

// Workspace of a plan build. Small plans: the layout and the transposition run one after the other over the same bytes.
// Large plans: [layout | vinfo (8 bytes per virtual row) | transposition] -- the layout sits between the two halves of
// the transposition (it needs the row lengths only), so everything is alive at once.
static size_t build_ws_bytes(int rows, int e, int transposed, const PlanSizes& z) {
    const size_t a = mccnn_rowplan_workspace_bytes(rows, e);
    const size_t b = transposed ? mccnn_transpose_neighbors_workspace_bytes(rows, e) : 0;
    if (z.small || rows <= 0 || e <= 0) return a > b ? a : b;
    return a + align_up((size_t)z.vcap * sizeof(int2)) + b;
}
size_t mccnn_rowplan_build_workspace_bytes(int rows, int e, int transposed) {
    return build_ws_bytes(rows, e, transposed, plan_sizes(rows > 0 ? rows : 0, e > 0 ? e : 0));
}

}  // extern "C"

namespace mccnn {
bool transpose_small(int e, int n);
size_t transpose_count_bytes(int n);
int transpose_count(const int2* pk, int e, int n, int* start_t, char* blk, int* slot, hipStream_t s);
int transpose_fill(const int2* pk, int e, const int* start_t, const int* slot, int* tmp, hipStream_t s);
}
// A large plan: layout, then ONE scattering pass in the order its source is read (see plan_scatter_*). For a transposed
// plan whose list does not exist yet the transposition is part of the build: count -> layout -> fill -> rank + scatter.
static int build_large(int transposed, const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs, const float* samples,
                       const int* start_idx, const int* packed, const float* aabb_min, const float* aabb_max, int n, int m, int e,
                       int batch_size, float radius, int scale_inv, int avg, const int* order, void* rec_edges, int rec_ready,
                       int* start_t, int* perm_t, int tlist_ready, void* plan_buffer, void* ws, size_t ws_bytes, hipStream_t s) {
    const int rows = transposed ? n : m;
    // the geometry (points, densities, samples, boxes, radius) is only read where records are EVALUATED: with records
    // ready (rec_ready) both plans are pure permutations of them, as the non-inline path always accepted
    if (!rec_edges || !start_idx || !packed) return MCCNN_E_BADARG;
    if (!rec_ready && (batch_size <= 0 || !(radius > 0.0f) || !sorted_pts || !sorted_batch_ids || !pdfs || !samples || !aabb_min || !aabb_max))
        return MCCNN_E_BADARG;
    long long off[6], total, cap, srows;
    int S;
    int rc = mccnn_rowplan_buffer(rows, e, off, &total, &S, &cap, &srows);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(plan_buffer);
    int* vrow = reinterpret_cast<int*>(base + off[0]);
    int* vcode = reinterpret_cast<int*>(base + off[1]);
    int* sliceOff = reinterpret_cast<int*>(base + off[2]);
    int* vposRow = reinterpret_cast<int*>(base + off[3]);
    int* other = reinterpret_cast<int*>(base + off[4]);
    float4* rec = reinterpret_cast<float4*>(base + off[5]);
    const PlanSizes z = plan_sizes(rows, e);
    const size_t lay = mccnn_rowplan_workspace_bytes(rows, e);
    Arena ar(ws, ws_bytes);
    char* layWs = ar.take<char>(lay);
    int2* vinfo = ar.take<int2>((size_t)z.vcap);
    if (!layWs || !vinfo) return MCCNN_E_WORKSPACE;
    ConvArgs a = {};
    a.pts = sorted_pts; a.bids = sorted_batch_ids; a.pdfs = pdfs; a.samples = samples; a.start = start_idx;
    a.packed = reinterpret_cast<const int2*>(packed); a.mn = aabb_min; a.mx = aabb_max;
    a.n = n; a.m = m; a.e = e; a.radius = radius; a.invRadius = radius > 0.0f ? 1.0f / radius : 0.0f; a.scaleInv = scale_inv; a.avg = avg;
    a.B = batch_size;
    RowPlan p = {vrow, vcode, sliceOff, vposRow, nullptr, nullptr, rows, z.S};
    const int eb = ceil_div(e, 256);
    float4* recE = reinterpret_cast<float4*>(rec_edges);
    if (!transposed) {
        rc = mccnn_rowplan_layout(start_idx, rows, e, order, vrow, vcode, sliceOff, vposRow, layWs, lay, s);
        if (rc) return rc;
        const int grid = z.S * ceil_div(z.L, MCCNN_FILL_CHUNK);
        if (rec_ready) plan_fill_tiles<false><<<grid, 256, 0, s>>>(a, recE, nullptr, start_idx, rows, p, z.slots, rec, other, z.L);
        else plan_fill_tiles<true><<<grid, 256, 0, s>>>(a, nullptr, recE, start_idx, rows, p, z.slots, rec, other, z.L);
        MCCNN_LAUNCHED();
        return 0;
    }
    int *slot = nullptr, *tmp = nullptr;
    const bool fused = !tlist_ready && !transpose_small(e, n);
    if (fused) {
        char* blk = ar.take<char>(transpose_count_bytes(n));
        slot = ar.take<int>((size_t)e);
        tmp = ar.take<int>((size_t)e);
        if (!blk || !slot || !tmp) return MCCNN_E_WORKSPACE;
        rc = transpose_count(a.packed, e, n, start_t, blk, slot, s);
        if (rc) return rc;
    } else if (!tlist_ready) {
        char* tws = ar.take<char>(mccnn_transpose_neighbors_workspace_bytes(n, e));
        if (!tws) return MCCNN_E_WORKSPACE;
        rc = mccnn_transpose_neighbors(packed, e, n, start_t, perm_t, tws, mccnn_transpose_neighbors_workspace_bytes(n, e), s);
        if (rc) return rc;
    }
    rc = mccnn_rowplan_layout(start_t, rows, e, nullptr, vrow, vcode, sliceOff, vposRow, layWs, lay, s);
    if (rc) return rc;
    if (fused) {
        rc = transpose_fill(a.packed, e, start_t, slot, tmp, s);
        if (rc) return rc;
    }
    plan_bases<<<ceil_div((long long)z.S * 64, 256), 256, 0, s>>>(p, start_t, rows, e, z.L, z.slots, vinfo, rec, other);
    MCCNN_LAUNCHED();
    if (!rec_ready) {
        rc = launch_edge_records(a, recE, s);
        if (rc) return rc;
    }
    if (fused) plan_scatter_tr<true><<<eb, 256, 0, s>>>(recE, a.packed, e, n, start_t, tmp, perm_t, vposRow, vinfo, z.L, rec, other);
    else plan_scatter_tr<false><<<eb, 256, 0, s>>>(recE, a.packed, e, n, start_t, nullptr, perm_t, vposRow, vinfo, z.L, rec, other);
    MCCNN_LAUNCHED();
    return 0;
}

// ---- the same chain for a BATCH of large plans: items + one launch per phase (exec.hip mccnn_geometry_prebuild_batch)
namespace mccnn {
bool plan_large_batchable(int rows, int e, int n, int transposed, int tlist_ready) {
    if (rows <= 0 || e <= 0) return false;
    const PlanSizes z = plan_sizes(rows, e);
    if (z.small || z.slots > 0x7fffffffLL || z.vcap > 0x7fffffffLL) return false;
    // (a list short enough for the single-workgroup transposition under a large plan: left to the single chain)
    if (transposed && !tlist_ready && (transpose_small(e, n) || (long long)n > 2048LL * 1024)) return false;
    return true;
}
size_t plan_large_ws_bytes(int rows, int e, int n, int transposed, int tlist_ready) {
    const PlanSizes z = plan_sizes(rows, e);
    size_t b = align_up(mccnn_rowplan_workspace_bytes(rows, e)) + align_up((size_t)z.vcap * sizeof(int2));
    if (transposed && !tlist_ready) b += align_up(mccnn_transpose_neighbors_workspace_bytes(n, e));
    return b + 256;
}
int plan_large_item(int transposed, const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs, const float* samples,
                    const int* start_idx, const int* packed, const float* aabb_min, const float* aabb_max, int n, int m, int e,
                    int batch_size, float radius, int scale_inv, int avg, const int* order, void* rec_edges, int rec_ready, int* start_t,
                    int* perm_t, int tlist_ready, void* plan_buffer, void* ws, size_t ws_bytes, LargeItem& it, TrChainItem* tc,
                    ScanItem* sc, bool* use_chain, SpanBatch& spans) {
    const int rows = transposed ? n : m;
    *use_chain = false;
    if (!plan_large_batchable(rows, e, n, transposed, tlist_ready)) return MCCNN_E_BADARG;
    if (!rec_edges || !start_idx || !packed || !plan_buffer) return MCCNN_E_BADARG;
    if (!rec_ready && (batch_size <= 0 || !(radius > 0.0f) || !sorted_pts || !sorted_batch_ids || !pdfs || !samples || !aabb_min || !aabb_max))
        return MCCNN_E_BADARG;
    if (transposed && (!start_t || !perm_t)) return MCCNN_E_BADARG;
    if (!ws || ws_bytes < plan_large_ws_bytes(rows, e, n, transposed, tlist_ready)) return MCCNN_E_WORKSPACE;
    if (spans.count + 3 > (int)(sizeof(spans.sp) / sizeof(spans.sp[0]))) return MCCNN_E_WORKSPACE;
    long long off[6], total, cap, srows;
    int S;
    int rc = mccnn_rowplan_buffer(rows, e, off, &total, &S, &cap, &srows);
    if (rc) return rc;
    const PlanSizes z = plan_sizes(rows, e);
    char* base = reinterpret_cast<char*>(plan_buffer);
    Arena ar(ws, ws_bytes);
    // the layout's workspace, cut as mccnn_rowplan_layout cuts it
    int* vcnt = ar.take<int>((size_t)rows + 1);
    int* vposP = ar.take<int>((size_t)rows + 1);
    int* vlistRow = ar.take<int>((size_t)z.vcap);
    int* sliceSlots = ar.take<int>((size_t)z.S);
    void* scan1 = ar.take<char>(layout_scan1_bytes(rows));
    int2* vinfo = ar.take<int2>((size_t)z.vcap);
    if (!vcnt || !vposP || !vlistRow || !sliceSlots || !scan1 || !vinfo) return MCCNN_E_WORKSPACE;
    const int tiles = ceil_div(rows, SCAN_TILE);
    if ((size_t)(tiles + 1) * 8 > layout_scan1_bytes(rows) || (size_t)(z.windows + 1) * 8 > align_up((size_t)z.S * 4)) return MCCNN_E_WORKSPACE;
    it = LargeItem{};
    it.rowStart = transposed ? start_t : start_idx;
    it.order = transposed ? nullptr : order;
    it.vcnt = vcnt; it.st1 = reinterpret_cast<unsigned long long*>(scan1); it.st2 = reinterpret_cast<unsigned long long*>(sliceSlots);
    it.vlistRow = vlistRow; it.vTotal = vposP + rows;
    it.vrow = reinterpret_cast<int*>(base + off[0]); it.vcode = reinterpret_cast<int*>(base + off[1]);
    it.sliceOff = reinterpret_cast<int*>(base + off[2]); it.vposRow = reinterpret_cast<int*>(base + off[3]);
    it.oth = reinterpret_cast<int*>(base + off[4]); it.rec = reinterpret_cast<float4*>(base + off[5]);
    it.cap = z.slots;
    it.pts = sorted_pts; it.bids = sorted_batch_ids; it.pdfs = pdfs; it.samples = samples; it.start = start_idx;
    it.packed = reinterpret_cast<const int2*>(packed); it.mn = aabb_min; it.mx = aabb_max;
    it.recE = reinterpret_cast<float4*>(rec_edges);
    it.vinfo = vinfo; it.startT = start_t; it.tmp = nullptr; it.permT = perm_t;
    it.rows = rows; it.e = e; it.n = n; it.m = m; it.B = batch_size; it.L = z.L; it.S = z.S; it.windows = z.windows; it.tiles = tiles;
    it.scaleInv = scale_inv; it.avg = avg; it.tr = transposed ? 1 : 0; it.eval = rec_ready ? 0 : 1; it.rank = 0;
    it.radius = radius;
    spans.sp[spans.count++] = clear_span(it.st1, (size_t)(tiles + 1) * 8);
    spans.sp[spans.count++] = clear_span(it.st2, (size_t)(z.windows + 1) * 8);
    if (transposed && !tlist_ready) {   // the transposition is part of the build: count -> scan -> [layout] -> fill -> rank in the scatter
        const size_t wb = mccnn_transpose_neighbors_workspace_bytes(n, e);
        char* tws = ar.take<char>(wb);
        if (!tws || !tc || !sc) return MCCNN_E_WORKSPACE;
        ClearSpan head;
        rc = tr_chain_item(*tc, *sc, head, packed, e, n, start_t, perm_t, tws, wb);
        if (rc) return rc;
        tc->norank = 1;
        spans.sp[spans.count++] = head;
        it.tmp = tc->tmp;
        it.rank = 1;
        *use_chain = true;
    }
    return 0;
}
int launch_plan_large_batch(const LargeBatch& lb, int count, int phase, hipStream_t s) {
    BatchBlocks bb;
    bb.count = count;
    int run = 0;
    for (int k = 0; k < count; ++k) {
        const LargeItem& t = lb.it[k];
        bb.first[k] = run;
        int blocks = 0;
        switch (phase) {
            case LARGE_VR_COUNT: blocks = ceil_div(t.rows, 256); break;
            case LARGE_VR_SCAN: blocks = t.tiles; break;
            case LARGE_SELL_SORT: blocks = t.windows; break;
            case LARGE_BASES: blocks = t.tr ? (int)ceil_div((long long)t.S * 64, 256) : 0; break;
            case LARGE_RECORDS: blocks = (t.tr && t.eval) ? ceil_div(t.e, 256) : 0; break;
            case LARGE_FILL: blocks = t.tr ? 0 : t.S * ceil_div(t.L, MCCNN_FILL_CHUNK); break;
            case LARGE_SCATTER: blocks = t.tr ? ceil_div(t.e, 256) : 0; break;
        }
        run += blocks;
    }
    for (int k = count; k <= MCCNN_BATCH_MAX; ++k) bb.first[k] = run;
    if (run == 0) return 0;
    switch (phase) {
        case LARGE_VR_COUNT: vr_count_batch<<<run, 256, 0, s>>>(lb, bb); break;
        case LARGE_VR_SCAN: vr_scan_expand_batch<<<run, SCAN_THREADS, 0, s>>>(lb, bb); break;
        case LARGE_SELL_SORT: sell_sort_batch<<<run, 256, 0, s>>>(lb, bb); break;
        case LARGE_BASES: plan_bases_batch<<<run, 256, 0, s>>>(lb, bb); break;
        case LARGE_RECORDS: edge_records_batch<<<run, 256, 0, s>>>(lb, bb); break;
        case LARGE_FILL: plan_fill_tiles_batch<<<run, 256, 0, s>>>(lb, bb); break;
        case LARGE_SCATTER: plan_scatter_tr_batch<<<run, 256, 0, s>>>(lb, bb); break;
        default: return MCCNN_E_BADARG;
    }
    MCCNN_LAUNCHED();
    return 0;
}
}  // namespace mccnn

extern "C" {

int mccnn_rowplan_build(int transposed, const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs,
                        const float* samples, const int* start_idx, const int* packed, const float* aabb_min,
                        const float* aabb_max, int n, int m, int e, int batch_size, float radius, int scale_inv, int avg,
                        const int* order, void* rec_edges, int rec_ready, int* start_t, int* perm_t, int tlist_ready,
                        void* plan_buffer, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (n < 0 || m < 0 || e < 0 || !plan_buffer) return MCCNN_E_BADARG;
    const int rows = transposed ? n : m;
    const bool inl = rows > 0 && e > 0 && plan_inline_records(rows, e);
    if (!inl && !rec_edges) return MCCNN_E_BADARG;
    if (transposed && (!start_t || !perm_t)) return MCCNN_E_BADARG;
    if (!ws || ws_bytes < mccnn_rowplan_build_workspace_bytes(rows, e, transposed)) return MCCNN_E_WORKSPACE;
    if (rows > 0 && e > 0 && !plan_sizes(rows, e).small)
        return build_large(transposed, sorted_pts, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min, aabb_max, n, m, e,
                           batch_size, radius, scale_inv, avg, order, rec_edges, rec_ready, start_t, perm_t, tlist_ready,
                           plan_buffer, ws, ws_bytes, (hipStream_t)stream);
    long long off[6], total, cap, srows;
    int S;
    int rc = mccnn_rowplan_buffer(rows, e, off, &total, &S, &cap, &srows);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(plan_buffer);
    int* vrow = reinterpret_cast<int*>(base + off[0]);
    int* vcode = reinterpret_cast<int*>(base + off[1]);
    int* sliceOff = reinterpret_cast<int*>(base + off[2]);
    int* vposRow = reinterpret_cast<int*>(base + off[3]);
    int* other = reinterpret_cast<int*>(base + off[4]);
    void* rec = base + off[5];
    if (transposed && !tlist_ready) {
        rc = mccnn_transpose_neighbors(packed, e, n, start_t, perm_t, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    const int* rowStart = transposed ? start_t : start_idx;
    rc = mccnn_rowplan_layout(rowStart, rows, e, transposed ? nullptr : order, vrow, vcode, sliceOff, vposRow, ws, ws_bytes, stream);
    if (rc) return rc;
    if (inl) {
        if (batch_size <= 0 || !(radius > 0.0f) || !sorted_pts || !sorted_batch_ids || !pdfs || !samples || !start_idx || !packed ||
            !aabb_min || !aabb_max)
            return MCCNN_E_BADARG;
        ConvArgs a = {};
        a.pts = sorted_pts; a.bids = sorted_batch_ids; a.pdfs = pdfs; a.samples = samples; a.start = start_idx;
        a.packed = reinterpret_cast<const int2*>(packed); a.mn = aabb_min; a.mx = aabb_max;
        a.n = n; a.m = m; a.e = e; a.radius = radius; a.invRadius = 1.0f / radius; a.scaleInv = scale_inv; a.avg = avg;
        a.B = batch_size;
        hipStream_t s = (hipStream_t)stream;
        const PlanSizes z = plan_sizes(rows, e);
        RowPlan p = {vrow, vcode, sliceOff, vposRow, nullptr, nullptr, rows, z.S};
        const dim3 grid(z.S, ceil_div(z.L, MCCNN_FILL_CHUNK));
        if (transposed)
            sell_fill<true, true><<<grid, 256, 0, s>>>(nullptr, a.packed, e, rowStart, rows, perm_t, p, z.slots,
                                                       reinterpret_cast<float4*>(rec), other, z.L, a);
        else
            sell_fill<false, true><<<grid, 256, 0, s>>>(nullptr, a.packed, e, rowStart, rows, nullptr, p, z.slots,
                                                        reinterpret_cast<float4*>(rec), other, z.L, a);
        MCCNN_LAUNCHED();
        return 0;
    }
    if (!rec_ready) {
        rc = mccnn_edge_records(sorted_pts, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min, aabb_max, n, m, e,
                                batch_size, radius, scale_inv, avg, rec_edges, stream);
        if (rc) return rc;
    }
    return mccnn_rowplan_fill(transposed, rec_edges, packed, rows, e, rowStart, perm_t, vrow, vcode, sliceOff, vposRow, rec, other,
                              stream);
}

static bool rows_shape_ok(const ConvArgs& a, int combin, const void* p0, const void* p1) {
    return !combin && a.Fin % 8 == 0 && ((((uintptr_t)p0 | (uintptr_t)p1) & 15) == 0);
}

int mccnn_spatial_conv_fwd_rows(const float* sorted_pts, const void* sorted_feats, const int* sorted_batch_ids,
                                const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                                const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                                const float* w2, const float* b2, const float* w3, const float* b3, int n, int m, int e,
                                int num_feats, int batch_size, float radius, int scale_inv, int avg, int bf16,
                                const int* plan_vrow, const int* plan_vcode, const int* slice_off, const int* vpos_row,
                                const void* plan_rec, const int* plan_other, void* out, float* scratch,
                                const int* feat_index, mccnn_stream_t stream) {
    ConvArgs a;
    int rc = conv_fill_args(a, sorted_pts, (const float*)sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed,
                            aabb_min, aabb_max, w1, b1, w2, b2, w3, b3, n, m, e, num_feats, num_feats, 0, batch_size, radius,
                            scale_inv, avg);
    if (rc) return rc;
    if (m == 0) return 0;
    if (e == 0) return MCCNN_E_BADARG;  // empty lists take mccnn_spatial_conv_fwd
    if (!out || !scratch || !plan_vrow || !plan_vcode || !slice_off || !vpos_row || !plan_rec || !plan_other) return MCCNN_E_BADARG;
    if (!rows_shape_ok(a, 0, sorted_feats, out)) return MCCNN_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const PlanSizes z = plan_sizes(m, e);
    RowPlan p = {plan_vrow, plan_vcode, slice_off, vpos_row, reinterpret_cast<const float4*>(plan_rec), plan_other, m, z.S};
    const int qTiles = (a.nb + 3) / 4;
    const int spw = fwd_rows_spw(p.S, a.nb, m, e);
    const int groups = (p.S + spw - 1) / spw;
    const long long blocks = ((long long)groups * qTiles + 7) / 8 * 8;
    if (blocks > 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    if (bf16) dw_fwd_rows<4><<<(int)blocks, 256, 0, s>>>(a, p, (float*)out, scratch, qTiles, spw, groups, feat_index);
    else dw_fwd_rows<2><<<(int)blocks, 256, 0, s>>>(a, p, (float*)out, scratch, qTiles, spw, groups, feat_index);
    MCCNN_LAUNCHED();
    if (bf16) launch_combine<true>(start_idx, m, e, vpos_row, scratch, a.outF, out, z.L, s);
    else launch_combine<false>(start_idx, m, e, vpos_row, scratch, a.outF, out, z.L, s);
    MCCNN_LAUNCHED();
    return 0;
}

size_t mccnn_spatial_conv_bwd_rows_workspace_bytes(int n, int e, int num_feats) {
    if (n <= 0 || num_feats <= 0) return 256;
    const PlanSizes z = plan_sizes(n, e);
    const int nb = (num_feats + 7) / 8;
    const int spw = bwd_rows_spw(z.S, nb, n, e);
    const long long groups = (z.S + spw - 1) / spw;
    return align_up((size_t)groups * nb * 176 * sizeof(float)) + 256;
}

int mccnn_spatial_conv_bwd_rows(const float* sorted_pts, const void* sorted_feats, const int* sorted_batch_ids,
                                const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                                const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                                const float* w2, const float* b2, const float* w3, const float* b3, const void* out_grad,
                                int n, int m, int e, int num_feats, int batch_size, float radius, int scale_inv, int avg,
                                int bf16, const int* start_t, const int* plan_vrow, const int* plan_vcode,
                                const int* slice_off, const int* vpos_row, const void* plan_rec, const int* plan_other,
                                void* feat_grad, float* scratch, float* dw1, float* db1, float* dw2, float* db2, float* dw3,
                                float* db3, const int* feat_index, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    ConvArgs a;
    int rc = conv_fill_args(a, sorted_pts, (const float*)sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed,
                            aabb_min, aabb_max, w1, b1, w2, b2, w3, b3, n, m, e, num_feats, num_feats, 0, batch_size, radius,
                            scale_inv, avg);
    if (rc) return rc;
    if (!dw1 || !db1 || !dw2 || !db2 || !dw3 || !db3 || (n > 0 && !feat_grad)) return MCCNN_E_BADARG;
    if (n == 0 || m == 0 || e == 0) return MCCNN_E_BADARG;  // empty lists take mccnn_spatial_conv_bwd
    if (!out_grad || !scratch || !start_t || !plan_vrow || !plan_vcode || !slice_off || !vpos_row || !plan_rec || !plan_other)
        return MCCNN_E_BADARG;
    if (!rows_shape_ok(a, 0, sorted_feats, out_grad) || (((uintptr_t)feat_grad) & 15)) return MCCNN_E_SHAPE;
    if (!ws || ws_bytes < mccnn_spatial_conv_bwd_rows_workspace_bytes(n, e, num_feats)) return MCCNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const PlanSizes z = plan_sizes(n, e);
    RowPlan p = {plan_vrow, plan_vcode, slice_off, vpos_row, reinterpret_cast<const float4*>(plan_rec), plan_other, n, z.S};
    const int spw = bwd_rows_spw(p.S, a.nb, n, e);
    const int groups = (p.S + spw - 1) / spw;
    const int qTiles = (a.nb + 3) / 4;
    const long long blocks = ((long long)groups * qTiles + 7) / 8 * 8;
    if (blocks > 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    float* partials = reinterpret_cast<float*>(ws);
    const size_t lds = ((size_t)4 * MCCNN_WQ_BWD + 4 * 512) * sizeof(float);  // 4 blocks of weights + the parked feature pieces
    if (feat_index) {
        if (bf16) dw_bwd_rows<4, true><<<(int)blocks, 256, lds, s>>>(a, p, (const float*)out_grad, (float*)feat_grad, scratch, partials, spw, groups, feat_index);
        else dw_bwd_rows<2, true><<<(int)blocks, 256, lds, s>>>(a, p, (const float*)out_grad, (float*)feat_grad, scratch, partials, spw, groups, feat_index);
    } else if (bf16) dw_bwd_rows<4, false><<<(int)blocks, 256, lds, s>>>(a, p, (const float*)out_grad, (float*)feat_grad, scratch, partials, spw, groups, nullptr);
    else dw_bwd_rows<2, false><<<(int)blocks, 256, lds, s>>>(a, p, (const float*)out_grad, (float*)feat_grad, scratch, partials, spw, groups, nullptr);
    MCCNN_LAUNCHED();
    if (bf16) launch_combine_reduce<true>(start_t, n, e, vpos_row, scratch, a.Fin, feat_grad, z.L, s, feat_index, partials, groups, a.nb,
                                          dw1, db1, dw2, db2, dw3, db3);
    else launch_combine_reduce<false>(start_t, n, e, vpos_row, scratch, a.Fin, feat_grad, z.L, s, feat_index, partials, groups, a.nb,
                                      dw1, db1, dw2, db2, dw3, db3);
    MCCNN_LAUNCHED();
    return 0;
}

}  // extern "C"
