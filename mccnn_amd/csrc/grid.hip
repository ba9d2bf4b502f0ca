// Bounding boxes, grid-hash keys, stable counting sort into cells, cell table, row
// permutations. Replaces tf_ops/aabb_gpu.cu and tf_ops/sort_gpu.cu.
#include "batch.h"

namespace mccnn {

// ------------------------------------------------------------------ AABB (aabb_gpu.cu:57-140)
// Floats are mapped to order-preserving uint32 so that plain integer atomics give min/max
// (the reference hand-rolled CAS loops, aabb_gpu.cu:23-45).
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void aabb_init(unsigned* __restrict__ enc, int B) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 3 * B) {
        enc[t] = f2ord(FLT_MAX);            // running min
        enc[3 * B + t] = f2ord(-FLT_MAX);   // running max
    }
}

__global__ __launch_bounds__(256) void aabb_reduce(const float* __restrict__ pts, const int* __restrict__ bids,
                                                   int n, int B, unsigned* __restrict__ enc) {
    // Clouds are stored contiguously (utils/DataSet.py:816-823), so a wave -- and usually a whole workgroup -- sees
    // one batch id: reduce in-wave with shuffles, combine the 4 waves in LDS and issue 6 global atomics per
    // WORKGROUP (a single 100k-point cloud otherwise hammers the same 6 addresses from 1500 waves).
    __shared__ unsigned sh[4][7];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    bool act = i < n;
    int b = act ? bids[i] : -1;
    float x = 0.f, y = 0.f, z = 0.f;
    if (act) {
        x = pts[(size_t)i * 3];
        y = pts[(size_t)i * 3 + 1];
        z = pts[(size_t)i * 3 + 2];
    }
    int b0 = __shfl(b, 0, 64);
    bool uniform = __all(b == b0 || !act) && (__shfl(act ? 1 : 0, 0, 64) != 0) && b0 >= 0 && b0 < B;
    if (uniform) {
        float mnx = act ? x : FLT_MAX, mny = act ? y : FLT_MAX, mnz = act ? z : FLT_MAX;
        float mxx = act ? x : -FLT_MAX, mxy = act ? y : -FLT_MAX, mxz = act ? z : -FLT_MAX;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, d, 64));
            mny = fminf(mny, __shfl_xor(mny, d, 64));
            mnz = fminf(mnz, __shfl_xor(mnz, d, 64));
            mxx = fmaxf(mxx, __shfl_xor(mxx, d, 64));
            mxy = fmaxf(mxy, __shfl_xor(mxy, d, 64));
            mxz = fmaxf(mxz, __shfl_xor(mxz, d, 64));
        }
        if (lane_id() == 0) {
            sh[wave][0] = f2ord(mnx); sh[wave][1] = f2ord(mny); sh[wave][2] = f2ord(mnz);
            sh[wave][3] = f2ord(mxx); sh[wave][4] = f2ord(mxy); sh[wave][5] = f2ord(mxz);
            sh[wave][6] = (unsigned)b0;
        }
    } else {
        if (lane_id() == 0) sh[wave][6] = 0xffffffffu;  // mixed / empty wave: handled per lane below
        if (act && b >= 0 && b < B) {
            atomicMin(&enc[b * 3], f2ord(x));
            atomicMin(&enc[b * 3 + 1], f2ord(y));
            atomicMin(&enc[b * 3 + 2], f2ord(z));
            atomicMax(&enc[3 * B + b * 3], f2ord(x));
            atomicMax(&enc[3 * B + b * 3 + 1], f2ord(y));
            atomicMax(&enc[3 * B + b * 3 + 2], f2ord(z));
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        // merge consecutive waves with the same batch id, one atomic per run
        int w = 0;
        while (w < 4) {
            unsigned bb = sh[w][6];
            if (bb == 0xffffffffu) { ++w; continue; }
            unsigned v = sh[w][k];
            int w2 = w + 1;
            while (w2 < 4 && sh[w2][6] == bb) {
                v = (k < 3) ? min(v, sh[w2][k]) : max(v, sh[w2][k]);
                ++w2;
            }
            if (k < 3) atomicMin(&enc[bb * 3 + k], v);
            else atomicMax(&enc[3 * B + bb * 3 + (k - 3)], v);
            w = w2;
        }
    }
}

// One block. scale_inv == 0: every row gets the whole-batch box (aabb_gpu.cu:104-114).
__global__ __launch_bounds__(256) void aabb_finalize(const unsigned* __restrict__ enc, int B, int scaleInv,
                                                     float* __restrict__ mn, float* __restrict__ mx) {
    if (scaleInv) {
        for (int t = threadIdx.x; t < 3 * B; t += blockDim.x) {
            mn[t] = ord2f(enc[t]);
            mx[t] = ord2f(enc[3 * B + t]);
        }
        return;
    }
    __shared__ unsigned g[6];
    if (threadIdx.x < 3) g[threadIdx.x] = 0xffffffffu;
    if (threadIdx.x >= 3 && threadIdx.x < 6) g[threadIdx.x] = 0u;
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * B; t += blockDim.x) {
        atomicMin(&g[t % 3], enc[t]);
        atomicMax(&g[3 + t % 3], enc[3 * B + t]);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * B; t += blockDim.x) {
        mn[t] = ord2f(g[t % 3]);
        mx[t] = ord2f(g[3 + t % 3]);
    }
}

// The whole op in ONE launch for up to MCCNN_AABB_ONE_N points (a network step computes one box per batch: init, reduce and
// finalize as three launches were 12-15 us of a chain that is a few microseconds of work): one workgroup of 1024 threads,
// running minima / maxima per cloud in LDS (order-preserving integers, wave pre-reduction as in aabb_reduce), then the
// finalize step. Same values (min / max are exact and order-free). Larger batches keep the three launches: one workgroup
// walks its points at one memory latency per trip (32 768 points: 68 us in the pipelined cfg1 step against ~20 for the three
// launches, and the hierarchy's chain is what bounds that step: 0.44 -> 0.41 ms with the limit at 8 192; cfg0, 4 096 points,
// keeps the single launch: 0.151 against 0.158 ms. MCCNN_DEBUG=aabb_one_max=N moves the limit down for an A/B).
#define MCCNN_AABB_ONE_N 8192
#define MCCNN_AABB_ONE_B 1024
__global__ __launch_bounds__(1024) void aabb_one(const float* __restrict__ pts, const int* __restrict__ bids, int n, int B,
                                                 int scaleInv, float* __restrict__ mn, float* __restrict__ mx) {
    __shared__ unsigned enc[6 * MCCNN_AABB_ONE_B];
    __shared__ unsigned g[6];
    const int t = threadIdx.x;
    for (int k = t; k < 3 * B; k += 1024) {
        enc[k] = f2ord(FLT_MAX);
        enc[3 * B + k] = f2ord(-FLT_MAX);
    }
    if (t < 3) g[t] = 0xffffffffu;
    if (t >= 3 && t < 6) g[t] = 0u;
    __syncthreads();
    for (int i00 = 0; i00 < n; i00 += 4096) {
      // four trips' loads in flight at once
      int bb[4];
      float xx[4], yy[4], zz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
          const int i = i00 + u * 1024 + t;
          const bool a = i < n;
          bb[u] = a ? bids[i] : -1;
          xx[u] = a ? pts[(size_t)i * 3] : 0.f;
          yy[u] = a ? pts[(size_t)i * 3 + 1] : 0.f;
          zz[u] = a ? pts[(size_t)i * 3 + 2] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i00 + u * 1024 + t;
        const bool act = i < n;
        const int b = bb[u];
        const float x = xx[u], y = yy[u], z = zz[u];
        if (i00 + u * 1024 >= n) break;
        const int b0 = __shfl(b, 0, 64);
        const bool uniform = __all(b == b0 || !act) && (__shfl(act ? 1 : 0, 0, 64) != 0) && b0 >= 0 && b0 < B;
        if (uniform) {   // clouds are stored contiguously: one LDS atomic per wave and coordinate
            float mnx = act ? x : FLT_MAX, mny = act ? y : FLT_MAX, mnz = act ? z : FLT_MAX;
            float mxx = act ? x : -FLT_MAX, mxy = act ? y : -FLT_MAX, mxz = act ? z : -FLT_MAX;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                mnx = fminf(mnx, __shfl_xor(mnx, d, 64));
                mny = fminf(mny, __shfl_xor(mny, d, 64));
                mnz = fminf(mnz, __shfl_xor(mnz, d, 64));
                mxx = fmaxf(mxx, __shfl_xor(mxx, d, 64));
                mxy = fmaxf(mxy, __shfl_xor(mxy, d, 64));
                mxz = fmaxf(mxz, __shfl_xor(mxz, d, 64));
            }
            if (lane_id() == 0) {
                atomicMin(&enc[b0 * 3], f2ord(mnx)); atomicMin(&enc[b0 * 3 + 1], f2ord(mny)); atomicMin(&enc[b0 * 3 + 2], f2ord(mnz));
                atomicMax(&enc[3 * B + b0 * 3], f2ord(mxx)); atomicMax(&enc[3 * B + b0 * 3 + 1], f2ord(mxy));
                atomicMax(&enc[3 * B + b0 * 3 + 2], f2ord(mxz));
            }
        } else if (act && b >= 0 && b < B) {
            atomicMin(&enc[b * 3], f2ord(x)); atomicMin(&enc[b * 3 + 1], f2ord(y)); atomicMin(&enc[b * 3 + 2], f2ord(z));
            atomicMax(&enc[3 * B + b * 3], f2ord(x)); atomicMax(&enc[3 * B + b * 3 + 1], f2ord(y));
            atomicMax(&enc[3 * B + b * 3 + 2], f2ord(z));
        }
      }
    }
    __syncthreads();
    if (scaleInv) {
        for (int k = t; k < 3 * B; k += 1024) {
            mn[k] = ord2f(enc[k]);
            mx[k] = ord2f(enc[3 * B + k]);
        }
        return;
    }
    // scale_inv == 0: every row gets the whole-batch box (aabb_gpu.cu:104-114)
    for (int k = t; k < 3 * B; k += 1024) {
        atomicMin(&g[k % 3], enc[k]);
        atomicMax(&g[3 + k % 3], enc[3 * B + k]);
    }
    __syncthreads();
    for (int k = t; k < 3 * B; k += 1024) {
        mn[k] = ord2f(g[k % 3]);
        mx[k] = ord2f(g[3 + k % 3]);
    }
}

__global__ void num_cells_dev(const float* __restrict__ mn, const float* __restrict__ mx, float cellSize,
                              int* __restrict__ out) {
    // determine_cell_size, sort_gpu.cu:374-391 (batch 0 only)
    float ext = max_extent(mn, mx, 0);
    int nc = (int)(ext / cellSize);
    *out = nc == 0 ? 1 : nc;
}

__global__ __launch_bounds__(256) void check_bids(const int* __restrict__ bids, int n, int B, int* __restrict__ bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (bids[i] < 0 || bids[i] >= B)) atomicAdd(bad, 1);  // one atomic per wave (the compiler aggregates)
}

// ------------------------------------------------------------------ keys + histogram
// calc_key + update_counters fused (sort_gpu.cu:35-80). arrival[i] = rank in atomic arrival
// order; it is only used to park point ids in their cell segment, the final order is fixed
// by rank_in_cell below.
__device__ __forceinline__ void keys_hist_body(int blk, const float* __restrict__ pts, const int* __restrict__ bids,
                                               const float* __restrict__ mn, const float* __restrict__ mx, int n, int B, int nc,
                                               int* __restrict__ keys, int* __restrict__ cnt, int* __restrict__ arrival,
                                               const int* __restrict__ nDev) {
    int i = blk * 256 + threadIdx.x;
    if (nDev) n = *nDev;  // device-side point count (hierarchy levels chained without a host read-back)
    if (i >= n) return;
    int b = clamp_batch(bids[i], B);
    float cs = max_extent(mn, mx, b) / (float)nc;
    int x = cell_coord(pts[(size_t)i * 3], mn[b * 3], cs, nc);
    int y = cell_coord(pts[(size_t)i * 3 + 1], mn[b * 3 + 1], cs, nc);
    int z = cell_coord(pts[(size_t)i * 3 + 2], mn[b * 3 + 2], cs, nc);
    int key = b * nc * nc * nc + x * nc * nc + y * nc + z;  // sort_gpu.cu:59
    keys[i] = key;
    arrival[i] = atomicAdd(&cnt[key], 1);
}
__global__ __launch_bounds__(256) void keys_hist(const float* __restrict__ pts, const int* __restrict__ bids,
                                                 const float* __restrict__ mn, const float* __restrict__ mx,
                                                 int n, int B, int nc, int* __restrict__ keys, int* __restrict__ cnt,
                                                 int* __restrict__ arrival, const int* __restrict__ nDev,
                                                 ClearSpan x1, ClearSpan x2) {
    clear_span_dev(x1);  // what LATER kernels of the chain want zeroed (common.h)
    clear_span_dev(x2);
    keys_hist_body((int)blockIdx.x, pts, bids, mn, mx, n, B, nc, keys, cnt, arrival, nDev);
}

// The same for grids of FEW cells (a classification network's coarse convolutions put a whole cloud into one to 27
// cells): thousands of returning atomics on a few dozen addresses serialise at the L2 (5 277 points in 32 cells: 34 us).
// Ranks inside the workgroup come from LDS counters, one global atomic per (workgroup, occupied cell) fetches the base.
#define MCCNN_HIST_LDS_BINS 1024
__device__ __forceinline__ void keys_hist_lds_body(int blk, const float* __restrict__ pts, const int* __restrict__ bids,
                                                   const float* __restrict__ mn, const float* __restrict__ mx, int n, int B, int nc,
                                                   int C, int* __restrict__ keys, int* __restrict__ cnt, int* __restrict__ arrival,
                                                   const int* __restrict__ nDev, int* __restrict__ bins /* LDS, MCCNN_HIST_LDS_BINS */) {
    for (int c = threadIdx.x; c < C; c += 256) bins[c] = 0;
    __syncthreads();
    const int i = blk * 256 + threadIdx.x;
    if (nDev) n = *nDev;
    int key = -1, local = 0;
    if (i < n) {
        const int b = clamp_batch(bids[i], B);
        const float cs = max_extent(mn, mx, b) / (float)nc;
        const int x = cell_coord(pts[(size_t)i * 3], mn[b * 3], cs, nc);
        const int y = cell_coord(pts[(size_t)i * 3 + 1], mn[b * 3 + 1], cs, nc);
        const int z = cell_coord(pts[(size_t)i * 3 + 2], mn[b * 3 + 2], cs, nc);
        key = b * nc * nc * nc + x * nc * nc + y * nc + z;
        keys[i] = key;
        local = atomicAdd(&bins[key], 1);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int v = bins[c];
        if (v) bins[c] = atomicAdd(&cnt[c], v);  // the workgroup's base in this cell
    }
    __syncthreads();
    if (i < n) arrival[i] = bins[key] + local;
}
__global__ __launch_bounds__(256) void keys_hist_lds(const float* __restrict__ pts, const int* __restrict__ bids,
                                                     const float* __restrict__ mn, const float* __restrict__ mx,
                                                     int n, int B, int nc, int C, int* __restrict__ keys,
                                                     int* __restrict__ cnt, int* __restrict__ arrival,
                                                     const int* __restrict__ nDev, ClearSpan x1, ClearSpan x2) {
    clear_span_dev(x1);
    clear_span_dev(x2);
    __shared__ int bins[MCCNN_HIST_LDS_BINS];
    keys_hist_lds_body((int)blockIdx.x, pts, bids, mn, mx, n, B, nc, C, keys, cnt, arrival, nDev, bins);
}

__global__ __launch_bounds__(256) void park_ids(const int* __restrict__ keys, const int* __restrict__ start,
                                                const int* __restrict__ arrival, int n, int* __restrict__ slot,
                                                const int* __restrict__ nDev) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (nDev) n = *nDev;
    if (i < n) slot[start[keys[i]] + arrival[i]] = i;
}

// new_idx[id] = cellStart + #{ids in the same cell smaller than id}: the stable (sequential) order.
// One thread per POSITION of the parked list: neighbouring lanes sit in the same cell and scan the same
// addresses (broadcast loads, equal trip counts); a thread per point id would make every lane of a wave scan a
// different cell. Cost is quadratic in the cell occupancy -- fine for radius-sized cells (tens of points), slow
// for degenerate grids that put >1e4 points into one cell.
__global__ __launch_bounds__(256) void rank_in_cell(const int* __restrict__ keys, const int* __restrict__ start,
                                                    const int* __restrict__ slot, int n, int* __restrict__ newIdx,
                                                    const int* __restrict__ nDev) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (nDev) n = *nDev;
    if (p >= n) return;
    const int id = slot[p];
    const int k = keys[id];
    const int s0 = start[k], s1 = start[k + 1];
    int r = 0;
    int q = s0;
    for (; q + 4 <= s1; q += 4) {
        int a0 = slot[q], a1 = slot[q + 1], a2 = slot[q + 2], a3 = slot[q + 3];
        r += (a0 < id) + (a1 < id) + (a2 < id) + (a3 < id);
    }
    for (; q < s1; ++q) r += (slot[q] < id) ? 1 : 0;
    newIdx[id] = s0 + r;
}

// rank_in_cell, move_points and cell_table in ONE launch (the grid build of the native executor and of a hierarchy level,
// where both sort steps belong to one call): the thread of parked position p ranks its point inside the cell and moves it
// -- point, batch id, inverse permutation -- to the final position at once; the cell table needs neither the sorted keys
// nor a cleared table: entry c is [start[c], start[c + 1]) of the prefix sum when the cell holds a point and (0, 0) -- what
// the reference's memset leaves, sort_gpu.cu:492 -- when it does not (save_indexs, sort_gpu.cu:225-248, writes exactly
// these bounds). Seven launches of round 5 (memset, keys_hist, scan, park_ids, rank_in_cell, move_points, cell_table) are
// four: keys_hist, scan, park_ids, rank_move.
__device__ __forceinline__ void rank_move_body(int blk, int nblk, const int* __restrict__ keys, const int* __restrict__ start,
                                               const int* __restrict__ slot, int n, long long numCells,
                                               const float* __restrict__ pts, const int* __restrict__ bids,
                                               int* __restrict__ newIdx, float* __restrict__ oPts, int* __restrict__ oBids,
                                               int* __restrict__ inv, int2* __restrict__ cells, const int* __restrict__ nDev) {
    const long long t = (long long)blk * 256 + threadIdx.x;
    for (long long c = t; c < numCells; c += (long long)nblk * 256) {
        const int s0 = start[c], s1 = start[c + 1];
        cells[c] = s1 > s0 ? make_int2(s0, s1) : make_int2(0, 0);
    }
    if (nDev) n = *nDev;
    if (t >= n) return;
    const int id = slot[t];
    const int k = keys[id];
    const int s0 = start[k], s1 = start[k + 1];
    int r = 0;
    int q = s0;
    for (; q + 4 <= s1; q += 4) {
        int a0 = slot[q], a1 = slot[q + 1], a2 = slot[q + 2], a3 = slot[q + 3];
        r += (a0 < id) + (a1 < id) + (a2 < id) + (a3 < id);
    }
    for (; q < s1; ++q) r += (slot[q] < id) ? 1 : 0;
    const int f = s0 + r;
    newIdx[id] = f;
    oPts[(size_t)f * 3] = pts[(size_t)id * 3];
    oPts[(size_t)f * 3 + 1] = pts[(size_t)id * 3 + 1];
    oPts[(size_t)f * 3 + 2] = pts[(size_t)id * 3 + 2];
    oBids[f] = bids[id];
    if (inv) inv[f] = id;
}
__global__ __launch_bounds__(256) void rank_move(const int* __restrict__ keys, const int* __restrict__ start,
                                                 const int* __restrict__ slot, int n, long long numCells,
                                                 const float* __restrict__ pts, const int* __restrict__ bids,
                                                 int* __restrict__ newIdx, float* __restrict__ oPts, int* __restrict__ oBids,
                                                 int* __restrict__ inv, int2* __restrict__ cells,
                                                 const int* __restrict__ nDev, ClearSpan x1, ClearSpan x2) {
    clear_span_dev(x1);
    clear_span_dev(x2);
    rank_move_body((int)blockIdx.x, (int)gridDim.x, keys, start, slot, n, numCells, pts, bids, newIdx, oPts, oBids, inv, cells, nDev);
}

// ------------------------------------------------------------------ one launch per phase over a BATCH of counting sorts
// (mccnn_geometry_build_batch, exec.hip): item = the grid build of one geometry, or the visiting order of one geometry's
// foreign centres (newIdx == nullptr: no rank / move phase). The prefix sums of the items' cell counters are one launch of
// scan.hip's batch form in between.
__global__ __launch_bounds__(256) void grid_keys_batch(GridBatch gb, BatchBlocks bb) {
    __shared__ int bins[MCCNN_HIST_LDS_BINS];
    int local, blocks;
    const GridItem& g = gb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    if (g.ldsHist) keys_hist_lds_body(local, g.pts, g.bids, g.mn, g.mx, g.n, g.B, g.nc, g.C, g.keys, g.cnt, g.arrival, nullptr, bins);
    else keys_hist_body(local, g.pts, g.bids, g.mn, g.mx, g.n, g.B, g.nc, g.keys, g.cnt, g.arrival, nullptr);
}
__global__ __launch_bounds__(256) void grid_park_batch(GridBatch gb, BatchBlocks bb) {
    int local, blocks;
    const GridItem& g = gb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const int i = local * 256 + threadIdx.x;
    if (i < g.n) g.slot[g.start[g.keys[i]] + g.arrival[i]] = i;
}
__global__ __launch_bounds__(256) void grid_rank_move_batch(GridBatch gb, BatchBlocks bb) {
    int local, blocks;
    const GridItem& g = gb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    rank_move_body(local, blocks, g.keys, g.start, g.slot, g.n, g.C, g.pts, g.bids, g.newIdx, g.oPts, g.oBids, g.inv, g.cells, nullptr);
}

// ------------------------------------------------------------------ step 2
// One launch moves points, batch ids, keys (and feature rows of up to 4 floats) to their sorted positions, clears the
// cell table for cell_table() (sort_gpu.cu:492) and, if asked, writes the inverse permutation (the cell-coherent
// visiting order find_neighbors wants): at 100k points each of these as a launch of its own costs ~5 us of pure
// launch latency.
template <int FS>  // feature floats moved here (0: separate permute_rows launch)
__global__ __launch_bounds__(256) void move_points(const float* __restrict__ pts, const int* __restrict__ bids,
                                                   const float* __restrict__ feats, const int* __restrict__ keys,
                                                   const int* __restrict__ newIdx, int n, float* __restrict__ oPts,
                                                   int* __restrict__ oBids, float* __restrict__ oFeats,
                                                   int* __restrict__ sKeys, int* __restrict__ inv,
                                                   int2* __restrict__ cells, long long numCells,
                                                   const int* __restrict__ nDev) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long c = i; c < numCells; c += (long long)gridDim.x * blockDim.x) cells[c] = make_int2(0, 0);
    if (nDev) n = *nDev;
    if (i >= n) return;
    int p = newIdx[i];
    oPts[(size_t)p * 3] = pts[(size_t)i * 3];
    oPts[(size_t)p * 3 + 1] = pts[(size_t)i * 3 + 1];
    oPts[(size_t)p * 3 + 2] = pts[(size_t)i * 3 + 2];
    oBids[p] = bids[i];
    sKeys[p] = keys[i];
    if (inv) inv[p] = (int)i;
#pragma unroll
    for (int f = 0; f < FS; ++f) oFeats[(size_t)p * FS + f] = feats[(size_t)i * FS + f];
}

// save_indexs, sort_gpu.cu:225-248
__global__ __launch_bounds__(256) void cell_table(const int* __restrict__ sKeys, int n, int* __restrict__ cells,
                                                  const int* __restrict__ nDev) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (nDev) n = *nDev;
    if (p >= n) return;
    int k = sKeys[p];
    if (p == 0 || sKeys[p - 1] != k) cells[2 * (size_t)k] = p;
    if (p == n - 1 || sKeys[p + 1] != k) cells[2 * (size_t)k + 1] = p + 1;
}

// ------------------------------------------------------------------ small grids: one workgroup per op
// A coarse level of a hierarchy (a few thousand points, a few thousand cells) spends its grid build in LAUNCHES: memset,
// keys_hist, scan, park_ids, rank_in_cell for step 1, move_points, cell_table for step 2 -- 7 launches of 2-4 us of work
// each, and the host pays ~6 us to issue every one of them. Up to MCCNN_GRID_SMALL_N points in at most
// MCCNN_GRID_SMALL_C cells (beyond that one workgroup's serial trips cost more than the launches: 5 277 points took 75 us
// against ~15 us for the seven kernels) ONE workgroup of 1024 threads runs the same phases back to back with barriers in between:
// same arrays, same results (the order inside a cell is fixed by the ranks, not by the arrival order of the atomics).
#define MCCNN_GRID_SMALL_N 2048
#define MCCNN_GRID_SMALL_C 8192

__global__ __launch_bounds__(1024) void grid_small_step1(const float* __restrict__ pts, const int* __restrict__ bids,
                                                         const float* __restrict__ mn, const float* __restrict__ mx, int n,
                                                         int B, int nc, int C, int* __restrict__ keys,
                                                         int* __restrict__ newIdx, int* __restrict__ cnt,
                                                         int* __restrict__ start, int* __restrict__ slot,
                                                         const int* __restrict__ nDev) {
    __shared__ int wsum[17];
    const int t = threadIdx.x;
    if (nDev) n = *nDev;
    for (int c = t; c < C; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {  // keys_hist
        const int b = clamp_batch(bids[i], B);
        const float cs = max_extent(mn, mx, b) / (float)nc;
        const int x = cell_coord(pts[(size_t)i * 3], mn[b * 3], cs, nc);
        const int y = cell_coord(pts[(size_t)i * 3 + 1], mn[b * 3 + 1], cs, nc);
        const int z = cell_coord(pts[(size_t)i * 3 + 2], mn[b * 3 + 2], cs, nc);
        const int key = b * nc * nc * nc + x * nc * nc + y * nc + z;
        keys[i] = key;
        newIdx[i] = atomicAdd(&cnt[key], 1);  // arrival rank, replaced by the final position below
    }
    __syncthreads();
    {   // start = exclusive scan of the cell counts: a contiguous run of cells per thread
        const int per = (C + 1023) / 1024, c0 = min(C, t * per), c1 = min(C, c0 + per);
        int sum = 0;
        for (int c = c0; c < c1; ++c) sum += cnt[c];
        int tot;
        int run = block1024_excl_scan(sum, tot, wsum);
        for (int c = c0; c < c1; ++c) { const int v = cnt[c]; start[c] = run; run += v; }
        if (t == 0) start[C] = tot;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) slot[start[keys[i]] + newIdx[i]] = i;  // park_ids
    __syncthreads();
    for (int p = t; p < n; p += 1024) {  // rank_in_cell
        const int id = slot[p];
        const int k = keys[id];
        const int s0 = start[k], s1 = start[k + 1];
        int r = 0;
        for (int q = s0; q < s1; ++q) r += (slot[q] < id) ? 1 : 0;
        newIdx[id] = s0 + r;
    }
}

// Both sort steps of a small level in ONE workgroup (geometry only: the grid build of the native executor and of a
// hierarchy level): the phases of grid_small_step1, then -- still in the same launch -- the cell table straight from the
// prefix sum and every point moved by the thread that ranked it (see rank_move).
__global__ __launch_bounds__(1024) void grid_small_all(const float* __restrict__ pts, const int* __restrict__ bids,
                                                       const float* __restrict__ mn, const float* __restrict__ mx, int n,
                                                       int B, int nc, int C, int* __restrict__ keys, int* __restrict__ newIdx,
                                                       int* __restrict__ cnt, int* __restrict__ start, int* __restrict__ slot,
                                                       float* __restrict__ oPts, int* __restrict__ oBids, int* __restrict__ inv,
                                                       int2* __restrict__ cells, const int* __restrict__ nDev, ClearSpan x1,
                                                       ClearSpan x2) {
    __shared__ int wsum[17];
    clear_span_dev(x1);
    clear_span_dev(x2);
    const int t = threadIdx.x;
    if (nDev) n = *nDev;
    for (int c = t; c < C; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {  // keys_hist
        const int b = clamp_batch(bids[i], B);
        const float cs = max_extent(mn, mx, b) / (float)nc;
        const int x = cell_coord(pts[(size_t)i * 3], mn[b * 3], cs, nc);
        const int y = cell_coord(pts[(size_t)i * 3 + 1], mn[b * 3 + 1], cs, nc);
        const int z = cell_coord(pts[(size_t)i * 3 + 2], mn[b * 3 + 2], cs, nc);
        const int key = b * nc * nc * nc + x * nc * nc + y * nc + z;
        keys[i] = key;
        newIdx[i] = atomicAdd(&cnt[key], 1);  // arrival rank, replaced by the final position below
    }
    __syncthreads();
    {   // start = exclusive scan of the cell counts; the cell table on the way (non-empty: [start, start + count))
        const int per = (C + 1023) / 1024, c0 = min(C, t * per), c1 = min(C, c0 + per);
        int sum = 0;
        for (int c = c0; c < c1; ++c) sum += cnt[c];
        int tot;
        int run = block1024_excl_scan(sum, tot, wsum);
        for (int c = c0; c < c1; ++c) {
            const int v = cnt[c];
            start[c] = run;
            cells[c] = v > 0 ? make_int2(run, run + v) : make_int2(0, 0);
            run += v;
        }
        if (t == 0) start[C] = tot;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) slot[start[keys[i]] + newIdx[i]] = i;  // park_ids
    __syncthreads();
    for (int p = t; p < n; p += 1024) {  // rank_in_cell + move_points
        const int id = slot[p];
        const int k = keys[id];
        const int s0 = start[k], s1 = start[k + 1];
        int r = 0;
        for (int q = s0; q < s1; ++q) r += (slot[q] < id) ? 1 : 0;
        const int f = s0 + r;
        newIdx[id] = f;
        oPts[(size_t)f * 3] = pts[(size_t)id * 3];
        oPts[(size_t)f * 3 + 1] = pts[(size_t)id * 3 + 1];
        oPts[(size_t)f * 3 + 2] = pts[(size_t)id * 3 + 2];
        oBids[f] = bids[id];
        if (inv) inv[f] = id;
    }
}

template <int FS>
__global__ __launch_bounds__(1024) void grid_small_step2(const float* __restrict__ pts, const int* __restrict__ bids,
                                                         const float* __restrict__ feats, const int* __restrict__ keys,
                                                         const int* __restrict__ newIdx, int n, float* __restrict__ oPts,
                                                         int* __restrict__ oBids, float* __restrict__ oFeats,
                                                         int* __restrict__ sKeys, int* __restrict__ inv,
                                                         int2* __restrict__ cells, int C, const int* __restrict__ nDev) {
    const int t = threadIdx.x;
    if (nDev) n = *nDev;
    for (int c = t; c < C; c += 1024) cells[c] = make_int2(0, 0);  // sort_gpu.cu:492
    for (int i = t; i < n; i += 1024) {  // move_points
        const int p = newIdx[i];
        oPts[(size_t)p * 3] = pts[(size_t)i * 3];
        oPts[(size_t)p * 3 + 1] = pts[(size_t)i * 3 + 1];
        oPts[(size_t)p * 3 + 2] = pts[(size_t)i * 3 + 2];
        oBids[p] = bids[i];
        sKeys[p] = keys[i];
        if (inv) inv[p] = i;
#pragma unroll
        for (int f = 0; f < FS; ++f) oFeats[(size_t)p * FS + f] = feats[(size_t)i * FS + f];
    }
    __syncthreads();
    int* ct = reinterpret_cast<int*>(cells);
    for (int p = t; p < n; p += 1024) {  // cell_table (save_indexs, sort_gpu.cu:225-248)
        const int k = sKeys[p];
        if (p == 0 || sKeys[p - 1] != k) ct[2 * (size_t)k] = p;
        if (p == n - 1 || sKeys[p + 1] != k) ct[2 * (size_t)k + 1] = p + 1;
    }
}

// ------------------------------------------------------------------ row permutations
// One thread per VEC floats of a row; rows are F floats. GATHER: out[i] = in[idx[i]],
// else out[idx[i]] = in[i].
template <int VEC, bool GATHER>
__global__ __launch_bounds__(256) void permute_rows(const float* __restrict__ in, const int* __restrict__ idx,
                                                    long long total /* n*F/VEC */, int fv /* F/VEC */,
                                                    float* __restrict__ out) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    long long row = t / fv;
    int c = (int)(t - row * fv);
    long long other = idx[row];
    long long src = GATHER ? other * fv + c : t;
    long long dst = GATHER ? t : other * fv + c;
    if (VEC == 4) {
        reinterpret_cast<float4*>(out)[dst] = reinterpret_cast<const float4*>(in)[src];
    } else if (VEC == 2) {
        reinterpret_cast<float2*>(out)[dst] = reinterpret_cast<const float2*>(in)[src];
    } else {
        out[dst] = in[src];
    }
}

template <bool GATHER>
static int launch_permute(const float* in, const int* idx, int n, int F, float* out, hipStream_t s) {
    if (n <= 0) return 0;
    bool a16 = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    bool a8 = (((uintptr_t)in | (uintptr_t)out) & 7) == 0;
    if (F % 4 == 0 && a16) {
        long long total = (long long)n * (F / 4);
        permute_rows<4, GATHER><<<ceil_div(total, 256), 256, 0, s>>>(in, idx, total, F / 4, out);
    } else if (F % 2 == 0 && a8) {
        long long total = (long long)n * (F / 2);
        permute_rows<2, GATHER><<<ceil_div(total, 256), 256, 0, s>>>(in, idx, total, F / 2, out);
    } else {
        long long total = (long long)n * F;
        permute_rows<1, GATHER><<<ceil_div(total, 256), 256, 0, s>>>(in, idx, total, F, out);
    }
    MCCNN_LAUNCHED();
    return 0;
}

__global__ __launch_bounds__(256) void invert_perm(const int* __restrict__ newIdx, int n, int* __restrict__ inv,
                                                   const int* __restrict__ nDev) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (nDev) n = *nDev;
    if (i < n) inv[newIdx[i]] = i;
}
__global__ __launch_bounds__(256) void map_indexs(const int* __restrict__ in, int s, const int* __restrict__ inv,
                                                  int* __restrict__ out, const int* __restrict__ sDev) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (sDev) s = *sDev;
    if (i < s) out[i] = inv[in[i]];
}

}  // namespace mccnn

using namespace mccnn;

extern "C" {

int mccnn_check_batch_ids(const int* batch_ids, int n, int batch_size, int* bad_count_dev, mccnn_stream_t stream) {
    if (n < 0 || batch_size <= 0 || !bad_count_dev || (n > 0 && !batch_ids)) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    {
        int rc = launch_zero_words(bad_count_dev, 1, s);
        if (rc) return rc;
    }
    if (n == 0) return 0;
    check_bids<<<ceil_div(n, 256), 256, 0, s>>>(batch_ids, n, batch_size, bad_count_dev);
    MCCNN_LAUNCHED();
    return 0;
}

size_t mccnn_compute_aabb_workspace_bytes(int batch_size) {
    return align_up((size_t)(batch_size > 0 ? batch_size : 1) * 6 * sizeof(unsigned));
}

int mccnn_compute_aabb(const float* pts, const int* batch_ids, int n, int batch_size, int scale_inv,
                       float* aabb_min, float* aabb_max, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (!aabb_min || !aabb_max || batch_size <= 0 || n < 0 || (n > 0 && (!pts || !batch_ids))) return MCCNN_E_BADARG;
    if (!ws || ws_bytes < mccnn_compute_aabb_workspace_bytes(batch_size)) return MCCNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    static const int oneMax = debug_int("aabb_one_max", MCCNN_AABB_ONE_N);   // A/B switch, read once
    if (n <= oneMax && n <= MCCNN_AABB_ONE_N && batch_size <= MCCNN_AABB_ONE_B && small_kernels_on()) {
        aabb_one<<<1, 1024, 0, s>>>(pts, batch_ids, n, batch_size, scale_inv, aabb_min, aabb_max);
        MCCNN_LAUNCHED();
        return 0;
    }
    unsigned* enc = (unsigned*)ws;
    aabb_init<<<ceil_div(3 * batch_size, 256), 256, 0, s>>>(enc, batch_size);
    MCCNN_LAUNCHED();
    if (n > 0) {
        aabb_reduce<<<ceil_div(n, 256), 256, 0, s>>>(pts, batch_ids, n, batch_size, enc);
        MCCNN_LAUNCHED();
    }
    aabb_finalize<<<1, 256, 0, s>>>(enc, batch_size, scale_inv, aabb_min, aabb_max);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_num_cells(const float* aabb_min, const float* aabb_max, int batch_size, float cell_size, int scale_inv,
                    int* num_cells_host, mccnn_stream_t stream) {
    if (!num_cells_host || batch_size <= 0 || !(cell_size > 0.0f)) return MCCNN_E_BADARG;
    if (scale_inv) {  // sort_gpu.cu:404-408
        int nc = (int)(1.0f / cell_size);
        *num_cells_host = nc == 0 ? 1 : nc;
        return 0;
    }
    if (!aabb_min || !aabb_max) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float h[6];
    MCCNN_HIP(hipMemcpyAsync(h, aabb_min, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    MCCNN_HIP(hipMemcpyAsync(h + 3, aabb_max, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    MCCNN_HIP(hipStreamSynchronize(s));
    float ext = fmaxf(fmaxf(h[3] - h[0], h[4] - h[1]), h[5] - h[2]);
    int nc = (int)(ext / cell_size);
    *num_cells_host = nc == 0 ? 1 : nc;
    return 0;
}

int mccnn_aabb_extent(const float* aabb_min, const float* aabb_max, float* extent_host, mccnn_stream_t stream) {
    if (!aabb_min || !aabb_max || !extent_host) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float h[6];
    MCCNN_HIP(hipMemcpyAsync(h, aabb_min, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    MCCNN_HIP(hipMemcpyAsync(h + 3, aabb_max, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    MCCNN_HIP(hipStreamSynchronize(s));
    *extent_host = fmaxf(fmaxf(h[3] - h[0], h[4] - h[1]), h[5] - h[2]);
    return 0;
}

static long long total_cells(int B, int nc) { return (long long)B * nc * nc * nc; }

size_t mccnn_sort_step1_workspace_bytes(int n, int batch_size, int num_cells) {
    long long C = total_cells(batch_size, num_cells);
    if (C <= 0 || C >= 0x7fffffffLL) return 0;
    return align_up((size_t)C * 4) + align_up((size_t)(C + 1) * 4) + align_up((size_t)(n > 0 ? n : 1) * 4) +
           scan_workspace_bytes((int)C) + 256;
}

static int sort_step1_impl(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int n,
                           int batch_size, int num_cells, int* keys, int* new_idx, void* ws, size_t ws_bytes,
                           mccnn_stream_t stream, const int* n_dev) {
    if (n < 0 || batch_size <= 0 || num_cells <= 0 || !aabb_min || !aabb_max) return MCCNN_E_BADARG;
    if (n == 0) return 0;
    if (!pts || !batch_ids || !keys || !new_idx) return MCCNN_E_BADARG;
    long long C = total_cells(batch_size, num_cells);
    if (C >= 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    if (!ws || ws_bytes < mccnn_sort_step1_workspace_bytes(n, batch_size, num_cells)) return MCCNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Arena a(ws, ws_bytes);
    // counters and the scan's status words are neighbours: ONE memset clears both
    const size_t cntBytes = align_up((size_t)C * 4), scanBytes = scan_workspace_bytes((int)C);
    char* blk = a.take<char>(cntBytes + scanBytes);
    int* start = a.take<int>((size_t)C + 1);
    int* slot = a.take<int>((size_t)n);
    if (!blk || !start || !slot) return MCCNN_E_WORKSPACE;
    int* cnt = (int*)blk;
    void* scanws = blk + cntBytes;
    if (n <= MCCNN_GRID_SMALL_N && C <= MCCNN_GRID_SMALL_C && small_kernels_on()) {
        grid_small_step1<<<1, 1024, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, (int)C, keys, new_idx, cnt,
                                            start, slot, n_dev);
        MCCNN_LAUNCHED();
        return 0;
    }
    {   // the head of this chain: histogram counters and the scan's status words (neighbours: one span)
        int rc = launch_clear_spans(clear_span(blk, cntBytes + scan_status_bytes((int)C)), no_span(), no_span(), s);
        if (rc) return rc;
    }
    int blocks = ceil_div(n, 256);
    if (C <= MCCNN_HIST_LDS_BINS && n >= 4 * C)  // few cells, many points per cell
        keys_hist_lds<<<blocks, 256, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, (int)C, keys, cnt, new_idx, n_dev, no_span(), no_span());
    else
        keys_hist<<<blocks, 256, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, keys, cnt, new_idx, n_dev, no_span(), no_span());
    MCCNN_LAUNCHED();
    int rc = exclusive_scan_i32(cnt, start, (int)C, start + C, scanws, s, true);
    if (rc) return rc;
    park_ids<<<blocks, 256, 0, s>>>(keys, start, new_idx, n, slot, n_dev);
    MCCNN_LAUNCHED();
    rank_in_cell<<<blocks, 256, 0, s>>>(keys, start, slot, n, new_idx, n_dev);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_sort_step1(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int n,
                     int batch_size, int num_cells, int* keys, int* new_idx, void* ws, size_t ws_bytes,
                     mccnn_stream_t stream) {
    return sort_step1_impl(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, keys, new_idx, ws, ws_bytes, stream,
                           nullptr);
}

int mccnn_sort_step1_dn(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int n_cap,
                        const int* n_dev, int batch_size, int num_cells, int* keys, int* new_idx, void* ws,
                        size_t ws_bytes, mccnn_stream_t stream) {
    if (!n_dev) return MCCNN_E_BADARG;
    return sort_step1_impl(pts, batch_ids, aabb_min, aabb_max, n_cap, batch_size, num_cells, keys, new_idx, ws, ws_bytes,
                           stream, n_dev);
}

size_t mccnn_sort_step2_workspace_bytes(int n) { return align_up((size_t)(n > 0 ? n : 1) * 4); }

static int sort_step2_impl(const float* pts, const int* batch_ids, const float* feats, const int* keys, const int* new_idx,
                           int n, int num_feats, int batch_size, int num_cells, float* out_pts, int* out_batch_ids,
                           float* out_feats, int* cell_indexs, int* inv_idx, void* ws, size_t ws_bytes,
                           mccnn_stream_t stream, const int* n_dev, bool geometry_only = false) {
    // num_feats == 0 (geometry only: no feature rows are moved) is accepted by the device-count form and mccnn_build_grid
    if (n < 0 || batch_size <= 0 || num_cells <= 0 || num_feats < ((n_dev || geometry_only) ? 0 : 1) || !cell_indexs) return MCCNN_E_BADARG;
    long long C = total_cells(batch_size, num_cells);
    if (C >= 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        return launch_zero_words(cell_indexs, (size_t)C * 2, s);  // sort_gpu.cu:492
    }
    if (!pts || !batch_ids || !keys || !new_idx || !out_pts || !out_batch_ids || (num_feats > 0 && (!feats || !out_feats)))
        return MCCNN_E_BADARG;
    if (n_dev && num_feats > 4) return MCCNN_E_BADARG;  // wide feature rows are gathered after the sizes are known
    if (!ws || ws_bytes < mccnn_sort_step2_workspace_bytes(n)) return MCCNN_E_WORKSPACE;
    int* skeys = (int*)ws;
    int blocks = ceil_div(n, 256);
    int2* ct = reinterpret_cast<int2*>(cell_indexs);
    if (n <= MCCNN_GRID_SMALL_N && C <= MCCNN_GRID_SMALL_C && small_kernels_on()) {
        switch (num_feats <= 4 ? num_feats : 0) {
            case 1: grid_small_step2<1><<<1, 1024, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, (int)C, n_dev); break;
            case 2: grid_small_step2<2><<<1, 1024, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, (int)C, n_dev); break;
            case 3: grid_small_step2<3><<<1, 1024, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, (int)C, n_dev); break;
            case 4: grid_small_step2<4><<<1, 1024, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, (int)C, n_dev); break;
            default: grid_small_step2<0><<<1, 1024, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, (int)C, n_dev); break;
        }
        MCCNN_LAUNCHED();
        if (num_feats > 4) return launch_permute<false>(feats, new_idx, n, num_feats, out_feats, s);
        return 0;
    }
    switch (num_feats <= 4 ? num_feats : 0) {
        case 1: move_points<1><<<blocks, 256, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, C, n_dev); break;
        case 2: move_points<2><<<blocks, 256, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, C, n_dev); break;
        case 3: move_points<3><<<blocks, 256, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, C, n_dev); break;
        case 4: move_points<4><<<blocks, 256, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, C, n_dev); break;
        default: move_points<0><<<blocks, 256, 0, s>>>(pts, batch_ids, feats, keys, new_idx, n, out_pts, out_batch_ids, out_feats, skeys, inv_idx, ct, C, n_dev); break;
    }
    MCCNN_LAUNCHED();
    if (num_feats > 4) {
        int rc = launch_permute<false>(feats, new_idx, n, num_feats, out_feats, s);
        if (rc) return rc;
    }
    cell_table<<<blocks, 256, 0, s>>>(skeys, n, cell_indexs, n_dev);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_sort_step2(const float* pts, const int* batch_ids, const float* feats, const int* keys, const int* new_idx,
                     int n, int num_feats, int batch_size, int num_cells, float* out_pts, int* out_batch_ids,
                     float* out_feats, int* cell_indexs, int* inv_idx, void* ws, size_t ws_bytes,
                     mccnn_stream_t stream) {
    return sort_step2_impl(pts, batch_ids, feats, keys, new_idx, n, num_feats, batch_size, num_cells, out_pts, out_batch_ids,
                           out_feats, cell_indexs, inv_idx, ws, ws_bytes, stream, nullptr);
}

int mccnn_sort_step2_dn(const float* pts, const int* batch_ids, const int* keys, const int* new_idx, int n_cap,
                        const int* n_dev, int batch_size, int num_cells, float* out_pts, int* out_batch_ids,
                        int* cell_indexs, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (!n_dev) return MCCNN_E_BADARG;
    return sort_step2_impl(pts, batch_ids, nullptr, keys, new_idx, n_cap, 0, batch_size, num_cells, out_pts, out_batch_ids,
                           nullptr, cell_indexs, nullptr, ws, ws_bytes, stream, n_dev);
}

}  // extern "C"
namespace mccnn {
// sort_step2_dn that also leaves the inverse permutation (sorted position -> input row): mccnn_hierarchy_level hands it
// to the Poisson emit kernel, which then writes the transformed indices itself (poisson.hip)
int sort_step2_dn_inv(const float* pts, const int* batch_ids, const int* keys, const int* new_idx, int n_cap,
                      const int* n_dev, int batch_size, int num_cells, float* out_pts, int* out_batch_ids,
                      int* cell_indexs, int* inv_idx, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (!n_dev) return MCCNN_E_BADARG;
    return sort_step2_impl(pts, batch_ids, nullptr, keys, new_idx, n_cap, 0, batch_size, num_cells, out_pts, out_batch_ids,
                           nullptr, cell_indexs, inv_idx, ws, ws_bytes, stream, n_dev);
}
}  // namespace mccnn
extern "C" {

size_t mccnn_build_grid_workspace_bytes(int n, int batch_size, int num_cells) {
    const size_t a = mccnn_sort_step1_workspace_bytes(n, batch_size, num_cells);
    if (a == 0) return 0;
    const size_t b = mccnn_sort_step2_workspace_bytes(n);
    return align_up((size_t)(n > 0 ? n : 1) * 4) + (a > b ? a : b);
}

}  // extern "C"
namespace mccnn {
// Both sort steps of a grid build as ONE chain (geometry only), with the device-count form folded in: keys_hist -> prefix
// sum -> park_ids -> rank_move (four launches; one workgroup and one launch for small levels). `ws` as
// mccnn_build_grid_workspace_bytes sizes it. What has to be zero before the first kernel -- the histogram counters and the
// status words of the single-pass scan -- is the span grid_head_span() names: the caller either had an earlier kernel of
// ITS chain clear it (cleared = true) or this function launches the clear itself. x1 / x2: spans LATER stages of the
// caller's chain want cleared (a hierarchy level's sampling state ...): the first kernel here clears them on its way.
ClearSpan grid_head_span(int n, int batch_size, int num_cells, void* ws, size_t ws_bytes) {
    const long long C = total_cells(batch_size, num_cells);
    if (C <= 0 || C >= 0x7fffffffLL || !ws) return no_span();
    if (n <= MCCNN_GRID_SMALL_N && C <= MCCNN_GRID_SMALL_C && small_kernels_on()) return no_span();  // clears its counters itself
    Arena a(ws, ws_bytes);
    if (!a.take<int>((size_t)(n > 0 ? n : 1))) return no_span();
    const size_t cntBytes = align_up((size_t)C * 4);
    char* blk = a.take<char>(cntBytes + scan_workspace_bytes((int)C));
    if (!blk) return no_span();
    return clear_span(blk, cntBytes + scan_status_bytes((int)C));
}

int build_grid_fused(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int n,
                     int batch_size, int num_cells, int* new_idx, float* out_pts, int* out_batch_ids, int* cell_indexs,
                     int* inv_idx, void* ws, size_t ws_bytes, hipStream_t s, const int* n_dev, bool cleared, ClearSpan x1,
                     ClearSpan x2) {
    if (n < 0 || batch_size <= 0 || num_cells <= 0 || !aabb_min || !aabb_max || !cell_indexs) return MCCNN_E_BADARG;
    const long long C = total_cells(batch_size, num_cells);
    if (C >= 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    const size_t need = mccnn_build_grid_workspace_bytes(n, batch_size, num_cells);
    if (need == 0) return MCCNN_E_TOOLARGE;
    if (!ws || ws_bytes < need) return MCCNN_E_WORKSPACE;
    if (n == 0) {
        int rc = launch_clear_spans(x1, x2, no_span(), s);
        if (rc) return rc;
        return launch_zero_words(cell_indexs, (size_t)C * 2, s);  // sort_gpu.cu:492
    }
    if (!pts || !batch_ids || !new_idx || !out_pts || !out_batch_ids) return MCCNN_E_BADARG;
    Arena a(ws, ws_bytes);
    int* keys = a.take<int>((size_t)n);
    const size_t cntBytes = align_up((size_t)C * 4), scanBytes = scan_workspace_bytes((int)C);
    char* blk = a.take<char>(cntBytes + scanBytes);
    int* start = a.take<int>((size_t)C + 1);
    int* slot = a.take<int>((size_t)n);
    if (!keys || !blk || !start || !slot) return MCCNN_E_WORKSPACE;
    int* cnt = (int*)blk;
    void* scanws = blk + cntBytes;
    int2* ct = reinterpret_cast<int2*>(cell_indexs);
    if (n <= MCCNN_GRID_SMALL_N && C <= MCCNN_GRID_SMALL_C && small_kernels_on()) {
        grid_small_all<<<1, 1024, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, (int)C, keys, new_idx, cnt,
                                          start, slot, out_pts, out_batch_ids, inv_idx, ct, n_dev, x1, x2);
        MCCNN_LAUNCHED();
        return 0;
    }
    if (!cleared) {
        int rc = launch_clear_spans(clear_span(blk, cntBytes + scan_status_bytes((int)C)), no_span(), no_span(), s);
        if (rc) return rc;
    }
    const int blocks = ceil_div(n, 256);
    if (C <= MCCNN_HIST_LDS_BINS && n >= 4 * C)  // few cells, many points per cell
        keys_hist_lds<<<blocks, 256, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, (int)C, keys, cnt, new_idx, n_dev, x1, x2);
    else
        keys_hist<<<blocks, 256, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, keys, cnt, new_idx, n_dev, x1, x2);
    MCCNN_LAUNCHED();
    int rc = exclusive_scan_i32(cnt, start, (int)C, start + C, scanws, s, true);
    if (rc) return rc;
    park_ids<<<blocks, 256, 0, s>>>(keys, start, new_idx, n, slot, n_dev);
    MCCNN_LAUNCHED();
    rank_move<<<blocks, 256, 0, s>>>(keys, start, slot, n, C, pts, batch_ids, new_idx, out_pts, out_batch_ids, inv_idx, ct, n_dev,
                                     no_span(), no_span());
    MCCNN_LAUNCHED();
    return 0;
}

// ---- host side of the batch form: one item per counting sort, the same workspace layouts as the single calls above
bool grid_batch_eligible(int n, int batch_size, int num_cells) {
    const long long C = total_cells(batch_size, num_cells);
    return n > 0 && C > 0 && C <= 2048LL * 1024;   // (tiles of the cell counters' chained prefix sum)
}
// a grid build (ws: mccnn_build_grid_workspace_bytes)
int grid_batch_item(GridItem& g, ScanItem& sc, ClearSpan& head, const float* pts, const int* batch_ids, const float* aabb_min,
                    const float* aabb_max, int n, int batch_size, int num_cells, int* new_idx, float* out_pts,
                    int* out_batch_ids, int* cell_indexs, int* inv_idx, void* ws, size_t ws_bytes) {
    const long long C = total_cells(batch_size, num_cells);
    if (!grid_batch_eligible(n, batch_size, num_cells)) return MCCNN_E_TOOLARGE;
    if (!ws || ws_bytes < mccnn_build_grid_workspace_bytes(n, batch_size, num_cells)) return MCCNN_E_WORKSPACE;
    Arena a(ws, ws_bytes);
    int* keys = a.take<int>((size_t)n);
    const size_t cntBytes = align_up((size_t)C * 4), scanBytes = scan_workspace_bytes((int)C);
    char* blk = a.take<char>(cntBytes + scanBytes);
    int* start = a.take<int>((size_t)C + 1);
    int* slot = a.take<int>((size_t)n);
    if (!keys || !blk || !start || !slot) return MCCNN_E_WORKSPACE;
    const int tiles = ceil_div(C, 2048);
    g = GridItem{pts, batch_ids, aabb_min, aabb_max, keys, (int*)blk, new_idx /* arrival ranks, then the final positions */, start, slot,
                 new_idx, out_pts, out_batch_ids, inv_idx, reinterpret_cast<int2*>(cell_indexs), n, batch_size, num_cells, (int)C,
                 (C <= MCCNN_HIST_LDS_BINS && n >= 4 * C) ? 1 : 0};
    sc = ScanItem{(const int*)blk, start, reinterpret_cast<unsigned long long*>(blk + cntBytes), start + C, nullptr, (int)C, tiles};
    head = clear_span(blk, cntBytes + align_up((size_t)(tiles + 1) * 8));
    return 0;
}
// a visiting order (ws: visiting_order_workspace_bytes)
size_t visiting_order_workspace_bytes(int m, int batch_size, int num_cells);
int order_batch_item(GridItem& g, ScanItem& sc, ClearSpan& head, const float* pts, const int* batch_ids, const float* aabb_min,
                     const float* aabb_max, int m, int batch_size, int num_cells, int* order, void* ws, size_t ws_bytes) {
    const long long C = total_cells(batch_size, num_cells);
    if (!grid_batch_eligible(m, batch_size, num_cells)) return MCCNN_E_TOOLARGE;
    if (!ws || ws_bytes < visiting_order_workspace_bytes(m, batch_size, num_cells)) return MCCNN_E_WORKSPACE;
    Arena a(ws, ws_bytes);
    int* keys = a.take<int>((size_t)m);
    int* arrival = a.take<int>((size_t)m);
    const size_t cntBytes = align_up((size_t)C * 4), scanBytes = scan_workspace_bytes((int)C);
    char* blk = a.take<char>(cntBytes + scanBytes);
    int* start = a.take<int>((size_t)C + 1);
    if (!keys || !arrival || !blk || !start) return MCCNN_E_WORKSPACE;
    const int tiles = ceil_div(C, 2048);
    g = GridItem{pts, batch_ids, aabb_min, aabb_max, keys, (int*)blk, arrival, start, order, nullptr, nullptr, nullptr, nullptr, nullptr,
                 m, batch_size, num_cells, (int)C, (C <= MCCNN_HIST_LDS_BINS && m >= 4 * C) ? 1 : 0};
    sc = ScanItem{(const int*)blk, start, reinterpret_cast<unsigned long long*>(blk + cntBytes), start + C, nullptr, (int)C, tiles};
    head = clear_span(blk, cntBytes + align_up((size_t)(tiles + 1) * 8));
    return 0;
}
static BatchBlocks grid_blocks(const GridBatch& gb, int count, bool rankOnly) {
    BatchBlocks bb;
    bb.count = count;
    int run = 0;
    for (int k = 0; k < count; ++k) {
        bb.first[k] = run;
        if (!rankOnly || gb.it[k].newIdx) run += ceil_div(gb.it[k].n, 256);
    }
    for (int k = count; k <= MCCNN_BATCH_MAX; ++k) bb.first[k] = run;
    return bb;
}
// phase 0: keys + histogram; phase 1 (after the prefix sums): park; phase 2: rank + move + cell table (grid builds only)
int launch_grid_batch_phase(const GridBatch& gb, int count, int phase, hipStream_t s) {
    const BatchBlocks bb = grid_blocks(gb, count, phase == 2);
    const int grid = bb.first[count];
    if (grid == 0) return 0;
    if (phase == 0) grid_keys_batch<<<grid, 256, 0, s>>>(gb, bb);
    else if (phase == 1) grid_park_batch<<<grid, 256, 0, s>>>(gb, bb);
    else grid_rank_move_batch<<<grid, 256, 0, s>>>(gb, bb);
    MCCNN_LAUNCHED();
    return 0;
}

// A cell-coherent VISITING ORDER of points that are not the gridded ones (the centres of another level: Poisson samples
// arrive phase by phase, all over the scene): position -> point id, points of one cell consecutive. Speed only -- no result
// depends on the order inside a cell, so the arrival order of the histogram's atomics is kept (keys_hist, prefix sum,
// park_ids: three launches; the stable sort of mccnn_sort_step1 + an inversion were six). ws as
// mccnn_sort_step1_workspace_bytes(m) + m ints of keys + m ints of arrival ranks; `order` [m].
size_t visiting_order_workspace_bytes(int m, int batch_size, int num_cells) {
    const size_t c = mccnn_sort_step1_workspace_bytes(m, batch_size, num_cells);
    return c == 0 ? 0 : c + 2 * align_up((size_t)(m > 0 ? m : 1) * 4);
}
ClearSpan visiting_order_head_span(int m, int batch_size, int num_cells, void* ws, size_t ws_bytes) {
    const long long C = total_cells(batch_size, num_cells);
    if (C <= 0 || C >= 0x7fffffffLL || !ws) return no_span();
    Arena a(ws, ws_bytes);
    if (!a.take<int>((size_t)(m > 0 ? m : 1)) || !a.take<int>((size_t)(m > 0 ? m : 1))) return no_span();
    const size_t cntBytes = align_up((size_t)C * 4);
    char* blk = a.take<char>(cntBytes + scan_workspace_bytes((int)C));
    if (!blk) return no_span();
    return clear_span(blk, cntBytes + scan_status_bytes((int)C));
}
int visiting_order(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int m, int batch_size,
                   int num_cells, int* order, void* ws, size_t ws_bytes, hipStream_t s, bool cleared) {
    if (m <= 0 || !order) return MCCNN_E_BADARG;
    const long long C = total_cells(batch_size, num_cells);
    if (C >= 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    if (!ws || ws_bytes < visiting_order_workspace_bytes(m, batch_size, num_cells)) return MCCNN_E_WORKSPACE;
    Arena a(ws, ws_bytes);
    int* keys = a.take<int>((size_t)m);
    int* arrival = a.take<int>((size_t)m);
    const size_t cntBytes = align_up((size_t)C * 4), scanBytes = scan_workspace_bytes((int)C);
    char* blk = a.take<char>(cntBytes + scanBytes);
    int* start = a.take<int>((size_t)C + 1);
    if (!keys || !arrival || !blk || !start) return MCCNN_E_WORKSPACE;
    int* cnt = (int*)blk;
    if (!cleared) {
        int rc = launch_clear_spans(clear_span(blk, cntBytes + scan_status_bytes((int)C)), no_span(), no_span(), s);
        if (rc) return rc;
    }
    const int blocks = ceil_div(m, 256);
    if (C <= MCCNN_HIST_LDS_BINS && m >= 4 * C)
        keys_hist_lds<<<blocks, 256, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, m, batch_size, num_cells, (int)C, keys, cnt, arrival, nullptr, no_span(), no_span());
    else
        keys_hist<<<blocks, 256, 0, s>>>(pts, batch_ids, aabb_min, aabb_max, m, batch_size, num_cells, keys, cnt, arrival, nullptr, no_span(), no_span());
    MCCNN_LAUNCHED();
    int rc = exclusive_scan_i32(cnt, start, (int)C, start + C, blk + cntBytes, s, true);
    if (rc) return rc;
    park_ids<<<blocks, 256, 0, s>>>(keys, start, arrival, m, order, nullptr);
    MCCNN_LAUNCHED();
    return 0;
}
}  // namespace mccnn
extern "C" {

int mccnn_build_grid(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int n,
                     int batch_size, int num_cells, int* new_idx, float* out_pts, int* out_batch_ids, int* cell_indexs,
                     int* inv_idx, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    return build_grid_fused(pts, batch_ids, aabb_min, aabb_max, n, batch_size, num_cells, new_idx, out_pts, out_batch_ids,
                            cell_indexs, inv_idx, ws, ws_bytes, (hipStream_t)stream, nullptr, false, no_span(), no_span());
}

int mccnn_permute_gather(const float* in, const int* idx, int n_idx, int num_feats, float* out,
                         mccnn_stream_t stream) {
    if (n_idx < 0 || num_feats <= 0) return MCCNN_E_BADARG;
    if (n_idx == 0) return 0;
    if (!in || !idx || !out) return MCCNN_E_BADARG;
    return launch_permute<true>(in, idx, n_idx, num_feats, out, (hipStream_t)stream);
}

int mccnn_permute_scatter(const float* in, const int* idx, int n_idx, int num_feats, float* out, int n_out,
                          int zero_fill, mccnn_stream_t stream) {
    if (n_idx < 0 || num_feats <= 0 || n_out < 0) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (zero_fill && n_out > 0) {
        if (!out) return MCCNN_E_BADARG;
        // (scatter into zeros -- GetSampledFeaturesGrad, poisson_sampling.cu:158-172 after its cudaMemset: the zero fill is
        // part of the op; rows no index points at stay zero)
        int rc = launch_zero_words(out, (size_t)n_out * num_feats, s);
        if (rc) return rc;
    }
    if (n_idx == 0) return 0;
    if (!in || !idx || !out) return MCCNN_E_BADARG;
    return launch_permute<false>(in, idx, n_idx, num_feats, out, s);
}

size_t mccnn_transform_indexs_workspace_bytes(int n) { return align_up((size_t)(n > 0 ? n : 1) * 4); }

static int transform_indexs_impl(const int* in_idx, int s_count, const int* new_idx, int n, int* out_idx, void* ws,
                                 size_t ws_bytes, mccnn_stream_t stream, const int* s_dev, const int* n_dev) {
    if (s_count < 0 || n < 0) return MCCNN_E_BADARG;
    if (s_count == 0) return 0;
    if (!in_idx || !new_idx || !out_idx || n == 0) return MCCNN_E_BADARG;
    if (!ws || ws_bytes < mccnn_transform_indexs_workspace_bytes(n)) return MCCNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int* inv = (int*)ws;
    invert_perm<<<ceil_div(n, 256), 256, 0, s>>>(new_idx, n, inv, n_dev);
    MCCNN_LAUNCHED();
    map_indexs<<<ceil_div(s_count, 256), 256, 0, s>>>(in_idx, s_count, inv, out_idx, s_dev);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_transform_indexs(const int* in_idx, int s_count, const int* new_idx, int n, int* out_idx, void* ws,
                           size_t ws_bytes, mccnn_stream_t stream) {
    return transform_indexs_impl(in_idx, s_count, new_idx, n, out_idx, ws, ws_bytes, stream, nullptr, nullptr);
}

int mccnn_transform_indexs_dn(const int* in_idx, int s_cap, const int* s_dev, const int* new_idx, int n_cap,
                              const int* n_dev, int* out_idx, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (!s_dev || !n_dev) return MCCNN_E_BADARG;
    return transform_indexs_impl(in_idx, s_cap, new_idx, n_cap, out_idx, ws, ws_bytes, stream, s_dev, n_dev);
}

}  // extern "C"
