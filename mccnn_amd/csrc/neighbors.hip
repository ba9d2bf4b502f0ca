// Fixed-radius neighbour search over the 27-cell window and the per-edge kernel density
// estimate. Replaces tf_ops/find_neighbors.cu and tf_ops/compute_pdf.cu.
#include "common.h"

namespace mccnn {

struct CentreCtx {
    float cx, cy, cz, R, T;  // T: squared-distance threshold equivalent to sqrt(d2) < R (common.h)
    int b, x, y, z;
};

__device__ __forceinline__ CentreCtx centre_ctx(const float* __restrict__ centres, const int* __restrict__ cb,
                                                const float* __restrict__ mn, const float* __restrict__ mx,
                                                int i, int nc, float radius, int scaleInv) {
    CentreCtx c;
    c.b = cb[i];
    c.cx = centres[(size_t)i * 3];
    c.cy = centres[(size_t)i * 3 + 1];
    c.cz = centres[(size_t)i * 3 + 2];
    float ext = max_extent(mn, mx, c.b);
    float cs = ext / (float)nc;
    c.R = scaleInv ? radius * ext : radius;  // find_neighbors.cu:73
    c.T = sqrt_threshold(c.R);
    c.x = cell_coord(c.cx, mn[c.b * 3], cs, nc);
    c.y = cell_coord(c.cy, mn[c.b * 3 + 1], cs, nc);
    c.z = cell_coord(c.cz, mn[c.b * 3 + 2], cs, nc);
    return c;
}

// THREE threads per centre -- one per z-slab of the 27-cell window (table entries 9*slab .. 9*slab+8, so slab order is
// the table order of find_neighbors.cu:282-291) -- visited in `order` (identity when null). FILL == false counts per
// (centre, slab), FILL == true writes (j, i) rows at the scanned per-slab offsets; inside a slab: table order, then
// ascending j. One thread per centre left the chip three quarters empty at 100k centres (1.5 waves per SIMD) and
// latency-bound; a slab per thread triples the parallelism and shortens every serial walk by three.
// The 9 cell ranges of a slab are fetched with independent loads and the candidates of a cell 4 at a time; with a
// cell-coherent visiting order the lanes of a wave read the same cells (broadcast loads, equal trip counts).
template <bool FILL>
__global__ __launch_bounds__(256) void neigh_walk(const float* __restrict__ centres, const int* __restrict__ cb, int m,
                                                  const float4* __restrict__ pts4, const int* __restrict__ cells,
                                                  const float* __restrict__ mn, const float* __restrict__ mx, int nc,
                                                  float radius, int scaleInv, const int* __restrict__ order,
                                                  int* __restrict__ counts3, const int* __restrict__ base3,
                                                  int* __restrict__ packed) {
    int tix = blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= 3 * m) return;
    const int ci = tix / 3, slab = tix - ci * 3;
    const int i = order ? order[ci] : ci;
    CentreCtx c = centre_ctx(centres, cb, mn, mx, i, nc, radius, scaleInv);
    int k = 0;
    int2* dst = FILL ? reinterpret_cast<int2*>(packed) + base3[(size_t)i * 3 + slab] : nullptr;
    const size_t cellBase = (size_t)c.b * nc * nc * nc;
    const int2* ct = reinterpret_cast<const int2*>(cells);
    const int Z = c.z + 1 - slab;  // offsets o = 9*slab .. 9*slab+8 share dz = 1 - slab
    int2 rng[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        int X = c.x + 1 - (u % 3), Y = c.y + 1 - (u / 3);
        bool ok = X >= 0 && X < nc && Y >= 0 && Y < nc && Z >= 0 && Z < nc;
        rng[u] = ok ? ct[cellBase + (size_t)X * nc * nc + (size_t)Y * nc + Z] : make_int2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        const int j0 = rng[u].x, j1 = rng[u].y;
        for (int j = j0; j < j1; j += 4) {
            float d[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                int jj = min(j + v, j1 - 1);
                float4 p = pts4[jj];
                d[v] = point_dist2(p.x, p.y, p.z, c.cx, c.cy, c.cz);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (j + v < j1 && d[v] < c.T) {
                    if (FILL) dst[k] = make_int2(j + v, i);
                    ++k;
                }
            }
        }
    }
    if (!FILL) counts3[(size_t)i * 3 + slab] = k;
}

// start_idx[i] = offset of centre i's first slab
__global__ __launch_bounds__(256) void slab_to_start(const int* __restrict__ base3, int m, int* __restrict__ startIdx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) startIdx[i] = base3[(size_t)i * 3];
}

// [N,3] -> [N] float4: one 16-byte load per candidate instead of three 4-byte loads at a 12-byte stride
__global__ __launch_bounds__(256) void pad_points(const float* __restrict__ pts, int n, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2], 0.f);
}

__global__ __launch_bounds__(256) void invert_perm_k(const int* __restrict__ newIdx, int n, int* __restrict__ inv) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[newIdx[i]] = i;
}

// ------------------------------------------------------------------ compute_pdf.cu:40-94
// Mode 0: one thread per edge (j, i), the reference's arithmetic (3 double-precision exps per pair).
__global__ __launch_bounds__(256) void pdf_edges_ref(const float* __restrict__ pts, const int* __restrict__ bids,
                                                 const int* __restrict__ startIdx, int m,
                                                 const int2* __restrict__ packed, int e, const float* __restrict__ mn,
                                                 const float* __restrict__ mx, float window, float radius,
                                                 int scaleInv, float* __restrict__ pdfs) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= e) return;
    int2 pr = packed[t];
    int cur = pr.x, centre = pr.y;
    float cx = pts[(size_t)cur * 3], cy = pts[(size_t)cur * 3 + 1], cz = pts[(size_t)cur * 3 + 2];
    int b = bids[cur];
    float ext = max_extent(mn, mx, b);
    float R = scaleInv ? radius * ext : radius;
    int i0 = startIdx[centre];
    int i1 = (centre < m - 1) ? startIdx[centre + 1] : e;
    const float h = window;
    const float invH = 1 / h;
    const float invRadH = (float)(1.0 / (double)(R * h));  // compute_pdf.cu:74
    float pdf = 0.0f;
    for (int it = i0; it < i1; ++it) {
        size_t q = (size_t)packed[it].x * 3;
        float d0 = (pts[q] - cx) * invRadH;
        float d1 = (pts[q + 1] - cy) * invRadH;
        float d2 = (pts[q + 2] - cz) * invRadH;
        // compute_pdf.cu:85-88, double sub-expressions rounded to float per statement
        float g = (float)((double)invH * ((0.39894228) * exp((-0.5) * (double)d0 * (double)d0)));
        g = (float)((double)(g * invH) * ((0.39894228) * exp((-0.5) * (double)d1 * (double)d1)));
        g = (float)((double)(g * invH) * ((0.39894228) * exp((-0.5) * (double)d2 * (double)d2)));
        pdf += g;
    }
    pdfs[t] = pdf / ((float)i1 - i0);  // compute_pdf.cu:92
}

// ---- single-precision KDE (mode 1) ------------------------------------------------------------------
// Pre-pass: one float4 per edge with the neighbour's coordinates already scaled by 1/(R_b h); the pair loop
// then costs one (wave-broadcast) 16-byte load and ~9 VALU instructions per pair.
__global__ __launch_bounds__(256) void pdf_scaled_coords(const float* __restrict__ pts, const int* __restrict__ bids,
                                                         const int2* __restrict__ packed, int e,
                                                         const float* __restrict__ mn, const float* __restrict__ mx,
                                                         float window, float radius, int scaleInv,
                                                         float4* __restrict__ sc) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= e) return;
    int j = packed[t].x;
    float R = scaleInv ? radius * max_extent(mn, mx, bids[j]) : radius;
    float s = (float)(1.0 / (double)(R * window));
    const float* p = pts + (size_t)j * 3;
    sc[t] = make_float4(p[0] * s, p[1] * s, p[2] * s, 0.f);
}

// Mode 1, row form: one wave per centre. Lanes hold the row's own points (a), the loop runs over the row's points b
// with x_b wave-uniform: scalar loads into SGPRs, no vector memory or LDS traffic in the inner loop and no divergence
// between rows of different length (a thread per edge walks its row with one gather per pair and idles while a
// longer row in the same wave finishes: 136 -> 100 us on the 100k room). Same arithmetic and summation order as the
// thread-per-edge form it replaced: identical results. (Reading the pre-scaled coordinates per POINT through the
// neighbour index instead of the per-edge copy makes the scalar loads dependent: measured slower, 124 us.)
__global__ __launch_bounds__(256) void pdf_rows(const float4* __restrict__ sc, const int* __restrict__ startIdx, int m,
                                                int e, float window, float* __restrict__ pdfs) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int lane = threadIdx.x & 63;
    const int i0 = __builtin_amdgcn_readfirstlane(startIdx[row]);
    const int i1 = __builtin_amdgcn_readfirstlane((row < m - 1) ? startIdx[row + 1] : e);
    const int k = i1 - i0;
    if (k <= 0) return;
    const float c = -0.5f * 1.44269504088896f;  // exp(-x/2) = exp2(c x)
    const float invH = 1.0f / window;
    const float g1 = invH * 0.39894228f;
    const float norm = g1 * g1 * g1;
    const float4* __restrict__ rowp = sc + i0;
    for (int a0 = 0; a0 < k; a0 += 64) {
        const int a = a0 + lane;
        const float4 me = rowp[min(a, k - 1)];
        float acc = 0.f;
        int b = 0;
        for (; b + 4 <= k; b += 4) {
            const float4 q0 = rowp[b], q1 = rowp[b + 1], q2 = rowp[b + 2], q3 = rowp[b + 3];
            float dx, dy, dz;
            dx = q0.x - me.x; dy = q0.y - me.y; dz = q0.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = q1.x - me.x; dy = q1.y - me.y; dz = q1.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = q2.x - me.x; dy = q2.y - me.y; dz = q2.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = q3.x - me.x; dy = q3.y - me.y; dz = q3.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
        for (; b < k; ++b) {
            const float4 q0 = rowp[b];
            const float dx = q0.x - me.x, dy = q0.y - me.y, dz = q0.z - me.z;
            acc += __builtin_amdgcn_exp2f(c * fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
        if (a < k) pdfs[i0 + a] = (acc * norm) / ((float)i1 - i0);
    }
}

}  // namespace mccnn

using namespace mccnn;

extern "C" {

size_t mccnn_find_neighbors_workspace_bytes(int m, int n) {
    size_t m3 = 3 * (size_t)(m > 0 ? m : 1);
    return align_up(m3 * 4) + scan_workspace_bytes((int)m3) + align_up((size_t)(n > 0 ? n : 1) * sizeof(float4)) + 256;
}

struct NeighWs {
    int* counts3;  // per (centre, z-slab) hit counts, scanned in place to output offsets
    void* scanws;
    float4* pts4;
};
static bool neigh_ws(void* ws, size_t ws_bytes, int m, int n, NeighWs& w) {
    if (!ws || ws_bytes < mccnn_find_neighbors_workspace_bytes(m, n)) return false;
    size_t m3 = 3 * (size_t)(m > 0 ? m : 1);
    Arena a(ws, ws_bytes);
    w.counts3 = a.take<int>(m3);
    w.scanws = a.take<char>(scan_workspace_bytes((int)m3));
    w.pts4 = a.take<float4>((size_t)(n > 0 ? n : 1));
    return w.counts3 && w.scanws && w.pts4;
}

int mccnn_find_neighbors_count(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                               int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                               int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                               int* start_idx, int* total_dev, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (m < 0 || n < 0 || batch_size <= 0 || num_cells <= 0 || !(radius > 0.0f) || !total_dev) return MCCNN_E_BADARG;
    if ((long long)m * 3 >= 0x7fffffffLL) return MCCNN_E_TOOLARGE;
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) {
        MCCNN_HIP(hipMemsetAsync(total_dev, 0, sizeof(int), s));
        return 0;
    }
    if (!centres || !centre_batch_ids || !cell_indexs || !aabb_min || !aabb_max || !start_idx || (n > 0 && !sorted_pts))
        return MCCNN_E_BADARG;
    NeighWs w;
    if (!neigh_ws(ws, ws_bytes, m, n, w)) return MCCNN_E_WORKSPACE;
    if (n > 0) {
        pad_points<<<ceil_div(n, 256), 256, 0, s>>>(sorted_pts, n, w.pts4);
        MCCNN_LAUNCHED();
    }
    neigh_walk<false><<<ceil_div(3LL * m, 256), 256, 0, s>>>(centres, centre_batch_ids, m, w.pts4, cell_indexs, aabb_min,
                                                             aabb_max, num_cells, radius, scale_inv, centre_order,
                                                             w.counts3, nullptr, nullptr);
    MCCNN_LAUNCHED();
    int rc = exclusive_scan_i32(w.counts3, w.counts3, 3 * m, total_dev, w.scanws, s);
    if (rc) return rc;
    slab_to_start<<<ceil_div(m, 256), 256, 0, s>>>(w.counts3, m, start_idx);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_find_neighbors_fill(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                              int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max,
                              int batch_size, int num_cells, float radius, int scale_inv, const int* centre_order,
                              const int* start_idx, int e, int* packed, void* ws, size_t ws_bytes,
                              mccnn_stream_t stream) {
    if (m < 0 || n < 0 || e < 0 || batch_size <= 0 || num_cells <= 0 || !(radius > 0.0f)) return MCCNN_E_BADARG;
    if (m == 0 || e == 0) return 0;
    if (!centres || !centre_batch_ids || !sorted_pts || !cell_indexs || !aabb_min || !aabb_max || !start_idx || !packed)
        return MCCNN_E_BADARG;
    NeighWs w;
    // same workspace as the count call: it holds the padded points and the scanned per-slab offsets
    if (!neigh_ws(ws, ws_bytes, m, n, w)) return MCCNN_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    neigh_walk<true><<<ceil_div(3LL * m, 256), 256, 0, s>>>(centres, centre_batch_ids, m, w.pts4, cell_indexs, aabb_min,
                                                            aabb_max, num_cells, radius, scale_inv, centre_order, nullptr,
                                                            w.counts3, packed);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_invert_permutation(const int* new_idx, int n, int* inv, mccnn_stream_t stream) {
    if (n < 0) return MCCNN_E_BADARG;
    if (n == 0) return 0;
    if (!new_idx || !inv) return MCCNN_E_BADARG;
    invert_perm_k<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(new_idx, n, inv);
    MCCNN_LAUNCHED();
    return 0;
}

size_t mccnn_compute_pdf_workspace_bytes(int e, int mode) {
    return (mode != 0 && e > 0) ? align_up((size_t)e * sizeof(float4)) : 256;
}

int mccnn_compute_pdf(const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx, int m,
                      const int* packed, int e, const float* aabb_min, const float* aabb_max, int batch_size,
                      float window, float radius, int scale_inv, int mode, float* pdfs, void* ws, size_t ws_bytes,
                      mccnn_stream_t stream) {
    if (m < 0 || e < 0 || batch_size <= 0 || !(radius > 0.0f) || !(window > 0.0f)) return MCCNN_E_BADARG;
    if (e == 0) return 0;
    if (!sorted_pts || !sorted_batch_ids || !start_idx || !packed || !aabb_min || !aabb_max || !pdfs || m == 0)
        return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int2* pk = reinterpret_cast<const int2*>(packed);
    if (mode == 0)
        pdf_edges_ref<<<ceil_div(e, 256), 256, 0, s>>>(sorted_pts, sorted_batch_ids, start_idx, m, pk, e, aabb_min,
                                                      aabb_max, window, radius, scale_inv, pdfs);
    else {
        if (!ws || ws_bytes < mccnn_compute_pdf_workspace_bytes(e, mode)) return MCCNN_E_WORKSPACE;
        float4* sc = (float4*)ws;
        pdf_scaled_coords<<<ceil_div(e, 256), 256, 0, s>>>(sorted_pts, sorted_batch_ids, pk, e, aabb_min, aabb_max,
                                                          window, radius, scale_inv, sc);
        MCCNN_LAUNCHED();
        pdf_rows<<<ceil_div(m, 4), 256, 0, s>>>(sc, start_idx, m, e, window, pdfs);
    }
    MCCNN_LAUNCHED();
    return 0;
}

}  // extern "C"
