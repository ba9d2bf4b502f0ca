// TEST INFRASTRUCTURE ONLY -- CPU oracle for the Monte-Carlo convolution hot path.
//
// This file is a sequential, single-precision restatement of the algorithms of
// viscom-ulm/MCCNN's tf_ops CUDA kernels. It is *not* the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The
// product path (mccnn_amd/) never links, imports or calls anything in oracle/.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, is
// GPU-only (every op registered DEVICE_GPU) and needs TensorFlow 1.x + nvcc, so
// it can neither be imported nor compiled in this image. This oracle is pinned
// instead by (a) the structural known answers that do exist in the reference
// (the two 27-entry offset tables, the Gaussian constant, the numCells formula),
// (b) an independent NumPy float64 brute-force cross-check (tests/test_oracle_cpu.py),
// (c) finite-difference gradient checks and (d) invariants -- see tests/.
//
// Canonical order. Two reference kernels depend on atomic arrival order
// (sort_gpu.cu:170 intra-cell order; poisson_sampling.cu:115 output order). The
// oracle fixes the order a *sequential* execution of those kernels produces:
// ascending original index within a grid cell, and for Poisson sampling
// batch -> phase 0..26 -> cell in launch-linear thread order -> point order.
//
// Arithmetic: the geometry (cell coordinates, distances, KDE) is evaluated in the
// precision and order the reference source spells out (build with -ffp-contract=off);
// doubles appear only where the reference promotes (compute_pdf.cu:72-92). The kernel
// MLP follows the reference as nvcc compiles it: fused multiply-add chains, see
// layer1() / dot8() below.
//
// With -fopenmp the per-centre / per-point loops run in parallel (used only for
// the cpu_baseline timing leg); integer outputs are unchanged by that, float
// weight-gradient sums are re-associated.

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MLP 8  // BLOCK_MLP_SIZE, genCompileScript.py:20

namespace {

// find_neighbors.cu:282-291
const int kCellOffsets[27][3] = {
    {1, 1, 1},   {0, 1, 1},   {-1, 1, 1},  {1, 0, 1},   {0, 0, 1},   {-1, 0, 1},  {1, -1, 1},
    {0, -1, 1},  {-1, -1, 1}, {1, 1, 0},   {0, 1, 0},   {-1, 1, 0},  {1, 0, 0},   {0, 0, 0},
    {-1, 0, 0},  {1, -1, 0},  {0, -1, 0},  {-1, -1, 0}, {1, 1, -1},  {0, 1, -1},  {-1, 1, -1},
    {1, 0, -1},  {0, 0, -1},  {-1, 0, -1}, {1, -1, -1}, {0, -1, -1}, {-1, -1, -1}};

// poisson_sampling.cu:192-196
const int kCellOffsetsPool[27][3] = {
    {1, 1, -1},  {0, -1, 1},  {0, 1, 1},  {0, 1, 0},  {0, 0, 1},   {0, -1, 0},   {-1, 1, -1},
    {0, -1, -1}, {1, 0, 0},   {1, -1, 1}, {1, 0, 1},  {-1, 1, 1},  {-1, 0, 0},   {1, -1, -1},
    {0, 1, -1},  {-1, -1, 0}, {-1, 1, 0}, {0, 0, 0},  {0, 0, -1},  {1, 1, 0},    {1, 0, -1},
    {1, -1, 0},  {-1, 0, 1},  {1, 1, 1},  {-1, 0, -1}, {-1, -1, -1}, {-1, -1, 1}};

inline float maxExtent(const float* mn, const float* mx, int b) {
    // sort_gpu.cu:50-52 (same expression in every kernel)
    return std::max(std::max(mx[b * 3] - mn[b * 3], mx[b * 3 + 1] - mn[b * 3 + 1]),
                    mx[b * 3 + 2] - mn[b * 3 + 2]);
}

inline int cellCoord(float p, float mn, float cellSize, int nc) {
    // sort_gpu.cu:55
    return std::max(std::min((int)std::floor((p - mn) / cellSize), nc - 1), 0);
}

inline float relu(float x) { return x > 0.0f ? x : 0.0f; }  // max(x, 0.0), spatial_conv.cu:49

// Kernel-MLP arithmetic as the reference is COMPILED, not as a strict C reading of its source: nvcc contracts
// float `a*b + c` into one fused multiply-add by default (-fmad=true), and where the source promotes to double
// (`auxResult += max(x, 0.0)*w`, spatial_conv.cu:375,390: the double product of two floats is exact, the sum is
// rounded once on the assignment to the float accumulator) the result is a fused multiply-add as well. So every
// hidden-layer sum is a k-ordered fmaf chain that starts at 0, and the bias is added last, where the source adds it
// (spatial_conv.cu:57-62, 68-73, 372-378). This file is built with -ffp-contract=off: the chains are spelled out.
//
// Layer 1 (spatial_conv.cu:49-52, 363-366): c0*w0 + c1*w1 + c2*w2 + b, contracted left to right.
inline float layer1(const float d[3], const float* w, float b) {
    float t = d[0] * w[0];
    t = std::fmaf(d[1], w[1], t);
    t = std::fmaf(d[2], w[2], t);
    return t + b;
}
// sum_{k<8} x[k] * w[k*stride], sequential in k, accumulator starts at 0 (spatial_conv.cu:57-61, 428-433)
inline float dot8(const float* x, const float* w, int stride) {
    float a = 0.0f;
    for (int k = 0; k < MLP; ++k) a = std::fmaf(x[k], w[k * stride], a);
    return a;
}

}  // namespace

extern "C" {

int orc_block_size() { return MLP; }

const int* orc_cell_offsets() { return &kCellOffsets[0][0]; }
const int* orc_cell_offsets_pool() { return &kCellOffsetsPool[0][0]; }

// aabb_gpu.cu:57-117,121-140. The reference's tail-drop quirk (work nested in
// if(idx<N), aabb_gpu.cu:69-96) is not replicated.
int orc_compute_aabb(const float* pts, const int* bids, int n, int B, int scaleInv, float* mn,
                     float* mx) {
    for (int i = 0; i < B * 3; ++i) {
        mn[i] = FLT_MAX;
        mx[i] = -FLT_MAX;
    }
    for (int i = 0; i < n; ++i) {
        int b = bids[i];
        if (b < 0 || b >= B) return -2;
        for (int d = 0; d < 3; ++d) {
            mn[b * 3 + d] = std::fmin(mn[b * 3 + d], pts[i * 3 + d]);
            mx[b * 3 + d] = std::fmax(mx[b * 3 + d], pts[i * 3 + d]);
        }
    }
    if (!scaleInv) {  // aabb_gpu.cu:104-114: every row <- whole-batch box
        float gmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, gmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int b = 0; b < B; ++b)
            for (int d = 0; d < 3; ++d) {
                gmn[d] = std::fmin(gmn[d], mn[b * 3 + d]);
                gmx[d] = std::fmax(gmx[d], mx[b * 3 + d]);
            }
        for (int b = 0; b < B; ++b)
            for (int d = 0; d < 3; ++d) {
                mn[b * 3 + d] = gmn[d];
                mx[b * 3 + d] = gmx[d];
            }
    }
    return 0;
}

// sort_gpu.cu:374-420
int orc_num_cells(const float* mn, const float* mx, int B, float cellSize, int scaleInv) {
    (void)B;
    int nc;
    if (scaleInv) {
        nc = (int)(1.0f / cellSize);
    } else {
        float ext = maxExtent(mn, mx, 0);
        nc = (int)(ext / cellSize);
    }
    return nc == 0 ? 1 : nc;
}

// sort_gpu.cu:35-61 (keys) + :69-174 (counting sort destination, canonical =
// ascending original index inside a cell).
int orc_sort_step1(const float* pts, const int* bids, const float* mn, const float* mx, int n,
                   int B, int nc, int* keys, int* newIdx) {
    long long C = (long long)B * nc * nc * nc;
    if (C <= 0 || C > 0x7fffffffLL) return -3;
    std::vector<int> cnt((size_t)C + 1, 0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int b = bids[i];
        float cs = maxExtent(mn, mx, b) / (float)nc;
        int x = cellCoord(pts[i * 3], mn[b * 3], cs, nc);
        int y = cellCoord(pts[i * 3 + 1], mn[b * 3 + 1], cs, nc);
        int z = cellCoord(pts[i * 3 + 2], mn[b * 3 + 2], cs, nc);
        keys[i] = b * nc * nc * nc + x * nc * nc + y * nc + z;  // sort_gpu.cu:59
    }
    for (int i = 0; i < n; ++i) cnt[(size_t)keys[i] + 1]++;
    for (long long c = 0; c < C; ++c) cnt[c + 1] += cnt[c];
    for (int i = 0; i < n; ++i) newIdx[i] = cnt[keys[i]]++;
    return 0;
}

// sort_gpu.cu:192-248,473-497
int orc_sort_step2(const float* pts, const int* bids, const float* feats, const int* keys,
                   const int* newIdx, int n, int F, int B, int nc, float* oPts, int* oBids,
                   float* oFeats, int* cellIdx) {
    long long C = (long long)B * nc * nc * nc;
    std::memset(cellIdx, 0, sizeof(int) * 2 * (size_t)C);
    std::vector<int> skeys(n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int p = newIdx[i];
        oPts[p * 3] = pts[i * 3];
        oPts[p * 3 + 1] = pts[i * 3 + 1];
        oPts[p * 3 + 2] = pts[i * 3 + 2];
        for (int f = 0; f < F; ++f) oFeats[(size_t)p * F + f] = feats[(size_t)i * F + f];
        skeys[p] = keys[i];
        oBids[p] = bids[i];
    }
    for (int p = 0; p < n; ++p) {  // save_indexs, sort_gpu.cu:225-248
        int k = skeys[p];
        if (p == 0)
            cellIdx[2 * (size_t)k] = 0;
        else if (k != skeys[p - 1])
            cellIdx[2 * (size_t)k] = p;
        if (p + 1 >= n)
            cellIdx[2 * (size_t)k + 1] = n;
        else if (k != skeys[p + 1])
            cellIdx[2 * (size_t)k + 1] = p + 1;
    }
    return 0;
}

// out[i,:] = in[idx[i],:]  -- compute_gradients :260, sort_features_back :288,
// selectFeatureSamples poisson_sampling.cu:135
int orc_permute_gather(const float* in, const int* idx, int n, int F, float* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int f = 0; f < F; ++f) out[(size_t)i * F + f] = in[(size_t)idx[i] * F + f];
    return 0;
}

// out[idx[i],:] = in[i,:] -- sort_features_back_grad :311 (no zero fill) and
// selectFeatureSamplesGrad poisson_sampling.cu:158 (zero fill of nOut rows first).
int orc_permute_scatter(const float* in, const int* idx, int n, int F, float* out, int nOut,
                        int zeroFill) {
    if (zeroFill) std::memset(out, 0, sizeof(float) * (size_t)nOut * F);
    for (int i = 0; i < n; ++i)
        for (int f = 0; f < F; ++f) out[(size_t)idx[i] * F + f] = in[(size_t)i * F + f];
    return 0;
}

// sort_gpu.cu:332-362, sort_gpu.cc:496-533
int orc_transform_indexs(const int* inIdx, int s, const int* newIdx, int n, int* out) {
    std::vector<int> inv(n);
    for (int i = 0; i < n; ++i) inv[newIdx[i]] = i;
    for (int k = 0; k < s; ++k) out[k] = inv[inIdx[k]];
    return 0;
}

// find_neighbors.cu:40-113 (count), :199-264 (fill). Returns E via *total.
// startIdx is the exclusive prefix over centres; packed rows are (j, i).
static inline int walkNeighbors(const float* c, int b, const float* pts2, const int* cellIdx,
                                const float* mn, const float* mx, int nc, float radius,
                                int scaleInv, int centre, int* out) {
    float ext = maxExtent(mn, mx, b);
    float cs = ext / (float)nc;
    float R = scaleInv ? radius * ext : radius;
    int x = cellCoord(c[0], mn[b * 3], cs, nc);
    int y = cellCoord(c[1], mn[b * 3 + 1], cs, nc);
    int z = cellCoord(c[2], mn[b * 3 + 2], cs, nc);
    int k = 0;
    for (int o = 0; o < 27; ++o) {
        int cx = x + kCellOffsets[o][0], cy = y + kCellOffsets[o][1], cz = z + kCellOffsets[o][2];
        if (cx < 0 || cx >= nc || cy < 0 || cy >= nc || cz < 0 || cz >= nc) continue;
        size_t flat = (size_t)b * nc * nc * nc + (size_t)cx * nc * nc + (size_t)cy * nc + cz;
        int i0 = cellIdx[flat * 2], i1 = cellIdx[flat * 2 + 1];
        for (int j = i0; j < i1; ++j) {
            float dx = pts2[j * 3] - c[0], dy = pts2[j * 3 + 1] - c[1], dz = pts2[j * 3 + 2] - c[2];
            float d = std::sqrt(dx * dx + dy * dy + dz * dz);  // find_neighbors.cu:96
            if (d < R) {
                if (out) {
                    out[2 * k] = j;
                    out[2 * k + 1] = centre;
                }
                ++k;
            }
        }
    }
    return k;
}

int orc_find_neighbors_count(const float* centres, const int* cbids, int m, const float* pts2,
                             const int* cellIdx, const float* mn, const float* mx, int B, int nc,
                             float radius, int scaleInv, int* startIdx, int* total) {
    (void)B;
    std::vector<int> cnt(m);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < m; ++i)
        cnt[i] = walkNeighbors(&centres[i * 3], cbids[i], pts2, cellIdx, mn, mx, nc, radius,
                               scaleInv, i, nullptr);
    long long acc = 0;
    for (int i = 0; i < m; ++i) {
        startIdx[i] = (int)acc;
        acc += cnt[i];
    }
    if (acc > 0x7fffffffLL) return -3;
    *total = (int)acc;
    return 0;
}

int orc_find_neighbors_fill(const float* centres, const int* cbids, int m, const float* pts2,
                            const int* cellIdx, const float* mn, const float* mx, int B, int nc,
                            float radius, int scaleInv, const int* startIdx, int* packed) {
    (void)B;
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < m; ++i)
        walkNeighbors(&centres[i * 3], cbids[i], pts2, cellIdx, mn, mx, nc, radius, scaleInv, i,
                      &packed[2 * (size_t)startIdx[i]]);
    return 0;
}

// compute_pdf.cu:40-94. pts/bids are the *sorted* point list the packed j index.
int orc_compute_pdf(const float* pts, const int* bids, const int* startIdx, int m,
                    const int* packed, int e, const float* mn, const float* mx, float window,
                    float radius, int scaleInv, float* pdfs, int exactCount) {
    // exactCount: compute_pdf.cu:92 divides by `(float)end - start`, a FLOAT subtraction. Beyond 2^24 edges both offsets
    // round to multiples of 2 (then 4, ...): rows of 1..3 neighbours get a count of 0, 2 or 4 and the reference writes
    // inf or a value off by a factor -- a defect of the reference at sizes it was never run at (BASELINE cfg4 as one
    // batch of 8 rooms has 36 M edges). The default (0) keeps the reference expression bit for bit; 1 subtracts the
    // integers first, which is what the HIP kernels do at every size (identical below 2^24 edges).
#pragma omp parallel for schedule(dynamic, 1024)
    for (int t = 0; t < e; ++t) {
        int cur = packed[2 * (size_t)t];
        float cx = pts[cur * 3], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
        int b = bids[cur];
        float ext = maxExtent(mn, mx, b);
        float R = scaleInv ? radius * ext : radius;
        int centre = packed[2 * (size_t)t + 1];
        int i0 = startIdx[centre];
        int i1 = (centre < m - 1) ? startIdx[centre + 1] : e;
        const float h = window;
        const float invH = 1 / h;
        const float invRadH = (float)(1.0 / (R * h));  // compute_pdf.cu:74 (double divide of a float product)
        float pdf = 0.0f;
        for (int it = i0; it < i1; ++it) {
            int q = packed[2 * (size_t)it] * 3;
            float d0 = (pts[q] - cx) * invRadH;
            float d1 = (pts[q + 1] - cy) * invRadH;
            float d2 = (pts[q + 2] - cz) * invRadH;
            // compute_pdf.cu:85-88: double sub-expressions, rounded to float per statement
            float g = (float)(invH * ((0.39894228) * std::exp((-0.5) * d0 * d0)));
            g = (float)(g * invH * ((0.39894228) * std::exp((-0.5) * d1 * d1)));
            g = (float)(g * invH * ((0.39894228) * std::exp((-0.5) * d2 * d2)));
            pdf += g;
        }
        pdfs[t] = exactCount ? pdf / (float)(i1 - i0) : pdf / ((float)i1 - i0);  // compute_pdf.cu:92
    }
    return 0;
}

// poisson_sampling.cu:51-124 (selectSamples), :175-230 (launch order).
// Outputs are written in the canonical sequential order; returns S via *numSel.
int orc_poisson_sampling(const float* pts, const int* bids, int n, const int* cellIdx,
                         const float* mn, const float* mx, int B, int nc, float radius,
                         int scaleInv, float* oPts, int* oBids, int* oIdx, int* numSel) {
    (void)bids;
    std::vector<unsigned char> sel(n, 0);
    int numGroups = nc / 3 + ((nc % 3 != 0) ? 1 : 0);
    int numBlocks = numGroups / 4 + ((numGroups % 4 != 0) ? 1 : 0);
    int dim = numBlocks * 4;
    int s = 0;
    for (int b = 0; b < B; ++b) {
        float ext = maxExtent(mn, mx, b);
        float R = scaleInv ? radius * ext : radius;
        for (int ph = 0; ph < 27; ++ph) {
            // launch-linear order: block z,y,x then thread z,y,x, x fastest
            for (int bz = 0; bz < numBlocks; ++bz)
            for (int by = 0; by < numBlocks; ++by)
            for (int bx = 0; bx < numBlocks; ++bx)
            for (int tz = 0; tz < 4; ++tz)
            for (int ty = 0; ty < 4; ++ty)
            for (int tx = 0; tx < 4; ++tx) {
                int gx = tx + bx * 4, gy = ty + by * 4, gz = tz + bz * 4;
                (void)dim;
                int xC = gx * 3 + 1 + kCellOffsetsPool[ph][0];
                int yC = gy * 3 + 1 + kCellOffsetsPool[ph][1];
                int zC = gz * 3 + 1 + kCellOffsetsPool[ph][2];
                // poisson_sampling.cu:74 tests only the upper bound; a negative
                // coordinate cannot occur (3g+1-1 >= 0).
                if (!(xC < nc && yC < nc && zC < nc)) continue;
                size_t cell = (size_t)b * nc * nc * nc + (size_t)xC * nc * nc + (size_t)yC * nc + zC;
                int p0 = cellIdx[cell * 2], p1 = cellIdx[cell * 2 + 1];
                for (int i = p0; i < p1; ++i) {
                    float c0 = pts[i * 3], c1 = pts[i * 3 + 1], c2 = pts[i * 3 + 2];
                    bool collision = false;
                    for (int o = 0; o < 27 && !collision; ++o) {
                        int cx = xC + kCellOffsetsPool[o][0], cy = yC + kCellOffsetsPool[o][1],
                            cz = zC + kCellOffsetsPool[o][2];
                        if (cx < 0 || cx >= nc || cy < 0 || cy >= nc || cz < 0 || cz >= nc) continue;
                        size_t flat = (size_t)b * nc * nc * nc + (size_t)cx * nc * nc + (size_t)cy * nc + cz;
                        int j0 = cellIdx[flat * 2], j1 = cellIdx[flat * 2 + 1];
                        for (int j = j0; j < j1 && !collision; ++j) {
                            float dx = pts[j * 3] - c0, dy = pts[j * 3 + 1] - c1, dz = pts[j * 3 + 2] - c2;
                            float d = std::sqrt(dx * dx + dy * dy + dz * dz);
                            if (d < R && sel[j]) collision = true;
                        }
                    }
                    if (!collision) {
                        sel[i] = 1;
                        oPts[s * 3] = c0;
                        oPts[s * 3 + 1] = c1;
                        oPts[s * 3 + 2] = c2;
                        oBids[s] = b;
                        oIdx[s] = i;
                        ++s;
                    }
                }
            }
        }
    }
    *numSel = s;
    return 0;
}

// ---------------------------------------------------------------------------
// spatial_conv forward: spatial_conv.cu:24-79 (combin), :178-230 (no-combin),
// :104-176 / :254-325 (edge set-up), :796-871 (zero-init + launch).
// Weight layouts are the flat ones the kernels index (spatial_conv.cu:172):
//   w1[nu*3+d], b1[nu], w2[q*64 + n*8 + m], b2[nu], w3[q*64 + n*8 + m], b3[nu].
// Edge contributions are accumulated in edge order (one of the orders the
// reference's atomicAdd may produce).
// ---------------------------------------------------------------------------
int orc_spatial_conv_fwd(const float* pts, const float* feats, const int* bids, const float* pdfs,
                         const float* samples, const int* startIdx, const int* packed,
                         const float* mn, const float* mx, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3,
                         int n, int m, int e, int Fin, int Fout, int combin, float radius,
                         int scaleInv, int avg, float* out) {
    (void)n;
    int neuronsOut = combin ? Fin * Fout : Fin;
    int nb = neuronsOut / MLP + ((neuronsOut % MLP != 0) ? 1 : 0);
    int outF = combin ? Fout : Fin;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < m; ++i) {
        float* o = &out[(size_t)i * outF];
        for (int f = 0; f < outF; ++f) o[f] = 0.0f;
        int e0 = startIdx[i];
        int e1 = (i < m - 1) ? startIdx[i + 1] : e;
        float K = avg ? (float)(e1 - e0) : 1.0f;
        for (int t = e0; t < e1; ++t) {
            int j = packed[2 * (size_t)t];
            int b = bids[j];
            float ext = maxExtent(mn, mx, b);
            float R = scaleInv ? radius * ext : radius;
            float d[3] = {(pts[j * 3] - samples[i * 3]) / R, (pts[j * 3 + 1] - samples[i * 3 + 1]) / R,
                          (pts[j * 3 + 2] - samples[i * 3 + 2]) / R};
            float pdf = pdfs[t];
            for (int q = 0; q < nb; ++q) {
                int off = q * MLP;
                float h1[MLP], h2[MLP];
                for (int t8 = 0; t8 < MLP; ++t8) {
                    int nu = off + t8;
                    h1[t8] = relu(layer1(d, &w1[nu * 3], b1[nu]));
                }
                for (int t8 = 0; t8 < MLP; ++t8) h2[t8] = relu(dot8(h1, &w2[off * MLP + t8 * MLP], 1) + b2[off + t8]);
                for (int t8 = 0; t8 < MLP; ++t8) {
                    int nu = off + t8;
                    if (nu >= neuronsOut) continue;
                    float a = dot8(h2, &w3[off * MLP + t8 * MLP], 1);
                    a = a + b3[nu];
                    int fin = nu % Fin;
                    int fo = combin ? nu / Fin : fin;
                    o[fo] += (feats[(size_t)j * Fin + fin] * a) / (pdf * K);
                }
            }
        }
    }
    return 0;
}

// spatial_conv backward: spatial_conv.cu:327-445 (combin), :563-680 (no-combin),
// :873-966 (zero-init + launch). Gradients for padded output neurons
// (nu >= neuronsOut) are left uninitialised by the reference
// (spatial_conv.cu:921,924); the oracle writes zeros there.
int orc_spatial_conv_bwd(const float* pts, const float* feats, const int* bids, const float* pdfs,
                         const float* samples, const int* startIdx, const int* packed,
                         const float* mn, const float* mx, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3,
                         const float* outGrad, int n, int m, int e, int Fin, int Fout, int combin,
                         float radius, int scaleInv, int avg, float* featGrad, float* dw1,
                         float* db1, float* dw2, float* db2, float* dw3, float* db3) {
    int neuronsOut = combin ? Fin * Fout : Fin;
    int nb = neuronsOut / MLP + ((neuronsOut % MLP != 0) ? 1 : 0);
    int outF = combin ? Fout : Fin;
    size_t nn = (size_t)nb * MLP;
    std::memset(featGrad, 0, sizeof(float) * (size_t)n * Fin);
    std::memset(dw1, 0, sizeof(float) * 3 * nn);
    std::memset(db1, 0, sizeof(float) * nn);
    std::memset(dw2, 0, sizeof(float) * MLP * nn);
    std::memset(db2, 0, sizeof(float) * nn);
    std::memset(dw3, 0, sizeof(float) * MLP * nn);
    std::memset(db3, 0, sizeof(float) * nn);

#ifdef _OPENMP
    int nthr = omp_get_max_threads();
#else
    int nthr = 1;
#endif
    size_t wsz = 3 * nn + nn + MLP * nn + nn + MLP * nn + nn;
    // The reference sums the parameter gradients with float atomics in arrival order (spatial_conv.cu:399-444): the
    // result carries an order-dependent rounding error of ~1e-4 relative on large neighbour lists. The oracle keeps
    // the reference's float products and adds them in double, so that it is the order-independent target.
    std::vector<double> priv((size_t)nthr * wsz, 0.0);

#pragma omp parallel
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double* pw1 = &priv[(size_t)tid * wsz];
        double* pb1 = pw1 + 3 * nn;
        double* pw2 = pb1 + nn;
        double* pb2 = pw2 + MLP * nn;
        double* pw3 = pb2 + nn;
        double* pb3 = pw3 + MLP * nn;
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < m; ++i) {
            int e0 = startIdx[i];
            int e1 = (i < m - 1) ? startIdx[i + 1] : e;
            float K = avg ? (float)(e1 - e0) : 1.0f;
            const float* g = &outGrad[(size_t)i * outF];
            for (int t = e0; t < e1; ++t) {
                int j = packed[2 * (size_t)t];
                int b = bids[j];
                float ext = maxExtent(mn, mx, b);
                float R = scaleInv ? radius * ext : radius;
                float d[3] = {(pts[j * 3] - samples[i * 3]) / R, (pts[j * 3 + 1] - samples[i * 3 + 1]) / R,
                              (pts[j * 3 + 2] - samples[i * 3 + 2]) / R};
                float c = pdfs[t] * K;
                for (int q = 0; q < nb; ++q) {
                    int off = q * MLP;
                    float pre1[MLP], pre2[MLP], t3[MLP], t4[MLP];
                    float a1[MLP], a2[MLP];
                    for (int t8 = 0; t8 < MLP; ++t8) {
                        int nu = off + t8;
                        pre1[t8] = layer1(d, &w1[nu * 3], b1[nu]);
                        a1[t8] = relu(pre1[t8]);
                    }
                    for (int t8 = 0; t8 < MLP; ++t8) {
                        pre2[t8] = dot8(a1, &w2[off * MLP + t8 * MLP], 1) + b2[off + t8];
                        a2[t8] = relu(pre2[t8]);
                    }
                    int numOuts = std::min(neuronsOut - off, MLP);
                    for (int t8 = 0; t8 < numOuts; ++t8) {  // spatial_conv.cu:383-400
                        int nu = off + t8;
                        int fin = nu % Fin;
                        int fo = combin ? nu / Fin : fin;
                        float f = feats[(size_t)j * Fin + fin];
                        float og = g[fo];
                        float cf = (f * og) / c;
                        for (int k = 0; k < MLP; ++k) pw3[off * MLP + t8 * MLP + k] += cf * a2[k];
                        float a = dot8(a2, &w3[off * MLP + t8 * MLP], 1);
                        pb3[nu] += cf;
                        a = a + b3[nu];
#pragma omp atomic
                        featGrad[(size_t)j * Fin + fin] += og * a / c;
                    }
                    for (int t8 = 0; t8 < MLP; ++t8) {  // spatial_conv.cu:403-414
                        float a = 0.0f;
                        float cf = (pre2[t8] >= 0.0f) ? 1.0f : 0.0f;
                        for (int k = 0; k < numOuts; ++k) {
                            int nu = off + k;
                            int fin = nu % Fin;
                            int fo = combin ? nu / Fin : fin;
                            a = std::fmaf(g[fo] * feats[(size_t)j * Fin + fin], w3[off * MLP + t8 + k * MLP], a);
                        }
                        t3[t8] = (cf * a) / c;
                    }
                    for (int t8 = 0; t8 < MLP; ++t8) {  // :419-425
                        for (int k = 0; k < MLP; ++k) pw2[off * MLP + t8 * MLP + k] += t3[t8] * a1[k];
                        pb2[off + t8] += t3[t8];
                    }
                    for (int t8 = 0; t8 < MLP; ++t8) {  // :428-434
                        float cf = (pre1[t8] >= 0.0f) ? 1.0f : 0.0f;
                        t4[t8] = cf * dot8(t3, &w2[off * MLP + t8], MLP);
                    }
                    for (int t8 = 0; t8 < MLP; ++t8) {  // :439-444
                        for (int k = 0; k < 3; ++k) pw1[(off + t8) * 3 + k] += t4[t8] * d[k];
                        pb1[off + t8] += t4[t8];
                    }
                }
            }
        }
    }
    std::vector<double> tot(wsz, 0.0);
    for (int t = 0; t < nthr; ++t)
        for (size_t k = 0; k < wsz; ++k) tot[k] += priv[(size_t)t * wsz + k];
    {
        const double* p = tot.data();
        for (size_t k = 0; k < 3 * nn; ++k) dw1[k] = (float)p[k];
        p += 3 * nn;
        for (size_t k = 0; k < nn; ++k) db1[k] = (float)p[k];
        p += nn;
        for (size_t k = 0; k < MLP * nn; ++k) dw2[k] = (float)p[k];
        p += MLP * nn;
        for (size_t k = 0; k < nn; ++k) db2[k] = (float)p[k];
        p += nn;
        for (size_t k = 0; k < MLP * nn; ++k) dw3[k] = (float)p[k];
        p += MLP * nn;
        for (size_t k = 0; k < nn; ++k) db3[k] = (float)p[k];
    }
    return 0;
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
