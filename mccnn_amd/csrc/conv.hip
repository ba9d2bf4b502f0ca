// Monte-Carlo convolution, forward and backward. Replaces tf_ops/spatial_conv.cu.
//
// Math (spatial_conv.cu:24-79,327-445): for edge e=(j,i), delta=(p_j-c_i)/R_b,
// block q, lane n, neuron nu=8q+n:
//   h1[n] = relu(delta . w1[nu] + b1[nu]);  h2[n] = relu(sum_m h1[m] w2[q][n][m] + b2[nu]);
//   o[n]  = sum_m h2[m] w3[q][n][m] + b3[nu];   out[i,fo(nu)] += feat[j,fin(nu)] o[n] / (pdf_e K_i)
//
// Design (v1, VALU): the reference launches 8 threads per (edge, block) and scatters every
// product with a float atomicAdd. Here the CSR rows are contiguous (find_neighbors.cu:255-258),
// so a wave owns G consecutive centres = one contiguous edge range; lanes are edges, the three
// layers are explicit fmaf chains with wave-uniform weights (scalar loads), the sum over a
// centre's edges is a segmented wave scan, and every output row is written exactly once from
// an LDS tile: no atomics, no pre-zeroing, bit-reproducible.
#include "common.h"

namespace mccnn {

struct ConvArgs {
    const float* pts;
    const float* feats;
    const int* bids;
    const float* pdfs;
    const float* samples;
    const int* start;
    const int2* packed;
    const float* mn;
    const float* mx;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    int n, m, e, Fin, Fout, nb, neuronsOut, outF;
    float radius;
    int scaleInv, avg, G;
};

__device__ __forceinline__ float relu(float x) { return fmaxf(x, 0.0f); }

struct EdgeCtx {
    int j, i;
    float d0, d1, d2, inv;  // inv = 1 / (pdf * K)
};

__device__ __forceinline__ EdgeCtx load_edge(const ConvArgs& a, int t) {
    EdgeCtx c;
    int2 pr = a.packed[t];
    c.j = pr.x;
    c.i = pr.y;
    int b = a.bids[c.j];
    float ext = max_extent(a.mn, a.mx, b);
    float R = a.scaleInv ? a.radius * ext : a.radius;
    c.d0 = (a.pts[(size_t)c.j * 3] - a.samples[(size_t)c.i * 3]) / R;       // spatial_conv.cu:155-158
    c.d1 = (a.pts[(size_t)c.j * 3 + 1] - a.samples[(size_t)c.i * 3 + 1]) / R;
    c.d2 = (a.pts[(size_t)c.j * 3 + 2] - a.samples[(size_t)c.i * 3 + 2]) / R;
    int e0 = a.start[c.i];
    int e1 = (c.i < a.m - 1) ? a.start[c.i + 1] : a.e;
    float K = a.avg ? (float)(e1 - e0) : 1.0f;                                // :161-163
    c.inv = 1.0f / (a.pdfs[t] * K);
    return c;
}

// Kernel MLP of block q for one edge. pre1/pre2 are the pre-activations (needed by backward).
__device__ __forceinline__ void mlp_block(const ConvArgs& a, int q, float d0, float d1, float d2, float* pre1,
                                          float* pre2, float* o) {
    const float* w1 = a.w1 + q * 24;
    const float* b1 = a.b1 + q * 8;
    const float* w2 = a.w2 + q * 64;
    const float* b2 = a.b2 + q * 8;
    const float* w3 = a.w3 + q * 64;
    const float* b3 = a.b3 + q * 8;
#pragma unroll
    for (int n = 0; n < 8; ++n)
        pre1[n] = fmaf(d2, w1[n * 3 + 2], fmaf(d1, w1[n * 3 + 1], fmaf(d0, w1[n * 3], b1[n])));
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float s = b2[n];
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(relu(pre1[k]), w2[n * 8 + k], s);
        pre2[n] = s;
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float s = b3[n];
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(relu(pre2[k]), w3[n * 8 + k], s);
        o[n] = s;
    }
}

// FEAT: 0 generic, 1 combin with Fin == 1, 2 no-combin with Fin % 8 == 0 (vector loads)
template <bool COMBIN, int FEAT>
__global__ __launch_bounds__(256) void conv_fwd_valu(ConvArgs a, float* __restrict__ out) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int G = a.G, outF = a.outF;
    float* tile = lds + (size_t)wave * G * outF;
    const int c0 = (blockIdx.x * 4 + wave) * G;
    if (c0 >= a.m) return;
    const int c1 = min(c0 + G, a.m);
    const int eBeg = a.start[c0];
    const int eEnd = (c1 < a.m) ? a.start[c1] : a.e;
    for (int k = lane; k < G * outF; k += 64) tile[k] = 0.0f;
    __builtin_amdgcn_wave_barrier();

    for (int base = eBeg; base < eEnd; base += 64) {
        const int t = base + lane;
        const bool act = t < eEnd;
        EdgeCtx ec;
        if (act) {
            ec = load_edge(a, t);
        } else {
            ec.j = 0; ec.i = c0; ec.d0 = ec.d1 = ec.d2 = 0.f; ec.inv = 0.f;
        }
        const int key = act ? (ec.i - c0) : (G + lane);  // inactive lanes never merge
        unsigned same = 0;  // bit s: lane - 2^s belongs to my centre
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            int k2 = __shfl_up(key, 1 << s, 64);
            if (lane >= (1 << s) && k2 == key) same |= 1u << s;
        }
        const int knext = __shfl_down(key, 1, 64);
        const bool tail = act && (lane == 63 || knext != key);
        float f1 = 0.f;
        if (FEAT == 1) f1 = act ? a.feats[ec.j] * ec.inv : 0.f;

        for (int q = 0; q < a.nb; ++q) {
            float pre1[8], pre2[8], o[8], c[8];
            mlp_block(a, q, ec.d0, ec.d1, ec.d2, pre1, pre2, o);
            if (FEAT == 2) {
                const float4* fp = reinterpret_cast<const float4*>(a.feats + (size_t)ec.j * a.Fin + q * 8);
                float4 fa = fp[0], fb = fp[1];
                float f[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
#pragma unroll
                for (int n = 0; n < 8; ++n) c[n] = f[n] * o[n] * ec.inv;
            } else if (FEAT == 1) {
#pragma unroll
                for (int n = 0; n < 8; ++n) c[n] = (q * 8 + n < a.neuronsOut) ? f1 * o[n] : 0.f;
            } else {
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    int nu = q * 8 + n;
                    int fin = nu % a.Fin;
                    c[n] = (nu < a.neuronsOut) ? a.feats[(size_t)ec.j * a.Fin + fin] * o[n] * ec.inv : 0.f;
                }
            }
            // segmented inclusive scan over the lanes of one centre
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                bool take = (same >> s) & 1u;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    float v = __shfl_up(c[n], 1 << s, 64);
                    c[n] += take ? v : 0.f;
                }
            }
            if (tail) {
                float* row = tile + (size_t)key * outF;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    int nu = q * 8 + n;
                    if (nu < a.neuronsOut) {
                        int fo = COMBIN ? nu / a.Fin : nu;
                        row[fo] += c[n];
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int cnt = (c1 - c0) * outF;
    float* dst = out + (size_t)c0 * outF;
    for (int k = lane; k < cnt; k += 64) dst[k] = tile[k];
}

// ---------------------------------------------------------------------------------------
// Backward (spatial_conv.cu:327-445 / :563-680). A wave owns GB consecutive centres; for each
// block q it sweeps its edges keeping the 176 weight-gradient partial sums of that block in
// registers, reduces them across the wave once per (wave, q), combines the 4 waves of the
// workgroup in LDS and issues one global atomic per (workgroup, parameter). Feature gradients
// are scattered with float atomics (rows of different centres share j).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <bool COMBIN>
__global__ __launch_bounds__(256) void conv_bwd_valu(ConvArgs a, const float* __restrict__ outGrad,
                                                     float* __restrict__ featGrad, float* __restrict__ dw1,
                                                     float* __restrict__ db1, float* __restrict__ dw2,
                                                     float* __restrict__ db2, float* __restrict__ dw3,
                                                     float* __restrict__ db3) {
    __shared__ float red[4][176];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int G = a.G, outF = a.outF;
    const int c0 = (blockIdx.x * 4 + wave) * G;
    const int c1 = min(c0 + G, a.m);
    int eBeg = 0, eEnd = 0;
    if (c0 < a.m) {
        eBeg = a.start[c0];
        eEnd = (c1 < a.m) ? a.start[c1] : a.e;
    }
    for (int q = 0; q < a.nb; ++q) {
        float gw3[64], gb3[8], gw2[64], gb2[8], gw1[24], gb1[8];
#pragma unroll
        for (int k = 0; k < 64; ++k) { gw3[k] = 0.f; gw2[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { gb3[k] = 0.f; gb2[k] = 0.f; gb1[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 24; ++k) gw1[k] = 0.f;
        const float* w2 = a.w2 + q * 64;
        const float* w3 = a.w3 + q * 64;
        const int numOuts = min(a.neuronsOut - q * 8, 8);

        for (int base = eBeg; base < eEnd; base += 64) {
            const int t = base + lane;
            if (t >= eEnd) continue;
            EdgeCtx ec = load_edge(a, t);
            float pre1[8], pre2[8], o[8];
            mlp_block(a, q, ec.d0, ec.d1, ec.d2, pre1, pre2, o);
            float gf[8];  // g_n * f_n
            const float* grow = outGrad + (size_t)ec.i * outF;
            const float* frow = a.feats + (size_t)ec.j * a.Fin;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (n < numOuts) {
                    int nu = q * 8 + n;
                    int fin = COMBIN ? nu % a.Fin : nu;
                    int fo = COMBIN ? nu / a.Fin : nu;
                    float g = grow[fo], f = frow[fin];
                    gf[n] = g * f;
                    float u = gf[n] * ec.inv;  // (f g)/(pdf K), spatial_conv.cu:389
#pragma unroll
                    for (int k = 0; k < 8; ++k) gw3[n * 8 + k] = fmaf(u, relu(pre2[k]), gw3[n * 8 + k]);
                    gb3[n] += u;
                    atomicAdd(&featGrad[(size_t)ec.j * a.Fin + fin], g * o[n] * ec.inv);  // :400
                } else {
                    gf[n] = 0.f;
                }
            }
            float t3[8], t4[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // :403-414
                float s = 0.f;
#pragma unroll
                for (int n = 0; n < 8; ++n) s = fmaf(gf[n], w3[n * 8 + k], s);
                t3[k] = (pre2[k] >= 0.0f) ? s * ec.inv : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // :419-425
#pragma unroll
                for (int l = 0; l < 8; ++l) gw2[k * 8 + l] = fmaf(t3[k], relu(pre1[l]), gw2[k * 8 + l]);
                gb2[k] += t3[k];
            }
#pragma unroll
            for (int l = 0; l < 8; ++l) {  // :428-434
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s = fmaf(t3[k], w2[k * 8 + l], s);
                t4[l] = (pre1[l] >= 0.0f) ? s : 0.f;
            }
#pragma unroll
            for (int l = 0; l < 8; ++l) {  // :439-444
                gw1[l * 3] = fmaf(t4[l], ec.d0, gw1[l * 3]);
                gw1[l * 3 + 1] = fmaf(t4[l], ec.d1, gw1[l * 3 + 1]);
                gw1[l * 3 + 2] = fmaf(t4[l], ec.d2, gw1[l * 3 + 2]);
                gb1[l] += t4[l];
            }
        }
        // wave reduction -> LDS (layout: w1[24] b1[8] w2[64] b2[8] w3[64] b3[8])
#pragma unroll
        for (int k = 0; k < 24; ++k) { float v = wave_sum(gw1[k]); if (lane == 0) red[wave][k] = v; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { float v = wave_sum(gb1[k]); if (lane == 0) red[wave][24 + k] = v; }
#pragma unroll
        for (int k = 0; k < 64; ++k) { float v = wave_sum(gw2[k]); if (lane == 0) red[wave][32 + k] = v; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { float v = wave_sum(gb2[k]); if (lane == 0) red[wave][96 + k] = v; }
#pragma unroll
        for (int k = 0; k < 64; ++k) { float v = wave_sum(gw3[k]); if (lane == 0) red[wave][104 + k] = v; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { float v = wave_sum(gb3[k]); if (lane == 0) red[wave][168 + k] = v; }
        __syncthreads();
        if (threadIdx.x < 176) {
            int k = threadIdx.x;
            float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
            if (v != 0.0f) {
                if (k < 24) atomicAdd(&dw1[q * 24 + k], v);
                else if (k < 32) atomicAdd(&db1[q * 8 + k - 24], v);
                else if (k < 96) atomicAdd(&dw2[q * 64 + k - 32], v);
                else if (k < 104) atomicAdd(&db2[q * 8 + k - 96], v);
                else if (k < 168) atomicAdd(&dw3[q * 64 + k - 104], v);
                else atomicAdd(&db3[q * 8 + k - 168], v);
            }
        }
        __syncthreads();
    }
}

static int fill_args(ConvArgs& a, const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                     const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                     const float* aabb_min, const float* aabb_max, const float* w1, const float* b1, const float* w2,
                     const float* b2, const float* w3, const float* b3, int n, int m, int e, int Fin, int Fout,
                     int combin, int batch_size, float radius, int scale_inv, int avg) {
    if (n < 0 || m < 0 || e < 0 || Fin <= 0 || Fout <= 0 || batch_size <= 0 || !(radius > 0.0f)) return MCCNN_E_BADARG;
    if (!combin && Fout != Fin) return MCCNN_E_SHAPE;  // MCConvBuilder.py:328-333, spatial_conv.cc:292-296
    long long neurons = combin ? (long long)Fin * Fout : Fin;
    if (neurons > (1 << 24)) return MCCNN_E_TOOLARGE;
    a.neuronsOut = (int)neurons;
    a.nb = (a.neuronsOut + 7) / 8;
    if ((a.nb * 8) % Fin != 0) return MCCNN_E_SHAPE;    // spatial_conv.cc:290
    a.outF = combin ? Fout : Fin;
    a.pts = sorted_pts; a.feats = sorted_feats; a.bids = sorted_batch_ids; a.pdfs = pdfs; a.samples = samples;
    a.start = start_idx; a.packed = reinterpret_cast<const int2*>(packed); a.mn = aabb_min; a.mx = aabb_max;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3;
    a.n = n; a.m = m; a.e = e; a.Fin = Fin; a.Fout = Fout; a.radius = radius; a.scaleInv = scale_inv; a.avg = avg;
    if (m > 0 && (!samples || !start_idx || !aabb_min || !aabb_max || !w1 || !b1 || !w2 || !b2 || !w3 || !b3))
        return MCCNN_E_BADARG;
    if (e > 0 && (!sorted_pts || !sorted_feats || !sorted_batch_ids || !pdfs || !packed)) return MCCNN_E_BADARG;
    return 0;
}

}  // namespace mccnn

using namespace mccnn;

extern "C" {

size_t mccnn_spatial_conv_fwd_workspace_bytes(int, int, int, int, int) { return 0; }

int mccnn_spatial_conv_fwd(const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                           const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                           const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* w3, const float* b3, int n, int m, int e,
                           int num_in_feats, int num_out_feats, int combin, int batch_size, float radius,
                           int scale_inv, int avg, float* out, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    (void)ws; (void)ws_bytes;
    ConvArgs a;
    int rc = fill_args(a, sorted_pts, sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min,
                       aabb_max, w1, b1, w2, b2, w3, b3, n, m, e, num_in_feats, num_out_feats, combin, batch_size,
                       radius, scale_inv, avg);
    if (rc) return rc;
    if (m == 0) return 0;
    if (!out) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    int G = 2048 / a.outF;
    if (G > 32) G = 32;
    if (G < 1) G = 1;
    a.G = G;
    size_t lds = (size_t)4 * G * a.outF * sizeof(float);
    if (lds > 64 * 1024) return MCCNN_E_TOOLARGE;
    int blocks = ceil_div(m, 4 * G);
    bool vec = !combin && (a.Fin % 8 == 0) && ((((uintptr_t)sorted_feats) & 15) == 0);
    if (combin) {
        if (a.Fin == 1) conv_fwd_valu<true, 1><<<blocks, 256, lds, s>>>(a, out);
        else conv_fwd_valu<true, 0><<<blocks, 256, lds, s>>>(a, out);
    } else {
        if (vec) conv_fwd_valu<false, 2><<<blocks, 256, lds, s>>>(a, out);
        else conv_fwd_valu<false, 0><<<blocks, 256, lds, s>>>(a, out);
    }
    MCCNN_LAUNCHED();
    return 0;
}

size_t mccnn_spatial_conv_bwd_workspace_bytes(int, int, int, int, int, int) { return 0; }

int mccnn_spatial_conv_bwd(const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                           const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                           const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* w3, const float* b3, const float* out_grad,
                           int n, int m, int e, int num_in_feats, int num_out_feats, int combin, int batch_size,
                           float radius, int scale_inv, int avg, float* feat_grad, float* dw1, float* db1, float* dw2,
                           float* db2, float* dw3, float* db3, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    (void)ws; (void)ws_bytes;
    ConvArgs a;
    int rc = fill_args(a, sorted_pts, sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min,
                       aabb_max, w1, b1, w2, b2, w3, b3, n, m, e, num_in_feats, num_out_feats, combin, batch_size,
                       radius, scale_inv, avg);
    if (rc) return rc;
    if (!dw1 || !db1 || !dw2 || !db2 || !dw3 || !db3 || (n > 0 && !feat_grad)) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    size_t nn = (size_t)a.nb * 8;
    if (n > 0) MCCNN_HIP(hipMemsetAsync(feat_grad, 0, (size_t)n * a.Fin * sizeof(float), s));
    MCCNN_HIP(hipMemsetAsync(dw1, 0, 3 * nn * sizeof(float), s));
    MCCNN_HIP(hipMemsetAsync(db1, 0, nn * sizeof(float), s));
    MCCNN_HIP(hipMemsetAsync(dw2, 0, 8 * nn * sizeof(float), s));
    MCCNN_HIP(hipMemsetAsync(db2, 0, nn * sizeof(float), s));
    MCCNN_HIP(hipMemsetAsync(dw3, 0, 8 * nn * sizeof(float), s));
    MCCNN_HIP(hipMemsetAsync(db3, 0, nn * sizeof(float), s));
    if (m == 0 || e == 0) return 0;
    if (!out_grad) return MCCNN_E_BADARG;
    a.G = 32;
    int blocks = ceil_div(m, 4 * a.G);
    if (combin) conv_bwd_valu<true><<<blocks, 256, 0, s>>>(a, out_grad, feat_grad, dw1, db1, dw2, db2, dw3, db3);
    else conv_bwd_valu<false><<<blocks, 256, 0, s>>>(a, out_grad, feat_grad, dw1, db1, dw2, db2, dw3, db3);
    MCCNN_LAUNCHED();
    return 0;
}

}  // extern "C"
