// Monte-Carlo convolution, forward and backward. Replaces tf_ops/spatial_conv.cu.
//
// Math (spatial_conv.cu:24-79,327-445): for edge e=(j,i), delta=(p_j-c_i)/R_b,
// block q, lane n, neuron nu=8q+n:
//   h1[n] = relu(delta . w1[nu] + b1[nu]);  h2[n] = relu(sum_m h1[m] w2[q][n][m] + b2[nu]);
//   o[n]  = sum_m h2[m] w3[q][n][m] + b3[nu];   out[i,fo(nu)] += feat[j,fin(nu)] o[n] / (pdf_e K_i)
//
// The reference launches 8 threads per (edge, block) and scatters every product with a float atomicAdd. Here the CSR
// rows are contiguous (find_neighbors.cu:255-258): lanes are edges, the sum over a centre's edges is a segmented wave
// scan and every output row is written exactly once -- no atomics, no pre-zeroing, bit-reproducible.
//   conv_stream      forward (and, on the transposed list, the depth-wise feature gradient): MFMA kernel MLP,
//                    edge-balanced slices, wave-wide fused-DPP scan with carry            (NOTES.md section 5)
//   conv_bwd_mfma    backward: q-outer sweeps with the weight-gradient sums in VGPRs, deterministic partial rows
//   conv_fwd_valu / conv_bwd_valu   fallback for nb > MCCNN_LDS_MAX_NB: scalar-loaded weights, fmaf chains, LDS tile
//   conv_f1.hip      combin layers with ONE input feature take the factored kernels there
#include "conv_mfma.h"
#include "batch.h"
#include <cstdlib>
#include <type_traits>

namespace mccnn {

__device__ __forceinline__ float relu(float x) { return fmaxf(x, 0.0f); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct EdgeCtx {
    int j, i;
    float d0, d1, d2, inv;  // inv = 1 / (pdf * K)
};

__device__ __forceinline__ EdgeCtx load_edge(const ConvArgs& a, int t) {
    EdgeCtx c;
    int2 pr = a.packed[t];
    c.j = pr.x;
    c.i = pr.y;
    int b = clamp_batch(a.bids[c.j], a.B);
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
    // the chains the MFMA kernels issue (conv_mfma.h): k-ordered fma from zero, bias last
#pragma unroll
    for (int n = 0; n < 8; ++n) pre1[n] = fmaf(d2, w1[n * 3 + 2], fmaf(d1, w1[n * 3 + 1], d0 * w1[n * 3])) + b1[n];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(relu(pre1[k]), w2[n * 8 + k], s);
        pre2[n] = s + b2[n];
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(relu(pre2[k]), w3[n * 8 + k], s);
        o[n] = s + b3[n];
    }
}

// FEAT: 0 generic, 1 combin with Fin == 1, 2 no-combin with Fin % 8 == 0 (vector loads), 3 combin with Fin = 2..4
//       (MFMA kernels only), 4 = 2 with bf16 feature / output rows (MFMA kernels only)
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
// Forward, streaming form. Every wave owns a centre-aligned slice of the neighbour list holding ~E/W edges (W = the
// number of waves the chip keeps resident, so the launch is ONE balanced round: with a fixed number of centres per
// wave the non-uniform clouds left the last third of the launch half empty). Chunks are 64 consecutive edges of the
// slice, independent of centre boundaries; the output tile is a sliding window of G rows (row of centre c = c mod G)
// that is flushed -- each row exactly once, in order -- when a chunk reaches beyond it.
// ---------------------------------------------------------------------------------------
// TR = true runs the same reduction over the TRANSPOSED list (rows = neighbours j, CSR a.start = startT, edge ids
// through permT, geometry from the per-edge records): the depth-wise feature gradient
//   featGrad[j, nu] = sum over edges e=(j,i) of outGrad[i, nu] * o_e[nu] / (pdf_e K_i)          (spatial_conv.cu:400)
// is the forward convolution with the roles of centres and neighbours swapped (a.feats = outGrad, a.m = n).
template <bool COMBIN, int FEAT, bool TR>
__global__ __launch_bounds__(256) void conv_stream(ConvArgs a, float* __restrict__ out, int numWaves,
                                                   const float4* __restrict__ rec, const int* __restrict__ permT,
                                                   float4* __restrict__ recOut) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i4 = lane & 3;
    const int outF = a.outF;
    float* wl = lds;
    float* carry = lds + a.nb * MCCNN_WQ_FWD + (size_t)wave * (a.nb * 8);  // running sums of a centre that spans chunks
    stage_weights<MCCNN_WQ_FWD>(a, wl);
    __syncthreads();
    const int w = blockIdx.x * 4 + wave;
    if (w >= numWaves) return;
    const int tA = (int)(((long long)a.e * w) / numWaves);
    const int tB = (int)(((long long)a.e * (w + 1)) / numWaves);
    const int cA = (w == 0) ? 0 : wave_lower_bound(a.start, a.m, a.e, tA, lane);
    const int cB = (w == numWaves - 1) ? a.m : wave_lower_bound(a.start, a.m, a.e, tB, lane);
    if (cA >= cB) return;
    const int eBeg = a.start[cA];
    const int eEnd = (cB < a.m) ? a.start[cB] : a.e;
    const bool vecOut = (!COMBIN || FEAT == 1) && (outF & 3) == 0;
    constexpr bool BF = FEAT == 4;  // bf16 rows: `a.feats` and `out` hold 2-byte elements
    const unsigned short* feats16 = reinterpret_cast<const unsigned short*>(a.feats);
    unsigned short* out16 = reinterpret_cast<unsigned short*>(out);

    // rows of centres without neighbours are never reached by an edge: zero them where the gap shows up
    // (depth-wise layers wider than one launch's weight tile are run in column tiles: `out` then points at the tile's
    // first column, rows keep the full stride outF and only the tile's columns belong to this launch)
    auto zero_rows = [&](int c0, int c1) {
        const int width = COMBIN ? outF : min(a.nb * 8, a.neuronsOut);
        const int words = BF ? width / 2 : width, stride = BF ? outF / 2 : outF;  // 32-bit words (bf16 rows: % 8 == 0)
        for (int c = c0; c < c1; ++c)
            for (int f = 0; f < words; ++f) out[(size_t)c * stride + f] = 0.0f;
    };

    int keyLast = cA;     // key (= centre + 1) of the previous chunk's last edge; cA stands for "centre cA-1 is done"
    int carryKey = -1;    // key whose partial sums sit in carry[]
    int2 prN = make_int2(0, cA);
    float pdfN = 1.0f;
    int eN = 0;
    if (TR) {
        eN = permT[min(eBeg + lane, a.e - 1)];
    } else if (eBeg + lane < eEnd) {
        prN = a.packed[eBeg + lane];
        pdfN = a.pdfs[eBeg + lane];
    }
    float acc = 0.f;  // generic combin layers: running sum over the Fin neurons of one output feature
    for (int base = eBeg; base < eEnd; base += 64) {
        const int t = base + lane;
        const int nIn = min(64, eEnd - base);
        const bool in = lane < nIn;
        int ci, j;
        float d0, d1, d2, inv;
        if (TR) {
            const float4 rc = rec[eN];
            const int2 pr = a.packed[eN];
            eN = permT[min(t + 64, a.e - 1)];
            ci = pr.x;  // the row this edge adds to: its neighbour point
            j = pr.y;   // the row it reads: outGrad of its centre
            d0 = rc.x; d1 = rc.y; d2 = rc.z;
            inv = in ? rc.w : 0.0f;
        } else {
            const int2 pr = prN;
            const float pdf = pdfN;
            if (t + 64 < eEnd) { prN = a.packed[t + 64]; pdfN = a.pdfs[t + 64]; }
            ci = pr.y;
            j = pr.x;
            float invR = a.invRadius;
            if (a.scaleInv) invR = 1.0f / (a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[j], a.B)));
            const float* pp = a.pts + (size_t)j * 3;
            const float* cc = a.samples + (size_t)ci * 3;
            const float R = a.scaleInv ? a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[j], a.B)) : a.radius;
            d0 = div_exact(pp[0] - cc[0], R, invR); d1 = div_exact(pp[1] - cc[1], R, invR); d2 = div_exact(pp[2] - cc[2], R, invR);
            float K = 1.0f;
            if (a.avg) K = (float)(((ci + 1 < a.m) ? a.start[ci + 1] : a.e) - a.start[ci]);
            inv = in ? __builtin_amdgcn_rcpf(pdf * K) : 0.0f;
            // optional forward state: the per-edge record (delta, 1 / (pdf K)) the backward pass would otherwise
            // recompute in a pass of its own -- identical bits (edge_records spells the same expressions)
            if (recOut && in) recOut[t] = make_float4(d0, d1, d2, inv);
        }
        const int key = in ? ci + 1 : 0;
        const int cLast = __builtin_amdgcn_readlane(ci, nIn - 1);
        // the last row of this chunk goes on in the next one iff its CSR range reaches beyond the chunk
        const int rowEnd = (cLast + 1 < a.m) ? a.start[cLast + 1] : a.e;
        const bool cont = rowEnd > base + 64;
        int keyPrev = __shfl_up(key, 1, 64);
        if (lane == 0) keyPrev = keyLast;
        int keyNext = __shfl_down(key, 1, 64);
        const bool tail = in && ((lane == nIn - 1) ? !cont : (key != keyNext));
        if (in && key - keyPrev > 1) zero_rows(keyPrev, ci);  // rows keyPrev .. ci-1 have no edges
        float f1 = 0.f;
        if (FEAT == 1) f1 = in ? a.feats[j] * inv : 0.f;
        // wave-wide segmented inclusive scan: 4 steps inside the 16-lane rows, then rows 1,3 take the end of rows
        // 0,2 and rows 2,3 the end of row 1 -- always only into lanes of the same centre
        const float m1 = (key != 0 && dpp_i<DPP_ROW_SHR(1)>(key) == key) ? 1.f : 0.f;
        const float m2 = (key != 0 && dpp_i<DPP_ROW_SHR(2)>(key) == key) ? 1.f : 0.f;
        const float m4 = (key != 0 && dpp_i<DPP_ROW_SHR(4)>(key) == key) ? 1.f : 0.f;
        const float m8 = (key != 0 && dpp_i<DPP_ROW_SHR(8)>(key) == key) ? 1.f : 0.f;
        const float mA = (key != 0 && dpp_rows_i<DPP_ROW_BCAST15, 0xA>(key) == key) ? 1.f : 0.f;
        const float mB = (key != 0 && dpp_rows_i<DPP_ROW_BCAST31, 0xC>(key) == key) ? 1.f : 0.f;
        const bool haveCarry = carryKey >= 0;
        const float mC = (haveCarry && key == carryKey) ? 1.f : 0.f;
        float* orow = out + (size_t)ci * outF;

        int finQ = 0, foQ = 0;  // (nu % Fin, nu / Fin) of the block's first neuron, kept incrementally
        constexpr bool smallFin = COMBIN && FEAT == 3;  // combin, 2..4 input features
        float fs[4] = {0.f, 0.f, 0.f, 0.f};
        if (smallFin) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < a.Fin) fs[k] = a.feats[(size_t)j * a.Fin + k] * inv;
        }
        constexpr std::integral_constant<int, 1> ic1{};
        constexpr std::integral_constant<int, 0> ic0{};
        auto block = [&](const int q, const float4 fa, const float4 fb, auto finC, auto r0C) {
            constexpr int FIN_ = decltype(finC)::value, R0_ = decltype(r0C)::value;
            float pre1[8], a1[8], pre2[8], a2[8], o[8], c[8];
            MCCNN_PHASE();
            mlp_block_mfma(wl + q * MCCNN_WQ_FWD, i4, d0, d1, d2, pre1, a1, pre2, a2, o);
            if (FEAT == 2 || FEAT == 4) {
                float f[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
#pragma unroll
                for (int n = 0; n < 8; ++n) c[n] = (f[n] * inv) * o[n];
            } else if (FEAT == 1) {
#pragma unroll
                for (int n = 0; n < 8; ++n) c[n] = f1 * o[n];
            } else {
                if (COMBIN && smallFin) {
                    float ffn[8];
                    combin_pick<FIN_, R0_>(fs, nullptr, ffn, nullptr);  // fs = feature row * 1/(pdf K), once per chunk
#pragma unroll
                    for (int n = 0; n < 8; ++n) c[n] = (q * 8 + n < a.neuronsOut) ? ffn[n] * o[n] : 0.f;
                } else {
                    // neuron nu = fo * Fin + fin (combin) or nu = fin (depth-wise): fin advances with nu, no division
                    int fin = finQ;
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        int nu = q * 8 + n;
                        c[n] = (nu < a.neuronsOut) ? a.feats[(size_t)j * a.Fin + fin] * o[n] * inv : 0.f;
                        if (++fin == a.Fin) fin = 0;
                    }
                }
            }
            float* cq = carry + q * 8;
            f32x4 cv0 = {0.f, 0.f, 0.f, 0.f}, cv1 = cv0;
            if (haveCarry) { cv0 = *reinterpret_cast<f32x4*>(cq); cv1 = *reinterpret_cast<f32x4*>(cq + 4); }
            wave_seg_scan8(c, m1, m2, m4, m8, mA, mB);
#pragma unroll
            for (int n = 0; n < 8; ++n) c[n] = fmaf(mC, n < 4 ? cv0[n & 3] : cv1[n & 3], c[n]);
            if (cont && lane == 63) {
                *reinterpret_cast<f32x4*>(cq) = (f32x4){c[0], c[1], c[2], c[3]};
                *reinterpret_cast<f32x4*>(cq + 4) = (f32x4){c[4], c[5], c[6], c[7]};
            }
            if (!COMBIN || FEAT == 1) {
                if (!COMBIN && FEAT == 0) finQ += 8;  // depth-wise, scalar path: fin = nu
                if (tail) {
                    if (BF) {
                        reinterpret_cast<uint4*>(out16 + (size_t)ci * outF)[q] = f32x8_to_bf16(c);
                    } else {
                    float* dst = orow + q * 8;
                    // (a block whose last neurons are padding -- depth-wise rows of 4 features, Fout % 8 == 4 -- must
                    // not store its 8 lanes: the padded half would land in the NEXT row)
                    if (vecOut && q * 8 + 8 <= a.neuronsOut) {
                        reinterpret_cast<float4*>(dst)[0] = make_float4(c[0], c[1], c[2], c[3]);
                        reinterpret_cast<float4*>(dst)[1] = make_float4(c[4], c[5], c[6], c[7]);
                    } else {
#pragma unroll
                        for (int n = 0; n < 8; ++n)
                            if (q * 8 + n < a.neuronsOut) dst[n] = c[n];
                    }
                    }
                }
            } else {
                // output feature fo = sum over its Fin consecutive neurons (spatial_conv.cu:236): close it at the last one
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const int nu = q * 8 + n;
                    if (nu < a.neuronsOut) {
                        acc += c[n];
                        if (++finQ == a.Fin) {
                            if (tail) orow[foQ] = acc;
                            acc = 0.f;
                            finQ = 0;
                            ++foQ;
                        }
                    }
                }
            }
        };
        if (FEAT == 4) {
            // bf16 rows: 64 bytes (4 blocks) per load group, half the bytes of the f32 form per edge
            for (int q0 = 0; q0 < a.nb; q0 += 4) {
                const uint4* fp = reinterpret_cast<const uint4*>(feats16 + (size_t)j * a.Fin + q0 * 8);
                const int left = a.nb - q0;
                uint4 fl[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) fl[k] = (k < left) ? fp[k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < left) {
                        float f[8];
                        bf16x8_to_f32(fl[k], f);
                        block(q0 + k, make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]), ic1, ic0);
                    }
                }
            }
        } else if (FEAT == 2) {
            // the feature row is read one 128-byte line (4 blocks) at a time: fetching 32 bytes per block would pull
            // every line through L1 four times, and L1 does not hold 64 rows x 20 waves in between
            for (int q0 = 0; q0 < a.nb; q0 += 4) {
                const float4* fp = reinterpret_cast<const float4*>(a.feats + (size_t)j * a.Fin + q0 * 8);
                const int left = a.nb - q0;
                float4 fl[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) fl[k] = (k < 2 * left) ? fp[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                block(q0, fl[0], fl[1], ic1, ic0);
                if (left > 1) block(q0 + 1, fl[2], fl[3], ic1, ic0);
                if (left > 2) block(q0 + 2, fl[4], fl[5], ic1, ic0);
                if (left > 3) block(q0 + 3, fl[6], fl[7], ic1, ic0);
            }
        } else {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (COMBIN && FEAT == 3) {
                // static neuron -> fin pattern: r0 = (8 q) % Fin is 0 for Fin = 2, 4 and cycles 0, 2, 1 for Fin = 3
                if (a.Fin == 2) {
                    for (int q = 0; q < a.nb; ++q) block(q, z, z, std::integral_constant<int, 2>{}, ic0);
                } else if (a.Fin == 4) {
                    for (int q = 0; q < a.nb; ++q) block(q, z, z, std::integral_constant<int, 4>{}, ic0);
                } else {
                    for (int q = 0; q < a.nb; q += 3) {
                        block(q, z, z, std::integral_constant<int, 3>{}, ic0);
                        if (q + 1 < a.nb) block(q + 1, z, z, std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
                        if (q + 2 < a.nb) block(q + 2, z, z, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
                    }
                }
            } else {
                for (int q = 0; q < a.nb; ++q) block(q, z, z, ic1, ic0);
            }
        }
        carryKey = cont ? cLast + 1 : -1;
        keyLast = cLast + 1;
    }
    if (lane == 0) zero_rows(keyLast, cB);
}

// ---------------------------------------------------------------------------------------
// Backward, MFMA version. q-outer: for one MLP block the wave sweeps all its edges keeping the block's
// 176 weight-gradient partial sums in VGPRs (VALU FMAs run beside the MFMA chains), reduces them across the
// wave once per (wave, q) and stores them to a per-wave partial row; reduce_partials sums the rows in a fixed
// order (deterministic, no float atomics on the parameters). Feature gradients: for combin layers with
// Fin <= 4 they are accumulated per edge in LDS across all blocks and flushed with ONE atomic per (edge, fin);
// otherwise one atomic per (edge, neuron).
// ---------------------------------------------------------------------------------------

// Per-edge record (delta0, delta1, delta2, 1/(pdf K)) written once per backward call: the q-outer sweep re-reads
// every edge nb times, so the dependent gathers (packed -> pts/samples/start/pdf) and the set-up arithmetic are
// paid once and the sweep itself only issues coalesced, prefetchable loads.
__global__ __launch_bounds__(256) void edge_records(ConvArgs a, float4* __restrict__ rec) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.e) return;
    rec[t] = edge_record(a, t);
}

#ifndef MCCNN_BWD_OCC
#define MCCNN_BWD_OCC 2
#endif
// combin layers with 2..4 input features and at most this many blocks keep one plane of per-edge feature-gradient sums
// per block (E * Fin floats each) instead of a read-modify-write of one plane
// (Round 5, built and dropped: the per-edge sums of a wave's slice kept in LDS across the blocks -- read-modify-write by
// the lane that owns the edge, at most 19 chunks per wave so that two workgroups of 62 KB stay resident -- and one float
// atomic per (edge, fin) to featGrad from inside the wave's LAST block, every wave starting its round of the blocks
// somewhere else so that the atomics spread over the whole kernel: no planes in memory, no scatter_edge_featgrad.
// Parity-green; 3 -> 8 on the room under rocprofv3, same box: 433 + 11 us against 265 + 128 + 9 us (sweep, scatter,
// reduction of the partial rows). Of the 168 us the sweep loses, the finer slices cost 33 (the plane form at the same
// slicing: 297 us), dynamic LDS beyond ~20 KB per workgroup another 25-35 even when nothing touches it, and neither the
// atomics nor the LDS updates show up in ablation builds (434 / 437 us without them): the rest is the sweep itself
// compiled in its 18 (fin, first neuron, last block) forms at the 256-register limit.)
// (Round 5, second form, also dropped: the plane form's slicing with ONE read-modify-write plane and the last block adding
// the finished sums to featGrad with returnless atomics: sweep 289 -> 434 us against 289 + 130 (scatter_edge_featgrad);
// atomics from every block, no plane at all: 765 us. Float atomics to random rows retire at ~100 G/s on this part --
// 13.6 M of them are 130-145 us wherever they are issued; hoisting the plane read above the MFMA chains costs 150
// more spill instructions and 60 us.)
#define MCCNN_DF_PLANES 4
static inline int df_planes(int Fin, int nb) { return (Fin >= 2 && Fin <= 4 && nb <= MCCNN_DF_PLANES) ? nb : 1; }
// Waves own equal, contiguous EDGE ranges (cpw chunks of 64 edges each): nothing in the backward pass needs
// centre alignment, and equal edge counts remove the tail that centre-aligned ranges show on non-uniform clouds.
#ifndef MCCNN_BWD_OCC_COMBIN
#define MCCNN_BWD_OCC_COMBIN MCCNN_BWD_OCC
#endif
template <bool COMBIN, int FEAT, bool COOP>
__global__ __launch_bounds__(256, (COMBIN && FEAT != 1) ? MCCNN_BWD_OCC_COMBIN : MCCNN_BWD_OCC) void conv_bwd_mfma(ConvArgs a, const float4* __restrict__ rec,
                                                                    const float* __restrict__ outGrad,
                                                                    float* __restrict__ featGrad,
                                                                    float* __restrict__ dfE, int cpw,
                                                                    float* __restrict__ partials) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i4 = lane & 3;
    const int outF = a.outF;
    float* wl = lds;
    stage_weights<MCCNN_WQ_BWD>(a, wl);
    __syncthreads();
    // COOP (depth-wise layers): the four waves of a workgroup share ONE slice of 4*cpw chunks and split the blocks
    // (wave k takes q = k, k+4, ...). The 32-byte pieces of a feature / gradient row that four consecutive blocks
    // read are one 128-byte line: fetched by one wave, it is still in L1/L2 when the other three want it, instead of
    // crossing the fabric once per block sweep. Combin layers keep wave-private slices (per-edge dFeat RMW).
    const int slice = COOP ? blockIdx.x : blockIdx.x * 4 + wave;
    const int scpw = COOP ? 4 * cpw : cpw;
    const long long eBegL = (long long)slice * scpw * 64;
    if (eBegL >= a.e) return;
    const int eBeg = (int)eBegL;
    const int eEnd = (int)min((long long)a.e, eBegL + (long long)scpw * 64);
    float* prow = partials + (size_t)slice * a.nb * 176;

    for (int q = COOP ? __builtin_amdgcn_readfirstlane(wave) : 0; q < a.nb; q += COOP ? 4 : 1) {
        float gw3[64], gb3[8], gw2[64], gb2[8], gw1[24], gb1[8];
#pragma unroll
        for (int k = 0; k < 64; ++k) { gw3[k] = 0.f; gw2[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { gb3[k] = 0.f; gb2[k] = 0.f; gb1[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 24; ++k) gw1[k] = 0.f;
        const int numOuts = min(a.neuronsOut - q * 8, 8);
        constexpr bool smallFin = COMBIN && FEAT == 3;  // combin, 2..4 input features
        const int r0 = smallFin ? (q * 8) % a.Fin : 0, fo0 = smallFin ? (q * 8) / a.Fin : 0;

        // the chunk loop as a generic lambda: combin layers with 2..4 input features instantiate it once per static
        // neuron -> (fin, fo) pattern (FIN_, R0_) and pick the instance once per block, outside the loop
        auto sweep = [&](auto finC, auto r0C) {
        constexpr int FIN_ = decltype(finC)::value, R0_ = decltype(r0C)::value;
        int2 prN;
        float4 rcN;
        {
            int t0 = min(eBeg + lane, a.e - 1);
            prN = a.packed[t0];
            rcN = rec[t0];
        }
        for (int base = eBeg; base < eEnd; base += 64) {
            const int t = base + lane;
            const bool act = t < eEnd;
            const int2 pr = prN;
            const float4 rc = rcN;
            const int j = pr.x;
            const float inv = act ? rc.w : 0.f;
            // g_n and f_n first: the gathers fly while the MFMA chains run
            float g[8], ff[8];
            const float* grow = outGrad + (size_t)pr.y * outF;
            if (FEAT == 4) {
                // bf16 rows: the block's 8 out-gradients / features are one 16-byte piece each
                const uint4 gu = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(outGrad) + (size_t)pr.y * outF)[q];
                const uint4 fu = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.feats) + (size_t)j * a.Fin)[q];
                float gg[8], f8[8];
                bf16x8_to_f32(gu, gg);
                bf16x8_to_f32(fu, f8);
#pragma unroll
                for (int n = 0; n < 8; ++n) { g[n] = act ? gg[n] : 0.f; ff[n] = f8[n]; }
            } else if (FEAT == 2) {
                const float4* gp = reinterpret_cast<const float4*>(grow + q * 8);
                const float4* fp = reinterpret_cast<const float4*>(a.feats + (size_t)j * a.Fin + q * 8);
                float4 ga = gp[0], gb = gp[1], fa = fp[0], fb = fp[1];
                float gg[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                float f8[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
#pragma unroll
                for (int n = 0; n < 8; ++n) { g[n] = act ? gg[n] : 0.f; ff[n] = f8[n]; }
            } else if (FEAT == 1) {
                float f = a.feats[j];
                if (numOuts == 8 && (outF & 3) == 0) {
                    const float4* gp = reinterpret_cast<const float4*>(grow + q * 8);
                    float4 ga = gp[0], gb = gp[1];
                    float gg[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
#pragma unroll
                    for (int n = 0; n < 8; ++n) { g[n] = act ? gg[n] : 0.f; ff[n] = f; }
                } else {
#pragma unroll
                    for (int n = 0; n < 8; ++n) { g[n] = (act && n < numOuts) ? grow[q * 8 + n] : 0.f; ff[n] = f; }
                }
            } else if (COMBIN && smallFin) {
                // 2..4 input features: the feature row and the <= 5 out-gradients this block touches are loaded once,
                // the neuron -> (fin, fo) pattern is one of <= 4 static register selections (combin_select)
                // Unconditional loads (clamped column, static counts): nothing selects on a loaded value here, so the
                // first use of the gathers is after the forward MLP and their latency flies under its MFMA chains.
                // Lanes past the slice end carry inv = 0, which zeroes every term they feed; padded neurons
                // (n >= numOuts) are zeroed below by block-uniform selects.
                constexpr int NG = (R0_ + 8 + FIN_ - 1) / FIN_;  // output features the block's 8 neurons span
                float fs[4] = {0.f, 0.f, 0.f, 0.f}, gw[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                const float* frow = a.feats + (size_t)j * FIN_;
#pragma unroll
                for (int k = 0; k < FIN_; ++k) fs[k] = frow[k];
#pragma unroll
                for (int k = 0; k < NG; ++k) gw[k] = grow[min(fo0 + k, outF - 1)];
                combin_pick<FIN_, R0_>(fs, gw, ff, g);
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    if (n >= numOuts) { g[n] = 0.f; ff[n] = 0.f; }
            } else {
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    int nu = q * 8 + n;
                    int fin = COMBIN ? nu % a.Fin : nu;
                    int fo = COMBIN ? nu / a.Fin : nu;
                    bool ok = act && n < numOuts;
                    g[n] = ok ? grow[fo] : 0.f;
                    ff[n] = ok ? a.feats[(size_t)j * a.Fin + fin] : 0.f;
                }
            }
            float dfOld = 0.f;
            if (COMBIN && FEAT == 1 && act && q > 0) dfOld = dfE[t];
            // prefetch the next chunk AFTER this chunk's gathers: vmcnt retires in order, so the waits for g / f
            // leave these two loads in flight across the whole iteration
            {
                int tn = min(t + 64, a.e - 1);  // clamped: the prefetch stays branch-free (counted vmcnt, not vmcnt(0))
                prN = a.packed[tn];
                rcN = rec[tn];
            }
            float a1[8], a2[8], o[8];
            bool p1[8], p2[8];  // pre-activation >= 0 (ReLU' of the reference: spatial_conv.cu:404,429), kept as lane masks
            // keep the LDS weight reads inside the chunk loop: hoisted, they would pin ~100 VGPRs and spill the
            // 176 accumulators (the LDS pipe is idle here, re-reading is free)
            int woff = q * MCCNN_WQ_BWD;
            asm volatile("" : "+s"(woff));
            const float* wq = wl + woff;
            const f32x4* w4 = reinterpret_cast<const f32x4*>(wq);
            {
                float pre1[8], pre2[8];
                mlp_block_mfma(wq, i4, rc.x, rc.y, rc.z, pre1, a1, pre2, a2, o);
#pragma unroll
                for (int k = 0; k < 8; ++k) { p1[k] = pre1[k] >= 0.0f; p2[k] = pre2[k] >= 0.0f; }
            }
            // feature gradient: og * o / (pdf K)   (spatial_conv.cu:400)
            if (COMBIN) {
                if (FEAT == 1) {
                    float sfg = 0.f;
#pragma unroll
                    for (int n = 0; n < 8; ++n) sfg = fmaf(g[n], o[n], sfg);
                    sfg *= inv;
                    // this lane owns edge t: plain RMW across the blocks, ONE atomic per edge after the last block
                    if (act) {
                        if (q == a.nb - 1) atomicAdd(&featGrad[j], dfOld + sfg);
                        else dfE[t] = dfOld + sfg;
                    }
                } else if (smallFin) {
                    float sf[4] = {0.f, 0.f, 0.f, 0.f};
                    combin_fold_t<FIN_, R0_>(g, o, sf);  // sf[fin] = sum over the block's neurons with that fin of g o
                    if (act) {
                        if (a.nb <= MCCNN_DF_PLANES) {
                            // few blocks: every block has its own plane of per-edge sums -- stores only, nothing to
                            // read back inside the sweep; scatter_edge_featgrad adds the planes
                            float* d = dfE + ((size_t)q * a.e + t) * FIN_;
#pragma unroll
                            for (int f = 0; f < FIN_; ++f) d[f] = sf[f] * inv;
                        } else {
#pragma unroll
                            for (int f = 0; f < 4; ++f) {
                                if (f < a.Fin) {
                                    float* d = dfE + (size_t)t * a.Fin + f;
                                    const float old = (q == 0) ? 0.f : *d;  // Fin <= 4: every fin is first touched by block 0
                                    *d = old + sf[f] * inv;
                                }
                            }
                        }
                    }
                } else if (act) {
                    // several neurons of a block may share fin: fold them in registers first
                    for (int f = 0; f < a.Fin; ++f) {
                        float sfg = 0.f;
                        bool any = false;
#pragma unroll
                        for (int n = 0; n < 8; ++n) {
                            int nu = q * 8 + n;
                            if (n < numOuts && nu % a.Fin == f) { sfg = fmaf(g[n], o[n], sfg); any = true; }
                        }
                        if (any) {
                            float* d = dfE + (size_t)t * a.Fin + f;
                            float old = (q * 8 >= a.Fin || q > 0) ? *d : 0.f;
                            if (q == 0) old = 0.f;
                            // first touch of (t, f) happens in the block where nu == f, i.e. q == f / 8
                            *d = ((q == f / 8) ? 0.f : old) + sfg * inv;
                        }
                    }
                }
            }  // depth-wise layers: the feature gradient is computed by conv_stream<TR> on the transposed list
            float gf[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) gf[n] = g[n] * ff[n];
            // dW3 += u a2^T, db3 += u, u = g f / (pdf K)          (spatial_conv.cu:383-399)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                float u = gf[n] * inv;
#pragma unroll
                for (int k = 0; k < 8; ++k) gw3[n * 8 + k] = fmaf(u, a2[k], gw3[n * 8 + k]);
                gb3[n] += u;
            }
            // t3 = 1[pre2 >= 0] * W3^T (g f) / (pdf K)             (:403-414)
            float t3[8];
            MCCNN_PHASE();
            layer8<false>(w4 + 62, nullptr, i4, gf, t3);  // W3^T rows at float 248 -> f32x4 index 62
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
            MCCNN_PHASE();
            layer8<false>(w4 + 46, nullptr, i4, t3, t4);  // W2^T rows at float 184 -> f32x4 index 46
            MCCNN_PHASE();
            // dW1 += t4 delta^T, db1 += t4                         (:439-444)
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float v = p1[l] ? t4[l] : 0.f;
                gw1[l * 3] = fmaf(v, rc.x, gw1[l * 3]);
                gw1[l * 3 + 1] = fmaf(v, rc.y, gw1[l * 3 + 1]);
                gw1[l * 3 + 2] = fmaf(v, rc.z, gw1[l * 3 + 2]);
                gb1[l] += v;
            }
        }
        };
        if constexpr (smallFin) {
            switch (a.Fin * 4 + r0) {
                case 8: sweep(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}); break;
                case 9: sweep(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}); break;
                case 12: sweep(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}); break;
                case 13: sweep(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}); break;
                case 14: sweep(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}); break;
                case 16: sweep(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{}); break;
                case 17: sweep(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}); break;
                case 18: sweep(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}); break;
                case 19: sweep(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{}); break;
                default: break;
            }
        } else {
            sweep(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        }
        // transposing wave reduction; partial row layout: w1[24] b1[8] w2[64] b2[8] w3[64] b3[8]
        {
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
            float* pq = prow + q * 176;
            pq[32 + lane] = r2;
            pq[104 + lane] = r3;
            if (lane < 32) pq[lane] = rm;                 // w1, b1
            else if (lane < 40) pq[96 + lane - 32] = rm;  // b2
            else if (lane < 48) pq[168 + lane - 40] = rm; // b3
        }
    }
}

// ---------------------------------------------------------------------------------------
// Transposed neighbour list (CSR by neighbour j) and the depth-wise feature gradient.
//   featGrad[j, nu] = sum over edges e=(j,i) of outGrad[i, nu] * o_e[nu] / (pdf_e K_i)     (spatial_conv.cu:400)
// is the forward convolution with the roles of centres and neighbours swapped, so it runs through the same
// MFMA + segmented-reduction machinery on the transposed list: rows are written once, no float atomics
// (the reference -- and the first version here -- scatter one atomic per (edge, feature): E*Fin of them).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tr_count(const int2* __restrict__ packed, int e, int* __restrict__ cnt,
                                                int* __restrict__ slot) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < e) slot[t] = atomicAdd(&cnt[packed[t].x], 1);
}
// Lists whose edges fall on FEW points (up-sampling from a coarse level: 43 072 edges on 73 points took 72 us of
// serialised returning atomics): ranks inside the workgroup from LDS counters, one global atomic per (workgroup, point).
#define MCCNN_TR_LDS_BINS 512  // a workgroup's 256 consecutive edges must share points for this to pay
__global__ __launch_bounds__(256) void tr_count_lds(const int2* __restrict__ packed, int e, int n, int* __restrict__ cnt,
                                                    int* __restrict__ slot) {
    __shared__ int bins[MCCNN_TR_LDS_BINS];
    for (int j = threadIdx.x; j < n; j += 256) bins[j] = 0;
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int j = -1, local = 0;
    if (t < e) {
        j = packed[t].x;
        local = atomicAdd(&bins[j], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const int v = bins[k];
        if (v) bins[k] = atomicAdd(&cnt[k], v);
    }
    __syncthreads();
    if (t < e) slot[t] = bins[j] + local;
}
// (Workgroups in XCD-contiguous order: the 4-byte scatters of one row come from centres of one region of space, i.e. from
// one run of edge ids -- handed to ONE XCD, the lines of `tmp` fill up inside that XCD's L2 instead of leaving eight L2s
// as 32-byte partial writes: 137 MB of fabric writes for 18 MB of payload with the default interleaving.)
__global__ __launch_bounds__(256) void tr_fill(const int2* __restrict__ packed, int e, const int* __restrict__ startT,
                                               const int* __restrict__ slot, int* __restrict__ tmp) {
    int t = xcd_contiguous(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (t < e) tmp[startT[packed[t].x] + slot[t]] = t;
}
// stable order inside a row: ascending edge id (the arrival order above is arbitrary). One thread per POSITION of
// the row-grouped list, so neighbouring lanes belong to the same row and scan the same addresses (broadcast loads);
// a thread per edge id would make every lane scan a different row (64 cache lines per load).
__global__ __launch_bounds__(256) void tr_rank(const int2* __restrict__ packed, int e, const int* __restrict__ startT,
                                               const int* __restrict__ tmp, int* __restrict__ permT) {
    int p = xcd_contiguous(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (p >= e) return;
    int v = tmp[p];
    int j = packed[v].x;
    int s0 = startT[j], s1 = startT[j + 1];
    int r = 0;
    int q = s0;
    for (; q + 4 <= s1; q += 4) {
        int a0 = tmp[q], a1 = tmp[q + 1], a2 = tmp[q + 2], a3 = tmp[q + 3];
        r += (a0 < v) + (a1 < v) + (a2 < v) + (a3 < v);
    }
    for (; q < s1; ++q) r += (tmp[q] < v) ? 1 : 0;
    permT[s0 + r] = v;
}

// ---- the three kernels above over a BATCH of lists (mccnn_geometry_prebuild_batch): one launch per phase
__global__ __launch_bounds__(256) void tr_count_batch(TrChainBatch tb, BatchBlocks bb) {
    __shared__ int bins[MCCNN_TR_LDS_BINS];
    int local, blocks;
    const TrChainItem& c = tb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const int t = local * 256 + threadIdx.x;
    if (!c.lds) {
        if (t < c.e) c.slot[t] = atomicAdd(&c.cnt[c.packed[t].x], 1);
        return;
    }
    for (int j = threadIdx.x; j < c.n; j += 256) bins[j] = 0;
    __syncthreads();
    int j = -1, loc = 0;
    if (t < c.e) {
        j = c.packed[t].x;
        loc = atomicAdd(&bins[j], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < c.n; k += 256) {
        const int v = bins[k];
        if (v) bins[k] = atomicAdd(&c.cnt[k], v);
    }
    __syncthreads();
    if (t < c.e) c.slot[t] = bins[j] + loc;
}
__global__ __launch_bounds__(256) void tr_fill_batch(TrChainBatch tb, BatchBlocks bb) {
    int local, blocks;
    const TrChainItem& c = tb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const int t = xcd_contiguous(local, blocks) * 256 + threadIdx.x;
    if (t < c.e) c.tmp[c.startT[c.packed[t].x] + c.slot[t]] = t;
}
__global__ __launch_bounds__(256) void tr_rank_batch(TrChainBatch tb, BatchBlocks bb) {
    int local, blocks;
    const TrChainItem& c = tb.it[batch_item(bb, (int)blockIdx.x, local, blocks)];
    const int p = xcd_contiguous(local, blocks) * 256 + threadIdx.x;
    if (p >= c.e) return;
    const int v = c.tmp[p];
    const int j = c.packed[v].x;
    const int s0 = c.startT[j], s1 = c.startT[j + 1];
    int r = 0;
    int q = s0;
    for (; q + 4 <= s1; q += 4) {
        int a0 = c.tmp[q], a1 = c.tmp[q + 1], a2 = c.tmp[q + 2], a3 = c.tmp[q + 3];
        r += (a0 < v) + (a1 < v) + (a2 < v) + (a3 < v);
    }
    for (; q < s1; ++q) r += (c.tmp[q] < v) ? 1 : 0;
    c.permT[s0 + r] = v;
}
int tr_chain_item(TrChainItem& t, ScanItem& sc, ClearSpan& head, const int* packed, int e, int n, int* start_t, int* perm_t, void* ws,
                  size_t ws_bytes) {
    if (e <= 0 || n <= 0 || (long long)n > 2048LL * 1024 || !start_t || !perm_t) return MCCNN_E_BADARG;
    if (!ws || ws_bytes < mccnn_transpose_neighbors_workspace_bytes(n, e)) return MCCNN_E_WORKSPACE;
    Arena ar(ws, ws_bytes);
    const size_t cntBytes = align_up((size_t)n * 4);
    char* blk = ar.take<char>(cntBytes + scan_workspace_bytes(n));
    int* slot = ar.take<int>((size_t)e);
    int* tmp = ar.take<int>((size_t)e);
    if (!blk || !slot || !tmp) return MCCNN_E_WORKSPACE;
    const int tiles = ceil_div(n, 2048);
    t = TrChainItem{reinterpret_cast<const int2*>(packed), (int*)blk, slot, tmp, start_t, perm_t, e, n,
                    (n <= MCCNN_TR_LDS_BINS && e >= 4 * n) ? 1 : 0, 0};
    sc = ScanItem{(const int*)blk, start_t, reinterpret_cast<unsigned long long*>(blk + cntBytes), start_t + n, nullptr, n, tiles};
    head = clear_span(blk, cntBytes + align_up((size_t)(tiles + 1) * 8));
    return 0;
}
int launch_tr_chain_batch(const TrChainBatch& tb, int count, int phase, hipStream_t s) {
    BatchBlocks bb;
    bb.count = count;
    int run = 0;
    for (int k = 0; k < count; ++k) { bb.first[k] = run; run += (phase == 2 && tb.it[k].norank) ? 0 : ceil_div(tb.it[k].e, 256); }
    for (int k = count; k <= MCCNN_BATCH_MAX; ++k) bb.first[k] = run;
    if (run == 0) return 0;
    if (phase == 0) tr_count_batch<<<run, 256, 0, s>>>(tb, bb);
    else if (phase == 1) tr_fill_batch<<<run, 256, 0, s>>>(tb, bb);
    else tr_rank_batch<<<run, 256, 0, s>>>(tb, bb);
    MCCNN_LAUNCHED();
    return 0;
}

// The whole transposition of a SMALL list in one workgroup of 1024 threads (memset, tr_count, scan, tr_fill, tr_rank: five
// launches for a few microseconds of work on the coarse levels of a hierarchy; the host pays ~6 us to issue each).
#define MCCNN_TR_SMALL_E 8192
#define MCCNN_TR_SMALL_N 8192
__device__ __forceinline__ void tr_small_body(const int2* __restrict__ packed, int e, int n, int* __restrict__ cnt,
                                              int* __restrict__ slot, int* __restrict__ tmp, int* __restrict__ startT,
                                              int* __restrict__ permT, int* wsum /* LDS, 17 */) {
    const int t = threadIdx.x;
    for (int j = t; j < n; j += 1024) cnt[j] = 0;
    __syncthreads();
    for (int k = t; k < e; k += 1024) slot[k] = atomicAdd(&cnt[packed[k].x], 1);
    __syncthreads();
    {
        const int per = (n + 1023) / 1024, j0 = min(n, t * per), j1 = min(n, j0 + per);
        int sum = 0;
        for (int j = j0; j < j1; ++j) sum += cnt[j];
        int tot;
        int run = block1024_excl_scan(sum, tot, wsum);
        for (int j = j0; j < j1; ++j) { const int v = cnt[j]; startT[j] = run; run += v; }
        if (t == 0) startT[n] = tot;
    }
    __syncthreads();
    for (int k = t; k < e; k += 1024) tmp[startT[packed[k].x] + slot[k]] = k;
    __syncthreads();
    for (int p = t; p < e; p += 1024) {  // stable order inside a row: ascending edge id
        const int v = tmp[p];
        const int j = packed[v].x;
        const int s0 = startT[j], s1 = startT[j + 1];
        int r = 0;
        for (int q = s0; q < s1; ++q) r += (tmp[q] < v) ? 1 : 0;
        permT[s0 + r] = v;
    }
}
__global__ __launch_bounds__(1024) void tr_small(const int2* __restrict__ packed, int e, int n, int* __restrict__ cnt,
                                                 int* __restrict__ slot, int* __restrict__ tmp, int* __restrict__ startT,
                                                 int* __restrict__ permT) {
    __shared__ int wsum[17];
    tr_small_body(packed, e, n, cnt, slot, tmp, startT, permT, wsum);
}
// ... of a BATCH of small lists (mccnn_geometry_prebuild_batch): one workgroup per list, one launch
__global__ __launch_bounds__(1024) void tr_small_batch(TrSmallBatch tb) {
    __shared__ int wsum[17];
    const TrSmallItem& t = tb.it[blockIdx.x];
    tr_small_body(t.packed, t.e, t.n, t.cnt, t.slot, t.tmp, t.startT, t.permT, wsum);
}
int launch_tr_small_batch(const TrSmallBatch& tb, int count, hipStream_t s) {
    if (count <= 0) return 0;
    tr_small_batch<<<count, 1024, 0, s>>>(tb);
    MCCNN_LAUNCHED();
    return 0;
}

// combin layers: featGrad[j, f] += dfE[e, f] (one atomic per edge and input feature)
// (`planes` > 1: the per-edge sums of the blocks lie in separate planes of `total` floats each, added here)
__global__ __launch_bounds__(256) void scatter_edge_featgrad(const int2* __restrict__ packed, const float* __restrict__ dfE,
                                                             long long total, int Fin, int planes,
                                                             float* __restrict__ featGrad) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    long long e = t / Fin;
    int f = (int)(t - e * Fin);
    float v = dfE[t];
    for (int p = 1; p < planes; ++p) v += dfE[(size_t)p * total + t];
    atomicAdd(&featGrad[(size_t)packed[e].x * Fin + f], v);
}

// The same sum without atomics, when the caller supplies the transposed neighbour list (it costs more to build than
// the atomics it saves, but a caller that prefetches the geometry of the next batch builds it for free on its side
// stream): 16 lanes per point split the point's edges, every lane adds the planes of its edges, a 4-step butterfly adds
// the lanes. Every row is written exactly once, in a fixed order: bit-reproducible feature gradients.
template <int FIN>
__global__ __launch_bounds__(256) void gather_edge_featgrad(const int* __restrict__ startT, const int* __restrict__ permT,
                                                            const float* __restrict__ dfE, long long total, int planes,
                                                            int n, int e, float* __restrict__ featGrad) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(tid >> 4), sub = (int)(tid & 15);
    float acc[FIN];
#pragma unroll
    for (int f = 0; f < FIN; ++f) acc[f] = 0.f;
    if (j < n) {
        const int t0 = startT[j], t1 = (j + 1 < n) ? startT[j + 1] : e;
        for (int t = t0 + sub; t < t1; t += 16) {
            const long long base = (long long)permT[t] * FIN;
            for (int p = 0; p < planes; ++p) {
#pragma unroll
                for (int f = 0; f < FIN; ++f) acc[f] += dfE[(size_t)p * total + base + f];
            }
        }
    }
#pragma unroll
    for (int f = 0; f < FIN; ++f) {
        acc[f] += __shfl_xor(acc[f], 8, 64);
        acc[f] += __shfl_xor(acc[f], 4, 64);
        acc[f] += __shfl_xor(acc[f], 2, 64);
        acc[f] += __shfl_xor(acc[f], 1, 64);
    }
    if (j < n && sub == 0) {
#pragma unroll
        for (int f = 0; f < FIN; ++f) featGrad[(size_t)j * FIN + f] = acc[f];
    }
}

// Sums the per-wave partial rows in a fixed order and scatters them to the six gradient tensors (body: conv_mfma.h).
__global__ __launch_bounds__(1024) void reduce_partials(const float* __restrict__ partials, int numWaves, int nb,
                                                        float* __restrict__ dw1, float* __restrict__ db1,
                                                        float* __restrict__ dw2, float* __restrict__ db2,
                                                        float* __restrict__ dw3, float* __restrict__ db3, ClearSpan x1) {
    clear_span_dev(x1);  // (the feature gradient the scatter pass that follows adds into: common.h)
    reduce_partials_body(blockIdx.x, partials, numWaves, nb, dw1, db1, dw2, db2, dw3, db3);
}

// ---------------------------------------------------------------------------------------
// Backward (spatial_conv.cu:327-445 / :563-680). A wave owns GB consecutive centres; for each
// block q it sweeps its edges keeping the 176 weight-gradient partial sums of that block in
// registers, reduces them across the wave once per (wave, q), combines the 4 waves of the
// workgroup in LDS and issues one global atomic per (workgroup, parameter). Feature gradients
// are scattered with float atomics (rows of different centres share j).
// ---------------------------------------------------------------------------------------
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
    a.n = n; a.m = m; a.e = e; a.Fin = Fin; a.Fout = Fout; a.radius = radius; a.invRadius = 1.0f / radius; a.scaleInv = scale_inv; a.avg = avg; a.B = batch_size;
    if (m > 0 && (!samples || !start_idx || !aabb_min || !aabb_max || !w1 || !b1 || !w2 || !b2 || !w3 || !b3))
        return MCCNN_E_BADARG;
    if (e > 0 && (!sorted_pts || !sorted_feats || !sorted_batch_ids || !pdfs || !packed)) return MCCNN_E_BADARG;
    return 0;
}

int launch_edge_records(const ConvArgs& a, float4* rec, hipStream_t s) {
    edge_records<<<ceil_div(a.e, 256), 256, 0, s>>>(a, rec);
    MCCNN_LAUNCHED();
    return 0;
}

void launch_reduce_partials(const float* partials, int rows, int nb, float* dw1, float* db1, float* dw2, float* db2,
                            float* dw3, float* db3, hipStream_t s) {
    reduce_partials<<<ceil_div((long long)nb * 176, 16), 1024, 0, s>>>(partials, rows, nb, dw1, db1, dw2, db2, dw3, db3, no_span());
}

int conv_fill_args(ConvArgs& a, const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                   const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                   const float* aabb_min, const float* aabb_max, const float* w1, const float* b1, const float* w2,
                   const float* b2, const float* w3, const float* b3, int n, int m, int e, int Fin, int Fout, int combin,
                   int batch_size, float radius, int scale_inv, int avg) {
    return fill_args(a, sorted_pts, sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min, aabb_max, w1,
                     b1, w2, b2, w3, b3, n, m, e, Fin, Fout, combin, batch_size, radius, scale_inv, avg);
}

std::atomic<int>& conv_impl_override() {
    static std::atomic<int> v{(debug_int("force_valu", 0) ? 1 : 0) | (debug_int("no_f1", 0) ? 2 : 0)};
    return v;
}

bool transpose_small(int e, int n) {
    // (the rank phase of tr_small is quadratic in the row length: e * e / n bounds the work of the worst case -- all edges on
    // one row -- to ~1 M compares spread over 1 024 threads, a few microseconds)
    return e <= MCCNN_TR_SMALL_E && n <= MCCNN_TR_SMALL_N && (long long)e * e / n <= (1LL << 20) && small_kernels_on();
}
// The first half of a transposition (large lists), shared with the fused plan build of conv_rows.hip: row counts with
// every edge's arrival slot, their prefix sums (start_t[n] = e). blk = [align_up(4 n) counters][scan workspace].
size_t transpose_count_bytes(int n) { return align_up((size_t)n * 4) + scan_workspace_bytes(n); }
int transpose_count(const int2* pk, int e, int n, int* start_t, char* blk, int* slot, hipStream_t s) {
    const size_t cntBytes = align_up((size_t)n * 4);
    int* cnt = (int*)blk;
    {   // the head of a transposition: row counters + the scan's status words (neighbours: one span)
        int rc = launch_clear_spans(clear_span(blk, cntBytes + scan_status_bytes(n)), no_span(), no_span(), s);
        if (rc) return rc;
    }
    if (n <= MCCNN_TR_LDS_BINS && e >= 4 * n) tr_count_lds<<<ceil_div(e, 256), 256, 0, s>>>(pk, e, n, cnt, slot);
    else tr_count<<<ceil_div(e, 256), 256, 0, s>>>(pk, e, cnt, slot);
    MCCNN_LAUNCHED();
    return exclusive_scan_i32(cnt, start_t, n, start_t + n, blk + cntBytes, s, true);
}
// ... and the second: edge ids grouped by row, arrival order inside a row
int transpose_fill(const int2* pk, int e, const int* start_t, const int* slot, int* tmp, hipStream_t s) {
    tr_fill<<<ceil_div(e, 256), 256, 0, s>>>(pk, e, start_t, slot, tmp);
    MCCNN_LAUNCHED();
    return 0;
}

}  // namespace mccnn

using namespace mccnn;

// Launch of the streaming reduction kernel: as many waves as the chip keeps resident (one balanced round).
template <bool TR>
static int launch_conv_stream(const ConvArgs& a, bool combin, bool vec, bool bf16, float* out, const float4* rec,
                              const int* permT, hipStream_t s, float4* recOut = nullptr) {
    typedef void (*Kern)(ConvArgs, float*, int, const float4*, const int*, float4*);
    Kern fn;
    if (bf16) fn = TR ? conv_stream<false, 4, true> : conv_stream<false, 4, false>;
    else if (TR) fn = vec ? conv_stream<false, 2, true> : conv_stream<false, 0, true>;
    else if (combin) fn = (a.Fin == 1) ? conv_stream<true, 1, false> : (a.Fin <= 4 ? conv_stream<true, 3, false> : conv_stream<true, 0, false>);
    else fn = vec ? conv_stream<false, 2, false> : conv_stream<false, 0, false>;
    const size_t lds = ((size_t)a.nb * MCCNN_WQ_FWD + 4 * (size_t)a.nb * 8) * sizeof(float);
    const int numCU = num_cus();
    const int perCU = cached_blocks_per_cu(reinterpret_cast<const void*>(fn), lds);
    const long long chunks = ((long long)a.e + 63) / 64;
    long long W = (long long)numCU * perCU * 4;
    if (W > (chunks + 1) / 2) W = (chunks + 1) / 2;  // at least ~2 chunks per wave
    if (W < 1) W = 1;
    fn<<<(int)((W + 3) / 4), 256, lds, s>>>(a, out, (int)W, rec, permT, recOut);
    MCCNN_LAUNCHED();
    return 0;
}

extern "C" {

int mccnn_debug_conv_impl(int mask) { return conv_impl_override().exchange(mask & 3); }

// combin layers with one input feature take the factored path of conv_f1.hip (f1_*)
static bool f1_shape(int num_in_feats, int num_out_feats, int combin) {
    return combin && num_in_feats == 1 && num_out_feats > 0 && (num_out_feats + 7) / 8 <= MCCNN_LDS_MAX_NB &&
           (conv_impl_override().load(std::memory_order_relaxed) & 3) == 0;
}

// Forward state: one record (delta, 1 / (pdf K)) per edge -- every layer on the MFMA kernels -- followed, for combin
// layers with one input feature, by the per-centre sums (A, S) of the factored algorithm.
static size_t state_record_bytes(int e) { return align_up((size_t)(e > 0 ? e : 0) * sizeof(float4)); }
size_t mccnn_spatial_conv_state_bytes(int m, int e, int num_in_feats, int num_out_feats, int combin) {
    if (m <= 0 || e <= 0 || num_in_feats <= 0 || num_out_feats <= 0) return 0;
    if (f1_shape(num_in_feats, num_out_feats, combin)) return state_record_bytes(e) + f1_state_bytes(m, (num_out_feats + 7) / 8);
    const long long neurons = combin ? (long long)num_in_feats * num_out_feats : num_in_feats;
    const bool mfma = (!combin || (neurons + 7) / 8 <= MCCNN_LDS_MAX_NB) && (conv_impl_override().load(std::memory_order_relaxed) & 1) == 0;
    return mfma ? state_record_bytes(e) : 0;
}

size_t mccnn_spatial_conv_fwd_workspace_bytes(int m, int e, int num_in_feats, int num_out_feats, int combin) {
    (void)e;
    if (m > 0 && f1_shape(num_in_feats, num_out_feats, combin)) return f1_fwd_workspace_bytes(m, (num_out_feats + 7) / 8);
    return 256;
}  // none needed today

// Depth-wise layers of any width run on the MFMA kernels: block q only touches feature columns 8q .. 8q+7, so a layer
// wider than one launch's LDS weight tile is cut into column tiles (pointer offsets, unchanged row strides). Combin
// layers beyond MCCNN_LDS_MAX_NB blocks (Fin * Fout > 512: not in the reference's models) keep the VALU kernels.
static bool use_mfma(const ConvArgs& a, bool combin) {
    return (a.nb <= MCCNN_LDS_MAX_NB || !combin) && (conv_impl_override().load(std::memory_order_relaxed) & 1) == 0;
}
#define MCCNN_TILE_FWD 64  // blocks per forward launch  (184 floats of LDS per block)
#define MCCNN_TILE_BWD 48  // blocks per backward launch (312 floats of LDS per block: 60 KB)
static void tile_split(int nb, int maxTile, int& tiles, int& per) {
    tiles = (nb + maxTile - 1) / maxTile;
    per = (nb + tiles - 1) / tiles;
}
// the arguments of column tile [q0, q0 + nbT) of a depth-wise layer (feature rows `elem` bytes per element)
static ConvArgs tile_args(const ConvArgs& a, int q0, int nbT, size_t elem) {
    ConvArgs t = a;
    t.feats = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.feats) + (size_t)q0 * 8 * elem);
    t.w1 = a.w1 + (size_t)q0 * 24; t.b1 = a.b1 + (size_t)q0 * 8;
    t.w2 = a.w2 + (size_t)q0 * 64; t.b2 = a.b2 + (size_t)q0 * 8;
    t.w3 = a.w3 + (size_t)q0 * 64; t.b3 = a.b3 + (size_t)q0 * 8;
    t.nb = nbT;
    t.neuronsOut = a.neuronsOut - q0 * 8 < nbT * 8 ? a.neuronsOut - q0 * 8 : nbT * 8;
    return t;
}
static float* col_offset(float* p, int q0, size_t elem) { return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + (size_t)q0 * 8 * elem); }
static const float* col_offset_c(const float* p, int q0, size_t elem) { return col_offset(const_cast<float*>(p), q0, elem); }

// bf16 feature storage: depth-wise layers on the MFMA path only (rows of Fin % 8 == 0 two-byte elements, 16-byte aligned)
static bool bf16_shape_ok(const ConvArgs& a, int combin, const void* p0, const void* p1) {
    return !combin && a.Fin % 8 == 0 && ((((uintptr_t)p0 | (uintptr_t)p1) & 15) == 0) &&
           (conv_impl_override().load(std::memory_order_relaxed) & 1) == 0;
}

static int conv_fwd_impl(const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                         const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                         const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3, int n, int m, int e,
                         int num_in_feats, int num_out_feats, int combin, int batch_size, float radius,
                         int scale_inv, int avg, float* out, void* state, void* ws, size_t ws_bytes,
                         mccnn_stream_t stream, int bf16) {
    ConvArgs a;
    int rc = fill_args(a, sorted_pts, sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min,
                       aabb_max, w1, b1, w2, b2, w3, b3, n, m, e, num_in_feats, num_out_feats, combin, batch_size,
                       radius, scale_inv, avg);
    if (rc) return rc;
    if (m == 0) return 0;
    if (!out) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    bool vec = !combin && (a.Fin % 8 == 0) && ((((uintptr_t)sorted_feats) & 15) == 0);
    if (bf16 && !bf16_shape_ok(a, combin, sorted_feats, out)) return MCCNN_E_SHAPE;
    float4* recOut = state ? reinterpret_cast<float4*>(state) : nullptr;
    if (e > 0 && f1_shape(num_in_feats, num_out_feats, combin))
        return f1_forward(a, out, recOut, state ? (char*)state + state_record_bytes(e) : nullptr, ws, ws_bytes, s);
    if (use_mfma(a, combin != 0) && e > 0) {
        if (combin || a.nb <= MCCNN_TILE_FWD) return launch_conv_stream<false>(a, combin != 0, vec, bf16 != 0, out, nullptr, nullptr, s, recOut);
        int tiles, per;
        tile_split(a.nb, MCCNN_TILE_FWD, tiles, per);
        const size_t elem = bf16 ? 2 : sizeof(float);
        for (int q0 = 0; q0 < a.nb; q0 += per) {
            const ConvArgs t = tile_args(a, q0, a.nb - q0 < per ? a.nb - q0 : per, elem);
            int rc2 = launch_conv_stream<false>(t, false, vec, bf16 != 0, col_offset(out, q0, elem), nullptr, nullptr, s,
                                                q0 == 0 ? recOut : nullptr);  // the records do not depend on the tile
            if (rc2) return rc2;
        }
        return 0;
    }
    if (e == 0) {
        return launch_zero_words(out, (size_t)m * a.outF * (bf16 ? 2 : sizeof(float)) / 4, s);  // (bf16 rows: outF is even)
    }
    // fallback for very wide layers (nb > MCCNN_LDS_MAX_NB): VALU kernel with scalar-loaded weights
    int G = 2048 / a.outF;
    if (G > 32) G = 32;
    if (G < 1) G = 1;
    a.G = G;
    size_t lds = (size_t)4 * G * a.outF * sizeof(float);
    if (lds > 64 * 1024) return MCCNN_E_TOOLARGE;
    int blocks = ceil_div(m, 4 * G);
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

int mccnn_spatial_conv_fwd(const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                           const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                           const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* w3, const float* b3, int n, int m, int e,
                           int num_in_feats, int num_out_feats, int combin, int batch_size, float radius,
                           int scale_inv, int avg, float* out, void* state, void* ws, size_t ws_bytes,
                           mccnn_stream_t stream) {
    return conv_fwd_impl(sorted_pts, sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min, aabb_max,
                         w1, b1, w2, b2, w3, b3, n, m, e, num_in_feats, num_out_feats, combin, batch_size, radius, scale_inv,
                         avg, out, state, ws, ws_bytes, stream, 0);
}

int mccnn_spatial_conv_fwd_bf16(const float* sorted_pts, const void* sorted_feats_bf16, const int* sorted_batch_ids,
                                const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                                const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                                const float* w2, const float* b2, const float* w3, const float* b3, int n, int m, int e,
                                int num_feats, int batch_size, float radius, int scale_inv, int avg, void* out_bf16,
                                void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    return conv_fwd_impl(sorted_pts, (const float*)sorted_feats_bf16, sorted_batch_ids, pdfs, samples, start_idx, packed,
                         aabb_min, aabb_max, w1, b1, w2, b2, w3, b3, n, m, e, num_feats, num_feats, 0, batch_size, radius,
                         scale_inv, avg, (float*)out_bf16, nullptr, ws, ws_bytes, stream, 1);
}

#ifndef MCCNN_BWD_WAVES
#define MCCNN_BWD_WAVES 2048    // 256 CUs x 4 SIMDs x 2 resident waves
#endif
// chunks per wave at least: amortises the per-(wave, block) reduction of the 176 partial sums. 8 until round 6 -- which left a
// short list on a tenth of the chip: BASELINE cfg0 (4 096 points, ~100 k edges, nb = 3) ran 200 waves of 24 sweeps each,
// conv_bwd_mfma 46.0 us; with 2 (800 waves of 6 sweeps) 25.0 us, reduce_partials 4.8 us either way (rocprofv3, same box;
// 1 measures the same as 2). Lists of >= 16 k chunks (1 M edges) are partitioned by MCCNN_BWD_WAVES as before.
// MCCNN_DEBUG=bwd_min_chunks=N: A/B.
#define MCCNN_BWD_MIN_CHUNKS 2

// Larger inputs get R equal rounds of resident-many waves with <= 32-chunk slices (see f1_bwd_partition): the slices in
// flight have to stay inside the Infinity Cache, the q-outer sweep re-reads them nb times.
static void bwd_partition(int e, int& cpw, int& waves) {
    long long chunks = ((long long)e + 63) / 64;
    cpw = (int)((chunks + MCCNN_BWD_WAVES - 1) / MCCNN_BWD_WAVES);
    if (cpw > 48) {
        const long long rounds = (cpw + 31) / 32;
        cpw = (int)((chunks + (long long)MCCNN_BWD_WAVES * rounds - 1) / ((long long)MCCNN_BWD_WAVES * rounds));
    }
    static const int minChunks = debug_int("bwd_min_chunks", MCCNN_BWD_MIN_CHUNKS);   // A/B switch, read once
    if (cpw < minChunks) cpw = minChunks < 1 ? 1 : minChunks;
    waves = (int)((chunks + cpw - 1) / cpw);
    if (waves < 1) waves = 1;
}

size_t mccnn_transpose_neighbors_workspace_bytes(int n, int e) {
    if (n <= 0 || e <= 0) return 256;
    return align_up((size_t)n * 4) + 2 * align_up((size_t)e * 4) + scan_workspace_bytes(n) + 256;
}

int mccnn_transpose_neighbors(const int* packed, int e, int n, int* start_t, int* perm_t, void* ws, size_t ws_bytes,
                              mccnn_stream_t stream) {
    if (e < 0 || n < 0 || !start_t) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (e == 0 || n == 0) {
        return launch_zero_words(start_t, (size_t)n + 1, s);
    }
    if (!packed || !perm_t) return MCCNN_E_BADARG;
    if (!ws || ws_bytes < mccnn_transpose_neighbors_workspace_bytes(n, e)) return MCCNN_E_WORKSPACE;
    Arena ar(ws, ws_bytes);
    // row counters and the scan's status words are neighbours: ONE memset clears both
    const size_t cntBytes = align_up((size_t)n * 4);
    char* blk = ar.take<char>(cntBytes + scan_workspace_bytes(n));
    int* slot = ar.take<int>((size_t)e);
    int* tmp = ar.take<int>((size_t)e);
    if (!blk || !slot || !tmp) return MCCNN_E_WORKSPACE;
    int* cnt = (int*)blk;
    const int2* pk = reinterpret_cast<const int2*>(packed);
    // the rank phase is quadratic in the row length: one workgroup only when the rows are short enough for it
    if (transpose_small(e, n)) {
        tr_small<<<1, 1024, 0, s>>>(pk, e, n, cnt, slot, tmp, start_t, perm_t);
        MCCNN_LAUNCHED();
        return 0;
    }
    int rc = transpose_count(pk, e, n, start_t, blk, slot, s);
    if (rc) return rc;
    rc = transpose_fill(pk, e, start_t, slot, tmp, s);
    if (rc) return rc;
    tr_rank<<<ceil_div(e, 256), 256, 0, s>>>(pk, e, start_t, tmp, perm_t);
    MCCNN_LAUNCHED();
    return 0;
}

size_t mccnn_spatial_conv_bwd_workspace_bytes(int n, int m, int e, int num_in_feats, int num_out_feats, int combin) {
    if (e <= 0 || num_in_feats <= 0 || num_out_feats <= 0) return 256;
    long long neurons = combin ? (long long)num_in_feats * num_out_feats : num_in_feats;
    long long nb = (neurons + 7) / 8;
    if (m > 0 && f1_shape(num_in_feats, num_out_feats, combin)) return f1_bwd_workspace_bytes(m, e, (int)nb);
    int cpw, waves;
    bwd_partition(e, cpw, waves);
    size_t bytes = align_up((size_t)(((long long)waves + 3) / 4 * 4 * nb * 176) * sizeof(float));  // partial rows
    bytes += align_up((size_t)e * sizeof(float4));                                                // edge records
    if (combin) bytes += align_up((size_t)e * num_in_feats * df_planes(num_in_feats, (int)nb) * sizeof(float));  // per-edge dFeat
    else bytes += align_up((size_t)(n + 1) * 4) + align_up((size_t)e * 4) +                       // start_t, perm_t
                  mccnn_transpose_neighbors_workspace_bytes(n, e);
    return bytes + 256;
}

static int conv_bwd_impl(const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                         const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                         const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3, const float* out_grad,
                         int n, int m, int e, int num_in_feats, int num_out_feats, int combin, int batch_size,
                         float radius, int scale_inv, int avg, const void* state, const int* start_t,
                         const int* perm_t, float* feat_grad, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                         void* ws, size_t ws_bytes, mccnn_stream_t stream, int bf16) {
    ConvArgs a;
    int rc = fill_args(a, sorted_pts, sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min,
                       aabb_max, w1, b1, w2, b2, w3, b3, n, m, e, num_in_feats, num_out_feats, combin, batch_size,
                       radius, scale_inv, avg);
    if (rc) return rc;
    if (!dw1 || !db1 || !dw2 || !db2 || !dw3 || !db3 || (n > 0 && !feat_grad)) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    size_t nn = (size_t)a.nb * 8;
    bool vec = !combin && (a.Fin % 8 == 0) && ((((uintptr_t)sorted_feats | (uintptr_t)out_grad) & 15) == 0);
    int tilesB = 1, perB = a.nb;
    if (!combin) tile_split(a.nb, MCCNN_TILE_BWD, tilesB, perB);
    size_t lds = ((size_t)perB * MCCNN_WQ_BWD + 4 * 192) * sizeof(float);
    bool mfma = use_mfma(a, combin != 0) && lds <= 64 * 1024 && m > 0 && e > 0;
    if (bf16 && (!bf16_shape_ok(a, combin, sorted_feats, out_grad) || (((uintptr_t)feat_grad) & 15))) return MCCNN_E_SHAPE;
    // depth-wise MFMA path writes every feat_grad row itself (conv_stream over the transposed list); everything else
    // accumulates into it
    bool dfeatT = mfma && !combin;
    // (the factored Fin = 1 path clears feat_grad in its centre pass, which runs before the edges add to it)
    const bool f1Clears = mfma && m > 0 && e > 0 && f1_shape(num_in_feats, num_out_feats, combin);
    // (... and the transposed gather of combin layers with 2..4 input features writes every row)
    const bool gatherT = mfma && combin && a.Fin >= 2 && a.Fin <= 4 && start_t && perm_t && m > 0 && e > 0;
    // feat_grad of the paths that ACCUMULATE into it (float atomics of scatter_edge_featgrad, the VALU fallback): cleared
    // by reduce_partials -- the kernel of this chain that runs before the first add -- when the rows allow 16-byte stores,
    // by a launch of its own otherwise
    const size_t fgBytes = (size_t)n * a.Fin * (bf16 ? 2 : sizeof(float));
    ClearSpan fgSpan = no_span();
    if (n > 0 && !dfeatT && !f1Clears && !gatherT) {
        const bool scatterPath = mfma && m > 0 && e > 0 && combin && a.Fin > 1;   // sweep -> reduce_partials -> scatter
        if (scatterPath && (((uintptr_t)feat_grad) & 15) == 0 && fgBytes % 16 == 0) {
            fgSpan = clear_span(feat_grad, fgBytes);
        } else {
            int rc = launch_zero_words(feat_grad, fgBytes / 4, s);
            if (rc) return rc;
        }
    }
    if (!mfma || m == 0 || e == 0) {  // (degenerate lists and the VALU fallback: the six sums start from zero)
        float* gp[6] = {dw1, db1, dw2, db2, dw3, db3};
        const size_t gn[6] = {3 * nn, nn, 8 * nn, nn, 8 * nn, nn};
        for (int k = 0; k < 6; ++k) {
            int rc = launch_zero_words(gp[k], gn[k], s);
            if (rc) return rc;
        }
    }
    if (m == 0 || e == 0) return 0;
    if (!out_grad) return MCCNN_E_BADARG;
    const float4* recIn = state ? reinterpret_cast<const float4*>(state) : nullptr;
    if (mfma && f1_shape(num_in_feats, num_out_feats, combin))
        return f1_backward(a, out_grad, recIn, state ? (const char*)state + state_record_bytes(e) : nullptr, feat_grad, dw1, db1,
                           dw2, db2, dw3, db3, ws, ws_bytes, s);
    if (mfma) {
        if (!ws || ws_bytes < mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, num_in_feats, num_out_feats, combin))
            return MCCNN_E_WORKSPACE;
        int cpw, waves;
        bwd_partition(e, cpw, waves);
        int blocks = (waves + 3) / 4;
        Arena ar(ws, ws_bytes);
        float* partials = ar.take<float>((size_t)blocks * 4 * a.nb * 176);
        float4* rec = ar.take<float4>((size_t)e);
        float* dfE = combin ? ar.take<float>((size_t)e * a.Fin * df_planes(a.Fin, a.nb)) : nullptr;
        if (!partials || !rec || (combin && !dfE)) return MCCNN_E_WORKSPACE;
        a.G = 0;
        if (recIn) {
            rec = const_cast<float4*>(recIn);  // left there by the forward call of the same inputs
        } else {
            edge_records<<<ceil_div(e, 256), 256, 0, s>>>(a, rec);
            MCCNN_LAUNCHED();
        }
        const size_t elem = bf16 ? 2 : sizeof(float);
        for (int q0 = 0; q0 < a.nb; q0 += perB) {  // one pass for combin layers, column tiles for wide depth-wise layers
            const int nbT = a.nb - q0 < perB ? a.nb - q0 : perB;
            const ConvArgs t = combin ? a : tile_args(a, q0, nbT, elem);
            const float* og = combin ? out_grad : col_offset_c(out_grad, q0, elem);
            const size_t ldsT = ((size_t)nbT * MCCNN_WQ_BWD + 4 * 192) * sizeof(float);
            const bool coop = !combin && nbT >= 4;
            const int rows = coop ? blocks : waves;  // partial rows to sum
            if (combin) {
                if (a.Fin == 1) conv_bwd_mfma<true, 1, false><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
                else if (a.Fin <= 4) conv_bwd_mfma<true, 3, false><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
                else conv_bwd_mfma<true, 0, false><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
            } else if (bf16) {
                if (coop) conv_bwd_mfma<false, 4, true><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
                else conv_bwd_mfma<false, 4, false><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
            } else if (coop) {
                if (vec) conv_bwd_mfma<false, 2, true><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
                else conv_bwd_mfma<false, 0, true><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
            } else {
                if (vec) conv_bwd_mfma<false, 2, false><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
                else conv_bwd_mfma<false, 0, false><<<blocks, 256, ldsT, s>>>(t, rec, og, feat_grad, dfE, cpw, partials);
            }
            MCCNN_LAUNCHED();
            reduce_partials<<<ceil_div((long long)nbT * 176, 16), 1024, 0, s>>>(
                partials, rows, nbT, dw1 + (size_t)q0 * 24, db1 + (size_t)q0 * 8, dw2 + (size_t)q0 * 64, db2 + (size_t)q0 * 8,
                dw3 + (size_t)q0 * 64, db3 + (size_t)q0 * 8, q0 == 0 ? fgSpan : no_span());
            MCCNN_LAUNCHED();
        }
        if (combin && a.Fin > 1) {  // Fin == 1: the main kernel adds each edge's finished sum itself
            long long total = (long long)e * a.Fin;
            const int planes = df_planes(a.Fin, a.nb);
            if (start_t && perm_t && a.Fin <= 4) {  // transposed list at hand: deterministic gather, no atomics
                const int gblocks = ceil_div((long long)n * 16, 256);
                if (a.Fin == 2) gather_edge_featgrad<2><<<gblocks, 256, 0, s>>>(start_t, perm_t, dfE, total, planes, n, e, feat_grad);
                else if (a.Fin == 3) gather_edge_featgrad<3><<<gblocks, 256, 0, s>>>(start_t, perm_t, dfE, total, planes, n, e, feat_grad);
                else gather_edge_featgrad<4><<<gblocks, 256, 0, s>>>(start_t, perm_t, dfE, total, planes, n, e, feat_grad);
            } else {
                scatter_edge_featgrad<<<ceil_div(total, 256), 256, 0, s>>>(a.packed, dfE, total, a.Fin, planes, feat_grad);
            }
            MCCNN_LAUNCHED();
        } else if (combin) {
        } else if (dfeatT) {
            if (!start_t || !perm_t) {  // not supplied by the caller: build the transposed list here
                int* st = ar.take<int>((size_t)n + 1);
                int* pt = ar.take<int>((size_t)e);
                size_t tb = mccnn_transpose_neighbors_workspace_bytes(n, e);
                void* tws = ar.take<char>(tb);
                if (!st || !pt || !tws) return MCCNN_E_WORKSPACE;
                int rc2 = mccnn_transpose_neighbors(packed, e, n, st, pt, tws, tb, stream);
                if (rc2) return rc2;
                start_t = st;
                perm_t = pt;
            }
            bool vecD = (a.Fin % 8 == 0) && ((((uintptr_t)out_grad) & 15) == 0);
            int tilesF, perF;
            tile_split(a.nb, MCCNN_TILE_FWD, tilesF, perF);
            for (int q0 = 0; q0 < a.nb; q0 += perF) {
                ConvArgs t = a;  // rows = the n neighbour points, CSR = start_t, gathered rows = outGrad
                t.feats = out_grad;
                t = tile_args(t, q0, a.nb - q0 < perF ? a.nb - q0 : perF, elem);
                t.start = start_t;
                t.m = n;
                int rc3 = launch_conv_stream<true>(t, false, vecD, bf16 != 0, col_offset(feat_grad, q0, elem), rec, perm_t, s);
                if (rc3) return rc3;
            }
        } else {
            return MCCNN_E_TOOLARGE;  // unreachable: the depth-wise tile always fits for nb <= MCCNN_LDS_MAX_NB
        }
        return 0;
    }
    a.G = 32;
    int blocks = ceil_div(m, 4 * a.G);
    if (combin) conv_bwd_valu<true><<<blocks, 256, 0, s>>>(a, out_grad, feat_grad, dw1, db1, dw2, db2, dw3, db3);
    else conv_bwd_valu<false><<<blocks, 256, 0, s>>>(a, out_grad, feat_grad, dw1, db1, dw2, db2, dw3, db3);
    MCCNN_LAUNCHED();
    return 0;
}

int mccnn_spatial_conv_bwd(const float* sorted_pts, const float* sorted_feats, const int* sorted_batch_ids,
                           const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                           const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* w3, const float* b3, const float* out_grad,
                           int n, int m, int e, int num_in_feats, int num_out_feats, int combin, int batch_size,
                           float radius, int scale_inv, int avg, const void* state, const int* start_t,
                           const int* perm_t, float* feat_grad, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                           void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    return conv_bwd_impl(sorted_pts, sorted_feats, sorted_batch_ids, pdfs, samples, start_idx, packed, aabb_min, aabb_max,
                         w1, b1, w2, b2, w3, b3, out_grad, n, m, e, num_in_feats, num_out_feats, combin, batch_size, radius,
                         scale_inv, avg, state, start_t, perm_t, feat_grad, dw1, db1, dw2, db2, dw3, db3, ws, ws_bytes, stream, 0);
}

int mccnn_spatial_conv_bwd_bf16(const float* sorted_pts, const void* sorted_feats_bf16, const int* sorted_batch_ids,
                                const float* pdfs, const float* samples, const int* start_idx, const int* packed,
                                const float* aabb_min, const float* aabb_max, const float* w1, const float* b1,
                                const float* w2, const float* b2, const float* w3, const float* b3,
                                const void* out_grad_bf16, int n, int m, int e, int num_feats, int batch_size, float radius,
                                int scale_inv, int avg, const int* start_t, const int* perm_t, void* feat_grad_bf16,
                                float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, void* ws,
                                size_t ws_bytes, mccnn_stream_t stream) {
    return conv_bwd_impl(sorted_pts, (const float*)sorted_feats_bf16, sorted_batch_ids, pdfs, samples, start_idx, packed,
                         aabb_min, aabb_max, w1, b1, w2, b2, w3, b3, (const float*)out_grad_bf16, n, m, e, num_feats, num_feats,
                         0, batch_size, radius, scale_inv, avg, nullptr, start_t, perm_t, (float*)feat_grad_bf16, dw1, db1,
                         dw2, db2, dw3, db3, ws, ws_bytes, stream, 1);
}

}  // extern "C"
