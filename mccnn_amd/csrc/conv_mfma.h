// Shared device helpers of the MFMA convolution kernels (conv.hip, conv_f1.hip): argument block, the 4x4x1 MFMA form
// of the 8x8 kernel-MLP layers, scheduling phases, fused-DPP segmented scan, slice search, transposing butterfly.
#pragma once
#include "common.h"
#include <atomic>
#include <mutex>

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
    float radius, invRadius;
    int scaleInv, avg, G, B;
};

// ---------------------------------------------------------------------------------------
// MFMA kernels (nb <= MCCNN_LDS_MAX_NB).
//
// v_mfma_f32_4x4x1_16b_f32 computes 16 independent 4x4 rank-1 updates per wave:
//     D[lane 4b+j][reg r] += A(lane 4b+r) * B(lane 4b+j)          (layout probed on gfx950, profiles/)
// With lane = edge and reg = neuron this is exactly one k-step of an 8x8 block of the kernel MLP
// for 64 edges at once and with ZERO block-diagonal waste (a 16x16x4 tiling wastes half of every
// MFMA on the off-diagonal zeros): A = the weight W[r][k] (same for every quad), B = the lane's
// own activation h[k]. Two accumulators (neurons 0-3 / 4-7) x 8 k-steps = 16 MFMAs per layer. An f32 MFMA is
// bit for bit a k-ordered fmaf chain, and the chain is the one the reference compiles to (nvcc contracts
// `aux += h*w` into fma; spatial_conv.cu:57-62,372-378): it starts at ZERO and the bias comes LAST, as one more
// k-step with B = 1.0 (fma(b, 1, acc) == acc + b exactly). Pre-activations -- and with them every ReLU / ReLU'
// decision -- are therefore bit-identical to those of the test suite's sequential CPU restatement of the same chains.
// The transposed products of the backward pass (t3 = W3^T (g f), t4 = W2^T t3) use the same form
// with the transposed weight copies staged in LDS.
// ---------------------------------------------------------------------------------------
#define MCCNN_LDS_MAX_NB 64
// floats per MLP block in LDS: W1[8][4] b1[8] W2[8][8] b2[8] W3[8][8] b3[8] (+ W2^T[8][8] W3^T[8][8] for bwd)
#define MCCNN_WQ_FWD 184
#define MCCNN_WQ_BWD 312

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_SHL(n) (0x100 + (n))

// MFMA and VALU instructions of one wave are kept in separate PHASES: tools/issue_probe.hip shows that a wave
// alternating between the two pays ~7 extra cycles per switch (8 mfma + 16 fma cost 163 cycles interleaved, 108
// grouped = the sum of the parts), and the scheduler's default is to interleave. SALU and memory ops may cross.
#ifndef MCCNN_NO_PHASES
#define MCCNN_PHASE() __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80)
#else
#define MCCNN_PHASE()
#endif

// max(x, 0) as ONE instruction: v_med3_f32(x, 0, +inf). fmaxf() costs an extra canonicalising v_max on MFMA results,
// and med3 with a literal +inf is folded back into that pair -- so the +inf goes through an opaque SGPR.
__device__ __forceinline__ float relu1(float x) {
    float inf = __builtin_huge_valf();
    asm("" : "+s"(inf));
    return __builtin_amdgcn_fmed3f(x, 0.0f, inf);
}

// Stage the MLP tensors of all nb blocks into LDS in the per-block layout above.
template <int WQ>
__device__ __forceinline__ void stage_weights(const ConvArgs& a, float* wl) {
    for (int t = threadIdx.x; t < a.nb * WQ; t += blockDim.x) {
        int q = t / WQ, r = t - q * WQ;
        float v;
        if (r < 32) { int row = r >> 2, c = r & 3; v = (c < 3) ? a.w1[(q * 8 + row) * 3 + c] : a.b1[q * 8 + row]; }  // row = (w0, w1, w2, b1)
        else if (r < 40) v = a.b1[q * 8 + r - 32];
        else if (r < 104) v = a.w2[q * 64 + r - 40];
        else if (r < 112) v = a.b2[q * 8 + r - 104];
        else if (r < 176) v = a.w3[q * 64 + r - 112];
        else if (r < 184) v = a.b3[q * 8 + r - 176];
        else if (r < 248) { int k = r - 184; v = a.w2[q * 64 + (k & 7) * 8 + (k >> 3)]; }   // W2^T[l][m] = W2[m][l]
        else { int k = r - 248; v = a.w3[q * 64 + (k & 7) * 8 + (k >> 3)]; }                  // W3^T[m][n] = W3[n][m]
        wl[t] = v;
    }
}

// 1.0 in a VGPR the optimiser cannot see through (the bias k-step's B operand)
__device__ __forceinline__ float opaque_one() {
    float one = 1.0f;
    asm("" : "+v"(one));
    return one;
}

// One 8x8 layer for 64 edges: y = W x (+ bias), rows i4 / 4+i4 of W supplied by this lane. Two interleaved
// accumulation chains (neurons 0-3 / 4-7) that start at zero; splitting K into more independent chains was measured
// slower (tools/issue_probe.hip: a single dependent 4x4x1 chain already issues every ~15 cycles and is hidden from 2
// waves per SIMD up). BIAS: bias[i4] / bias[4 + i4] enter as a ninth k-step against B = 1.0.
template <bool BIAS>
__device__ __forceinline__ void layer8(const f32x4* __restrict__ wrows /* 16 x f32x4: row r at [2r],[2r+1] */,
                                       const float* __restrict__ bias, int i4, const float* x, float* y) {
#ifdef MCCNN_ABL_NOLDSW  // timing ablation (wrong results): the weight operands come from registers, no LDS read
    f32x4 al0 = {x[0], x[1], x[2], x[3]}, al1 = {x[4], x[5], x[6], x[7]};
    f32x4 ah0 = {x[1], x[2], x[3], x[4]}, ah1 = {x[5], x[6], x[7], x[0]};
#else
    f32x4 al0 = wrows[2 * i4], al1 = wrows[2 * i4 + 1];
    f32x4 ah0 = wrows[2 * (4 + i4)], ah1 = wrows[2 * (4 + i4) + 1];
#endif
    float al[8] = {al0.x, al0.y, al0.z, al0.w, al1.x, al1.y, al1.z, al1.w};
    float ah[8] = {ah0.x, ah0.y, ah0.z, ah0.w, ah1.x, ah1.y, ah1.z, ah1.w};
    float bl = 0.f, bh = 0.f;
#ifdef MCCNN_ABL_NOLDSW
    if (BIAS) { bl = x[0]; bh = x[1]; }
#else
    if (BIAS) { bl = bias[i4]; bh = bias[4 + i4]; }
#endif
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo = MFMA4(al[k], x[k], lo);
        hi = MFMA4(ah[k], x[k], hi);
    }
    if (BIAS) {
        const float one = opaque_one();
        lo = MFMA4(bl, one, lo);
        hi = MFMA4(bh, one, hi);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { y[r] = lo[r]; y[4 + r] = hi[r]; }
}

// The same layer with the lane's weight rows already in registers (kernels that keep one block's weights resident
// across a whole sweep: no LDS traffic in the loop). al / ah: rows i4 and 4 + i4 of W, bl / bh their biases.
template <bool BIAS>
__device__ __forceinline__ void layer8_regs(const float* al, const float* ah, float bl, float bh, const float* x, float* y) {
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo = MFMA4(al[k], x[k], lo);
        hi = MFMA4(ah[k], x[k], hi);
    }
    if (BIAS) {
        const float one = opaque_one();
        lo = MFMA4(bl, one, lo);
        hi = MFMA4(bh, one, hi);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { y[r] = lo[r]; y[4 + r] = hi[r]; }
}

// One block's forward weights as the lane needs them for the 4x4x1 form (44 VGPRs), read straight from the flat
// tensors (w1[nu][3], w2 / w3 [q][out][in]): rows i4 and 4 + i4 of every layer.
struct BlockWeights {
    float w1lo[4], w1hi[4];      // (w0, w1, w2, b1) of neurons i4 / 4 + i4
    float w2lo[8], w2hi[8], b2lo, b2hi;
    float w3lo[8], w3hi[8], b3lo, b3hi;
};
__device__ __forceinline__ void load_block_weights(const ConvArgs& a, int q, int i4, BlockWeights& w) {
    const int nl = q * 8 + i4, nh = nl + 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) { w.w1lo[c] = a.w1[nl * 3 + c]; w.w1hi[c] = a.w1[nh * 3 + c]; }
    w.w1lo[3] = a.b1[nl]; w.w1hi[3] = a.b1[nh];
    const float4* r2l = reinterpret_cast<const float4*>(a.w2 + (size_t)q * 64 + i4 * 8);
    const float4* r2h = reinterpret_cast<const float4*>(a.w2 + (size_t)q * 64 + (4 + i4) * 8);
    const float4* r3l = reinterpret_cast<const float4*>(a.w3 + (size_t)q * 64 + i4 * 8);
    const float4* r3h = reinterpret_cast<const float4*>(a.w3 + (size_t)q * 64 + (4 + i4) * 8);
    const float4 a0 = r2l[0], a1 = r2l[1], b0 = r2h[0], b1 = r2h[1], c0 = r3l[0], c1 = r3l[1], d0 = r3h[0], d1 = r3h[1];
    w.w2lo[0] = a0.x; w.w2lo[1] = a0.y; w.w2lo[2] = a0.z; w.w2lo[3] = a0.w; w.w2lo[4] = a1.x; w.w2lo[5] = a1.y; w.w2lo[6] = a1.z; w.w2lo[7] = a1.w;
    w.w2hi[0] = b0.x; w.w2hi[1] = b0.y; w.w2hi[2] = b0.z; w.w2hi[3] = b0.w; w.w2hi[4] = b1.x; w.w2hi[5] = b1.y; w.w2hi[6] = b1.z; w.w2hi[7] = b1.w;
    w.w3lo[0] = c0.x; w.w3lo[1] = c0.y; w.w3lo[2] = c0.z; w.w3lo[3] = c0.w; w.w3lo[4] = c1.x; w.w3lo[5] = c1.y; w.w3lo[6] = c1.z; w.w3lo[7] = c1.w;
    w.w3hi[0] = d0.x; w.w3hi[1] = d0.y; w.w3hi[2] = d0.z; w.w3hi[3] = d0.w; w.w3hi[4] = d1.x; w.w3hi[5] = d1.y; w.w3hi[6] = d1.z; w.w3hi[7] = d1.w;
    w.b2lo = a.b2[nl]; w.b2hi = a.b2[nh];
    w.b3lo = a.b3[nl]; w.b3hi = a.b3[nh];
}
// Kernel MLP of one block from register-resident weights: identical chains (and bits) as mlp_block_mfma.
__device__ __forceinline__ void mlp_block_regs(const BlockWeights& w, float d0, float d1, float d2, float* a1, float* a2, float* o) {
    float pre[8];
    {
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
        const float one = opaque_one();
        lo = MFMA4(w.w1lo[0], d0, lo);
        hi = MFMA4(w.w1hi[0], d0, hi);
        lo = MFMA4(w.w1lo[1], d1, lo);
        hi = MFMA4(w.w1hi[1], d1, hi);
        lo = MFMA4(w.w1lo[2], d2, lo);
        hi = MFMA4(w.w1hi[2], d2, hi);
        lo = MFMA4(w.w1lo[3], one, lo);
        hi = MFMA4(w.w1hi[3], one, hi);
#pragma unroll
        for (int r = 0; r < 4; ++r) { pre[r] = lo[r]; pre[4 + r] = hi[r]; }
    }
    MCCNN_PHASE();
#pragma unroll
    for (int r = 0; r < 8; ++r) a1[r] = relu1(pre[r]);
    MCCNN_PHASE();
    layer8_regs<true>(w.w2lo, w.w2hi, w.b2lo, w.b2hi, a1, pre);
    MCCNN_PHASE();
#pragma unroll
    for (int r = 0; r < 8; ++r) a2[r] = relu1(pre[r]);
    MCCNN_PHASE();
    layer8_regs<true>(w.w3lo, w.w3hi, w.b3lo, w.b3hi, a2, o);
    MCCNN_PHASE();
}

// Layer 1 for 64 edges: pre1 = ((d0 w0 + d1 w1) + d2 w2) + b1 as fma steps (spatial_conv.cu:49-52); the LDS row of
// neuron r is (w0, w1, w2, b1).
__device__ __forceinline__ void layer1_mfma(const f32x4* __restrict__ w, int i4, float d0, float d1, float d2, float* pre1) {
#ifdef MCCNN_ABL_NOLDSW
    f32x4 a1lo = {d0, d1, d2, d0}, a1hi = {d1, d2, d0, d1};
#else
    f32x4 a1lo = w[i4], a1hi = w[4 + i4];
#endif
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    const float one = opaque_one();
    lo = MFMA4(a1lo.x, d0, lo);
    hi = MFMA4(a1hi.x, d0, hi);
    lo = MFMA4(a1lo.y, d1, lo);
    hi = MFMA4(a1hi.y, d1, hi);
    lo = MFMA4(a1lo.z, d2, lo);
    hi = MFMA4(a1hi.z, d2, hi);
    lo = MFMA4(a1lo.w, one, lo);
    hi = MFMA4(a1hi.w, one, hi);
#pragma unroll
    for (int r = 0; r < 4; ++r) { pre1[r] = lo[r]; pre1[4 + r] = hi[r]; }
}

// Layers 1 and 2 of block q (weights at wq in LDS) for the 64 edges of a wave. a1 = relu(pre1), a2 = relu(pre2).
__device__ __forceinline__ void mlp_block_l12(const float* __restrict__ wq, int i4, float d0, float d1, float d2,
                                              float* pre1, float* a1, float* pre2, float* a2) {
    const f32x4* w = reinterpret_cast<const f32x4*>(wq);
    layer1_mfma(w, i4, d0, d1, d2, pre1);
    MCCNN_PHASE();
#pragma unroll
    for (int r = 0; r < 8; ++r) a1[r] = relu1(pre1[r]);
    MCCNN_PHASE();
    layer8<true>(w + 10, wq + 104, i4, a1, pre2);  // W2, b2
    MCCNN_PHASE();
#pragma unroll
    for (int r = 0; r < 8; ++r) a2[r] = relu1(pre2[r]);
    MCCNN_PHASE();
}

// Kernel MLP of block q (weights at wq in LDS) for the 64 edges of a wave.
// a1 = relu(pre1), a2 = relu(pre2) are returned because backward needs them.
__device__ __forceinline__ void mlp_block_mfma(const float* __restrict__ wq, int i4, float d0, float d1, float d2,
                                               float* pre1, float* a1, float* pre2, float* a2, float* o) {
    const f32x4* w = reinterpret_cast<const f32x4*>(wq);
    mlp_block_l12(wq, i4, d0, d1, d2, pre1, a1, pre2, a2);
    layer8<true>(w + 28, wq + 176, i4, a2, o);     // W3, b3
    MCCNN_PHASE();
}

// bf16 feature storage (depth-wise layers, extension: the reference is f32-only): rows are stored as bf16, every value is
// widened exactly to f32 on load, all arithmetic and accumulation stay f32, results are rounded to nearest-even on store
// (v_cvt_pk_bf16_f32). A block's 8 features are one 16-byte piece.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf16x8_to_f32(const uint4 u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned f32x2_to_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint4 f32x8_to_bf16(const float* c) {
    return make_uint4(f32x2_to_bf16(c[0], c[1]), f32x2_to_bf16(c[2], c[3]), f32x2_to_bf16(c[4], c[5]), f32x2_to_bf16(c[6], c[7]));
}

// Correctly rounded x / R from y = RN(1 / R) in three instructions (Markstein): q = RN(x y) is within one ulp,
// r = x - q R is exact in an fma, RN(q + r y) is the correctly rounded quotient. The reference divides
// (spatial_conv.cu:155-158); the quotient feeds layer 1, whose pre-activations have to be
// bit-identical (see above). Spelled with fma builtins: the library is compiled with -ffp-contract=off.
__device__ __forceinline__ float div_exact(float x, float R, float invR) {
    const float q = x * invR;
    const float r = __builtin_fmaf(-q, R, x);
    return __builtin_fmaf(r, invR, q);
}

// The record of edge t: (delta0, delta1, delta2, 1 / (pdf K)) -- delta = (point - centre) / R correctly rounded
// (spatial_conv.cu:155-158), K = the centre's neighbour count when the layer averages (spatial_conv.cu:160-166).
__device__ __forceinline__ float4 edge_record(const ConvArgs& a, int t) {
    const int2 pr = a.packed[t];
    float invR = a.invRadius;
    if (a.scaleInv) invR = 1.0f / (a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[pr.x], a.B)));
    const float* p = a.pts + (size_t)pr.x * 3;
    const float* c = a.samples + (size_t)pr.y * 3;
    const int e0 = a.start[pr.y];
    const int e1 = (pr.y < a.m - 1) ? a.start[pr.y + 1] : a.e;
    const float K = a.avg ? (float)(e1 - e0) : 1.0f;
    const float R = a.scaleInv ? a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[pr.x], a.B)) : a.radius;
    return make_float4(div_exact(p[0] - c[0], R, invR), div_exact(p[1] - c[1], R, invR), div_exact(p[2] - c[2], R, invR),
                       __builtin_amdgcn_rcpf(a.pdfs[t] * K));
}

// smallest c in [0, m] with S(c) >= t, S(c) = start[c] (c < m), S(m) = e. 64-ary: three dependent loads for m < 2^18.
__device__ __forceinline__ int wave_lower_bound(const int* __restrict__ start, int m, int e, int t, int lane) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int span = hi - lo;
        const int step = (span + 63) >> 6;
        const int p = min(lo + lane * step, hi);
        const int v = (p < m) ? start[p] : e;
        const unsigned long long b = __ballot(v >= t);
        const int f = b ? (int)__builtin_ctzll(b) : 64;
        const int nlo = (f == 0) ? lo : min(lo + (f - 1) * step, hi) + 1;
        const int nhi = (f == 0) ? lo : ((f == 64) ? hi : min(lo + f * step, hi));
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_rows_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_rows_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xf, false);
}

// Segmented inclusive scan of 8 values over the 64 lanes: v += m_step * v[lane - step]. The DPP read is folded
// into the multiply-add (v_fmac_f32_dpp) -- from C++ the compiler emits v_mov_b32_dpp + v_fma (VOP3 cannot carry
// DPP) -- and the 8 values advance in lock step, so an instruction never reads a register written less than 8
// instructions earlier (the VALU-write -> DPP-read hazard needs 2 wait states; inline asm gets no automatic nops).
#define MCCNN_SCAN_STEP(CTRL, M)                                                         \
    "v_fmac_f32_dpp %0, %0, %" #M " " CTRL "\n v_fmac_f32_dpp %1, %1, %" #M " " CTRL "\n" \
    "v_fmac_f32_dpp %2, %2, %" #M " " CTRL "\n v_fmac_f32_dpp %3, %3, %" #M " " CTRL "\n" \
    "v_fmac_f32_dpp %4, %4, %" #M " " CTRL "\n v_fmac_f32_dpp %5, %5, %" #M " " CTRL "\n" \
    "v_fmac_f32_dpp %6, %6, %" #M " " CTRL "\n v_fmac_f32_dpp %7, %7, %" #M " " CTRL "\n"
__device__ __forceinline__ void wave_seg_scan8(float* c, float m1, float m2, float m4, float m8, float mA, float mB) {
    asm("s_nop 1\n"
        MCCNN_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", 8)
        MCCNN_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", 9)
        MCCNN_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", 10)
        MCCNN_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", 11)
        MCCNN_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", 12)
        MCCNN_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf", 13)
        "s_nop 1\n"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
        : "v"(m1), "v"(m2), "v"(m4), "v"(m8), "v"(mA), "v"(mB));
}

// Transposing butterfly: 64 per-lane values -> lane l ends with sum over all lanes of v[l].
// 63 exchanges instead of the 64 x 6 of a per-value wave_sum. (Template recursion keeps every index static:
// a runtime-indexed register array would be demoted to scratch.)
template <int S>
__device__ __forceinline__ void butterfly_step(float* v, int lane) {
    const bool upper = (lane & S) != 0;
    // three passes so that the S cross-lane exchanges of a step are in flight together (written as one loop the
    // compiler waits for every ds_bpermute before issuing the next: ~100 cycles each, 63 per reduction)
    float send[S], keep[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        send[k] = upper ? v[k] : v[k + S];
        keep[k] = upper ? v[k + S] : v[k];
    }
#pragma unroll
    for (int k = 0; k < S; ++k) send[k] = __shfl_xor(send[k], S, 64);
#pragma unroll
    for (int k = 0; k < S; ++k) v[k] = keep[k] + send[k];
}
__device__ __forceinline__ float wave_reduce64(float* v, int lane) {
    butterfly_step<32>(v, lane);
    butterfly_step<16>(v, lane);
    butterfly_step<8>(v, lane);
    butterfly_step<4>(v, lane);
    butterfly_step<2>(v, lane);
    butterfly_step<1>(v, lane);
    return v[0];
}

// Combin layers with 2..4 input features: neuron nu = fo * Fin + fin, so inside a block the feature index of neuron n
// is (r0 + n) % Fin and its output feature fo0 + (r0 + n) / Fin with r0 = (8 q) % Fin, fo0 = (8 q) / Fin. Fin and r0 are
// wave-uniform: one branch selects a fully static register pattern (no per-neuron gathers, no select chains).
template <int FIN, int R0>
__device__ __forceinline__ void combin_pick(const float* __restrict__ fs, const float* __restrict__ gw, float* ff, float* g) {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        ff[n] = fs[(R0 + n) % FIN];
        if (gw) g[n] = gw[(R0 + n) / FIN];
    }
}
__device__ __forceinline__ void combin_select(int Fin, int r0, const float* fs, const float* gw, float* ff, float* g) {
    switch (Fin * 4 + r0) {
        case 2 * 4 + 0: combin_pick<2, 0>(fs, gw, ff, g); break;
        case 2 * 4 + 1: combin_pick<2, 1>(fs, gw, ff, g); break;
        case 3 * 4 + 0: combin_pick<3, 0>(fs, gw, ff, g); break;
        case 3 * 4 + 1: combin_pick<3, 1>(fs, gw, ff, g); break;
        case 3 * 4 + 2: combin_pick<3, 2>(fs, gw, ff, g); break;
        case 4 * 4 + 0: combin_pick<4, 0>(fs, gw, ff, g); break;
        case 4 * 4 + 1: combin_pick<4, 1>(fs, gw, ff, g); break;
        case 4 * 4 + 2: combin_pick<4, 2>(fs, gw, ff, g); break;
        default: combin_pick<4, 3>(fs, gw, ff, g); break;
    }
}

template <int FIN, int R0>
__device__ __forceinline__ void combin_fold_t(const float* g, const float* o, float* sf) {
#pragma unroll
    for (int n = 0; n < 8; ++n) sf[(R0 + n) % FIN] = fmaf(g[n], o[n], sf[(R0 + n) % FIN]);
}
__device__ __forceinline__ void combin_fold(int Fin, int r0, const float* g, const float* o, float* sf) {
    switch (Fin * 4 + r0) {
        case 2 * 4 + 0: combin_fold_t<2, 0>(g, o, sf); break;
        case 2 * 4 + 1: combin_fold_t<2, 1>(g, o, sf); break;
        case 3 * 4 + 0: combin_fold_t<3, 0>(g, o, sf); break;
        case 3 * 4 + 1: combin_fold_t<3, 1>(g, o, sf); break;
        case 3 * 4 + 2: combin_fold_t<3, 2>(g, o, sf); break;
        case 4 * 4 + 0: combin_fold_t<4, 0>(g, o, sf); break;
        case 4 * 4 + 1: combin_fold_t<4, 1>(g, o, sf); break;
        case 4 * 4 + 2: combin_fold_t<4, 2>(g, o, sf); break;
        default: combin_fold_t<4, 3>(g, o, sf); break;
    }
}

// Resident 256-thread workgroups per CU for a kernel / dynamic-LDS size. The query is a driver call (~10 us), so the
// answers are cached; the table is append-only behind a mutex (the ops may be called from several host threads).
inline int cached_blocks_per_cu(const void* fn, size_t lds) {
    struct Entry { const void* fn; size_t lds; int n; };
    static Entry cache[32];
    static int used = 0;
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> g(mu);
        for (int i = 0; i < used; ++i)
            if (cache[i].fn == fn && cache[i].lds == lds) return cache[i].n;
    }
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, lds) != hipSuccess || n < 1) n = 1;
    std::lock_guard<std::mutex> g(mu);
    if (used < 32) cache[used++] = {fn, lds, n};
    return n;
}

// CUs of the current device (constant per process for a homogeneous node; initialised once, thread-safe).
inline int num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
            v = 256;
        return v;
    }();
    return n;
}

// Implementation override for A/B tests of the conv kernels: bit 0 = VALU fallback kernels, bit 1 = general MFMA
// kernels for one-input-feature layers. Read ONCE from the environment (MCCNN_FORCE_VALU / MCCNN_NO_F1) when the
// library is first used; tests switch it through mccnn_debug_conv_impl(). Not part of the product configuration.
std::atomic<int>& conv_impl_override();

// conv_f1.hip: combin layers with one input feature (layer 3 factored out of the edge sum)
size_t f1_state_bytes(int m, int nb);
size_t f1_fwd_workspace_bytes(int m, int nb);
size_t f1_bwd_workspace_bytes(int m, int e, int nb);
int f1_forward(const ConvArgs& a, float* out, float4* rec_out, void* state, void* ws, size_t ws_bytes, hipStream_t s);
int f1_backward(const ConvArgs& a, const float* out_grad, const float4* rec_in, const void* state, float* feat_grad, float* dw1, float* db1,
                float* dw2, float* db2, float* dw3, float* db3, void* ws, size_t ws_bytes, hipStream_t s);

// Sums the per-wave partial rows in a fixed order and scatters them to the six gradient tensors: the work of workgroup
// `bid` of 1024 threads (the reduce_partials kernel of conv.hip, and the first workgroups of rows_combine_reduce).
__device__ __forceinline__ void reduce_partials_body(int bid, const float* __restrict__ partials, int numWaves, int nb,
                                                        float* __restrict__ dw1, float* __restrict__ db1,
                                                        float* __restrict__ dw2, float* __restrict__ db2,
                                                        float* __restrict__ dw3, float* __restrict__ db3) {
    // 16 consecutive parameters x 64 row slices per workgroup, 4 independent accumulators per thread: the sum over
    // ~2000 rows is a chain of dependent load latencies, 256 rows in flight per parameter make it 8 links long
    __shared__ float acc[64][17];
    const int K = nb * 176;
    const int kk = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int k = bid * 16 + kk;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (k < K) {
        int w = sl;
        for (; w + 192 < numWaves; w += 256) {
            s0 += partials[(size_t)w * K + k];
            s1 += partials[(size_t)(w + 64) * K + k];
            s2 += partials[(size_t)(w + 128) * K + k];
            s3 += partials[(size_t)(w + 192) * K + k];
        }
        for (; w < numWaves; w += 64) s0 += partials[(size_t)w * K + k];
    }
    acc[sl][kk] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && k < K) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) v += acc[i][kk];
        int q = k / 176, r = k - q * 176;
        if (r < 24) dw1[q * 24 + r] = v;
        else if (r < 32) db1[q * 8 + r - 24] = v;
        else if (r < 96) dw2[q * 64 + r - 32] = v;
        else if (r < 104) db2[q * 8 + r - 96] = v;
        else if (r < 168) dw3[q * 64 + r - 104] = v;
        else db3[q * 8 + r - 168] = v;
    }
}


}  // namespace mccnn
