// Monte-Carlo convolution for combin layers with ONE input feature (the first layer of every MCCNN network: constant
// or intensity input at the finest level -- the layer with the most points and edges). Same results as the general
// kernels of conv.hip (spatial_conv.cu:24-79 forward, :327-445 backward) up to float summation order.
//
// With Fin = 1 the edge weight s_e = f_j / (pdf_e K_i) is a scalar, so the third MLP layer commutes with the sum over
// a centre's edges (block q, 8 neurons, a2_e = second hidden layer of edge e):
//     out_i[q]  = sum_e s_e (W3 a2_e + b3)   =  W3 A_i + b3 S_i,        A_i = sum_e s_e a2_e,   S_i = sum_e s_e
//     dW3       = sum_i g_i (x) A_i,         db3 = sum_i g_i S_i
//     t3_e      = 1[pre2_e >= 0] * (W3^T g_i) s_e  =  1[pre2_e >= 0] * G_i s_e
//     dFeat_j  += (G_i . a2_e + g_i . b3) / (pdf_e K_i)
// Layer 3 and everything that multiplies by it moves from the E edges to the M centres: per (64-edge chunk, block) the
// forward issues 22 MFMAs instead of 38, the backward 38 instead of 70 and 104 accumulator FMAs instead of 176.
// The forward leaves (A, S) in an optional state buffer for the backward; without it the backward recomputes them.
#include "conv_mfma.h"

namespace mccnn {

// segmented inclusive wave scan of ONE value (once per chunk: the nops cover the VALU-write -> DPP-read hazard)
__device__ __forceinline__ float wave_seg_scan1(float v, float m1, float m2, float m4, float m8, float mA, float mB) {
    asm("s_nop 1\n"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_fmac_f32_dpp %0, %0, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_fmac_f32_dpp %0, %0, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_fmac_f32_dpp %0, %0, %4 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
        "v_fmac_f32_dpp %0, %0, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
        "v_fmac_f32_dpp %0, %0, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n"
        : "+v"(v)
        : "v"(m1), "v"(m2), "v"(m4), "v"(m8), "v"(mA), "v"(mB));
    return v;
}

// ---------------------------------------------------------------------------------------
// Forward, edge pass: A_i = sum_e s_e a2_e (nb*8 floats per centre), S_i = sum_e s_e. Same streaming structure as
// conv_stream (conv.hip): edge-balanced centre-aligned slices, chunks of 64 consecutive edges, wave-wide segmented
// scan with carry, the last lane of a centre stores its finished sums.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f1_fwd_edges(ConvArgs a, float* __restrict__ A, float* __restrict__ S,
                                                    int numWaves, float4* __restrict__ recOut) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i4 = lane & 3;
    const int rowA = a.nb * 8;
    float* wl = lds;
    float* carry = lds + a.nb * MCCNN_WQ_FWD + (size_t)wave * rowA;
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

    auto zero_rows = [&](int c0, int c1) {  // centres without neighbours
        for (int c = c0; c < c1; ++c) {
            for (int f = 0; f < rowA; ++f) A[(size_t)c * rowA + f] = 0.0f;
            S[c] = 0.0f;
        }
    };

    int keyLast = cA;
    int carryKey = -1;
    float carryS = 0.f;
    int2 prN = make_int2(0, cA);
    float pdfN = 1.0f;
    if (eBeg + lane < eEnd) { prN = a.packed[eBeg + lane]; pdfN = a.pdfs[eBeg + lane]; }
    for (int base = eBeg; base < eEnd; base += 64) {
        const int t = base + lane;
        const int nIn = min(64, eEnd - base);
        const bool in = lane < nIn;
        const int2 pr = prN;
        const float pdf = pdfN;
        if (t + 64 < eEnd) { prN = a.packed[t + 64]; pdfN = a.pdfs[t + 64]; }
        const int ci = pr.y, j = pr.x;
        float invR = a.invRadius;
        if (a.scaleInv) invR = 1.0f / (a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[j], a.B)));
        const float* pp = a.pts + (size_t)j * 3;
        const float* cc = a.samples + (size_t)ci * 3;
        const float R = a.scaleInv ? a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[j], a.B)) : a.radius;
        const float d0 = div_exact(pp[0] - cc[0], R, invR), d1 = div_exact(pp[1] - cc[1], R, invR), d2 = div_exact(pp[2] - cc[2], R, invR);
        float K = 1.0f;
        if (a.avg) K = (float)(((ci + 1 < a.m) ? a.start[ci + 1] : a.e) - a.start[ci]);
        const float inv = in ? __builtin_amdgcn_rcpf(pdf * K) : 0.0f;
        const float s = in ? a.feats[j] * inv : 0.0f;
        if (recOut && in) recOut[t] = make_float4(d0, d1, d2, inv);  // the record f1_edge_records would compute (same bits)
        const int key = in ? ci + 1 : 0;
        const int cLast = __builtin_amdgcn_readlane(ci, nIn - 1);
        const int rowEnd = (cLast + 1 < a.m) ? a.start[cLast + 1] : a.e;
        const bool cont = rowEnd > base + 64;
        int keyPrev = __shfl_up(key, 1, 64);
        if (lane == 0) keyPrev = keyLast;
        int keyNext = __shfl_down(key, 1, 64);
        const bool tail = in && ((lane == nIn - 1) ? !cont : (key != keyNext));
        if (in && key - keyPrev > 1) zero_rows(keyPrev, ci);
        const float m1 = (key != 0 && dpp_i<DPP_ROW_SHR(1)>(key) == key) ? 1.f : 0.f;
        const float m2 = (key != 0 && dpp_i<DPP_ROW_SHR(2)>(key) == key) ? 1.f : 0.f;
        const float m4 = (key != 0 && dpp_i<DPP_ROW_SHR(4)>(key) == key) ? 1.f : 0.f;
        const float m8 = (key != 0 && dpp_i<DPP_ROW_SHR(8)>(key) == key) ? 1.f : 0.f;
        const float mA = (key != 0 && dpp_rows_i<DPP_ROW_BCAST15, 0xA>(key) == key) ? 1.f : 0.f;
        const float mB = (key != 0 && dpp_rows_i<DPP_ROW_BCAST31, 0xC>(key) == key) ? 1.f : 0.f;
        const bool haveCarry = carryKey >= 0;
        const float mC = (haveCarry && key == carryKey) ? 1.f : 0.f;

        float sv = wave_seg_scan1(s, m1, m2, m4, m8, mA, mB);
        sv = fmaf(mC, carryS, sv);
        if (tail) S[ci] = sv;
        carryS = __shfl(sv, 63, 64);
        float* arow = A + (size_t)ci * rowA;

        for (int q = 0; q < a.nb; ++q) {
            float pre1[8], a1[8], pre2[8], a2[8], c[8];
            MCCNN_PHASE();
            mlp_block_l12(wl + q * MCCNN_WQ_FWD, i4, d0, d1, d2, pre1, a1, pre2, a2);
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = s * a2[k];
            float* cq = carry + q * 8;
            f32x4 cv0 = {0.f, 0.f, 0.f, 0.f}, cv1 = cv0;
            if (haveCarry) { cv0 = *reinterpret_cast<f32x4*>(cq); cv1 = *reinterpret_cast<f32x4*>(cq + 4); }
            wave_seg_scan8(c, m1, m2, m4, m8, mA, mB);
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = fmaf(mC, k < 4 ? cv0[k & 3] : cv1[k & 3], c[k]);
            if (cont && lane == 63) {
                *reinterpret_cast<f32x4*>(cq) = (f32x4){c[0], c[1], c[2], c[3]};
                *reinterpret_cast<f32x4*>(cq + 4) = (f32x4){c[4], c[5], c[6], c[7]};
            }
            if (tail) {
                float4* dst = reinterpret_cast<float4*>(arow + q * 8);
                dst[0] = make_float4(c[0], c[1], c[2], c[3]);
                dst[1] = make_float4(c[4], c[5], c[6], c[7]);
            }
        }
        carryKey = cont ? cLast + 1 : -1;
        keyLast = cLast + 1;
    }
    if (lane == 0) zero_rows(keyLast, cB);
}

// ---------------------------------------------------------------------------------------
// Forward, edge pass with FOUR consecutive edges per lane (256 edges per iteration of a wave). The segmented sums of
// f1_fwd_edges cost 6 fused-DPP steps per value and 64 edges; here a lane first sums its own four edges in registers
// (3 fma per value), ONE 6-step scan runs over the lane totals, and the incoming prefix is added back to the lane's
// edges (3 fma per value): 12 + 6 + 8 instructions per value and 256 edges instead of 24 + ..., the per-edge work
// (MLP layers 1 and 2 on the matrix cores, ReLU, the product with s_e) is unchanged. The weights of a block are read
// from LDS once per iteration (registers) instead of once per 64 edges. Same slices, carries and tails as above.
// ---------------------------------------------------------------------------------------
#define DPP_WAVE_SHR1 0x138
#define DPP_WAVE_SHL1 0x130
struct L12Regs {
    f32x4 w1lo, w1hi;  // (w0, w1, w2, b1) of neurons i4 / 4 + i4
    float w2lo[8], w2hi[8], b2lo, b2hi;
};
__device__ __forceinline__ void load_l12(const float* __restrict__ wq, int i4, L12Regs& w) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(wq);
    w.w1lo = w4[i4];
    w.w1hi = w4[4 + i4];
    const f32x4 a0 = w4[10 + 2 * i4], a1 = w4[10 + 2 * i4 + 1], b0 = w4[10 + 2 * (4 + i4)], b1 = w4[10 + 2 * (4 + i4) + 1];
    w.w2lo[0] = a0.x; w.w2lo[1] = a0.y; w.w2lo[2] = a0.z; w.w2lo[3] = a0.w; w.w2lo[4] = a1.x; w.w2lo[5] = a1.y; w.w2lo[6] = a1.z; w.w2lo[7] = a1.w;
    w.w2hi[0] = b0.x; w.w2hi[1] = b0.y; w.w2hi[2] = b0.z; w.w2hi[3] = b0.w; w.w2hi[4] = b1.x; w.w2hi[5] = b1.y; w.w2hi[6] = b1.z; w.w2hi[7] = b1.w;
    w.b2lo = wq[104 + i4];
    w.b2hi = wq[104 + 4 + i4];
}

#ifndef MCCNN_F1_X4_OCC
#define MCCNN_F1_X4_OCC 1
#endif
#ifndef MCCNN_F1_X4_GROUP
#define MCCNN_F1_X4_GROUP 4
#endif
#ifdef MCCNN_F1_X4_NOPHASE
#define X4_PHASE()
#else
#define X4_PHASE() MCCNN_PHASE()
#endif
__global__ __launch_bounds__(256, MCCNN_F1_X4_OCC) void f1_fwd_edges4(ConvArgs a, float* __restrict__ A, float* __restrict__ S,
                                                     int numWaves, float4* __restrict__ recOut) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i4 = lane & 3;
    const int rowA = a.nb * 8;
    float* wl = lds;
    float* carry = lds + a.nb * MCCNN_WQ_FWD + (size_t)wave * rowA;
    // the lane's four records are 64 contiguous bytes, but a store instruction of 16 B per lane at a 64-byte stride touches 64
    // lines: they go through LDS and leave as four fully coalesced 1 KB stores (the 7 M-edge list of BASELINE cfg2, 2 blocks:
    // forward 0.29 -> 0.23 ms; only when the caller keeps the records, i.e. a backward pass follows)
    float4* recStage = reinterpret_cast<float4*>(lds + a.nb * MCCNN_WQ_FWD + 4 * (size_t)rowA) + (size_t)wave * 256;
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

    auto zero_rows = [&](int c0, int c1) {  // centres without neighbours
        for (int c = c0; c < c1; ++c) {
            for (int f = 0; f < rowA; ++f) A[(size_t)c * rowA + f] = 0.0f;
            S[c] = 0.0f;
        }
    };

    int keyLast = cA;    // key (centre + 1) of the last edge seen so far; cA before the first
    int carryKey = -1;   // >= 0: the last centre of the previous iteration continues (its sums so far are the carry)
    float carryS = 0.f;
    // (Measured and dropped: a second iteration of look-ahead -- the rows the NEXT iteration's list entries point to
    // (point, centre, feature, row bounds) gathered before this iteration's block loop: 200 VGPRs / 2 waves per SIMD, same
    // forward time 0.175 ms, pipelined step 0.607 against 0.602 ms. The loop's skeleton -- gathers, records, tail stores --
    // measures 60-75 us on its own (MCCNN_ABL_X4_NOSCAN + NOMLP), of which the stores are ~12 us (NOTAIL / NOREC) and the
    // prologue 5 us (NOLOOP); kernel MLP 63 us, segmented sums 18 us.)
    int2 prN[4];
    float pdfN[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = min(eBeg + 4 * lane + u, a.e - 1);
        prN[u] = a.packed[t];
        pdfN[u] = a.pdfs[t];
    }
#ifdef MCCNN_ABL_X4_NOLOOP  // timing ablation only: the kernel's prologue alone
    if (eBeg >= 0) return;
#endif
    for (int base = eBeg; base < eEnd; base += 256) {
        const int lastPos = min(base + 256, eEnd) - 1;  // the last edge of this iteration
        float d0[4], d1[4], d2[4], sE[4];
        int key[4], ci[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = base + 4 * lane + u;
            in[u] = t < eEnd;
            const int2 pr = prN[u];
            const float pdf = pdfN[u];
            const int j = pr.x;
            ci[u] = pr.y;
            float invR = a.invRadius, R = a.radius;
            if (a.scaleInv) {
                R = a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[j], a.B));
                invR = 1.0f / R;
            }
            const float* pp = a.pts + (size_t)j * 3;
            const float* cc = a.samples + (size_t)pr.y * 3;
            d0[u] = div_exact(pp[0] - cc[0], R, invR);
            d1[u] = div_exact(pp[1] - cc[1], R, invR);
            d2[u] = div_exact(pp[2] - cc[2], R, invR);
            float K = 1.0f;
            if (a.avg) K = (float)(((pr.y + 1 < a.m) ? a.start[pr.y + 1] : a.e) - a.start[pr.y]);
            const float inv = in[u] ? __builtin_amdgcn_rcpf(pdf * K) : 0.0f;
            sE[u] = in[u] ? a.feats[j] * inv : 0.0f;
#ifndef MCCNN_ABL_X4_NOREC
            if (recOut) recStage[4 * lane + u] = make_float4(d0[u], d1[u], d2[u], inv);  // the record f1_edge_records would compute
#endif
            key[u] = in[u] ? pr.y + 1 : 0;
        }
#ifndef MCCNN_ABL_X4_NOREC
        if (recOut) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = base + 64 * k + lane;
                if (t < eEnd) recOut[t] = recStage[64 * k + lane];
            }
        }
#endif
        if (base + 256 < eEnd) {  // the list entries of the next iteration
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(base + 256 + 4 * lane + u, a.e - 1);
                prN[u] = a.packed[t];
                pdfN[u] = a.pdfs[t];
            }
        }
        // the centre of the last edge: does its row go on behind this iteration?
        const int lp = lastPos - base;
        const int lu = lp & 3;
        const int kSel = (lu == 0) ? key[0] : (lu == 1) ? key[1] : (lu == 2) ? key[2] : key[3];
        const int keyEnd = __builtin_amdgcn_readlane(kSel, lp >> 2);  // centre + 1
        const int rowEnd = (keyEnd < a.m) ? a.start[keyEnd] : a.e;
        const bool cont = rowEnd > lastPos + 1;
        // key of the edge before / behind the lane's four (lane 0: the last edge of the previous iteration)
        const int kPrev = __builtin_amdgcn_update_dpp(keyLast, key[3], DPP_WAVE_SHR1, 0xf, 0xf, false);
        const int kNext = __builtin_amdgcn_update_dpp(0, key[0], DPP_WAVE_SHL1, 0xf, 0xf, true);
        const bool haveCarry = carryKey >= 0;
        float nh[4], op[4];  // edge u continues the segment of edge u - 1 / belongs to the segment that enters the lane
        bool tail[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kp = (u == 0) ? kPrev : key[u - 1];
            const int kn = (u == 3) ? kNext : key[u + 1];
            if (in[u] && key[u] - kp > 1) zero_rows(kp, ci[u]);
            nh[u] = (key[u] != 0 && key[u] == kp) ? 1.f : 0.f;
            op[u] = (key[u] != 0 && key[u] == kPrev && lane != 0) ? 1.f : 0.f;
            const int t = base + 4 * lane + u;
            tail[u] = in[u] && ((t == lastPos) ? !cont : (key[u] != kn));
        }
        // lane-level masks of the scan over the lane totals: lanes l - d + 1 .. l hold no segment head
        const int k3 = key[3];
        const float m1 = (k3 != 0 && dpp_i<DPP_ROW_SHR(1)>(k3) == k3) ? 1.f : 0.f;
        const float m2 = (k3 != 0 && dpp_i<DPP_ROW_SHR(2)>(k3) == k3) ? 1.f : 0.f;
        const float m4 = (k3 != 0 && dpp_i<DPP_ROW_SHR(4)>(k3) == k3) ? 1.f : 0.f;
        const float m8 = (k3 != 0 && dpp_i<DPP_ROW_SHR(8)>(k3) == k3) ? 1.f : 0.f;
        const float mA = (k3 != 0 && dpp_rows_i<DPP_ROW_BCAST15, 0xA>(k3) == k3) ? 1.f : 0.f;
        const float mB = (k3 != 0 && dpp_rows_i<DPP_ROW_BCAST31, 0xC>(k3) == k3) ? 1.f : 0.f;
        const float mC = (haveCarry && lane == 0) ? 1.f : 0.f;  // the carry enters at the first edge of the iteration

        {   // S_i = sum of s_e: the same three stages on one value
            float r0 = fmaf(mC, carryS, sE[0]);
            float r1 = fmaf(r0, nh[1], sE[1]);
            float r2 = fmaf(r1, nh[2], sE[2]);
            float r3 = fmaf(r2, nh[3], sE[3]);
            r3 = wave_seg_scan1(r3, m1, m2, m4, m8, mA, mB);
            const float cin = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r3), DPP_WAVE_SHR1, 0xf, 0xf, true));
            r0 = fmaf(op[0], cin, r0);
            r1 = fmaf(op[1], cin, r1);
            r2 = fmaf(op[2], cin, r2);
            if (tail[0]) S[ci[0]] = r0;
            if (tail[1]) S[ci[1]] = r1;
            if (tail[2]) S[ci[2]] = r2;
            if (tail[3]) S[ci[3]] = r3;
            carryS = __shfl(r3, 63, 64);
        }

        for (int q = 0; q < a.nb; ++q) {
            L12Regs wr;
            load_l12(wl + q * MCCNN_WQ_FWD, i4, wr);
            float c[4][8];
            X4_PHASE();
            const float one = opaque_one();
#ifdef MCCNN_ABL_X4_NOMLP  // timing ablation only (wrong results): no kernel MLP
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 8; ++r) c[u][r] = sE[u] * (r < 4 ? wr.w2lo[r] : wr.w2hi[r]) + (r & 1 ? d0[u] : d1[u]) + one;
#else
#pragma unroll
            for (int g = 0; g < 4; g += MCCNN_F1_X4_GROUP) {  // MCCNN_F1_X4_GROUP edges per MFMA / VALU phase
                float h[MCCNN_F1_X4_GROUP][8];
#pragma unroll
                for (int v = 0; v < MCCNN_F1_X4_GROUP; ++v) {
                    const int u = g + v;
                    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
                    lo = MFMA4(wr.w1lo.x, d0[u], lo);
                    hi = MFMA4(wr.w1hi.x, d0[u], hi);
                    lo = MFMA4(wr.w1lo.y, d1[u], lo);
                    hi = MFMA4(wr.w1hi.y, d1[u], hi);
                    lo = MFMA4(wr.w1lo.z, d2[u], lo);
                    hi = MFMA4(wr.w1hi.z, d2[u], hi);
                    lo = MFMA4(wr.w1lo.w, one, lo);
                    hi = MFMA4(wr.w1hi.w, one, hi);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { h[v][r] = lo[r]; h[v][4 + r] = hi[r]; }
                }
                X4_PHASE();
#pragma unroll
                for (int v = 0; v < MCCNN_F1_X4_GROUP; ++v)
#pragma unroll
                    for (int r = 0; r < 8; ++r) h[v][r] = relu1(h[v][r]);
                X4_PHASE();
#pragma unroll
                for (int v = 0; v < MCCNN_F1_X4_GROUP; ++v) layer8_regs<true>(wr.w2lo, wr.w2hi, wr.b2lo, wr.b2hi, h[v], c[g + v]);
                X4_PHASE();
#pragma unroll
                for (int v = 0; v < MCCNN_F1_X4_GROUP; ++v)
#pragma unroll
                    for (int r = 0; r < 8; ++r) c[g + v][r] = sE[g + v] * relu1(c[g + v][r]);
            }
#endif
            float* cq = carry + q * 8;
            {
                f32x4 cv0 = {0.f, 0.f, 0.f, 0.f}, cv1 = cv0;
                if (haveCarry) { cv0 = *reinterpret_cast<f32x4*>(cq); cv1 = *reinterpret_cast<f32x4*>(cq + 4); }
#pragma unroll
                for (int k = 0; k < 8; ++k) c[0][k] = fmaf(mC, k < 4 ? cv0[k & 3] : cv1[k & 3], c[0][k]);
            }
#ifndef MCCNN_ABL_X4_NOSCAN  // timing ablation only (wrong results): no segmented sums
            // the lane's own four edges, the scan over the lane totals, the prefix that enters the lane
#pragma unroll
            for (int u = 1; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) c[u][k] = fmaf(c[u - 1][k], nh[u], c[u][k]);
            wave_seg_scan8(c[3], m1, m2, m4, m8, mA, mB);
            float cin[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                cin[k] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[3][k]), DPP_WAVE_SHR1, 0xf, 0xf, true));
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) c[u][k] = fmaf(op[u], cin[k], c[u][k]);
#endif
            if (cont && lane == 63) {
                *reinterpret_cast<f32x4*>(cq) = (f32x4){c[3][0], c[3][1], c[3][2], c[3][3]};
                *reinterpret_cast<f32x4*>(cq + 4) = (f32x4){c[3][4], c[3][5], c[3][6], c[3][7]};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#ifdef MCCNN_ABL_X4_NOTAIL  // timing ablation only (wrong results): the rows are stored by the last block only
                if (tail[u] && q == a.nb - 1) {
#else
                if (tail[u]) {
#endif
                    float4* dst = reinterpret_cast<float4*>(A + (size_t)ci[u] * rowA + q * 8);
                    dst[0] = make_float4(c[u][0], c[u][1], c[u][2], c[u][3]);
                    dst[1] = make_float4(c[u][4], c[u][5], c[u][6], c[u][7]);
                }
            }
        }
        carryKey = cont ? keyEnd : -1;
        keyLast = keyEnd;
    }
    if (lane == 0) zero_rows(keyLast, cB);
}

// Forward, centre pass: out_i[8q+n] = W3_q[n] . A_i[q] + b3_q[n] S_i. One thread per (centre, block): consecutive lanes
// read / write consecutive 32-byte pieces of a row; weights from LDS.
__global__ __launch_bounds__(256) void f1_fwd_centres(ConvArgs a, const float* __restrict__ A,
                                                      const float* __restrict__ S, float* __restrict__ out) {
    extern __shared__ float lds[];  // per block: W3[8][8], b3[8]
    for (int t = threadIdx.x; t < a.nb * 72; t += blockDim.x) {
        int q = t / 72, r = t - q * 72;
        lds[t] = (r < 64) ? a.w3[q * 64 + r] : a.b3[q * 8 + r - 64];
    }
    __syncthreads();
    const long long tix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tix >= (long long)a.m * a.nb) return;
    const int i = (int)(tix / a.nb), q = (int)(tix - (long long)i * a.nb);
    const float Si = S[i];
    const float4* ap = reinterpret_cast<const float4*>(A + tix * 8);
    const float4 x0 = ap[0], x1 = ap[1];
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const float* wq = lds + q * 72;
    float o[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float acc = wq[64 + n] * Si;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = fmaf(wq[n * 8 + k], x[k], acc);
        o[n] = acc;
    }
    float* dst = out + (size_t)i * a.outF + q * 8;
    if ((a.outF & 3) == 0 && q * 8 + 8 <= a.outF) {
        reinterpret_cast<float4*>(dst)[0] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4*>(dst)[1] = make_float4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (q * 8 + n < a.outF) dst[n] = o[n];
    }
}

__global__ __launch_bounds__(256) void f1_edge_records(ConvArgs a, float4* __restrict__ rec) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.e) return;
    int2 pr = a.packed[t];
    float invR = a.invRadius;
    if (a.scaleInv) invR = 1.0f / (a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[pr.x], a.B)));
    const float* p = a.pts + (size_t)pr.x * 3;
    const float* c = a.samples + (size_t)pr.y * 3;
    int e0 = a.start[pr.y];
    int e1 = (pr.y < a.m - 1) ? a.start[pr.y + 1] : a.e;
    float K = a.avg ? (float)(e1 - e0) : 1.0f;
    const float R = a.scaleInv ? a.radius * max_extent(a.mn, a.mx, clamp_batch(a.bids[pr.x], a.B)) : a.radius;
    rec[t] = make_float4(div_exact(p[0] - c[0], R, invR), div_exact(p[1] - c[1], R, invR), div_exact(p[2] - c[2], R, invR),
                         __builtin_amdgcn_rcpf(a.pdfs[t] * K));
}

// ---------------------------------------------------------------------------------------
// Backward, edge pass (q-outer, edge-balanced waves, see conv_bwd_mfma): dW2, db2, dW1, db1 partial sums in VGPRs
// (104 accumulators), per-edge feature gradient accumulated across blocks with a plain read-modify-write and added to
// featGrad by the last block. Partial row per (wave, block): w1[24] b1[8] w2[64] b2[8].
// ---------------------------------------------------------------------------------------
#define MCCNN_F1_ROW 104
#ifndef MCCNN_F1_OCC
#define MCCNN_F1_OCC 2
#endif
__global__ __launch_bounds__(256, MCCNN_F1_OCC) void f1_bwd_edges(ConvArgs a, const float4* __restrict__ rec,
                                                       const float* __restrict__ G, const float* __restrict__ gb,
                                                       float* __restrict__ featGrad, float* __restrict__ dfE, int cpw,
                                                       float* __restrict__ partials) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i4 = lane & 3;
    float* wl = lds;
    stage_weights<MCCNN_WQ_BWD>(a, wl);
    __syncthreads();
    const int waveGlobal = blockIdx.x * 4 + wave;
#ifdef MCCNN_ABL_SAMESLICE  // timing ablation only (wrong results): every wave sweeps the first slice -> all loads hit the caches
    const long long eBegL = ((long long)waveGlobal * cpw * 64 < a.e) ? 0 : a.e;
#else
    const long long eBegL = (long long)waveGlobal * cpw * 64;
#endif
    if (eBegL >= a.e) return;
    const int eBeg = (int)eBegL;
    const int eEnd = (int)min((long long)a.e, eBegL + (long long)cpw * 64);
    float* prow = partials + (size_t)waveGlobal * a.nb * MCCNN_F1_ROW;

    for (int q = 0; q < a.nb; ++q) {
        float gw2[64], gb2[8], gw1[24], gb1[8];
#pragma unroll
        for (int k = 0; k < 64; ++k) gw2[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { gb2[k] = 0.f; gb1[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 24; ++k) gw1[k] = 0.f;
        const bool last = (q == a.nb - 1);

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
            const int j = pr.x, ci = pr.y;
            const float inv = act ? rc.w : 0.f;
            // G is block-major, [q][centre][8]: inside a sweep the 32-byte pieces of consecutive centres are adjacent
            const float4* gp = reinterpret_cast<const float4*>(G + ((size_t)q * a.m + ci) * 8);
#ifdef MCCNN_ABL_NOGATHER  // timing ablation only (wrong results): the kernel with every gather already in registers
            const float4 g0 = make_float4(rc.x, rc.y, rc.z, rc.w), g1 = make_float4(rc.y, rc.z, rc.x, rc.w);
            const float f = rc.z;
            float dfOld = 0.f, gbi = 0.f;
            (void)gp;
#else
            const float4 g0 = gp[0], g1 = gp[1];
            const float f = a.feats[j];
            float dfOld = 0.f, gbi = 0.f;
            if (act && q > 0) dfOld = dfE[t];
            if (last) gbi = gb[ci];
#endif
            {
                int tn = min(t + 64, a.e - 1);  // clamped: branch-free prefetch of the next chunk
                prN = a.packed[tn];
                rcN = rec[tn];
            }
            const float Gq[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float s = f * inv;
            float a1[8], a2[8];
            bool p1[8], p2[8];  // pre-activation >= 0 (ReLU' of the reference: spatial_conv.cu:404,429)
            int woff = q * MCCNN_WQ_BWD;
            asm volatile("" : "+s"(woff));  // keep the LDS weight reads inside the chunk loop (see conv_bwd_mfma)
            const float* wq = wl + woff;
            const f32x4* w4 = reinterpret_cast<const f32x4*>(wq);
            {
                float pre1[8], pre2[8];
                mlp_block_l12(wq, i4, rc.x, rc.y, rc.z, pre1, a1, pre2, a2);
#pragma unroll
                for (int k = 0; k < 8; ++k) { p1[k] = pre1[k] >= 0.0f; p2[k] = pre2[k] >= 0.0f; }
            }
            // feature gradient: (sum_q G_i[q] . a2_e[q] + g_i . b3) / (pdf K)
            float sfg = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sfg = fmaf(Gq[k], a2[k], sfg);
#ifndef MCCNN_ABL_NODF  // timing ablation only: no per-edge feature-gradient traffic
            if (act) {
                if (last) atomicAdd(&featGrad[j], (dfOld + sfg + gbi) * inv);
                else dfE[t] = dfOld + sfg;
            }
#else
            asm volatile("" ::"v"(sfg), "v"(dfOld), "v"(gbi));
#endif
            // t3 = 1[pre2 >= 0] * G_i s_e ; dW2 += t3 a1^T, db2 += t3
            float t3[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t3[k] = p2[k] ? Gq[k] * s : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int l = 0; l < 8; ++l) gw2[k * 8 + l] = fmaf(t3[k], a1[l], gw2[k * 8 + l]);
                gb2[k] += t3[k];
            }
            // t4 = 1[pre1 >= 0] * W2^T t3 ; dW1 += t4 delta^T, db1 += t4
            float t4[8];
            MCCNN_PHASE();
            layer8<false>(w4 + 46, nullptr, i4, t3, t4);  // W2^T rows at float 184 -> f32x4 index 46
            MCCNN_PHASE();
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float v = p1[l] ? t4[l] : 0.f;
                gw1[l * 3] = fmaf(v, rc.x, gw1[l * 3]);
                gw1[l * 3 + 1] = fmaf(v, rc.y, gw1[l * 3 + 1]);
                gw1[l * 3 + 2] = fmaf(v, rc.z, gw1[l * 3 + 2]);
                gb1[l] += v;
            }
        }
        {
            float r2 = wave_reduce64(gw2, lane);
            float misc[64];
#pragma unroll
            for (int k = 0; k < 24; ++k) misc[k] = gw1[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) { misc[24 + k] = gb1[k]; misc[32 + k] = gb2[k]; }
#pragma unroll
            for (int k = 40; k < 64; ++k) misc[k] = 0.f;
            float rm = wave_reduce64(misc, lane);
            float* pq = prow + q * MCCNN_F1_ROW;
            pq[32 + lane] = r2;
            if (lane < 32) pq[lane] = rm;                 // w1, b1
            else if (lane < 40) pq[96 + lane - 32] = rm;  // b2
        }
    }
}

// Backward, centre pass (runs before the edge pass): G_i[q] = W3_q^T g_i[q] and gb_i = g_i . b3 for the edges, and the
// layer-3 gradients dW3 = sum_i g_i (x) A_i, db3 = sum_i g_i S_i. Waves own slices of centres, q-outer with the 72 sums
// of a block in registers; partial row per (wave, block): w3[64] b3[8].
#define MCCNN_F1_ROWC 72
// One wave per (slice of centres, block): nb times more waves than a q-outer sweep per slice. The pass is a chain of
// dependent latencies per wave (load -> 137 FMAs -> stores, then two 6-step butterflies), not arithmetic, so its time
// is the length of that chain: with the blocks spread over waves it is 1 / nb of it (41 -> 12 us on the 100k room).
// The bias term of the feature gradient, gb_i = g_i . b3 over the whole row, has waves of its own (pseudo-block
// q == nb: one ascending fma chain per centre, no sums to reduce), so the blocks stay independent.
__global__ __launch_bounds__(256) void f1_bwd_centres(ConvArgs a, const float* __restrict__ outGrad,
                                                      const float* __restrict__ A, const float* __restrict__ S,
                                                      int cPerWave, int numSlices, float* __restrict__ G,
                                                      float* __restrict__ gb, float* __restrict__ partials,
                                                      float* __restrict__ featGrad) {
    // the edge pass ADDS to the feature gradient: cleared here (this pass runs first), no memset launch of its own
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x)
        featGrad[i] = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + wave;  // block-major inside a slice: neighbouring waves read the same rows
    // wave-uniform by construction; readfirstlane tells the compiler, so the block's weights below are scalar loads
    const int w = __builtin_amdgcn_readfirstlane(wid / (a.nb + 1));
    const int q = __builtin_amdgcn_readfirstlane(wid - w * (a.nb + 1));
    if (w >= numSlices) return;
    const int c0 = w * cPerWave;
    const int c1 = min(a.m, c0 + cPerWave);
    if (q == a.nb) {
#ifdef MCCNN_ABL_NODOT  // timing ablation only (wrong results)
        return;
#endif
        for (int i = c0 + lane; i < c1; i += 64) {
            const float* row = outGrad + (size_t)i * a.outF;
            float gbv = 0.f;
            int f = 0;
            if ((a.outF & 3) == 0) {
                for (; f + 4 <= a.outF; f += 4) {
                    const float4 gg = *reinterpret_cast<const float4*>(row + f);
                    gbv = fmaf(gg.x, a.b3[f], gbv);
                    gbv = fmaf(gg.y, a.b3[f + 1], gbv);
                    gbv = fmaf(gg.z, a.b3[f + 2], gbv);
                    gbv = fmaf(gg.w, a.b3[f + 3], gbv);
                }
            }
            for (; f < a.outF; ++f) gbv = fmaf(row[f], a.b3[f], gbv);
            gb[i] = gbv;
        }
        return;
    }
    const int rowA = a.nb * 8;
    const bool vec = (a.outF & 3) == 0 && q * 8 + 8 <= a.outF;
    float wq[64];  // W3[8][8] of this block: wave-uniform -> scalar registers
#pragma unroll
    for (int r = 0; r < 64; ++r) wq[r] = a.w3[q * 64 + r];
    float acc[64], accb[8];
#pragma unroll
    for (int k = 0; k < 64; ++k) acc[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) accb[k] = 0.f;
    for (int i = c0 + lane; i < c1; i += 64) {
        const float* grow = outGrad + (size_t)i * a.outF + q * 8;
        float g[8];
        if (vec) {
            const float4 g0 = reinterpret_cast<const float4*>(grow)[0], g1 = reinterpret_cast<const float4*>(grow)[1];
            g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        } else {
#pragma unroll
            for (int n = 0; n < 8; ++n) g[n] = (q * 8 + n < a.outF) ? grow[n] : 0.f;
        }
        const float4* ap = reinterpret_cast<const float4*>(A + (size_t)i * rowA + q * 8);
        const float4 x0 = ap[0], x1 = ap[1];
        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float Si = S[i];
        float Gk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) Gk[k] = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[n * 8 + k] = fmaf(g[n], x[k], acc[n * 8 + k]);
                Gk[k] = fmaf(wq[n * 8 + k], g[n], Gk[k]);
            }
            accb[n] = fmaf(g[n], Si, accb[n]);
        }
        float4* dst = reinterpret_cast<float4*>(G + ((size_t)q * a.m + i) * 8);  // block-major: 2 KB contiguous per wave
        dst[0] = make_float4(Gk[0], Gk[1], Gk[2], Gk[3]);
        dst[1] = make_float4(Gk[4], Gk[5], Gk[6], Gk[7]);
    }
#ifdef MCCNN_ABL_NOBFLY  // timing ablation only (wrong results)
    float r3 = acc[lane & 7] + acc[8 + (lane & 7)], rb = accb[lane & 7];
    for (int k = 16; k < 64; ++k) r3 += acc[k];
#else
    float r3 = wave_reduce64(acc, lane);
    float misc[64];
#pragma unroll
    for (int k = 0; k < 8; ++k) misc[k] = accb[k];
#pragma unroll
    for (int k = 8; k < 64; ++k) misc[k] = 0.f;
    float rb = wave_reduce64(misc, lane);
#endif
    float* pq = partials + ((size_t)w * a.nb + q) * MCCNN_F1_ROWC;
    pq[lane] = r3;
    if (lane < 8) pq[64 + lane] = rb;
}

// Sums the partial rows of both passes in a fixed order (deterministic parameter gradients, no float atomics).
__global__ __launch_bounds__(1024) void f1_reduce(const float* __restrict__ pe, int rowsE, const float* __restrict__ pc,
                                                 int rowsC, int nb, float* __restrict__ dw1, float* __restrict__ db1,
                                                 float* __restrict__ dw2, float* __restrict__ db2,
                                                 float* __restrict__ dw3, float* __restrict__ db3) {
    __shared__ float acc[64][17];  // 16 parameters x 64 row slices per workgroup (see reduce_partials, conv.hip)
    const int K = nb * 176;
    const int kk = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + kk;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = 0, r = 0;
    if (k < K) {
        q = k / 176;
        r = k - q * 176;
        const bool fromE = r < MCCNN_F1_ROW;
        const float* src = fromE ? pe + q * MCCNN_F1_ROW + r : pc + q * MCCNN_F1_ROWC + (r - MCCNN_F1_ROW);
        const size_t stride = (size_t)nb * (fromE ? MCCNN_F1_ROW : MCCNN_F1_ROWC);
        const int rows = fromE ? rowsE : rowsC;
        int w = sl;
        for (; w + 192 < rows; w += 256) {
            s0 += src[(size_t)w * stride];
            s1 += src[(size_t)(w + 64) * stride];
            s2 += src[(size_t)(w + 128) * stride];
            s3 += src[(size_t)(w + 192) * stride];
        }
        for (; w < rows; w += 64) s0 += src[(size_t)w * stride];
    }
    acc[sl][kk] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && k < K) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) v += acc[i][kk];
        if (r < 24) dw1[q * 24 + r] = v;
        else if (r < 32) db1[q * 8 + r - 24] = v;
        else if (r < 96) dw2[q * 64 + r - 32] = v;
        else if (r < 104) db2[q * 8 + r - 96] = v;
        else if (r < 168) dw3[q * 64 + r - 104] = v;
        else db3[q * 8 + r - 168] = v;
    }
}

// ------------------------------------------------------------------------------------ host side
// The q-outer sweep re-reads a wave's slice nb times (32 B per edge and sweep). With one slice per resident wave the
// slices of a launch together must stay inside the 256 MB Infinity Cache or every sweep comes from HBM (8 rooms, 36 M
// edges: 3.5 ms instead of 8 x 0.4): larger inputs get R equal rounds of resident-many waves with <= 32-chunk slices,
// dispatched in slice order, so the waves in flight always cover one window of ~130 MB.
static void f1_bwd_partition(int e, int& cpw, int& waves) {
    const long long chunks = ((long long)e + 63) / 64;
    const long long resident = (long long)num_cus() * 4 * MCCNN_F1_OCC;
    cpw = (int)((chunks + resident - 1) / resident);
    if (cpw > 48) {
        const long long rounds = (cpw + 31) / 32;
        cpw = (int)((chunks + resident * rounds - 1) / (resident * rounds));
    }
    // (at least 2 chunks per wave -- 8 until round 6, which ran a short list on a fraction of the chip: conv.hip MCCNN_BWD_MIN_CHUNKS;
    // the same A/B key)
    static const int minChunks = debug_int("bwd_min_chunks", 2);
    if (cpw < minChunks) cpw = minChunks < 1 ? 1 : minChunks;
    waves = (int)((chunks + cpw - 1) / cpw);
    if (waves < 1) waves = 1;
}
#ifndef MCCNN_F1_CENTRE_WAVES
#define MCCNN_F1_CENTRE_WAVES 1024
#endif
static void f1_centre_partition(int m, int& cPerWave, int& waves) {
    cPerWave = (m + MCCNN_F1_CENTRE_WAVES - 1) / MCCNN_F1_CENTRE_WAVES;
    cPerWave = (cPerWave + 63) / 64 * 64;
    waves = (m + cPerWave - 1) / cPerWave;
}

size_t f1_state_bytes(int m, int nb) { return align_up((size_t)m * nb * 8 * sizeof(float)) + align_up((size_t)m * sizeof(float)); }

static void f1_state_split(void* state, int m, int nb, float*& A, float*& S) {
    A = (float*)state;
    S = (float*)((char*)state + align_up((size_t)m * nb * 8 * sizeof(float)));
}

static int f1_run_edges(const ConvArgs& a, float* A, float* S, float4* recOut, hipStream_t s) {
    const size_t lds = ((size_t)a.nb * MCCNN_WQ_FWD + 4 * (size_t)a.nb * 8) * sizeof(float);
    // long lists: four edges per lane (one scan per 256 edges); short ones keep 64-edge chunks (more waves to spread)
    if (a.e >= g_f1_x4_min_edges.load(std::memory_order_relaxed)) {
        const size_t lds4 = lds + 4 * 256 * sizeof(float4);  // + the record stage of the four waves
        const int perCU4 = cached_blocks_per_cu(reinterpret_cast<const void*>(f1_fwd_edges4), lds4);
        const long long iters = ((long long)a.e + 255) / 256;
        static const int wpc = debug_int("f1_x4_waves_per_cu", 0);
        long long W4 = (long long)num_cus() * (wpc > 0 ? wpc : perCU4 * 4);
        if (W4 > (iters + 1) / 2) W4 = (iters + 1) / 2;
        if (W4 < 1) W4 = 1;
        f1_fwd_edges4<<<(int)((W4 + 3) / 4), 256, lds4, s>>>(a, A, S, (int)W4, recOut);
        MCCNN_LAUNCHED();
        return 0;
    }
    const int perCU = cached_blocks_per_cu(reinterpret_cast<const void*>(f1_fwd_edges), lds);
    const long long chunks = ((long long)a.e + 63) / 64;
    long long W = (long long)num_cus() * perCU * 4;
    if (W > (chunks + 1) / 2) W = (chunks + 1) / 2;
    if (W < 1) W = 1;
    f1_fwd_edges<<<(int)((W + 3) / 4), 256, lds, s>>>(a, A, S, (int)W, recOut);
    MCCNN_LAUNCHED();
    return 0;
}

size_t f1_fwd_workspace_bytes(int m, int nb) { return f1_state_bytes(m, nb) + 256; }

int f1_forward(const ConvArgs& a, float* out, float4* rec_out, void* state, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!state) {
        if (!ws || ws_bytes < f1_fwd_workspace_bytes(a.m, a.nb)) return MCCNN_E_WORKSPACE;
        state = ws;
    }
    float *A, *S;
    f1_state_split(state, a.m, a.nb, A, S);
    int rc = f1_run_edges(a, A, S, rec_out, s);
    if (rc) return rc;
    f1_fwd_centres<<<ceil_div((long long)a.m * a.nb, 256), 256, (size_t)a.nb * 72 * sizeof(float), s>>>(a, A, S, out);
    MCCNN_LAUNCHED();
    return 0;
}

size_t f1_bwd_workspace_bytes(int m, int e, int nb) {
    int cpw, wavesE, cpwC, wavesC;
    f1_bwd_partition(e, cpw, wavesE);
    f1_centre_partition(m, cpwC, wavesC);
    size_t b = f1_state_bytes(m, nb);                                               // (A, S) when the caller kept no state
    b += align_up((size_t)m * nb * 8 * sizeof(float)) + align_up((size_t)m * sizeof(float));  // G, gb
    b += align_up((size_t)e * sizeof(float4)) + align_up((size_t)e * sizeof(float));          // records, per-edge dFeat
    b += align_up((size_t)(wavesE + 3) / 4 * 4 * nb * MCCNN_F1_ROW * sizeof(float));
    b += align_up((size_t)(wavesC + 3) / 4 * 4 * nb * MCCNN_F1_ROWC * sizeof(float));
    return b + 256;
}

// feat_grad and the six parameter gradients are fully written (feat_grad is cleared by the centre pass).
int f1_backward(const ConvArgs& a, const float* out_grad, const float4* rec_in, const void* state, float* feat_grad, float* dw1, float* db1,
                float* dw2, float* db2, float* dw3, float* db3, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < f1_bwd_workspace_bytes(a.m, a.e, a.nb)) return MCCNN_E_WORKSPACE;
    int cpw, wavesE, cPerWave, wavesC;
    f1_bwd_partition(a.e, cpw, wavesE);
    f1_centre_partition(a.m, cPerWave, wavesC);
    const int blocksE = (wavesE + 3) / 4, blocksC = (wavesC + 3) / 4;
    Arena ar(ws, ws_bytes);
    void* own = ar.take<char>(f1_state_bytes(a.m, a.nb));
    float* G = ar.take<float>((size_t)a.m * a.nb * 8);
    float* gb = ar.take<float>((size_t)a.m);
    float4* rec = ar.take<float4>((size_t)a.e);
    float* dfE = ar.take<float>((size_t)a.e);
    float* pe = ar.take<float>((size_t)blocksE * 4 * a.nb * MCCNN_F1_ROW);
    float* pc = ar.take<float>((size_t)blocksC * 4 * a.nb * MCCNN_F1_ROWC);
    if (!own || !G || !gb || !rec || !dfE || !pe || !pc) return MCCNN_E_WORKSPACE;
    float *A, *S;
    if (state) {
        f1_state_split(const_cast<void*>(state), a.m, a.nb, A, S);
    } else {
        f1_state_split(own, a.m, a.nb, A, S);
        int rc = f1_run_edges(a, A, S, nullptr, s);
        if (rc) return rc;
    }
    f1_bwd_centres<<<ceil_div((long long)wavesC * (a.nb + 1), 4), 256, 0, s>>>(a, out_grad, A, S, cPerWave, wavesC, G, gb, pc, feat_grad);
    MCCNN_LAUNCHED();
    const float4* recUse = rec;
    if (rec_in) {
        recUse = rec_in;  // written by the forward call of the same inputs
    } else {
        f1_edge_records<<<ceil_div(a.e, 256), 256, 0, s>>>(a, rec);
        MCCNN_LAUNCHED();
    }
    const size_t ldsE = (size_t)a.nb * MCCNN_WQ_BWD * sizeof(float);
    f1_bwd_edges<<<blocksE, 256, ldsE, s>>>(a, recUse, G, gb, feat_grad, dfE, cpw, pe);
    MCCNN_LAUNCHED();
    f1_reduce<<<ceil_div((long long)a.nb * 176, 16), 1024, 0, s>>>(pe, wavesE, pc, wavesC, a.nb, dw1, db1, dw2, db2, dw3, db3);
    MCCNN_LAUNCHED();
    return 0;
}

}  // namespace mccnn
