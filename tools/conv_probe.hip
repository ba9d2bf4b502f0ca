// Steady-state cost of ONE block of the factored forward pass (conv_f1.hip f1_fwd_edges: layer 1 = 8 MFMA, layer 2 = 18
// MFMA with the bias k-step, 16 ReLUs, s * a2, fused-DPP segmented scan of 8 values) for 64 edges, isolated from memory,
// in several instruction orders. Answers: what do the MFMA -> VALU hand-offs cost, and does interleaving two chunks of
// the same wave hide them?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 tools/conv_probe.hip -o tools/conv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#define N_IT 2000
#define PH() __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80)

__device__ __forceinline__ float relu1(float x, float inf) { return __builtin_amdgcn_fmed3f(x, 0.0f, inf); }

#define SCAN_STEP(CTRL, M)                                                               \
    "v_fmac_f32_dpp %0, %0, %" #M " " CTRL "\n v_fmac_f32_dpp %1, %1, %" #M " " CTRL "\n" \
    "v_fmac_f32_dpp %2, %2, %" #M " " CTRL "\n v_fmac_f32_dpp %3, %3, %" #M " " CTRL "\n" \
    "v_fmac_f32_dpp %4, %4, %" #M " " CTRL "\n v_fmac_f32_dpp %5, %5, %" #M " " CTRL "\n" \
    "v_fmac_f32_dpp %6, %6, %" #M " " CTRL "\n v_fmac_f32_dpp %7, %7, %" #M " " CTRL "\n"
__device__ __forceinline__ void seg_scan8(float* c, float m1, float m2, float m4, float m8, float mA, float mB) {
    asm("s_nop 1\n" SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", 8)
        SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", 9)
        SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", 10)
        SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", 11)
        SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", 12) SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf", 13) "s_nop 1\n"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
        : "v"(m1), "v"(m2), "v"(m4), "v"(m8), "v"(mA), "v"(mB));
}

struct Blk {  // activations of one 64-edge chunk in flight
    f32x4 lo, hi;
    float a1[8], a2[8];
};
__device__ __forceinline__ void l1(Blk& b, const f32x4* w, int i4, float d0, float d1, float d2, float one) {
    f32x4 a1lo = w[i4], a1hi = w[4 + i4];
    f32x4 lo = {0, 0, 0, 0}, hi = lo;
    lo = MFMA4(a1lo.x, d0, lo); hi = MFMA4(a1hi.x, d0, hi);
    lo = MFMA4(a1lo.y, d1, lo); hi = MFMA4(a1hi.y, d1, hi);
    lo = MFMA4(a1lo.z, d2, lo); hi = MFMA4(a1hi.z, d2, hi);
    lo = MFMA4(a1lo.w, one, lo); hi = MFMA4(a1hi.w, one, hi);
    b.lo = lo; b.hi = hi;
}
__device__ __forceinline__ void l2(Blk& b, const f32x4* wrows, const float* bias, int i4, float one) {
    f32x4 al0 = wrows[2 * i4], al1 = wrows[2 * i4 + 1];
    f32x4 ah0 = wrows[2 * (4 + i4)], ah1 = wrows[2 * (4 + i4) + 1];
    float al[8] = {al0.x, al0.y, al0.z, al0.w, al1.x, al1.y, al1.z, al1.w};
    float ah[8] = {ah0.x, ah0.y, ah0.z, ah0.w, ah1.x, ah1.y, ah1.z, ah1.w};
    float bl = bias[i4], bh = bias[4 + i4];
    f32x4 lo = {0, 0, 0, 0}, hi = lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo = MFMA4(al[k], b.a1[k], lo); hi = MFMA4(ah[k], b.a1[k], hi); }
    lo = MFMA4(bl, one, lo); hi = MFMA4(bh, one, hi);
    b.lo = lo; b.hi = hi;
}
template <int RELU>  // 0: none (copy), 1: med3, 2: fmaxf
__device__ __forceinline__ void act(const Blk& b, float* dst, float inf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dst[r] = RELU == 1 ? relu1(b.lo[r], inf) : RELU == 2 ? fmaxf(b.lo[r], 0.f) : b.lo[r];
        dst[4 + r] = RELU == 1 ? relu1(b.hi[r], inf) : RELU == 2 ? fmaxf(b.hi[r], 0.f) : b.hi[r];
    }
}

// MODE: 0 = one chunk per iteration, phases (the kernel's order); 1 = two chunks interleaved phase by phase;
//       2 = one chunk, no ReLU (ablation: MFMA results read by the next MFMA only through a copy);
//       3 = one chunk, fmaxf ReLU; 4 = one chunk, no scan (ablation); 5 = one chunk, no phases (compiler order)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, const float* win, float seed, float inf, int nb) {
    extern __shared__ float lds[];
    for (int t = threadIdx.x; t < nb * 184; t += 256) lds[t] = win[t % 184] + seed;
    __syncthreads();
    const int lane = threadIdx.x & 63, i4 = lane & 3;
    float d0 = lane * 0.01f + seed, d1 = 0.3f - lane * 0.003f, d2 = seed;
    float one = 1.0f;
    asm("" : "+v"(one));
    const float m1 = (lane & 1) ? 1.f : 0.f, m2 = (lane & 2) ? 1.f : 0.f, m4 = (lane & 4) ? 1.f : 0.f, m8 = (lane & 8) ? 1.f : 0.f;
    const float mA = (lane & 16) ? 1.f : 0.f, mB = (lane & 32) ? 1.f : 0.f;
    float accum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int RELU = MODE == 2 ? 0 : MODE == 3 ? 2 : 1;
    for (int it = 0; it < N_IT; ++it) {
        int q = it % nb;
        int woff = q * 184;
        asm volatile("" : "+s"(woff));
        const float* wq = lds + woff;
        const f32x4* w = reinterpret_cast<const f32x4*>(wq);
        if (MODE == 1) {
            Blk A, B;
            float cA[8], cB[8];
            PH();
            l1(A, w, i4, d0, d1, d2, one);
            l1(B, w, i4, d1, d2, d0, one);
            PH();
            act<1>(A, A.a1, inf);
            act<1>(B, B.a1, inf);
            PH();
            l2(A, w + 10, wq + 104, i4, one);
            l2(B, w + 10, wq + 104, i4, one);
            PH();
            act<1>(A, A.a2, inf);
            act<1>(B, B.a2, inf);
#pragma unroll
            for (int n = 0; n < 8; ++n) { cA[n] = A.a2[n] * d1; cB[n] = B.a2[n] * d2; }
            seg_scan8(cA, m1, m2, m4, m8, mA, mB);
            seg_scan8(cB, m1, m2, m4, m8, mA, mB);
#pragma unroll
            for (int n = 0; n < 8; ++n) accum[n] += cA[n] + cB[n];
        } else {
            Blk A;
            float c[8];
            if (MODE != 5) PH();
            l1(A, w, i4, d0, d1, d2, one);
            if (MODE != 5) PH();
            act<RELU>(A, A.a1, inf);
            if (MODE != 5) PH();
            l2(A, w + 10, wq + 104, i4, one);
            if (MODE != 5) PH();
            act<RELU>(A, A.a2, inf);
#pragma unroll
            for (int n = 0; n < 8; ++n) c[n] = A.a2[n] * d1;
            if (MODE != 4) seg_scan8(c, m1, m2, m4, m8, mA, mB);
#pragma unroll
            for (int n = 0; n < 8; ++n) accum[n] += c[n];
        }
        d0 += 1e-6f;
    }
    float s = 0;
    for (int n = 0; n < 8; ++n) s += accum[n];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wavesPerSimd, const float* win, float* out) {
    int blocks = 256 * wavesPerSimd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, 256, 8 * 184 * 4>>>(out, win, 0.5f, __builtin_huge_valf(), 8);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256, 8 * 184 * 4>>>(out, win, 0.25f, __builtin_huge_valf(), 8);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double chunks = (MODE == 1 ? 2.0 : 1.0) * N_IT;
    printf("%-52s waves/SIMD %d: %8.1f cycles per 64-edge block (@2.4 GHz)\n", name, wavesPerSimd,
           ms * 1e-3 * 2.4e9 / ((double)wavesPerSimd * chunks));
}

int main() {
    float *win, *out;
    hipMalloc(&win, 184 * 4);
    hipMemset(win, 0, 184 * 4);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 4, 7}) {
        run<0>("26 mfma + 16 med3 + 8 mul + scan48 (kernel order)", w, win, out);
        run<1>("same, two chunks interleaved phase by phase", w, win, out);
        run<2>("no relu", w, win, out);
        run<3>("fmaxf relu", w, win, out);
        run<4>("no scan", w, win, out);
        run<5>("compiler order (no phases)", w, win, out);
    }
    return 0;
}
