// Steady-state cost of ONE forward MLP block (38 MFMA + ReLUs + scale + segmented DPP scan) for 64 edges, isolated
// from memory: variants drop the LDS weight reads / the scan / the ReLUs to see what the instruction mix itself costs.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/block_probe.hip -o tools/block_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#define N_IT 4000
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#ifdef PHASES
#define PH() __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80)
#else
#define PH()
#endif
__device__ __forceinline__ float relu1(float x, float inf) {
#ifdef MED3
    return __builtin_amdgcn_fmed3f(x, 0.0f, inf);
#else
    return fmaxf(x, 0.f);
#endif
}

__device__ __forceinline__ void layer8(const f32x4* wrows, f32x4 lo, f32x4 hi, int i4, const float* x, float* y) {
    f32x4 al0 = wrows[2 * i4], al1 = wrows[2 * i4 + 1];
    f32x4 ah0 = wrows[2 * (4 + i4)], ah1 = wrows[2 * (4 + i4) + 1];
    float al[8] = {al0.x, al0.y, al0.z, al0.w, al1.x, al1.y, al1.z, al1.w};
    float ah[8] = {ah0.x, ah0.y, ah0.z, ah0.w, ah1.x, ah1.y, ah1.z, ah1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo = MFMA4(al[k], x[k], lo); hi = MFMA4(ah[k], x[k], hi); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { y[r] = lo[r]; y[4 + r] = hi[r]; }
}

// MODE bit0: LDS weight reads inside the loop (else hoisted: registers)   bit1: ReLU   bit2: scale + scan
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, const float* win, float seed, float inf, int nb) {
    extern __shared__ float lds[];
    for (int t = threadIdx.x; t < nb * 184; t += 256) lds[t] = win[t % 184] + seed;
    __syncthreads();
    const int lane = threadIdx.x & 63, i4 = lane & 3;
    float d0 = lane * 0.01f + seed, d1 = 0.3f - lane * 0.003f, d2 = seed;
    float m1 = (lane & 1) ? 1.f : 0.f, m2 = (lane & 2) ? 1.f : 0.f, m4 = (lane & 4) ? 1.f : 0.f, m8 = (lane & 8) ? 1.f : 0.f;
    float accum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < N_IT; ++it) {
        int q = it % nb;
        int woff = q * 184;
        if (MODE & 1) asm volatile("" : "+s"(woff));
        const f32x4* w = reinterpret_cast<const f32x4*>(lds + ((MODE & 1) ? woff : 0));
        PH();
        f32x4 a1lo = w[i4], a1hi = w[4 + i4];
        f32x4 lo = w[8], hi = w[9];
        lo = MFMA4(a1lo.x, d0, lo); hi = MFMA4(a1hi.x, d0, hi);
        lo = MFMA4(a1lo.y, d1, lo); hi = MFMA4(a1hi.y, d1, hi);
        lo = MFMA4(a1lo.z, d2, lo); hi = MFMA4(a1hi.z, d2, hi);
        PH();
        float a1[8], pre2[8], a2[8], o[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { a1[r] = (MODE & 2) ? relu1(lo[r], inf) : lo[r]; a1[4 + r] = (MODE & 2) ? relu1(hi[r], inf) : hi[r]; }
        PH();
        layer8(w + 10, w[26], w[27], i4, a1, pre2);
        PH();
#pragma unroll
        for (int r = 0; r < 8; ++r) a2[r] = (MODE & 2) ? relu1(pre2[r], inf) : pre2[r];
        PH();
        layer8(w + 28, w[44], w[45], i4, a2, o);
        PH();
        if (MODE & 4) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                float v = o[n] * d1;
                v = fmaf(m1, dpp_f<0x111>(v), v);
                v = fmaf(m2, dpp_f<0x112>(v), v);
                v = fmaf(m4, dpp_f<0x114>(v), v);
                v = fmaf(m8, dpp_f<0x118>(v), v);
                accum[n] += v;
            }
        } else {
#pragma unroll
            for (int n = 0; n < 8; ++n) accum[n] += o[n];
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
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256, 8 * 184 * 4>>>(out, win, 0.5f, __builtin_huge_valf(), 8);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256, 8 * 184 * 4>>>(out, win, 0.25f, __builtin_huge_valf(), 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d: %8.1f cycles per block (@2.4 GHz)\n", name, wavesPerSimd, ms * 1e-3 * 2.4e9 / ((double)wavesPerSimd * N_IT));
}

int main() {
    float *win, *out; hipMalloc(&win, 184 * 4); hipMemset(win, 0, 184 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 4, 8}) {
        run<0>("38 mfma only (weights in regs)", w, win, out);
        run<1>("+ LDS weight reads", w, win, out);
        run<3>("+ LDS + relu", w, win, out);
        run<7>("+ LDS + relu + scale/scan (full block)", w, win, out);
        run<6>("relu + scan, weights in regs", w, win, out);
    }
    return 0;
}
