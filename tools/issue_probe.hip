// Issue-cost probe for gfx950: cycles per wave-instruction for the instruction mix used by the conv kernels.
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_IT 2000

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float seed, long long* clk) {
    long long c0 = __builtin_readcyclecounter();
    float a = threadIdx.x * 0.001f + seed, b = 1.0f + threadIdx.x * 0.002f;
    float v[16];
    f32x4 acc[8];
    f32x2 p[8];
    for (int k = 0; k < 16; ++k) v[k] = a + k;
    for (int k = 0; k < 8; ++k) { acc[k] = (f32x4){a, b, a, b}; p[k] = (f32x2){a + k, b}; }
    for (int it = 0; it < N_IT; ++it) {
        if (KIND == 0) {  // 16 independent v_fma_f32
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], b, a);
        } else if (KIND == 1) {  // 8 independent v_pk_fma_f32
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = __builtin_elementwise_fma(p[k], (f32x2){b, b}, (f32x2){a, a});
        } else if (KIND == 2) {  // 8 independent mfma 4x4x1
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
        } else if (KIND == 3) {  // 8 mfma + 16 fma interleaved (same wave)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
                v[2 * k] = fmaf(v[2 * k], b, a);
                v[2 * k + 1] = fmaf(v[2 * k + 1], b, a);
            }
        } else if (KIND == 4) {  // 2 dependent chains of mfma (like layer8)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc[1], 0, 0, 0);
            }
        } else if (KIND == 5) {  // 16 v_mov_dpp + add (row_shr:1)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x111, 0xf, 0xf, true));
        } else if (KIND == 6) {  // 16 v_max (relu)
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = __builtin_amdgcn_fmed3f(v[k] - 1.0f, 0.0f, __builtin_huge_valf());
        } else if (KIND == 7) {  // 1 dependent chain of mfma
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0);
        } else if (KIND == 8) {  // 8 mfma then 16 fma, grouped (phases kept apart by scheduling barriers)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], b, a);
            __builtin_amdgcn_sched_barrier(0);
        } else if (KIND == 9) {  // 8 mfma, each followed by 2 fma that READ its result (dependent, like relu after a layer)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
                v[2 * k] = fmaf(v[2 * k], b, acc[(k + 4) & 7][0]);
                v[2 * k + 1] = fmaf(v[2 * k + 1], b, acc[(k + 4) & 7][1]);
            }
        } else if (KIND == 10) {  // 1 mfma : 1 fma alternating, independent
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                v[k] = fmaf(v[k], b, a);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (KIND == 11) {  // 16 ds_read_b128-like LDS reads + 8 mfma (LDS pipe in parallel?)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
        }
    }
    float s = 0;
    if (out == nullptr) return;
    for (int k = 0; k < 16; ++k) s += v[k];
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3] + p[k][0] + p[k][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = __builtin_readcyclecounter() - c0;
}

template <int KIND>
void run(const char* name, int instrPerIter, int blocksPerCU) {
    float* out;
    int blocks = 256 * blocksPerCU;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    long long* clk; hipMalloc(&clk, 8);
    probe<KIND><<<blocks, 256>>>(out, 0.5f, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<KIND><<<blocks, 256>>>(out, 0.25f, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocksPerCU (4 waves per block, 4 SIMDs per CU)
    double instrPerSimd = (double)blocksPerCU * N_IT * instrPerIter;
    long long hc = 0; hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
    // s_memtime ticks of workgroup 0 (constant 100 MHz on gfx9) -- its lifetime vs the launch shows how many rounds ran
    printf("%-40s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (@2.4 GHz)  [wg0 %lld ticks]\n", name, blocksPerCU, ms,
           ms * 1e-3 * 2.4e9 / instrPerSimd, hc);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 x16 (indep)", 16, w);
        run<1>("v_pk_fma_f32 x8 (indep)", 8, w);
        run<2>("mfma_4x4x1 x8 (indep)", 8, w);
        run<3>("8 mfma + 16 fma interleaved", 24, w);
        run<4>("mfma 2 dependent chains x8", 8, w);
        run<7>("mfma 1 dependent chain x8", 8, w);
        run<8>("8 mfma | 16 fma grouped", 24, w);
        run<9>("8 mfma + 16 fma reading mfma results", 24, w);
        run<10>("8 x (mfma, fma) alternating", 16, w);
        run<5>("mov_dpp+add x16", 32, w);
        run<6>("sub+med3 x16", 32, w);
    }
    return 0;
}
