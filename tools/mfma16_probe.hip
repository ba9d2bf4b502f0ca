// Does v_mfma_f32_16x16x4_f32 (32 cycles per instruction per SIMD) leave the VALU free, unlike the 4x4x1 form (whose
// cost ADDS to the VALU work of the same wave, profiles/r01_issue_probe.txt)? Cycles per loop iteration per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma16_probe.hip -o tools/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_IT 2000

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float seed) {
    __shared__ f32x4 lds[1024];
    float a = threadIdx.x * 0.001f + seed, b = 1.0f + threadIdx.x * 0.002f;
    float v[32];
    f32x4 acc[8], sm[4];
    f32x2 p[16];
    for (int k = 0; k < 32; ++k) v[k] = a + k;
    for (int k = 0; k < 8; ++k) acc[k] = (f32x4){a, b, a, b};
    for (int k = 0; k < 4; ++k) sm[k] = (f32x4){b, a, b, a};
    for (int k = 0; k < 16; ++k) p[k] = (f32x2){a + k, b};
    lds[threadIdx.x] = acc[0];
    __syncthreads();
    for (int it = 0; it < N_IT; ++it) {
        if (KIND == 0) {  // 8 independent 16x16x4
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
        } else if (KIND == 1) {  // 8 x (16x16x4, 4 independent v_fma)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * k + j] = fmaf(v[4 * k + j], b, a);
            }
        } else if (KIND == 2) {  // the same, grouped: 8 mfma | 32 fma
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = fmaf(v[k], b, a);
            __builtin_amdgcn_sched_barrier(0);
        } else if (KIND == 3) {  // 32 v_fma alone
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = fmaf(v[k], b, a);
        } else if (KIND == 4) {  // 8 x (16x16x4, 8 v_fma): VALU 8 x 8 x ~3 = 192 < 256
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[4 * k + j] = fmaf(v[4 * k + j], b, a); v[(4 * k + j + 16) & 31] = fmaf(v[(4 * k + j + 16) & 31], a, b); }
            }
        } else if (KIND == 5) {  // 8 x (16x16x4, 2 x 4x4x1): both on the matrix pipe
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
                sm[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[k & 3], 0, 0, 0);
                sm[(k + 2) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, sm[(k + 2) & 3], 0, 0, 0);
            }
        } else if (KIND == 6) {  // 16 x 4x4x1 alone
#pragma unroll
            for (int k = 0; k < 16; ++k) sm[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, sm[k & 3], 0, 0, 0);
        } else if (KIND == 7) {  // 8 x (16x16x4, 2 pk_fma)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
                p[2 * k] = __builtin_elementwise_fma(p[2 * k], (f32x2){b, b}, (f32x2){a, a});
                p[2 * k + 1] = __builtin_elementwise_fma(p[2 * k + 1], (f32x2){b, b}, (f32x2){a, a});
            }
        } else if (KIND == 8) {  // 8 x (16x16x4 whose A operand comes from LDS: ds_write_b128 + ds_read_b32 per mfma)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                lds[(threadIdx.x + 64 * k) & 1023] = acc[(k + 4) & 7];
                const float x = reinterpret_cast<float*>(lds)[(threadIdx.x * 4 + k) & 4095];
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b, acc[k], 0, 0, 0);
            }
        } else if (KIND == 9) {  // 4 dependent 16x16x4 on ONE accumulator x 2 chains
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[1], 0, 0, 0);
            }
        }
    }
    float s = 0;
    if (out == nullptr) return;
    for (int k = 0; k < 32; ++k) s += v[k];
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 4; ++k) s += sm[k][0] + sm[k][3];
    for (int k = 0; k < 16; ++k) s += p[k][0] + p[k][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x][1];
}

template <int KIND>
void run(const char* name, int blocksPerCU) {
    float* out;
    int blocks = 256 * blocksPerCU;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    probe<KIND><<<blocks, 256>>>(out, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<KIND><<<blocks, 256>>>(out, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s waves/SIMD %d: %.3f ms -> %.1f cycles per iteration per wave, %.1f per SIMD (@2.4 GHz)\n", name, blocksPerCU, ms,
           ms * 1e-3 * 2.4e9 / N_IT, ms * 1e-3 * 2.4e9 / N_IT / blocksPerCU);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("8 x mfma16x16x4 (indep)", w);
        run<3>("32 x v_fma alone", w);
        run<1>("8 x (mfma16, 4 v_fma) interleaved", w);
        run<2>("8 mfma16 | 32 v_fma grouped", w);
        run<4>("8 x (mfma16, 8 v_fma) interleaved", w);
        run<6>("16 x mfma4x4x1 alone", w);
        run<5>("8 x (mfma16, 2 mfma4x4x1)", w);
        run<7>("8 x (mfma16, 2 v_pk_fma)", w);
        run<8>("8 x (ds_write_b128, ds_read_b32, mfma16)", w);
        run<9>("2 dependent chains x 4 mfma16", w);
    }
    return 0;
}
