// Probe: lane/register layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* outA, float* outB) {
    int l = threadIdx.x;
    f32x4 z = {0, 0, 0, 0};
    f32x4 dA = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);
    f32x4 dB = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        outA[l * 4 + r] = dA[r];
        outB[l * 4 + r] = dB[r];
    }
}

template <int KIND>
__global__ void rate(float* out, int iters) {
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND == 0) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k], 0, 0, 0);
            else acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
        }
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *dA, *dB;
    hipMalloc(&dA, 256 * 4);
    hipMalloc(&dB, 256 * 4);
    layout<<<1, 64>>>(dA, dB);
    std::vector<float> hA(256), hB(256);
    hipMemcpy(hA.data(), dA, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(hB.data(), dB, 1024, hipMemcpyDeviceToHost);
    printf("4x4x1: D[lane][reg] = A(from lane) * B(from lane)\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) printf("  r%d: A<-%2d B<-%2d", r, (int)hA[l * 4 + r] - 1, (int)hB[l * 4 + r] - 1);
        printf("\n");
    }
    float* out;
    int blocks = 256 * 4, threads = 256;
    hipMalloc(&out, blocks * threads * 4);
    for (int kind = 0; kind < 2; ++kind) {
        int iters = 20000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        if (kind == 0) rate<0><<<blocks, threads>>>(out, 100); else rate<1><<<blocks, threads>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (kind == 0) rate<0><<<blocks, threads>>>(out, iters); else rate<1><<<blocks, threads>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flopPer = kind == 0 ? 512.0 : 2048.0;
        double total = (double)blocks * (threads / 64) * iters * 8 * flopPer;
        printf("%s: %.3f ms, %.1f TFLOP/s, %.2f cycles/instr/SIMD @2.4GHz (4 waves/SIMD resident)\n",
               kind == 0 ? "mfma_f32_4x4x1_16b" : "mfma_f32_16x16x4", ms, total / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)blocks * (threads / 64) * iters * 8 / (256.0 * 4)));
    }
    return 0;
}
