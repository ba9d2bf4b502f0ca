#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, float* q16a, float* q16b, float* q32a, float* q32b) {
    float x = p[threadIdx.x];
    unsigned u = __builtin_bit_cast(unsigned, x);
    (void)u;
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    q16a[threadIdx.x] = a; q16b[threadIdx.x] = b;
    float c = x, d = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    q32a[threadIdx.x] = c; q32b[threadIdx.x] = d;
}
int main() {
    float h[64], *d; for (int i = 0; i < 64; ++i) h[i] = i;
    hipMalloc(&d, 5 * 64 * 4); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, d + 64, d + 128, d + 192, d + 256);
    float o[256]; hipMemcpy(o, d + 64, 1024, hipMemcpyDeviceToHost);
    const char* n[4] = {"16 dst", "16 src", "32 dst", "32 src"};
    for (int a = 0; a < 4; ++a) { printf("%s:", n[a]); for (int i = 0; i < 64; ++i) printf(" %d", (int)o[a * 64 + i]); printf("\n"); }
}
