// Host cost of a kernel launch on this box: hipcc --offload-arch=gfx950 -O2 tools/launch_probe.hip -o /tmp/launch_probe && /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[200]; };
__global__ void k0(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k1(Big a, int* p) { if (p && threadIdx.x == 9999) *p = a.b[0]; }
int main() {
    int* d;
    hipMalloc(&d, 4);
    hipStream_t s;
    hipStreamCreate(&s);
    Big big = {};
    for (int rep = 0; rep < 3; ++rep) {
        for (int mode = 0; mode < 3; ++mode) {
            const int N = 2000;
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 0) k0<<<1, 64, 0, s>>>(d);
                else if (mode == 1) k1<<<64, 256, 0, s>>>(big, d);
                else hipMemsetAsync(d, 0, 4, s);
            }
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(s);
            auto t2 = std::chrono::steady_clock::now();
            printf("%s: %.2f us per launch to enqueue, %.2f us per launch until done\n",
                   mode == 0 ? "empty kernel, 8 B of arguments" : (mode == 1 ? "64 workgroups, 208 B of arguments" : "hipMemsetAsync 4 B"),
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
    }
    return 0;
}
