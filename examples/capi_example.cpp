// Stand-alone consumer of the C-ABI (include/mccnn.h): no Python, no torch -- only the HIP runtime for memory.
// Runs compute_aabb -> sort -> find_neighbors -> compute_pdf -> spatial_conv forward on a random cloud and prints
// a checksum; tests/test_gpu_capi_example.py compares the checksum path against the oracle on the same input.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/capi_example.cpp -Lmccnn_amd/lib -lmccnn_hip \
//         -Wl,-rpath,$PWD/mccnn_amd/lib -o examples/capi_example && ./examples/capi_example 4096 0.1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mccnn.h"

#define CK(x) do { int rc__ = (x); if (rc__ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc__, mccnn_error_string(rc__)); return 1; } } while (0)
#define HK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

template <typename T> static T* dalloc(size_t n) { void* p = nullptr; hipMalloc(&p, (n ? n : 1) * sizeof(T)); return (T*)p; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const float radius = argc > 2 ? (float)atof(argv[2]) : 0.1f;
    const int B = 1, Fin = 3, Fout = 8, nb = 3, scaleInv = 1;
    // deterministic inputs (the test regenerates the same numbers): LCG in [0,1)
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) * (1.0f / 16777216.0f); };
    std::vector<float> pts(n * 3), feats((size_t)n * Fin), w1(3 * 8 * nb), b1(8 * nb), w2(64 * nb), b2(8 * nb), w3(64 * nb), b3(8 * nb);
    for (auto& v : pts) v = rnd();
    for (auto& v : feats) v = 2 * rnd() - 1;
    for (auto* w : {&w1, &w2, &w3}) for (auto& v : *w) v = rnd() - 0.5f;
    for (auto* b : {&b1, &b2, &b3}) for (auto& v : *b) v = 0.1f * (rnd() - 0.5f);
    std::vector<int> bids(n, 0);
    hipStream_t s; HK(hipStreamCreate(&s));
    float *dP = dalloc<float>(n * 3), *dF = dalloc<float>((size_t)n * Fin), *mn = dalloc<float>(3), *mx = dalloc<float>(3);
    int* dB = dalloc<int>(n);
    HK(hipMemcpy(dP, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dF, feats.data(), feats.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dB, bids.data(), n * 4, hipMemcpyHostToDevice));
    size_t wsb = mccnn_compute_aabb_workspace_bytes(B); void* ws = dalloc<char>(wsb);
    CK(mccnn_compute_aabb(dP, dB, n, B, scaleInv, mn, mx, ws, wsb, s));
    int nc = 0; CK(mccnn_num_cells(mn, mx, B, radius, scaleInv, &nc, s));
    int *keys = dalloc<int>(n), *idx = dalloc<int>(n);
    wsb = mccnn_sort_step1_workspace_bytes(n, B, nc); void* ws1 = dalloc<char>(wsb);
    CK(mccnn_sort_step1(dP, dB, mn, mx, n, B, nc, keys, idx, ws1, wsb, s));
    float *sP = dalloc<float>(n * 3), *sF = dalloc<float>((size_t)n * Fin); int *sB = dalloc<int>(n), *cells = dalloc<int>((size_t)B * nc * nc * nc * 2);
    wsb = mccnn_sort_step2_workspace_bytes(n); void* ws2 = dalloc<char>(wsb);
    CK(mccnn_sort_step2(dP, dB, dF, keys, idx, n, Fin, B, nc, sP, sB, sF, cells, /*inv_idx=*/nullptr, ws2, wsb, s));
    int *start = dalloc<int>(n), *total = dalloc<int>(1);
    wsb = mccnn_find_neighbors_workspace_bytes(n, n); void* ws3 = dalloc<char>(wsb);
    CK(mccnn_find_neighbors_count(dP, dB, n, sP, n, cells, mn, mx, B, nc, radius, scaleInv, nullptr, start, total, ws3, wsb, s));
    int E = 0; HK(hipMemcpyAsync(&E, total, 4, hipMemcpyDeviceToHost, s)); HK(hipStreamSynchronize(s));
    int* packed = dalloc<int>((size_t)E * 2);
    CK(mccnn_find_neighbors_fill(dP, dB, n, sP, n, cells, mn, mx, B, nc, radius, scaleInv, nullptr, start, E, packed, ws3, wsb, s));
    float* pdfs = dalloc<float>(E);
    size_t wsp = mccnn_compute_pdf_workspace_bytes(E, 1); void* ws4 = dalloc<char>(wsp);
    CK(mccnn_compute_pdf(sP, sB, start, n, packed, E, mn, mx, B, 0.2f, radius, scaleInv, 1, pdfs, ws4, wsp, s));
    float *dw1 = dalloc<float>(w1.size()), *db1 = dalloc<float>(b1.size()), *dw2 = dalloc<float>(w2.size()), *db2 = dalloc<float>(b2.size()),
          *dw3 = dalloc<float>(w3.size()), *db3 = dalloc<float>(b3.size()), *out = dalloc<float>((size_t)n * Fout);
    HK(hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(db1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(db2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dw3, w3.data(), w3.size() * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(db3, b3.data(), b3.size() * 4, hipMemcpyHostToDevice));
    size_t wsc = mccnn_spatial_conv_fwd_workspace_bytes(n, E, Fin, Fout, 1); void* ws5 = dalloc<char>(wsc);
    CK(mccnn_spatial_conv_fwd(sP, sF, sB, pdfs, dP, start, packed, mn, mx, dw1, db1, dw2, db2, dw3, db3, n, n, E, Fin, Fout, 1, B,
                              radius, scaleInv, 1, out, /*state=*/nullptr, ws5, wsc, s));
    std::vector<float> hout((size_t)n * Fout);
    HK(hipMemcpyAsync(hout.data(), out, hout.size() * 4, hipMemcpyDeviceToHost, s)); HK(hipStreamSynchronize(s));
    double sum = 0, asum = 0;
    for (float v : hout) { sum += v; asum += v < 0 ? -v : v; }
    printf("arch=%s block=%d n=%d nc=%d E=%d out_sum=%.6e out_abs_sum=%.6e\n", mccnn_arch(), mccnn_block_size(), n, nc, E, sum, asum);
    return 0;
}
