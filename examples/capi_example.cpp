// Stand-alone consumer of the C-ABI (include/mccnn.h): no Python, no torch -- only the HIP runtime for memory.
// Runs compute_aabb -> sort -> find_neighbors -> compute_pdf -> spatial_conv forward AND backward on a random cloud, prints
// per-tensor sums and (third argument) dumps every result tensor; then runs the same layer through the native step
// executor (mccnn_geometry_* / mccnn_conv_*) and prints how far the two are apart. tests/test_gpu_capi_example.py compares
// the dumped tensors with the oracle on the same input, element by element.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/capi_example.cpp -Lmccnn_amd/lib -lmccnn_hip \
//         -Wl,-rpath,$PWD/mccnn_amd/lib -o examples/capi_example && ./examples/capi_example 4096 0.1 [dump.bin]
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

    // ---- SpatialConvGrad: feature gradient + the six kernel-MLP gradients for a deterministic out-gradient
    std::vector<float> og((size_t)n * Fout);
    for (auto& v : og) v = 2 * rnd() - 1;
    float* dOG = dalloc<float>(og.size());
    HK(hipMemcpy(dOG, og.data(), og.size() * 4, hipMemcpyHostToDevice));
    float *fgS = dalloc<float>((size_t)n * Fin), *fg = dalloc<float>((size_t)n * Fin);
    float *gw1 = dalloc<float>(w1.size()), *gb1 = dalloc<float>(b1.size()), *gw2 = dalloc<float>(w2.size()), *gb2 = dalloc<float>(b2.size()),
          *gw3 = dalloc<float>(w3.size()), *gb3 = dalloc<float>(b3.size());
    size_t wsg = mccnn_spatial_conv_bwd_workspace_bytes(n, n, E, Fin, Fout, 1); void* ws6 = dalloc<char>(wsg);
    CK(mccnn_spatial_conv_bwd(sP, sF, sB, pdfs, dP, start, packed, mn, mx, dw1, db1, dw2, db2, dw3, db3, dOG, n, n, E, Fin, Fout, 1, B,
                              radius, scaleInv, 1, /*state=*/nullptr, /*start_t=*/nullptr, /*perm_t=*/nullptr, fgS, gw1, gb1, gw2, gb2,
                              gw3, gb3, ws6, wsg, s));
    CK(mccnn_permute_gather(fgS, idx, n, Fin, fg, s));  // SortPointsStep2Grad: back into the order of the input points
    struct Piece { const char* name; float* dev; size_t count; };
    const Piece pieces[] = {{"out", out, (size_t)n * Fout}, {"feat_grad", fg, (size_t)n * Fin}, {"dw1", gw1, w1.size()},
                            {"db1", gb1, b1.size()}, {"dw2", gw2, w2.size()}, {"db2", gb2, b2.size()}, {"dw3", gw3, w3.size()},
                            {"db3", gb3, b3.size()}};
    FILE* dump = argc > 3 ? fopen(argv[3], "wb") : nullptr;
    std::vector<std::vector<float>> host;
    for (const Piece& p : pieces) {
        std::vector<float> h(p.count);
        HK(hipMemcpyAsync(h.data(), p.dev, p.count * 4, hipMemcpyDeviceToHost, s)); HK(hipStreamSynchronize(s));
        double sm = 0;
        for (float v : h) sm += v;
        printf("%s_sum=%.6e ", p.name, sm);
        if (dump) fwrite(h.data(), 4, h.size(), dump);
        host.push_back(std::move(h));
    }
    printf("\n");
    if (dump) fclose(dump);

    // ---- the same layer through the native step executor: ONE call for the geometry, one per direction
    const int cap = E + E / 16 + 64;
    int* totalHost = nullptr; HK(hipHostMalloc((void**)&totalHost, sizeof(int), hipHostMallocDefault));
    mccnn_geometry_t* g = mccnn_geometry_create();
    size_t gb = mccnn_geometry_bytes(n, n, B, nc, cap, 1); void* gbuf = dalloc<char>(gb);
    CK(mccnn_geometry_build(g, dP, dB, n, dP, dB, n, mn, mx, B, nc, radius, scaleInv, 0.2f, 1, cap, nullptr, gbuf, gb, totalHost, s));
    if (mccnn_geometry_edges(g, -1) != E) { fprintf(stderr, "geometry: E differs\n"); return 1; }
    float *out2 = dalloc<float>((size_t)n * Fout), *fg2 = dalloc<float>((size_t)n * Fin);
    for (int dir = 0; dir < 2; ++dir) {
        int mask = 0, edges = 0; long long need[4], wsb2 = 0, svb = 0;
        static void* saved = nullptr; static size_t savedBytes = 0;
        CK(mccnn_conv_prepare(g, dF, Fin, Fout, 1, 0, dir, 1, &mask, need, &wsb2, &svb, &edges));
        for (int k = 0; k < 4; ++k)
            if (mask & (1 << k)) CK(mccnn_geometry_attach(g, 1 << k, dalloc<char>((size_t)need[k]), (size_t)need[k]));
        void* wsx = dalloc<char>((size_t)wsb2);
        if (dir == 0) {
            saved = dalloc<char>((size_t)svb); savedBytes = (size_t)svb;
            CK(mccnn_conv_forward(g, dF, Fin, Fout, 1, 1, 0, 1, dw1, db1, dw2, db2, dw3, db3, out2, saved, savedBytes, wsx, (size_t)wsb2, s));
        } else {
            CK(mccnn_conv_backward(g, dF, saved, savedBytes, dOG, Fin, Fout, 1, 1, 0, 1, dw1, db1, dw2, db2, dw3, db3, fg2, gw1, gb1, gw2,
                                   gb2, gw3, gb3, wsx, (size_t)wsb2, s));
        }
    }
    double worst = 0;
    const Piece again[] = {{"out", out2, (size_t)n * Fout}, {"feat_grad", fg2, (size_t)n * Fin}, {"dw1", gw1, w1.size()},
                           {"db1", gb1, b1.size()}, {"dw2", gw2, w2.size()}, {"db2", gb2, b2.size()}, {"dw3", gw3, w3.size()},
                           {"db3", gb3, b3.size()}};
    for (size_t k = 0; k < 8; ++k) {
        std::vector<float> h(again[k].count);
        HK(hipMemcpyAsync(h.data(), again[k].dev, h.size() * 4, hipMemcpyDeviceToHost, s)); HK(hipStreamSynchronize(s));
        double scale = 1e-30, d = 0;
        for (size_t t = 0; t < h.size(); ++t) {
            const double a = host[k][t] < 0 ? -host[k][t] : host[k][t], e = h[t] - host[k][t];
            if (a > scale) scale = a;
            if ((e < 0 ? -e : e) > d) d = e < 0 ? -e : e;
        }
        if (d / scale > worst) worst = d / scale;
    }
    printf("executor_vs_ops_max_rel=%.3e\n", worst);
    mccnn_geometry_destroy(g);
    return 0;
}
