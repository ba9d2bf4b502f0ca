// Batch form of a step's geometries (exec.hip: mccnn_geometry_build_batch): ONE launch per kernel kind over up to
// MCCNN_BATCH_MAX geometries. The per-geometry arguments of every kind travel by value in the kernel arguments (BatchBlocks,
// ScanBatch, SpanBatch: common.h); kernels and item builders live next to their single forms (grid.hip, neighbors.hip, scan.hip).
#pragma once
#include "common.h"

namespace mccnn {

// one counting sort: the grid build of a geometry, or the visiting order of its foreign centres (newIdx == nullptr)
struct GridItem {
    const float* pts; const int* bids; const float* mn; const float* mx;
    int* keys; int* cnt; int* arrival; int* start; int* slot;
    int* newIdx; float* oPts; int* oBids; int* inv; int2* cells;
    int n, B, nc, C, ldsHist;
};
struct GridBatch { GridItem it[MCCNN_BATCH_MAX]; };

// one neighbour search (count pass / fill pass)
struct NeighItem {
    const float* centres; const int* cb; const float* pts; const int* cells; const float* mn; const float* mx; const int* order;
    int* cnt; unsigned long long* masks; int* startIdx; int* packed; unsigned long long* zeroWords;
    int m, B, nc, scaleInv, capacity, G, numZero;
    float radius, Tabs;
};
struct NeighBatch { NeighItem it[MCCNN_BATCH_MAX]; };

// one kernel-density estimate
struct PdfItem {
    const float* pts; const int* bids; const int2* packed; const int* startIdx; const float* mn; const float* mx; float* pdfs;
    const int* eDev;
    int m, e, B, scaleInv, rowsPerWave;
    float window, radius;
};
struct PdfBatch { PdfItem it[MCCNN_BATCH_MAX]; };

// ---- small row plans of a step (mccnn_geometry_prebuild_batch): transposition, layout and fill of every small list as one
// launch each. (12 items per launch: the fill's item carries the geometry an inline record is evaluated from.)
#define MCCNN_PLAN_BATCH_MAX 12
struct TrSmallItem { const int2* packed; int* cnt; int* slot; int* tmp; int* startT; int* permT; int e, n; };
struct TrSmallBatch { TrSmallItem it[MCCNN_BATCH_MAX]; };
struct PlanSmallItem { const int* rowStart; const int* order; int* vrow; int* vcode; int* sliceOff; int* vposRow; int rows, e, L, S; };
struct PlanSmallBatch { PlanSmallItem it[MCCNN_BATCH_MAX]; };
struct SellFillItem {
    // the list and its geometry (records are evaluated inline: conv_mfma.h edge_record)
    const float* pts; const int* bids; const float* pdfs; const float* samples; const int* start; const int2* packed;
    const float* mn; const float* mx;
    const int* rowStart; const int* permT;
    // the plan
    const int* vrow; const int* vcode; const int* sliceOff; const int* vposRow; float4* rec; int* oth;
    long long cap;
    int n, m, e, B, scaleInv, avg, rows, S, L, chunks;
    float radius;
};
struct SellFillBatch { SellFillItem it[MCCNN_PLAN_BATCH_MAX]; };
// the transposition of a list too long for one workgroup: count -> prefix sum (scan.hip's batch form) -> fill -> rank
struct TrChainItem { const int2* packed; int* cnt; int* slot; int* tmp; int* startT; int* permT; int e, n, lds, norank /* ranked by the plan's scatter */; };
struct TrChainBatch { TrChainItem it[MCCNN_BATCH_MAX]; };
int tr_chain_item(TrChainItem& t, ScanItem& sc, ClearSpan& head, const int* packed, int e, int n, int* start_t, int* perm_t, void* ws,
                  size_t ws_bytes);                                                                 // conv.hip (ws: mccnn_transpose_neighbors_workspace_bytes)
int launch_tr_chain_batch(const TrChainBatch& tb, int count, int phase, hipStream_t s);            // 0 count, 1 fill, 2 rank
// a LARGE row plan (every plan plan_small does not lay out): layout (vr_count -> vr_scan_expand -> sell_sort), then the
// forward plan's tile fill or the transposed plan's bases + scatter (conv_rows.hip build_large has the single-plan chain)
#define MCCNN_LARGE_BATCH_MAX 8
struct LargeItem {
    const int* rowStart; const int* order;
    int* vcnt; unsigned long long* st1; unsigned long long* st2; int* vlistRow; int* vTotal;
    int* vrow; int* vcode; int* sliceOff; int* vposRow; int* oth; float4* rec;
    long long cap;
    const float* pts; const int* bids; const float* pdfs; const float* samples; const int* start; const int2* packed;
    const float* mn; const float* mx;
    float4* recE;                                    // per-edge records in edge order (evaluated here when `eval`)
    int2* vinfo; const int* startT; const int* tmp; int* permT;   // transposed plans
    int rows, e, n, m, B, L, S, windows, tiles, scaleInv, avg;
    int tr, eval /* 0 records ready, 1 evaluate (fwd: in the fill; tr: a pass of its own) */, rank /* tr: the list is ranked in the scatter */;
    float radius;
};
struct LargeBatch { LargeItem it[MCCNN_LARGE_BATCH_MAX]; };
bool plan_large_batchable(int rows, int e, int n, int transposed, int tlist_ready);                 // conv_rows.hip
size_t plan_large_ws_bytes(int rows, int e, int n, int transposed, int tlist_ready);
// -> item + (transposed, list not ready) the chain item of its transposition (*use_chain) + the spans its head has to clear
int plan_large_item(int transposed, const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs, const float* samples,
                    const int* start_idx, const int* packed, const float* aabb_min, const float* aabb_max, int n, int m, int e,
                    int batch_size, float radius, int scale_inv, int avg, const int* order, void* rec_edges, int rec_ready, int* start_t,
                    int* perm_t, int tlist_ready, void* plan_buffer, void* ws, size_t ws_bytes, LargeItem& it, TrChainItem* tc,
                    ScanItem* sc, bool* use_chain, SpanBatch& spans);
enum { LARGE_VR_COUNT = 0, LARGE_VR_SCAN, LARGE_SELL_SORT, LARGE_BASES, LARGE_RECORDS, LARGE_FILL, LARGE_SCATTER };
int launch_plan_large_batch(const LargeBatch& lb, int count, int phase, hipStream_t s);

int launch_tr_small_batch(const TrSmallBatch& tb, int count, hipStream_t s);                       // conv.hip
int launch_plan_small_batch(const PlanSmallBatch& pb, int count, hipStream_t s);                   // conv_rows.hip
int launch_sell_fill_batch(const SellFillBatch& fb, int count, int transposed, hipStream_t s);     // conv_rows.hip

bool grid_batch_eligible(int n, int batch_size, int num_cells);
int grid_batch_item(GridItem& g, ScanItem& sc, ClearSpan& head, const float* pts, const int* batch_ids, const float* aabb_min,
                    const float* aabb_max, int n, int batch_size, int num_cells, int* new_idx, float* out_pts,
                    int* out_batch_ids, int* cell_indexs, int* inv_idx, void* ws, size_t ws_bytes);
int order_batch_item(GridItem& g, ScanItem& sc, ClearSpan& head, const float* pts, const int* batch_ids, const float* aabb_min,
                     const float* aabb_max, int m, int batch_size, int num_cells, int* order, void* ws, size_t ws_bytes);
int launch_grid_batch_phase(const GridBatch& gb, int count, int phase, hipStream_t s);   // 0 keys + histogram, 1 park, 2 rank + move + cells
bool neigh_batch_eligible(int m, int n);
int neigh_batch_item(NeighItem& it, ScanItem& sc, const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts,
                     int n, const int* cell_indexs, const float* aabb_min, const float* aabb_max, int batch_size, int num_cells,
                     float radius, int scale_inv, const int* order, int* start_idx, int e_capacity, int* packed, int* total_dev,
                     int* total_host, void* ws, size_t ws_bytes);
int launch_neigh_batch(const NeighBatch& nbt, int count, int mode, hipStream_t s);        // 0 count, 1 fill
void pdf_batch_item(PdfItem& it, const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx, int m, const int* packed,
                    int e_capacity, const int* e_dev, const float* aabb_min, const float* aabb_max, int batch_size, float window,
                    float radius, int scale_inv, float* pdfs);
int launch_pdf_batch(const PdfBatch& pb, int count, hipStream_t s);

}  // namespace mccnn
