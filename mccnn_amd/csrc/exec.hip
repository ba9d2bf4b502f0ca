// exec.hip -- the native step executor: what ConvolutionBuilder.create_convolution (MCConvBuilder.py:336-427) does
// around the kernels, behind TWO kinds of C-ABI calls instead of a dozen Python-mediated ones:
//   * mccnn_geometry_build : sort_points_step1/2 of the input level, find_neighbors (count, scan, fill into a buffer sized by
//     the caller's guess) and compute_pdf, enqueued back to back into ONE caller-provided device buffer, without a host
//     wait -- the geometry one (level, radius, output level) cache entry of the builder stands for (:349-391);
//   * mccnn_conv_forward / mccnn_conv_backward : the convolution of one layer over a geometry, INCLUDING the feature sort
//     (sort_features / its gradient, MCConvModuleSrc:35-45), the choice of the kernel family (row-per-lane depth-wise,
//     factored Fin = 1, edge streaming) and the row plans / transposed list a family needs, built on first use and kept
//     with the geometry for every further layer over the same list.
// No kernels live here: every launch goes through the op-level entry points of this library (mccnn.h), so the results are
// those of the op-by-op chain bit for bit. The library still allocates no device memory: buffers come from the caller,
// sized by the *_bytes queries.
#include "batch.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

namespace mccnn {
std::atomic<int>& conv_impl_override();  // conv.hip (mccnn_debug_conv_impl)
// grid.hip: the grid build as one chain of four launches, the visiting order of foreign centres as three; what each wants
// cleared before its first kernel is a span the caller may clear together with others (one launch per geometry)
ClearSpan grid_head_span(int n, int batch_size, int num_cells, void* ws, size_t ws_bytes);
int build_grid_fused(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int n,
                     int batch_size, int num_cells, int* new_idx, float* out_pts, int* out_batch_ids, int* cell_indexs,
                     int* inv_idx, void* ws, size_t ws_bytes, hipStream_t s, const int* n_dev, bool cleared, ClearSpan x1,
                     ClearSpan x2);
size_t visiting_order_workspace_bytes(int m, int batch_size, int num_cells);
ClearSpan visiting_order_head_span(int m, int batch_size, int num_cells, void* ws, size_t ws_bytes);
int visiting_order(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int m, int batch_size,
                   int num_cells, int* order, void* ws, size_t ws_bytes, hipStream_t s, bool cleared);
// neighbors.hip: count + (scan) + fill; lists of few centres scan their counts inside the fill pass
// conv_rows.hip: small row plans as batch items
bool plan_batchable(int rows, int e, int transposed);
size_t plan_batch_tr_ws_bytes(int n, int e);
int plan_batch_items(int transposed, const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs, const float* samples,
                     const int* start_idx, const int* packed, const float* aabb_min, const float* aabb_max, int n, int m, int e,
                     int batch_size, float radius, int scale_inv, int avg, const int* order, int* start_t, int* perm_t,
                     int tlist_ready, void* plan_buffer, void* tws, size_t tws_bytes, TrSmallItem* tr, bool* use_tr,
                     PlanSmallItem& lay, SellFillItem& fill);
bool transpose_small(int e, int n);
int find_neighbors_chain(const float* centres, const int* centre_batch_ids, int m, const float* sorted_pts, int n,
                         const int* cell_indexs, const float* aabb_min, const float* aabb_max, int batch_size, int num_cells,
                         float radius, int scale_inv, const int* centre_order, int* start_idx, int e_capacity, int* packed,
                         int* total_dev, int* total_host, void* ws, size_t ws_bytes, mccnn_stream_t stream);
}

using namespace mccnn;

namespace {

struct Plan {
    char* buf = nullptr;
    size_t bytes = 0;
    bool built = false;
    int avg = -1;
    long long off[6] = {0, 0, 0, 0, 0, 0};
    long long total = 0, slots = 0, scratch_rows = 0;
    int S = 0;
};

inline size_t al(size_t x) { return align_up(x); }

}  // namespace

struct mccnn_geometry {
    // inputs (borrowed: the caller keeps them alive as long as the geometry is used)
    const float* pts = nullptr;
    const int* bids = nullptr;
    const float* centres = nullptr;
    const int* cbids = nullptr;
    const float* mn = nullptr;
    const float* mx = nullptr;
    int n = 0, m = 0, B = 0, nc = 0, scale_inv = 0, use_pdf = 1, same_level = 0;
    float radius = 0.f, window = 0.f;
    // the geometry buffer
    char* buf = nullptr;
    size_t bytes = 0;
    float* s_pts = nullptr;
    int *s_bids = nullptr, *cells = nullptr, *new_idx = nullptr, *inv_idx = nullptr;
    int *start = nullptr, *packed = nullptr, *total_dev = nullptr, *order = nullptr, *order_buf = nullptr;
    float* pdfs = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    const mccnn_geometry* grid_of = nullptr;  // the grid arrays belong to another geometry (same level, radius)
    // edge count
    int e_cap = 0;
    int e = -1;                      // known once the pinned word has been read
    volatile int* total_host = nullptr;
    bool built = false;
    // attachments
    Plan plan[2];                    // [0] forward (rows = centres), [1] transposed (rows = points)
    char* tl_buf = nullptr;          // transposed list: start_t [n + 1] | perm_t [e]
    size_t tl_bytes = 0;
    bool tl_built = false;
    char* rec_buf = nullptr;         // per-edge records (16 B per edge) in edge order
    size_t rec_bytes = 0;
    int rec_avg = -1;
};

namespace {

enum { NEED_PLAN_FWD = 1, NEED_PLAN_TR = 2, NEED_TLIST = 4, NEED_RECORDS = 8 };

struct GeoLayout {
    size_t s_pts, s_bids, cells, new_idx, inv_idx, start, packed, pdfs, total, order, ws, ws_bytes, total_bytes;
};

size_t cells_of(int B, int nc) { return (size_t)B * nc * nc * nc; }
constexpr int MCCNN_ORDER_MIN_M = 16384;  // foreign centres get a visiting order of their own from this many on

// byte offsets of the pieces of a geometry buffer; total_bytes == 0: the grid does not fit 32-bit keys
GeoLayout geo_layout(int n, int m, int B, int nc, int e_cap, bool with_grid) {
    GeoLayout L;
    memset(&L, 0, sizeof(L));
    const size_t n1 = n > 0 ? n : 1, m1 = m > 0 ? m : 1, e1 = e_cap > 0 ? e_cap : 1;
    size_t o = 0;
    if (with_grid) {
        L.s_pts = o; o += al(n1 * 12);
        L.s_bids = o; o += al(n1 * 4);
        L.cells = o; o += al(cells_of(B, nc) * 8);
        L.new_idx = o; o += al(n1 * 4);
        L.inv_idx = o; o += al(n1 * 4);
    }
    L.start = o; o += al(m1 * 4);
    L.packed = o; o += al(e1 * 8);
    L.pdfs = o; o += al(e1 * 4);
    L.total = o; o += 256;
    L.order = o; o += al(m1 * 4);
    size_t w = mccnn_find_neighbors_workspace_bytes(m, n);
    // the grid build and the visiting order of foreign centres (>= MCCNN_ORDER_MIN_M of them) run before the search and
    // are cleared by ONE launch at the head of the chain: side by side, not aliased (the search and the KDE reuse the space)
    size_t head = 0;
    if (with_grid) {
        const size_t g = mccnn_build_grid_workspace_bytes(n, B, nc);
        if (g == 0) return L;
        head += al(g);
    }
    if (m >= MCCNN_ORDER_MIN_M) {
        const size_t c = visiting_order_workspace_bytes(m, B, nc);
        if (c == 0) return L;
        head += al(c);
    }
    if (head > w) w = head;
    if (mccnn_sort_step1_workspace_bytes(m, B, nc) == 0) return L;   // (32-bit keys)
    const size_t p = mccnn_compute_pdf_workspace_bytes(e_cap, 1);
    if (p > w) w = p;
    L.ws = o;
    L.ws_bytes = al(w) + 256;
    o += L.ws_bytes;
    L.total_bytes = o;
    return L;
}

int read_mask() { return conv_impl_override().load(std::memory_order_relaxed); }

bool env_flag(const char* name, bool dflt) {
    const char* v = getenv(name);
    return v ? (strcmp(v, "0") != 0) : dflt;
}
bool row_kernels_on() { static const bool on = env_flag("MCCNN_ROW_KERNELS", true); return on; }
float rows_min_degree() { static const float d = (float)debug_float("rows_min_degree", 16.0); return d; }
int unsorted_max_points() { static const int v = debug_int("unsorted_max_points", 32768); return v; }

// Row-per-lane kernels for this (layer, list, direction)? The measured rule of the Python op surface
// (MCConvModule._rows_shape): every list of <= 500 k edges; forward of larger lists unless the layer is narrow and the
// gathered rows miss the L2s; backward of larger lists only for wide layers with long rows.
bool rows_shape(bool combin, int fin, const void* feats, int rows, int n_points, int e, bool backward) {
    if (!row_kernels_on() || read_mask() != 0 || combin || fin % 8 != 0 || e <= 0 || rows <= 0 || (((uintptr_t)feats) & 15)) return false;
    static const int force = debug_int("rows_force", 0);  // A/B: 1 = rows wherever they apply
    if (force == 1) return true;
    if (e <= 500000) return true;
    if (!backward) return !(fin <= 128 && n_points >= 65536);
    return fin >= 256 && (float)e / (float)rows >= rows_min_degree();
}

std::atomic<long long> g_wait_ns{0};  // host time spent waiting for edge totals (mccnn_debug_wait_ns)

// the edge total: stored by the prefix sum of the count pass straight into the caller's pinned word
thread_local bool t_count_waits = true;  // (helper threads that wait instead of the caller's thread switch it off)

int wait_edges(mccnn_geometry* g, int spin_us) {
    if (g->e >= 0) return g->e;
    if (!g->total_host) return -1;
    int v = *g->total_host;
    if (v < 0 && spin_us != 0) {
        const auto t0 = std::chrono::steady_clock::now();
        struct Acc {
            std::chrono::steady_clock::time_point t;
            ~Acc() { if (t_count_waits) g_wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(), std::memory_order_relaxed); }
        } acc{t0};
        for (;;) {
            for (int k = 0; k < 256 && v < 0; ++k) v = *g->total_host;
            if (v >= 0) break;
            const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (spin_us > 0 && us > spin_us) break;
            if (us > 200) std::this_thread::yield();
        }
    }
    if (v >= 0) g->e = v;
    return v;
}

const mccnn_geometry* grid_owner(const mccnn_geometry* g) { return g->grid_of ? g->grid_of : g; }

struct Shape {
    int fin, fout, combin, bf16, outF, nb;
    size_t elem;      // bytes per feature element
    int words;        // 32-bit words per feature row
};
bool make_shape(Shape& s, int fin, int fout, int combin, int bf16) {
    if (fin <= 0 || fout <= 0) return false;
    s.fin = fin; s.fout = fout; s.combin = combin ? 1 : 0; s.bf16 = bf16 ? 1 : 0;
    s.outF = combin ? fout : fin;
    const long long neurons = combin ? (long long)fin * fout : fin;
    s.nb = (int)((neurons + 7) / 8);
    s.elem = bf16 ? 2 : 4;
    if (bf16 && (combin || fin % 8 != 0)) return false;
    s.words = bf16 ? fin / 2 : fin;
    return true;
}

// what the forward pass of a layer does with the feature rows
struct FwdMode {
    bool rows_fwd, rows_bwd, unsorted;
};
FwdMode fwd_mode(const mccnn_geometry* g, const Shape& s, const void* feats, int e) {
    FwdMode f;
    // (the sorted copy of the rows lives in a 256-byte aligned buffer of this layer's own: only the unsorted mode reads
    // the caller's rows in place and depends on their alignment)
    f.rows_fwd = rows_shape(s.combin, s.fin, nullptr, g->m, g->n, e, false);
    f.rows_bwd = rows_shape(s.combin, s.fin, nullptr, g->n, g->n, e, true);
    // small levels whose layer takes the row kernels in both directions: the rows are read where they lie (feat_index)
    f.unsorted = g->n <= unsorted_max_points() && e > 0 && g->m > 0 && f.rows_fwd && f.rows_bwd && ((((uintptr_t)feats) & 15) == 0);
    return f;
}

int plan_prepare(mccnn_geometry* g, int tr, int e) {
    Plan& p = g->plan[tr];
    const int rows = tr ? g->n : g->m;
    return mccnn_rowplan_buffer(rows, e, p.off, &p.total, &p.S, &p.slots, &p.scratch_rows);
}

size_t tlist_bytes(int n, int e) { return al((size_t)(n + 1) * 4) + al((size_t)(e > 0 ? e : 1) * 4); }
int* tl_start(const mccnn_geometry* g) { return reinterpret_cast<int*>(g->tl_buf); }
int* tl_perm(const mccnn_geometry* g) { return reinterpret_cast<int*>(g->tl_buf + al((size_t)(g->n + 1) * 4)); }

int ensure_tlist(mccnn_geometry* g, int e, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (g->tl_built) return 0;
    if (!g->tl_buf || g->tl_bytes < tlist_bytes(g->n, e)) return MCCNN_E_WORKSPACE;
    int rc = mccnn_transpose_neighbors(g->packed, e, g->n, tl_start(g), tl_perm(g), ws, ws_bytes, stream);
    if (rc) return rc;
    g->tl_built = true;
    return 0;
}

int ensure_plan(mccnn_geometry* g, int tr, int e, int avg, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    Plan& p = g->plan[tr];
    if (p.built && p.avg == avg) return 0;
    int rc = plan_prepare(g, tr, e);
    if (rc) return rc;
    if (!p.buf || p.bytes < (size_t)p.total) return MCCNN_E_WORKSPACE;
    const int rows = tr ? g->n : g->m;
    const bool inl = mccnn_rowplan_inline_records(rows, e) != 0;
    int rec_ready = 1;
    void* rec = nullptr;
    if (!inl) {
        if (!g->rec_buf || g->rec_bytes < (size_t)e * 16) return MCCNN_E_WORKSPACE;
        rec = g->rec_buf;
        rec_ready = g->rec_avg == avg ? 1 : 0;
    }
    int tready = 1;
    int *st = nullptr, *pt = nullptr;
    if (tr) {
        if (!g->tl_buf || g->tl_bytes < tlist_bytes(g->n, e)) return MCCNN_E_WORKSPACE;
        st = tl_start(g);
        pt = tl_perm(g);
        tready = g->tl_built ? 1 : 0;
    }
    const mccnn_geometry* go = grid_owner(g);
    // visiting order of the rows of a forward plan: the search's own centre order
    const int* order = tr ? nullptr : (g->same_level ? go->inv_idx : (g->order ? g->order : nullptr));
    rc = mccnn_rowplan_build(tr, go->s_pts, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn, g->mx, g->n, g->m, e,
                             g->B, g->radius, g->scale_inv, avg, order, rec, rec_ready, st, pt, tready, p.buf, ws, ws_bytes,
                             stream);
    if (rc) return rc;
    if (!inl) g->rec_avg = avg;
    if (tr) g->tl_built = true;
    p.built = true;
    p.avg = avg;
    return 0;
}

size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

}  // namespace

extern "C" {

mccnn_geometry_t* mccnn_geometry_create(void) { return new (std::nothrow) mccnn_geometry(); }
void mccnn_geometry_destroy(mccnn_geometry_t* g) { delete g; }

size_t mccnn_geometry_bytes(int n, int m, int batch_size, int num_cells, int e_capacity, int with_grid) {
    if (n < 0 || m < 0 || batch_size <= 0 || num_cells <= 0 || e_capacity < 0) return 0;
    if (cells_of(batch_size, num_cells) > 0x7fffffffULL) return 0;
    return geo_layout(n, m, batch_size, num_cells, e_capacity, with_grid != 0).total_bytes;
}

}  // extern "C"
namespace {
// Argument checks + the struct of a geometry over the caller's buffer (nothing is launched). `in_batch`: the grid owner may
// be a geometry set up earlier in the same batch (not built yet).
int geometry_setup(mccnn_geometry_t* g, const float* pts, const int* batch_ids, int n, const float* centres,
                   const int* centre_batch_ids, int m, const float* aabb_min, const float* aabb_max, int batch_size, int num_cells,
                   float radius, int scale_inv, float window, int use_pdf, int e_capacity, const mccnn_geometry_t* grid_from,
                   void* buffer, size_t buffer_bytes, int* total_host, bool in_batch) {
    if (!g || !pts || !batch_ids || !centres || !centre_batch_ids || !aabb_min || !aabb_max || !buffer || !total_host)
        return MCCNN_E_BADARG;
    if (n <= 0 || m <= 0 || batch_size <= 0 || num_cells <= 0 || !(radius > 0.f) || e_capacity <= 0) return MCCNN_E_BADARG;
    if (use_pdf && !(window > 0.f)) return MCCNN_E_BADARG;
    if (grid_from && (grid_from->n != n || grid_from->nc != num_cells || grid_from->B != batch_size ||
                      !(grid_owner(grid_from)->built || (in_batch && grid_owner(grid_from)->s_pts))))
        return MCCNN_E_BADARG;
    const bool with_grid = grid_from == nullptr;
    const GeoLayout L = geo_layout(n, m, batch_size, num_cells, e_capacity, with_grid);
    if (L.total_bytes == 0) return MCCNN_E_TOOLARGE;
    if (buffer_bytes < L.total_bytes || (((uintptr_t)buffer) & 255)) return MCCNN_E_WORKSPACE;
    *g = mccnn_geometry();
    g->pts = pts; g->bids = batch_ids; g->centres = centres; g->cbids = centre_batch_ids; g->mn = aabb_min; g->mx = aabb_max;
    g->n = n; g->m = m; g->B = batch_size; g->nc = num_cells; g->scale_inv = scale_inv ? 1 : 0; g->use_pdf = use_pdf ? 1 : 0;
    g->radius = radius; g->window = window;
    g->same_level = (centres == pts && m == n) ? 1 : 0;
    g->buf = (char*)buffer; g->bytes = buffer_bytes;
    char* b = g->buf;
    if (with_grid) {
        g->s_pts = (float*)(b + L.s_pts); g->s_bids = (int*)(b + L.s_bids); g->cells = (int*)(b + L.cells);
        g->new_idx = (int*)(b + L.new_idx); g->inv_idx = (int*)(b + L.inv_idx);
    } else {
        g->grid_of = grid_owner(grid_from);
    }
    g->start = (int*)(b + L.start); g->packed = (int*)(b + L.packed); g->pdfs = (float*)(b + L.pdfs);
    g->total_dev = (int*)(b + L.total);
    g->order_buf = (int*)(b + L.order);
    g->ws = b + L.ws; g->ws_bytes = L.ws_bytes;
    g->e_cap = e_capacity; g->e = -1; g->total_host = total_host;
    *total_host = -1;  // armed: the prefix sum of the count pass overwrites it
    return 0;
}

// the two workspaces that are alive side by side at the head of a geometry's chain: grid build | visiting order
struct HeadWs { char* gws; size_t gwb; char* ows; size_t owb; bool with_grid, own_order; };
int head_ws(const mccnn_geometry_t* g, HeadWs& h) {
    h.with_grid = g->grid_of == nullptr;
    // visiting order of the centres (speed only): the grid's own order when the centres are the gridded points; for the
    // points of another level (pooling / up-sampling: Poisson samples arrive phase by phase, all over the scene) a
    // cell-coherent order of their own in THIS grid -- small lists are searched in ~10 us either way
    h.own_order = !g->same_level && g->m >= MCCNN_ORDER_MIN_M;
    Arena ha(g->ws, g->ws_bytes);
    h.gws = h.ows = nullptr;
    h.gwb = h.with_grid ? al(mccnn_build_grid_workspace_bytes(g->n, g->B, g->nc)) : 0;
    h.owb = h.own_order ? al(visiting_order_workspace_bytes(g->m, g->B, g->nc)) : 0;
    if (h.with_grid && !(h.gws = ha.take<char>(h.gwb))) return MCCNN_E_WORKSPACE;
    if (h.own_order && !(h.ows = ha.take<char>(h.owb))) return MCCNN_E_WORKSPACE;
    return 0;
}

// the chain of ONE geometry: [head clear] grid build (4 launches) [visiting order (3)] count (scan) fill KDE
int geometry_issue_single(mccnn_geometry_t* g, mccnn_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const int n = g->n, m = g->m, batch_size = g->B, num_cells = g->nc;
    HeadWs h;
    int rc = head_ws(g, h);
    if (rc) return rc;
    // ONE clear at the head of the chain (histogram counters + scan status words of both counting sorts); everything
    // later in the chain is cleared by a kernel of the chain
    rc = launch_clear_spans(h.with_grid ? grid_head_span(n, batch_size, num_cells, h.gws, h.gwb) : no_span(),
                            h.own_order ? visiting_order_head_span(m, batch_size, num_cells, h.ows, h.owb) : no_span(), no_span(), s);
    if (rc) return rc;
    if (h.with_grid) {
        rc = build_grid_fused(g->pts, g->bids, g->mn, g->mx, n, batch_size, num_cells, g->new_idx, g->s_pts, g->s_bids, g->cells,
                              g->inv_idx, h.gws, h.gwb, s, nullptr, true, no_span(), no_span());
        if (rc) return rc;
    }
    const mccnn_geometry* go = grid_owner(g);
    const int* order = nullptr;
    if (g->same_level) {
        order = go->inv_idx;
    } else if (h.own_order) {
        g->order = g->order_buf;
        rc = visiting_order(g->centres, g->cbids, g->mn, g->mx, m, batch_size, num_cells, g->order, h.ows, h.owb, s, true);
        if (rc) return rc;
        order = g->order;
    }
    rc = find_neighbors_chain(g->centres, g->cbids, m, go->s_pts, n, go->cells, g->mn, g->mx, batch_size, num_cells, g->radius,
                              g->scale_inv, order, g->start, g->e_cap, g->packed, g->total_dev, const_cast<int*>((volatile int*)g->total_host),
                              g->ws, g->ws_bytes, stream);
    if (rc) return rc;
    if (g->use_pdf) {
        rc = mccnn_compute_pdf_dn(go->s_pts, go->s_bids, g->start, m, g->packed, g->e_cap, g->total_dev, g->mn, g->mx, batch_size,
                                  g->window, g->radius, g->scale_inv, g->pdfs, g->ws, g->ws_bytes, stream);
        if (rc) return rc;
    } else {  // MCConvBuilder.py:388-390: a tensor of ones
        rc = launch_fill_words(g->pdfs, (size_t)g->e_cap, 0x3f800000u, s);
        if (rc) return rc;
    }
    g->built = true;
    return 0;
}

// ONE launch per kernel kind over a chunk of <= MCCNN_BATCH_MAX geometries (and <= MCCNN_BATCH_MAX counting sorts): head
// clear, keys + histogram, prefix sums of the cell counters, park, rank + move + cell tables, count pass, prefix sums of
// the counts, fill pass, KDE -- nine launches whatever the number of geometries (a step of BASELINE cfg4 has fourteen).
int geometry_issue_chunk(mccnn_geometry_t* const* gs, int count, mccnn_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GridBatch gridB;
    ScanBatch scanG, scanN;
    NeighBatch neighB;
    PdfBatch pdfB;
    SpanBatch spans;
    spans.count = 0;
    int nGrid = 0, nPdf = 0;
    int rc;
    for (int k = 0; k < count; ++k) {
        mccnn_geometry_t* g = gs[k];
        HeadWs h;
        rc = head_ws(g, h);
        if (rc) return rc;
        if (h.with_grid) {
            ClearSpan head;
            rc = grid_batch_item(gridB.it[nGrid], scanG.it[nGrid], head, g->pts, g->bids, g->mn, g->mx, g->n, g->B, g->nc, g->new_idx,
                                 g->s_pts, g->s_bids, g->cells, g->inv_idx, h.gws, h.gwb);
            if (rc) return rc;
            spans.sp[spans.count++] = head;
            ++nGrid;
        }
        if (h.own_order) {
            ClearSpan head;
            g->order = g->order_buf;
            rc = order_batch_item(gridB.it[nGrid], scanG.it[nGrid], head, g->centres, g->cbids, g->mn, g->mx, g->m, g->B, g->nc, g->order,
                                  h.ows, h.owb);
            if (rc) return rc;
            spans.sp[spans.count++] = head;
            ++nGrid;
        }
    }
    rc = launch_clear_batch(spans, s);
    if (rc) return rc;
    if (nGrid) {
        if ((rc = launch_grid_batch_phase(gridB, nGrid, 0, s))) return rc;
        if ((rc = launch_scan_batch(scanG, nGrid, s))) return rc;
        if ((rc = launch_grid_batch_phase(gridB, nGrid, 1, s))) return rc;
        if ((rc = launch_grid_batch_phase(gridB, nGrid, 2, s))) return rc;
    }
    for (int k = 0; k < count; ++k) {
        mccnn_geometry_t* g = gs[k];
        const mccnn_geometry* go = grid_owner(g);
        const int* order = g->same_level ? go->inv_idx : g->order;
        rc = neigh_batch_item(neighB.it[k], scanN.it[k], g->centres, g->cbids, g->m, go->s_pts, g->n, go->cells, g->mn, g->mx, g->B, g->nc,
                              g->radius, g->scale_inv, order, g->start, g->e_cap, g->packed, g->total_dev,
                              const_cast<int*>((volatile int*)g->total_host), g->ws, g->ws_bytes);
        if (rc) return rc;
        if (g->use_pdf)
            pdf_batch_item(pdfB.it[nPdf++], go->s_pts, go->s_bids, g->start, g->m, g->packed, g->e_cap, g->total_dev, g->mn, g->mx, g->B,
                           g->window, g->radius, g->scale_inv, g->pdfs);
    }
    if ((rc = launch_neigh_batch(neighB, count, 0, s))) return rc;
    if ((rc = launch_scan_batch(scanN, count, s))) return rc;
    if ((rc = launch_neigh_batch(neighB, count, 1, s))) return rc;
    if (nPdf && (rc = launch_pdf_batch(pdfB, nPdf, s))) return rc;
    for (int k = 0; k < count; ++k) {
        mccnn_geometry_t* g = gs[k];
        if (!g->use_pdf) {  // MCConvBuilder.py:388-390: a tensor of ones
            rc = launch_fill_words(g->pdfs, (size_t)g->e_cap, 0x3f800000u, s);
            if (rc) return rc;
        }
        g->built = true;
    }
    return 0;
}
}  // namespace
extern "C" {

int mccnn_geometry_build(mccnn_geometry_t* g, const float* pts, const int* batch_ids, int n, const float* centres,
                         const int* centre_batch_ids, int m, const float* aabb_min, const float* aabb_max, int batch_size,
                         int num_cells, float radius, int scale_inv, float window, int use_pdf, int e_capacity,
                         const mccnn_geometry_t* grid_from, void* buffer, size_t buffer_bytes, int* total_host,
                         mccnn_stream_t stream) {
    int rc = geometry_setup(g, pts, batch_ids, n, centres, centre_batch_ids, m, aabb_min, aabb_max, batch_size, num_cells, radius,
                            scale_inv, window, use_pdf, e_capacity, grid_from, buffer, buffer_bytes, total_host, false);
    if (rc) return rc;
    return geometry_issue_single(g, stream);
}

// Several geometries of a step at once (a grid owner BEFORE the geometries that share its grid): one launch per kernel
// kind over chunks of the requests; a geometry too large for the batch form (> 2 M cells or centres) takes its own chain.
int mccnn_geometry_build_batch(const mccnn_geometry_request* req, int count, mccnn_stream_t stream) {
    if (!req || count < 0) return MCCNN_E_BADARG;
    for (int k = 0; k < count; ++k) {
        const mccnn_geometry_request& r = req[k];
        int rc = geometry_setup(r.geometry, r.pts, r.batch_ids, r.n, r.centres, r.centre_batch_ids, r.m, r.aabb_min, r.aabb_max,
                                r.batch_size, r.num_cells, r.radius, r.scale_inv, r.window, r.use_pdf, r.e_capacity, r.grid_from,
                                r.buffer, r.buffer_bytes, r.total_host, true);
        if (rc) return rc;
    }
    mccnn_geometry_t* chunk[MCCNN_BATCH_MAX];
    int nc = 0, sorts = 0;
    auto flush = [&]() -> int {
        int rc = nc ? geometry_issue_chunk(chunk, nc, stream) : 0;
        nc = 0; sorts = 0;
        return rc;
    };
    for (int k = 0; k < count; ++k) {
        mccnn_geometry_t* g = req[k].geometry;
        const bool with_grid = g->grid_of == nullptr, own_order = !g->same_level && g->m >= MCCNN_ORDER_MIN_M;
        const bool ok = grid_batch_eligible(with_grid ? g->n : 1, g->B, g->nc) && (!own_order || grid_batch_eligible(g->m, g->B, g->nc)) &&
                        neigh_batch_eligible(g->m, g->n);
        if (!ok) {   // its own chain -- behind the chunk in flight (it may share a grid set up there), and before what follows
            int rc = flush();
            if (!rc) rc = geometry_issue_single(g, stream);
            if (rc) return rc;
            continue;
        }
        const int need = (with_grid ? 1 : 0) + (own_order ? 1 : 0);
        if (nc == MCCNN_BATCH_MAX || sorts + need > MCCNN_BATCH_MAX) {
            int rc = flush();
            if (rc) return rc;
        }
        chunk[nc++] = g;
        sorts += need;
    }
    return flush();
}

long long mccnn_debug_wait_ns(void) { return g_wait_ns.load(std::memory_order_relaxed); }

int mccnn_debug_wait_accounting(int on) {
    const int prev = t_count_waits ? 1 : 0;
    t_count_waits = on != 0;
    return prev;
}

int mccnn_geometry_edges(mccnn_geometry_t* g, int wait_us) {
    if (!g || !g->built) return MCCNN_E_BADARG;
    const int e = wait_edges(g, wait_us);
    return e < 0 ? -1 : e;
}

int mccnn_geometry_info(const mccnn_geometry_t* g, long long out[16]) {
    if (!g || !g->built || !out) return MCCNN_E_BADARG;
    const mccnn_geometry* go = grid_owner(g);
    out[0] = (long long)(uintptr_t)go->s_pts; out[1] = (long long)(uintptr_t)go->s_bids; out[2] = (long long)(uintptr_t)go->cells;
    out[3] = (long long)(uintptr_t)go->new_idx; out[4] = (long long)(uintptr_t)go->inv_idx;
    out[5] = (long long)(uintptr_t)g->start; out[6] = (long long)(uintptr_t)g->packed; out[7] = (long long)(uintptr_t)g->pdfs;
    out[8] = g->n; out[9] = g->m; out[10] = g->nc; out[11] = g->e_cap; out[12] = g->e; out[13] = g->B;
    out[14] = (long long)(uintptr_t)g->total_dev; out[15] = g->grid_of ? 0 : 1;
    return 0;
}

int mccnn_geometry_attach(mccnn_geometry_t* g, int what, void* buffer, size_t bytes) {
    if (!g || !g->built || !buffer || (((uintptr_t)buffer) & 255)) return MCCNN_E_BADARG;
    switch (what) {
        case NEED_PLAN_FWD: g->plan[0] = Plan(); g->plan[0].buf = (char*)buffer; g->plan[0].bytes = bytes; return 0;
        case NEED_PLAN_TR: g->plan[1] = Plan(); g->plan[1].buf = (char*)buffer; g->plan[1].bytes = bytes; return 0;
        case NEED_TLIST: g->tl_buf = (char*)buffer; g->tl_bytes = bytes; g->tl_built = false; return 0;
        case NEED_RECORDS: g->rec_buf = (char*)buffer; g->rec_bytes = bytes; g->rec_avg = -1; return 0;
        default: return MCCNN_E_BADARG;
    }
}

// Pieces built AHEAD of the layers that need them (ConvolutionBuilder.prefetch_geometry(transposed=True): the transposed
// list and the transposed row plan of a depth-wise layer's backward pass on a side stream, under the forward passes).
}  // extern "C"
namespace {
size_t piece_bytes(mccnn_geometry* g, int what, int e) {
    switch (what) {
        case NEED_PLAN_FWD: return plan_prepare(g, 0, e) ? 0 : (size_t)g->plan[0].total;
        case NEED_PLAN_TR: return plan_prepare(g, 1, e) ? 0 : (size_t)g->plan[1].total;
        case NEED_TLIST: return tlist_bytes(g->n, e);
        case NEED_RECORDS: return al((size_t)e * 16);
        default: return 0;
    }
}
}  // namespace
extern "C" {

int mccnn_geometry_piece_bytes(mccnn_geometry_t* g, int what, long long* bytes, long long* ws_bytes) {
    if (!g || !g->built || !bytes || !ws_bytes) return MCCNN_E_BADARG;
    const int e = wait_edges(g, -1);
    if (e < 0) return MCCNN_E_BADARG;
    if (e > g->e_cap) return MCCNN_E_CAPACITY;
    *bytes = (long long)piece_bytes(g, what, e);
    size_t w = 256;
    if (what == NEED_TLIST) w = mccnn_transpose_neighbors_workspace_bytes(g->n, e);
    if (what == NEED_PLAN_FWD) w = mccnn_rowplan_build_workspace_bytes(g->m, e, 0);
    if (what == NEED_PLAN_TR) w = mccnn_rowplan_build_workspace_bytes(g->n, e, 1);
    *ws_bytes = (long long)(al(w) + 512);
    return *bytes > 0 ? 0 : MCCNN_E_BADARG;
}

// The same sizes for a geometry that is not built yet (n points, m centres, a list of at most e_cap edges): what a caller
// allocates when it asks for pieces before the edge total exists.
int mccnn_geometry_piece_bound(int n, int m, int e_cap, int what, long long* bytes, long long* ws_bytes) {
    if (n < 0 || m < 0 || e_cap < 0 || !bytes || !ws_bytes) return MCCNN_E_BADARG;
    long long b = 0, w = 256;
    int rc = 0;
    switch (what) {
        case NEED_PLAN_FWD: rc = mccnn_rowplan_bound(m, e_cap, 0, &b, &w); break;
        case NEED_PLAN_TR: rc = mccnn_rowplan_bound(n, e_cap, 1, &b, &w); break;
        case NEED_TLIST: b = (long long)tlist_bytes(n, e_cap); w = (long long)mccnn_transpose_neighbors_workspace_bytes(n, e_cap); break;
        case NEED_RECORDS: b = (long long)al((size_t)(e_cap > 0 ? e_cap : 1) * 16); break;
        default: return MCCNN_E_BADARG;
    }
    if (rc) return rc;
    *bytes = (long long)al((size_t)b);
    *ws_bytes = (long long)(al((size_t)w) + 512);
    return 0;
}

int mccnn_geometry_prebuild(mccnn_geometry_t* g, int what, int avg, void* ws, size_t ws_bytes, mccnn_stream_t stream) {
    if (!g || !g->built || !ws) return MCCNN_E_BADARG;
    const int e = wait_edges(g, -1);
    if (e < 0) return MCCNN_E_BADARG;
    if (e > g->e_cap) return MCCNN_E_CAPACITY;
    if (e == 0) return 0;
    avg = avg ? 1 : 0;
    int rc = 0;
    // (the transposed plan of a large list builds the transposed list itself when there is none yet -- its rank pass
    // writes the plan on the way, conv_rows.hip plan_scatter_tr -- so the plan goes first and the list call that
    // follows finds its work done; the forward plan evaluates the edge records both plans permute)
    if (what & NEED_PLAN_FWD) rc = ensure_plan(g, 0, e, avg, ws, ws_bytes, stream);
    if (!rc && (what & NEED_PLAN_TR)) rc = ensure_plan(g, 1, e, avg, ws, ws_bytes, stream);
    if (!rc && (what & NEED_TLIST)) rc = ensure_tlist(g, e, ws, ws_bytes, stream);
    return rc;
}

// The pieces of SEVERAL geometries at once (the row plans / transposed lists a step's layers will ask for, built ahead), one
// launch per kernel KIND per flush:
//   head clear (status words, row counters of every chain of the flush)
//   transposition chains of lists too long for one workgroup: tr_count, scan, tr_fill, tr_rank (a large transposed plan's
//     chain is ranked by its scatter instead)                                  | tr_small: single-workgroup transpositions
//   small plans (single-workgroup layout, records evaluated inline): plan_small, sell_fill forward, sell_fill transposed
//   large plans: vr_count, vr_scan_expand, sell_sort, plan_bases (transposed), edge_records (transposed plans whose forward
//     plan did not leave the records), plan_fill_tiles (forward; evaluates the records), plan_scatter_tr (transposed)
// A flush holds <= MCCNN_PLAN_BATCH_MAX small plans / chains and <= MCCNN_LARGE_BATCH_MAX large plans (the items travel in
// the kernel arguments); what no flush can take (a small layout over > 262 144 edges, a list beyond the chained scan, a
// geometry without room for a piece) builds as mccnn_geometry_prebuild does, behind the flushes, on the shared region at the
// head of `ws`. Launch errors aside, the flags of a geometry (plan built, list built, records evaluated) are set when its
// items are QUEUED: later items of the same call rely on them (a transposed plan on the list queued before it).
// ws: mccnn_geometry_prebuild_batch_ws_bytes. Waits for the edge totals.
size_t mccnn_geometry_prebuild_batch_ws_bytes(mccnn_geometry_t* const* geoms, const int* what, int count) {
    if (!geoms || !what || count < 0) return 0;
    size_t single = 256, trs = 0;
    for (int k = 0; k < count; ++k) {
        mccnn_geometry_t* g = geoms[k];
        if (!g || !g->built) return 0;
        const int e = wait_edges(g, -1);
        if (e <= 0 || e > g->e_cap) continue;
        for (int bit = 1; bit <= 4; bit <<= 1) {
            if (!(what[k] & bit)) continue;
            long long b = 0, w = 0;
            if (mccnn_geometry_piece_bytes(g, bit, &b, &w) == 0 && (size_t)w > single) single = (size_t)w;
        }
        if ((what[k] & 6) && !g->tl_built)
            trs += transpose_small(e, g->n) ? al(plan_batch_tr_ws_bytes(g->n, e)) : al(mccnn_transpose_neighbors_workspace_bytes(g->n, e));
        for (int tr = 0; tr < 2; ++tr)
            if ((what[k] & (tr ? 2 : 1)) && plan_large_batchable(tr ? g->n : g->m, e, g->n, tr, 0))
                trs += al(plan_large_ws_bytes(tr ? g->n : g->m, e, g->n, tr, 0));
    }
    return al(single) + trs + 512;
}

int mccnn_geometry_prebuild_batch(mccnn_geometry_t* const* geoms, const int* what, int count, int avg, void* ws, size_t ws_bytes,
                                  mccnn_stream_t stream) {
    if (!geoms || !what || count < 0 || !ws) return MCCNN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    avg = avg ? 1 : 0;
    Arena ar(ws, ws_bytes);
    // the region the individual builds share (one after the other, stream-ordered) comes first
    size_t single = 256;
    for (int k = 0; k < count; ++k) {
        mccnn_geometry_t* g = geoms[k];
        if (!g || !g->built) return MCCNN_E_BADARG;
        const int e = wait_edges(g, -1);
        if (e < 0) return MCCNN_E_BADARG;
        if (e == 0 || e > g->e_cap) continue;
        for (int bit = 1; bit <= 4; bit <<= 1) {
            if (!(what[k] & bit)) continue;
            long long b = 0, w = 0;
            if (mccnn_geometry_piece_bytes(g, bit, &b, &w) == 0 && (size_t)w > single) single = (size_t)w;
        }
    }
    char* sws = ar.take<char>(al(single));
    if (!sws) return MCCNN_E_WORKSPACE;
    TrSmallBatch trB;
    TrChainBatch chB;
    ScanBatch chScan;
    SpanBatch chSpans;
    chSpans.count = 0;
    PlanSmallBatch layB;
    SellFillBatch fwdB, trfB;
    LargeBatch lgB;
    int nTr = 0, nCh = 0, nLay = 0, nFwd = 0, nTrf = 0, nLg = 0;
    static const int dbgSync = debug_int("plan_batch_sync", 0);   // debugging: a synchronisation and a line per launch
    auto mark = [&](const char* what, int k) {
        if (!dbgSync) return;
        const hipError_t e = hipStreamSynchronize(s);
        fprintf(stderr, "mccnn batch: %s %d -> %d\n", what, k, (int)e);
    };
    auto flush = [&]() -> int {
        int rc = 0;
        if (dbgSync) fprintf(stderr, "mccnn batch: flush nCh %d nTr %d nLay %d nFwd %d nTrf %d nLg %d spans %d\n", nCh, nTr, nLay, nFwd, nTrf, nLg, chSpans.count);
        if (chSpans.count) rc = launch_clear_batch(chSpans, s);   // the head of the batch: status words and row counters of every chain in it
        mark("clear", 0);
        if (!rc && nCh) {   // transpositions too long for one workgroup: count, prefix sums, fill, rank -- one launch each
            rc = launch_tr_chain_batch(chB, nCh, 0, s);
            mark("tr_count", 0);
            if (!rc) rc = launch_scan_batch(chScan, nCh, s);
            mark("tr_scan", 0);
            if (!rc) rc = launch_tr_chain_batch(chB, nCh, 1, s);
            mark("tr_fill", 0);
            if (!rc) rc = launch_tr_chain_batch(chB, nCh, 2, s);
            mark("tr_rank", 0);
        }
        if (!rc) rc = launch_tr_small_batch(trB, nTr, s);
        mark("tr_small", 0);
        if (!rc) rc = launch_plan_small_batch(layB, nLay, s);
        mark("plan_small", 0);
        if (!rc) rc = launch_sell_fill_batch(fwdB, nFwd, 0, s);
        if (!rc) rc = launch_sell_fill_batch(trfB, nTrf, 1, s);
        mark("sell_fill", 0);
        // the large plans: layout, then tile fill (forward) / bases + records + scatter (transposed)
        for (int ph = LARGE_VR_COUNT; ph <= LARGE_SCATTER && !rc && nLg; ++ph) {
            rc = launch_plan_large_batch(lgB, nLg, ph, s);
            mark("large", ph);
        }
        nTr = nCh = nLay = nFwd = nTrf = nLg = 0;
        chSpans.count = 0;
        return rc;
    };
    // the transposed list of g, by the batch: single-workgroup form or chain item; false = not possible here
    auto batch_tlist = [&](mccnn_geometry_t* g, int e) -> int {
        if (transpose_small(e, g->n)) {
            char* w = ar.take<char>(al(plan_batch_tr_ws_bytes(g->n, e)));
            if (!w) return MCCNN_E_WORKSPACE;
            trB.it[nTr++] = TrSmallItem{reinterpret_cast<const int2*>(g->packed), reinterpret_cast<int*>(w),
                                        reinterpret_cast<int*>(w + al((size_t)g->n * 4)),
                                        reinterpret_cast<int*>(w + al((size_t)g->n * 4) + al((size_t)e * 4)), tl_start(g), tl_perm(g), e, g->n};
            return 0;
        }
        const size_t wb = al(mccnn_transpose_neighbors_workspace_bytes(g->n, e));
        char* w = ar.take<char>(wb);
        if (!w) return MCCNN_E_WORKSPACE;
        ClearSpan head;
        int rc = tr_chain_item(chB.it[nCh], chScan.it[nCh], head, g->packed, e, g->n, tl_start(g), tl_perm(g), w, wb);
        if (rc) return rc;
        chSpans.sp[chSpans.count++] = head;
        ++nCh;
        return 0;
    };
    struct Later { mccnn_geometry_t* g; int what; };
    Later later[256];
    int nLater = 0;
    for (int k = 0; k < count; ++k) {
        mccnn_geometry_t* g = geoms[k];
        const int e = g->e;
        if (e <= 0 || e > g->e_cap) continue;   // (an overflowing list is rebuilt by its first layer: nothing to build ahead)
        const mccnn_geometry* go = grid_owner(g);
        int rest = what[k] & 7;
        if (nLay + 2 > MCCNN_PLAN_BATCH_MAX || nTr + 1 > MCCNN_PLAN_BATCH_MAX || nCh + 1 > MCCNN_PLAN_BATCH_MAX) {
            int rc = flush();
            if (rc) return rc;
        }
        const bool tl_room = g->tl_buf && g->tl_bytes >= tlist_bytes(g->n, e) && (long long)g->n <= 2048LL * 1024;
        // the transposed list: on its own (a combin layer's deterministic feature gradient) or under a small transposed plan
        const bool tr_plan_small = (rest & 2) && plan_batchable(g->n, e, 1) && !(g->plan[1].built && g->plan[1].avg == avg);
        if (((rest & 4) || tr_plan_small) && !g->tl_built && tl_room && (!(rest & 2) || tr_plan_small)) {
            int rc = batch_tlist(g, e);
            if (rc) return rc;
            g->tl_built = true;
        }
        if (g->tl_built) rest &= ~4;
        for (int tr = 0; tr < 2; ++tr) {
            const int bit = tr ? 2 : 1;
            if (!(rest & bit)) continue;
            Plan& p = g->plan[tr];
            if (p.built && p.avg == avg) { rest &= ~bit; continue; }
            const int rows = tr ? g->n : g->m;
            if (!plan_batchable(rows, e, tr)) continue;
            if (plan_prepare(g, tr, e)) continue;
            if (!p.buf || p.bytes < (size_t)p.total) continue;
            if (tr && (!g->tl_buf || g->tl_bytes < tlist_bytes(g->n, e))) continue;
            if (tr && !g->tl_built) continue;   // (no room for the list: the individual chain below reports it)
            const int* order = tr ? nullptr : (g->same_level ? go->inv_idx : (g->order ? g->order : nullptr));
            bool use_tr = false;
            SellFillItem& fi = tr ? trfB.it[nTrf] : fwdB.it[nFwd];
            int rc = plan_batch_items(tr, go->s_pts, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn, g->mx, g->n, g->m, e, g->B,
                                      g->radius, g->scale_inv, avg, order, tr ? tl_start(g) : nullptr, tr ? tl_perm(g) : nullptr,
                                      1 /* the list: transposed above, ahead of every layout of this flush */, p.buf, nullptr, 0, nullptr,
                                      &use_tr, layB.it[nLay], fi);
            if (rc) continue;   // (left to the individual chain below)
            ++nLay;
            if (tr) ++nTrf; else ++nFwd;
            if (tr) g->tl_built = true;
            p.built = true;
            p.avg = avg;
            rest &= ~bit;
            if (tr) rest &= ~4;
        }
        // the large plans of the batch (records in the geometry's own array: evaluated by the forward plan's fill, or by a
        // pass of the batch when the transposed plan comes alone)
        for (int tr = 0; tr < 2; ++tr) {
            const int bit = tr ? 2 : 1;
            if (!(rest & bit)) continue;
            Plan& p = g->plan[tr];
            const int rows = tr ? g->n : g->m;
            static const int largeOn = debug_int("plan_large_batch", 1);   // A/B: 0 = large plans take their own chains
            if (!largeOn) continue;
            if (!plan_large_batchable(rows, e, g->n, tr, g->tl_built ? 1 : 0)) continue;
            if (plan_prepare(g, tr, e)) continue;
            if (!p.buf || p.bytes < (size_t)p.total) continue;
            if (!g->rec_buf || g->rec_bytes < (size_t)e * 16) continue;
            if (tr && (!g->tl_buf || g->tl_bytes < tlist_bytes(g->n, e))) continue;
            if (nLg + 1 > MCCNN_LARGE_BATCH_MAX || nCh + 1 > MCCNN_PLAN_BATCH_MAX || chSpans.count + 3 > 3 * MCCNN_BATCH_MAX) {
                int rc = flush();
                if (rc) return rc;
            }
            const size_t wb = al(plan_large_ws_bytes(rows, e, g->n, tr, g->tl_built ? 1 : 0));
            char* w = ar.take<char>(wb);
            if (!w) continue;   // (left to the individual chain below, on the shared region)
            const int* order = tr ? nullptr : (g->same_level ? go->inv_idx : (g->order ? g->order : nullptr));
            bool use_chain = false;
            const int spans0 = chSpans.count;
            int rc = plan_large_item(tr, go->s_pts, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn, g->mx, g->n, g->m, e, g->B,
                                     g->radius, g->scale_inv, avg, order, g->rec_buf, g->rec_avg == avg ? 1 : 0, tr ? tl_start(g) : nullptr,
                                     tr ? tl_perm(g) : nullptr, g->tl_built ? 1 : 0, p.buf, w, wb, lgB.it[nLg], &chB.it[nCh], &chScan.it[nCh],
                                     &use_chain, chSpans);
            if (rc) { chSpans.count = spans0; continue; }
            ++nLg;
            if (use_chain) ++nCh;
            g->rec_avg = avg;
            if (tr) g->tl_built = true;
            p.built = true;
            p.avg = avg;
            rest &= ~bit;
            if (tr) rest &= ~4;
        }
        if (rest && nLater < 256) later[nLater++] = Later{g, rest};
    }
    int rc = flush();
    if (rc) return rc;
    for (int k = 0; k < nLater; ++k) {   // the large ones (and what did not fit): their own chains, one after the other
        mccnn_geometry_t* g = later[k].g;
        const int e = g->e, w = later[k].what;
        if (w & NEED_PLAN_FWD) rc = ensure_plan(g, 0, e, avg, sws, al(single), stream);
        if (!rc && (w & NEED_PLAN_TR)) rc = ensure_plan(g, 1, e, avg, sws, al(single), stream);
        if (!rc && (w & NEED_TLIST)) rc = ensure_tlist(g, e, sws, al(single), stream);
        if (rc) return rc;
    }
    return 0;
}

// What one mccnn_conv_forward / _backward call of this layer shape needs from the caller: which pieces the geometry does
// not hold yet (mask, need_bytes[k] for bit k), the scratch of the call, and (forward) what has to be kept for the
// backward pass.
struct Req {
    int mask;
    long long need_bytes[4];
    size_t ws, saved;
};
static int requirements(mccnn_geometry* g, const Shape& s, const void* feats, int backward, int flags, int e, Req& r) {
    const int n = g->n, m = g->m;
    const FwdMode f = fwd_mode(g, s, feats, e);
    int mask = 0;
    for (int k = 0; k < 4; ++k) r.need_bytes[k] = 0;
    size_t ws = 256, saved = 0;
    auto want_plan = [&](int tr) -> int {
        Plan& p = g->plan[tr];
        int rc = plan_prepare(g, tr, e);
        if (rc) return rc;
        if (!p.buf || p.bytes < (size_t)p.total) { mask |= tr ? NEED_PLAN_TR : NEED_PLAN_FWD; r.need_bytes[tr] = p.total; }
        const int rows = tr ? n : m;
        if (!mccnn_rowplan_inline_records(rows, e) && (!g->rec_buf || g->rec_bytes < (size_t)e * 16)) {
            mask |= NEED_RECORDS;
            r.need_bytes[3] = (long long)al((size_t)e * 16);
        }
        // (budgeted even when a plan exists: a layer with another `avg` flag rebuilds it -- ensure_plan -- and the flag is
        // not known here; the caller's scratch only ever grows, so this costs one allocation per process)
        const size_t b = mccnn_rowplan_build_workspace_bytes(rows, e, tr);
        if (b > ws) ws = b;
        return 0;
    };
    auto want_tlist = [&]() {
        if (!g->tl_buf || g->tl_bytes < tlist_bytes(n, e)) { mask |= NEED_TLIST; r.need_bytes[2] = (long long)tlist_bytes(n, e); }
        if (!g->tl_built) {
            const size_t b = mccnn_transpose_neighbors_workspace_bytes(n, e);
            if (b > ws) ws = b;
        }
    };
    if (!backward) {
        if (!f.unsorted) saved += al((size_t)n * s.fin * s.elem);  // the sorted feature rows
        if (f.rows_fwd) {
            int rc = want_plan(0);
            if (rc) return rc;
            const size_t scratch = al((size_t)g->plan[0].scratch_rows * s.outF * 4);
            // (the plan is built and its workspace released before the layer's own scratch is used: they share `ws`)
            ws = max3(ws, scratch, 256);
        } else if (!s.bf16) {
            const size_t w = mccnn_spatial_conv_fwd_workspace_bytes(m, e, s.fin, s.fout, s.combin);
            if (w > ws) ws = w;
            if (flags & 1) saved += al(mccnn_spatial_conv_state_bytes(m, e, s.fin, s.fout, s.combin));
        }
    } else {
        const bool rows_bwd = f.rows_bwd && m > 0;
        size_t fg_sorted = f.unsorted && rows_bwd ? 0 : al((size_t)n * s.fin * s.elem);  // the gradient rows in grid order
        size_t sort_again = 0;
        if (rows_bwd) {
            int rc = want_plan(1);
            if (rc) return rc;
            want_tlist();
            const size_t w = al(mccnn_spatial_conv_bwd_rows_workspace_bytes(n, e, s.fin)) + al((size_t)g->plan[1].scratch_rows * s.fin * 4);
            ws = max3(ws, w + fg_sorted, 256);
            // mccnn_conv_backward leaves the row kernels when the out-gradient it is handed is not 16-byte aligned (the
            // pointer is not known here): the scratch covers the streaming path of the same layer as well
            const size_t w2 = al(mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, s.fin, s.fout, s.combin));
            const size_t tb = e > 0 ? al(mccnn_transpose_neighbors_workspace_bytes(n, e)) : 0;
            const size_t full = al((size_t)n * s.fin * s.elem);
            ws = max3(ws, (w2 > tb ? w2 : tb) + full + (f.unsorted ? full : 0), 256);
        } else {
            if (f.unsorted) sort_again = al((size_t)n * s.fin * s.elem);  // (an out-gradient the row kernel cannot take)
            size_t tb = 0;
            if (e > 0 && (!s.combin || ((flags & 2) && s.fin >= 2 && s.fin <= 4))) {
                want_tlist();
                tb = al(mccnn_transpose_neighbors_workspace_bytes(n, e));
            }
            const size_t w = al(mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, s.fin, s.fout, s.combin));
            ws = max3(ws, (w > tb ? w : tb) + fg_sorted + sort_again, 256);
        }
    }
    r.mask = mask;
    r.ws = ws + 512;
    r.saved = saved;
    return 0;
}

// Waits for the edge total of the geometry (the one host wait of a convolution).
int mccnn_conv_prepare(mccnn_geometry_t* g, const void* feats, int num_in_feats, int num_out_feats, int combin, int bf16,
                       int backward, int flags, int* need_mask, long long need_bytes[4], long long* ws_bytes,
                       long long* saved_bytes, int* edges) {
    if (!g || !g->built || !need_mask || !need_bytes || !ws_bytes || !saved_bytes || !edges) return MCCNN_E_BADARG;
    Shape s;
    if (!make_shape(s, num_in_feats, num_out_feats, combin, bf16)) return MCCNN_E_SHAPE;
    const int e = wait_edges(g, -1);
    if (e < 0) return MCCNN_E_BADARG;
    *edges = e;
    if (e > g->e_cap) return MCCNN_E_CAPACITY;
    Req r;
    int rc = requirements(g, s, feats, backward, flags, e, r);
    if (rc) return rc;
    *need_mask = r.mask;
    for (int k = 0; k < 4; ++k) need_bytes[k] = r.need_bytes[k];
    *ws_bytes = (long long)r.ws;
    *saved_bytes = (long long)r.saved;
    return 0;
}

int mccnn_conv_forward(mccnn_geometry_t* g, const void* feats, int num_in_feats, int num_out_feats, int combin, int avg,
                       int bf16, int flags, const float* w1, const float* b1, const float* w2, const float* b2,
                       const float* w3, const float* b3, void* out, void* saved, size_t saved_bytes, void* ws,
                       size_t ws_bytes, mccnn_stream_t stream) {
    if (!g || !g->built || !feats || !out || !ws) return MCCNN_E_BADARG;
    Shape s;
    if (!make_shape(s, num_in_feats, num_out_feats, combin, bf16)) return MCCNN_E_SHAPE;
    const int e = wait_edges(g, -1);
    if (e < 0) return MCCNN_E_BADARG;
    if (e > g->e_cap) return MCCNN_E_CAPACITY;
    const int n = g->n, m = g->m;
    const mccnn_geometry* go = grid_owner(g);
    const FwdMode f = fwd_mode(g, s, feats, e);
    avg = avg ? 1 : 0;
    {   // nothing is launched unless everything this call needs is there (the caller may call optimistically, with the
        // sizes of the last batch, and only ask mccnn_conv_prepare when this says no)
        Req r;
        int rc = requirements(g, s, feats, 0, flags, e, r);
        if (rc) return rc;
        if (r.mask || ws_bytes < r.ws || (r.saved && (!saved || saved_bytes < r.saved))) return MCCNN_E_WORKSPACE;
    }
    // sort_features (MCConvModuleSrc:35-36): sorted[new_idx[i]] = feats[i]
    const void* rows_in = feats;
    char* sv = (char*)saved;
    size_t sv_off = 0;
    if (!f.unsorted) {
        const size_t b = al((size_t)n * s.fin * s.elem);
        if (!saved || saved_bytes < b) return MCCNN_E_WORKSPACE;
        int rc = mccnn_permute_scatter((const float*)feats, go->new_idx, n, s.words, (float*)sv, n, 0, stream);
        if (rc) return rc;
        rows_in = sv;
        sv_off = b;
    }
    if (f.rows_fwd) {
        int rc = ensure_plan(g, 0, e, avg, ws, ws_bytes, stream);
        if (rc) return rc;
        const Plan& p = g->plan[0];
        if (ws_bytes < (size_t)p.scratch_rows * s.outF * 4) return MCCNN_E_WORKSPACE;
        char* pb = p.buf;
        return mccnn_spatial_conv_fwd_rows(go->s_pts, rows_in, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn, g->mx,
                                           w1, b1, w2, b2, w3, b3, n, m, e, s.fin, g->B, g->radius, g->scale_inv, avg, s.bf16,
                                           (const int*)(pb + p.off[0]), (const int*)(pb + p.off[1]), (const int*)(pb + p.off[2]),
                                           (const int*)(pb + p.off[3]), pb + p.off[5], (const int*)(pb + p.off[4]), out,
                                           (float*)ws, f.unsorted ? go->inv_idx : nullptr, stream);
    }
    if (s.bf16)
        return mccnn_spatial_conv_fwd_bf16(go->s_pts, rows_in, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn, g->mx, w1,
                                           b1, w2, b2, w3, b3, n, m, e, s.fin, g->B, g->radius, g->scale_inv, avg, out, ws, ws_bytes,
                                           stream);
    void* state = nullptr;
    if (flags & 1) {
        const size_t sb = mccnn_spatial_conv_state_bytes(m, e, s.fin, s.fout, s.combin);
        if (sb) {
            if (!saved || saved_bytes < sv_off + sb) return MCCNN_E_WORKSPACE;
            state = sv + sv_off;
        }
    }
    return mccnn_spatial_conv_fwd(go->s_pts, (const float*)rows_in, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn,
                                  g->mx, w1, b1, w2, b2, w3, b3, n, m, e, s.fin, s.fout, s.combin, g->B, g->radius, g->scale_inv,
                                  avg, (float*)out, state, ws, ws_bytes, stream);
}

int mccnn_conv_backward(mccnn_geometry_t* g, const void* feats, const void* saved, size_t saved_bytes, const void* out_grad,
                        int num_in_feats, int num_out_feats, int combin, int avg, int bf16, int flags, const float* w1,
                        const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                        void* feat_grad, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, void* ws,
                        size_t ws_bytes, mccnn_stream_t stream) {
    if (!g || !g->built || !feats || !out_grad || !feat_grad || !ws) return MCCNN_E_BADARG;
    Shape s;
    if (!make_shape(s, num_in_feats, num_out_feats, combin, bf16)) return MCCNN_E_SHAPE;
    const int e = wait_edges(g, -1);
    if (e < 0) return MCCNN_E_BADARG;
    if (e > g->e_cap) return MCCNN_E_CAPACITY;
    const int n = g->n, m = g->m;
    const mccnn_geometry* go = grid_owner(g);
    const FwdMode f = fwd_mode(g, s, feats, e);
    avg = avg ? 1 : 0;
    {
        Req r;
        int rc = requirements(g, s, feats, 1, flags, e, r);
        if (rc) return rc;
        if (r.mask || ws_bytes < r.ws) return MCCNN_E_WORKSPACE;
    }
    const size_t rows_b = al((size_t)n * s.fin * s.elem);
    const char* sv = (const char*)saved;
    size_t sv_off = 0;
    const void* rows_in = feats;  // unsorted mode: the rows of the unsorted points
    if (!f.unsorted) {
        if (!saved || saved_bytes < rows_b) return MCCNN_E_WORKSPACE;
        rows_in = sv;
        sv_off = rows_b;
    }
    Arena a(ws, ws_bytes);
    const bool rows_bwd = f.rows_bwd && m > 0 && ((((uintptr_t)out_grad) & 15) == 0);
    int rc;
    if (rows_bwd) {
        // ONE sweep over the transposed row plan: feature gradient and the six parameter gradients
        rc = ensure_plan(g, 1, e, avg, ws, ws_bytes, stream);  // (its workspace is free again when the sweep starts)
        if (rc) return rc;
        const Plan& p = g->plan[1];
        void* fg = feat_grad;
        if (!f.unsorted) {
            fg = a.take<char>(rows_b);
            if (!fg) return MCCNN_E_WORKSPACE;
        }
        float* scratch = a.take<float>((size_t)p.scratch_rows * s.fin);
        const size_t wb = mccnn_spatial_conv_bwd_rows_workspace_bytes(n, e, s.fin);
        char* w = a.take<char>(wb);
        if (!scratch || !w) return MCCNN_E_WORKSPACE;
        char* pb = p.buf;
        rc = mccnn_spatial_conv_bwd_rows(go->s_pts, rows_in, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn, g->mx, w1,
                                         b1, w2, b2, w3, b3, out_grad, n, m, e, s.fin, g->B, g->radius, g->scale_inv, avg, s.bf16,
                                         tl_start(g), (const int*)(pb + p.off[0]), (const int*)(pb + p.off[1]),
                                         (const int*)(pb + p.off[2]), (const int*)(pb + p.off[3]), pb + p.off[5],
                                         (const int*)(pb + p.off[4]), fg, scratch, dw1, db1, dw2, db2, dw3, db3,
                                         f.unsorted ? go->inv_idx : nullptr, w, wb, stream);
        if (rc) return rc;
        if (!f.unsorted)  // back into the order the features arrived in: fg_in[i] = fg_sorted[new_idx[i]]
            return mccnn_permute_gather((const float*)fg, go->new_idx, n, s.words, (float*)feat_grad, stream);
        return 0;
    }
    if (f.unsorted) {  // the streaming kernels want the rows in grid order after all
        char* sorted = a.take<char>(rows_b);
        if (!sorted) return MCCNN_E_WORKSPACE;
        rc = mccnn_permute_scatter((const float*)feats, go->new_idx, n, s.words, (float*)sorted, n, 0, stream);
        if (rc) return rc;
        rows_in = sorted;
    }
    char* fg = a.take<char>(rows_b);
    if (!fg) return MCCNN_E_WORKSPACE;
    const int *st = nullptr, *pt = nullptr;
    if (e > 0 && (!s.combin || ((flags & 2) && s.fin >= 2 && s.fin <= 4) || (s.combin && s.fin >= 2 && s.fin <= 4 && g->tl_built))) {
        // depth-wise layers need the transposed list; combin layers with 2..4 input features gather their feature
        // gradient through it when the caller asks for the deterministic form or the list exists already
        const size_t tb = mccnn_transpose_neighbors_workspace_bytes(n, e);
        if (!g->tl_built) {
            char* tw = a.base + a.off;  // free until the convolution's own scratch is carved out below
            if (a.cap - a.off < tb) return MCCNN_E_WORKSPACE;
            rc = ensure_tlist(g, e, tw, a.cap - a.off, stream);
            if (rc) return rc;
        }
        st = tl_start(g);
        pt = tl_perm(g);
    }
    const size_t wb = mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, s.fin, s.fout, s.combin);
    char* w = a.take<char>(wb);
    if (!w) return MCCNN_E_WORKSPACE;
    if (s.bf16) {
        rc = mccnn_spatial_conv_bwd_bf16(go->s_pts, rows_in, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn, g->mx, w1,
                                         b1, w2, b2, w3, b3, out_grad, n, m, e, s.fin, g->B, g->radius, g->scale_inv, avg, st, pt,
                                         fg, dw1, db1, dw2, db2, dw3, db3, w, wb, stream);
    } else {
        const void* state = nullptr;
        if (flags & 1) {
            const size_t sb = mccnn_spatial_conv_state_bytes(m, e, s.fin, s.fout, s.combin);
            // (written by the forward call only when IT ran the edge-streaming / factored kernels)
            if (sb && saved && saved_bytes >= sv_off + sb && !f.unsorted && !f.rows_fwd) state = sv + sv_off;
        }
        rc = mccnn_spatial_conv_bwd(go->s_pts, (const float*)rows_in, go->s_bids, g->pdfs, g->centres, g->start, g->packed, g->mn,
                                    g->mx, w1, b1, w2, b2, w3, b3, (const float*)out_grad, n, m, e, s.fin, s.fout, s.combin, g->B,
                                    g->radius, g->scale_inv, avg, state, st, pt, (float*)fg, dw1, db1, dw2, db2, dw3, db3, w, wb,
                                    stream);
    }
    if (rc) return rc;
    return mccnn_permute_gather((const float*)fg, go->new_idx, n, s.words, (float*)feat_grad, stream);
}

}  // extern "C"
