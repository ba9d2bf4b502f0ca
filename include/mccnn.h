/*
 * mccnn.h -- C-ABI of the MI355X-native Monte-Carlo convolution library
 * (libmccnn_hip.so, built from mccnn_amd/csrc/ by hipcc for gfx950).
 *
 * This is the drop-in boundary for the reference's tf_ops operators
 * (viscom-ulm/MCCNN, tf_ops .cc/.cu files behind tf_ops/MCConvModuleSrc). One entry
 * point (or a count/fill pair where the output size is data dependent) replaces
 * one registered TF op; each declaration cites the reference interface it
 * replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - the library never allocates, frees or synchronises, except where a
 *    function documents a host read-back (num_cells with scale_inv == 0);
 *    scratch memory is caller-provided (`ws`, size from the matching
 *    *_workspace_bytes query) -- the reference instead cudaMalloc/cudaFree'd
 *    inside its launchers (find_neighbors.cu:298-309, sort_gpu.cu:410-418,
 *    poisson_sampling.cu:202-223);
 *  - every launch goes to the explicit `stream` (a hipStream_t passed as void*;
 *    the reference used the legacy default stream);
 *  - return value: 0 = ok, < 0 = argument error (MCCNN_E_*), > 0 = hipError_t.
 *    The reference aborted the process on CUDA errors (cuda_kernel_utils.h:17-24);
 *    this library never does;
 *  - data layout is the reference's flattened ragged batch:
 *    points [N,3] f32, batch ids [N] i32 (the [N,1] tensor, flat), features
 *    [N,F] f32 row-major, aabb [B,3] f32, cell table [B,nc,nc,nc,2] i32,
 *    start indices [M] i32, packed neighbours [E,2] i32 rows (j, i);
 *  - kernel-MLP weights are addressed FLAT exactly like the reference kernels
 *    (spatial_conv.cu:172): w1[nu*3+d], b1[nu], w2[q*64+n*8+m], b2[nu],
 *    w3[q*64+n*8+m], b3[nu], nu = 8q+n -- although the Python variables are
 *    declared [3,8nb] / [8,8nb] (MCConvBuilder.py:407-419) there is no transpose.
 *  - canonical order: inside a grid cell points keep ascending original index;
 *    Poisson samples are emitted batch -> phase -> cell (launch-linear) -> point.
 *    Both are one valid outcome of the reference's atomics (sort_gpu.cu:170,
 *    poisson_sampling.cu:115) and are reproducible.
 */
#ifndef MCCNN_H_
#define MCCNN_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mccnn_stream_t; /* hipStream_t */

#define MCCNN_OK 0
#define MCCNN_E_BADARG (-1)     /* null pointer / non-positive size            */
#define MCCNN_E_BATCHID (-2)    /* batch id outside [0,B): raised by the binding from mccnn_check_batch_ids */
#define MCCNN_E_TOOLARGE (-3)   /* B*nc^3 or E does not fit int32              */
#define MCCNN_E_WORKSPACE (-4)  /* workspace smaller than *_workspace_bytes    */
#define MCCNN_E_SHAPE (-5)      /* MLP shape rules of spatial_conv.cc:258-300  */

/* generated get_block_size(), genCompileScript.py:46-47 (BLOCK_MLP_SIZE = 8) */
int mccnn_block_size(void);
/* library/ABI version, arch string ("gfx950") */
int mccnn_abi_version(void);
const char* mccnn_arch(void);
const char* mccnn_error_string(int code);

/* Batch ids index per-cloud tables (boxes, cell grids) in every op. The reference never checks them
 * (an id outside [0, batch_size) reads / writes out of bounds: sort_gpu.cu:46-59, aabb_gpu.cu:97-103). Here
 * every kernel clamps the id before it indexes (memory-safe, result unspecified for invalid input) and this
 * asynchronous check counts the invalid ids into *bad_count_dev, for a binding that wants to raise
 * MCCNN_E_BATCHID (costs the caller one 4-byte read-back; mccnn_amd.MCConvModule.CHECK_BATCH_IDS). */
int mccnn_check_batch_ids(const int* batch_ids, int n, int batch_size, int* bad_count_dev,
                          mccnn_stream_t stream);

/* ComputeAabb -- aabb_gpu.cc:22-86, aabb_gpu.cu:57-140.
 * scale_inv == 0: every row receives the whole-batch box (aabb_gpu.cu:104-114). */
size_t mccnn_compute_aabb_workspace_bytes(int batch_size);
int mccnn_compute_aabb(const float* pts, const int* batch_ids, int n, int batch_size, int scale_inv,
                       float* aabb_min, float* aabb_max, void* ws, size_t ws_bytes,
                       mccnn_stream_t stream);

/* determineNumCells -- sort_gpu.cu:374-420. scale_inv != 0: pure host arithmetic.
 * scale_inv == 0: reads the box of batch 0 back to the host (one stream sync,
 * like the reference's cudaMemcpy D2H at sort_gpu.cu:417). */
int mccnn_num_cells(const float* aabb_min, const float* aabb_max, int batch_size, float cell_size,
                    int scale_inv, int* num_cells_host, mccnn_stream_t stream);
/* The box extent mccnn_num_cells(scale_inv = 0) divides by -- max over the axes of aabb_max[0] - aabb_min[0] (batch 0:
 * with scale_inv = 0 every row holds the whole-batch box, aabb_gpu.cu:104-114) -- read back ONCE (24 bytes, this call
 * synchronises the stream). A binding that builds several grids over the same boxes (every level of a PointHierarchy,
 * every convolution radius) computes num_cells = max(1, (int)(extent / cell_size)) in float on the host from it
 * instead of paying the reference's read-back (sort_gpu.cu:410-419) per grid. */
int mccnn_aabb_extent(const float* aabb_min, const float* aabb_max, float* extent_host,
                      mccnn_stream_t stream);

/* SortPointsStep1 -- sort_gpu.cc:23,184-268, sort_gpu.cu:35-174,436-471.
 * keys[i] = b*nc^3 + x*nc^2 + y*nc + z; new_idx[i] = destination of point i in
 * the cell-sorted list (stable: ascending i inside a cell). */
size_t mccnn_sort_step1_workspace_bytes(int n, int batch_size, int num_cells);
int mccnn_sort_step1(const float* pts, const int* batch_ids, const float* aabb_min,
                     const float* aabb_max, int n, int batch_size, int num_cells, int* keys,
                     int* new_idx, void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* SortPointsStep2 -- sort_gpu.cc:40,270-378, sort_gpu.cu:192-248,473-497.
 * Applies the permutation and builds the (first,last+1) cell table; empty cell = (0,0).
 * inv_idx: optional [n] output, inv_idx[new_idx[i]] = i -- the cell-coherent visiting order
 * mccnn_find_neighbors_* take as centre_order for same-level searches; may be NULL. */
size_t mccnn_sort_step2_workspace_bytes(int n);
int mccnn_sort_step2(const float* pts, const int* batch_ids, const float* feats, const int* keys,
                     const int* new_idx, int n, int num_feats, int batch_size, int num_cells,
                     float* out_pts, int* out_batch_ids, float* out_feats, int* cell_indexs,
                     int* inv_idx, void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* out[i,:] = in[idx[i],:], i < n_idx.  Replaces SortPointsStep2Grad (sort_gpu.cu:260),
 * SortFeaturesBack (:288) and GetSampledFeatures (poisson_sampling.cu:135). */
int mccnn_permute_gather(const float* in, const int* idx, int n_idx, int num_feats, float* out,
                         mccnn_stream_t stream);
/* out[idx[i],:] = in[i,:], i < n_idx; zero_fill != 0 first clears out[0:n_out,:].
 * Replaces SortFeaturesBackGrad (sort_gpu.cu:311; this is Python's sort_features(),
 * MCConvModuleSrc:35-36) and GetSampledFeaturesGrad (poisson_sampling.cu:158,262-274). */
int mccnn_permute_scatter(const float* in, const int* idx, int n_idx, int num_feats, float* out,
                          int n_out, int zero_fill, mccnn_stream_t stream);

/* TransformIndexs -- sort_gpu.cc:96,496-533, sort_gpu.cu:332-362.
 * out[s] = inv[in_idx[s]] with inv[new_idx[i]] = i.  ws: n ints. */
size_t mccnn_transform_indexs_workspace_bytes(int n);
int mccnn_transform_indexs(const int* in_idx, int s, const int* new_idx, int n, int* out_idx,
                           void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* FindNeighbors -- find_neighbors.cc:25,80-185, find_neighbors.cu:40-372.
 * count: start_idx[i] = exclusive prefix of the per-centre neighbour counts and
 * *total_dev = E.  total_dev is any DEVICE-ACCESSIBLE int: device memory, or a pinned host
 * word (hipHostMalloc) -- then no device-to-host copy is needed and a host that polls the
 * word sees E as soon as the count retires (the Python layer does this).  The caller reads
 * E, allocates packed[E,2] and calls fill with the same arguments.  Row order: centre ascending; inside a
 * centre the 27-cell table order of find_neighbors.cu:282-291, then ascending j.
 * centre_order (optional, may be NULL): a permutation of 0..m-1 giving the order in
 * which threads visit the centres. It never changes the result; a spatially coherent
 * order (e.g. the inverse of new_idx when the centres are the gridded points) makes
 * neighbouring lanes walk the same cells, which is several times faster. */
size_t mccnn_find_neighbors_workspace_bytes(int m, int n);
int mccnn_find_neighbors_count(const float* centres, const int* centre_batch_ids, int m,
                               const float* sorted_pts, int n, const int* cell_indexs,
                               const float* aabb_min, const float* aabb_max, int batch_size,
                               int num_cells, float radius, int scale_inv, const int* centre_order,
                               int* start_idx, int* total_dev, void* ws, size_t ws_bytes,
                               mccnn_stream_t stream);
/* e: capacity of `packed` in rows. Normally the total the count call produced; a caller that wants
 * to launch the fill before it has read that total back (to hide the read-back behind the kernel)
 * may pass a guess: rows beyond the capacity are simply not written, and if the total turns out
 * larger the call is repeated with a big enough buffer. ws: the workspace of the count call,
 * UNTOUCHED in between -- the count pass leaves the hit masks of every (centre, 64-candidate round)
 * there and the fill pass only compacts them (one traversal of the candidate windows, not two as in
 * find_neighbors.cu:268-372). */
/* The same count with a SECOND destination of the total (extension): total_dev stays on the device for
 * mccnn_compute_pdf_dn, total_host is a pinned host word the caller polls -- no device-to-host copy is enqueued. */
int mccnn_find_neighbors_count2(const float* centres, const int* centre_batch_ids, int m,
                                const float* sorted_pts, int n, const int* cell_indexs,
                                const float* aabb_min, const float* aabb_max, int batch_size, int num_cells,
                                float radius, int scale_inv, const int* centre_order, int* start_idx,
                                int* total_dev, int* total_host, void* ws, size_t ws_bytes,
                                mccnn_stream_t stream);
int mccnn_find_neighbors_fill(const float* centres, const int* centre_batch_ids, int m,
                              const float* sorted_pts, int n, const int* cell_indexs,
                              const float* aabb_min, const float* aabb_max, int batch_size,
                              int num_cells, float radius, int scale_inv, const int* centre_order,
                              const int* start_idx, int e, int* packed, void* ws, size_t ws_bytes,
                              mccnn_stream_t stream);
/* inv[new_idx[i]] = i -- the visiting order above for same-level searches (sort_gpu.cu:332-345). */
int mccnn_invert_permutation(const int* new_idx, int n, int* inv, mccnn_stream_t stream);

/* ComputePDF -- compute_pdf.cc:25,57-142, compute_pdf.cu:40-119.
 * mode 0: the reference's arithmetic (double exp per axis, float rounding per
 *         statement, compute_pdf.cu:72-92);
 * mode 1: single precision, the pair sums of a row as 16 x 16 Gram-matrix tiles on the
 *         matrix cores (coordinates relative to the row's first point); agrees with
 *         mode 0 to ~1e-5 relative per value, inside the 1e-4 tolerance of the feature
 *         path; needs no workspace beyond the 256-byte minimum;
 * mode 2: single precision on the VALU (subtract-first pair loop, ~1e-6 relative);
 *         workspace: 16 bytes per edge. */
size_t mccnn_compute_pdf_workspace_bytes(int e, int mode);
int mccnn_compute_pdf(const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx,
                      int m, const int* packed, int e, const float* aabb_min, const float* aabb_max,
                      int batch_size, float window, float radius, int scale_inv, int mode,
                      float* pdfs, void* ws, size_t ws_bytes, mccnn_stream_t stream);
/* ComputePDF with the edge count read from DEVICE memory (single-precision KDE only): packed / pdfs hold
 * e_capacity rows, *e_dev (<= e_capacity expected) of them are valid. Lets a caller that guessed the size of
 * the neighbour list enqueue search and KDE back to back without waiting for the count
 * (ConvolutionBuilder.prefetch_geometry); if *e_dev turns out larger than e_capacity the rows are cut and
 * the caller repeats with the exact size. Workspace: mccnn_compute_pdf_workspace_bytes(e_capacity, 1). */
int mccnn_compute_pdf_dn(const float* sorted_pts, const int* sorted_batch_ids, const int* start_idx, int m,
                         const int* packed, int e_capacity, const int* e_dev, const float* aabb_min,
                         const float* aabb_max, int batch_size, float window, float radius, int scale_inv,
                         float* pdfs, void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* PoissonSampling -- poisson_sampling.cc:26,109-211, poisson_sampling.cu:51-230.
 * count: runs the 27 colour phases, leaves the selection in ws and writes the
 * number of samples S to *total_dev.  fill: emits pts[S,3], batch ids[S] and
 * indices[S] (into the SORTED list) in canonical order.  ws must be preserved
 * between the two calls.
 * mode 0: one launch per phase (27 grid-wide barriers).  mode 1: all phases in one launch, cells
 * wait on per-cell flags of the earlier-phase cells of their window (same samples, same order).
 * The waits are bounded; if one times out *total_dev is set to -1 and the caller repeats the
 * count with mode 0. */
size_t mccnn_poisson_sampling_workspace_bytes(int n, int batch_size, int num_cells);
int mccnn_poisson_sampling_count(const float* sorted_pts, const int* sorted_batch_ids, int n,
                                 const int* cell_indexs, const float* aabb_min,
                                 const float* aabb_max, int batch_size, int num_cells, float radius,
                                 int scale_inv, int mode, int* total_dev, void* ws, size_t ws_bytes,
                                 mccnn_stream_t stream);
int mccnn_poisson_sampling_fill(const float* sorted_pts, int n, const int* cell_indexs,
                                int batch_size, int num_cells, int s, float* out_pts,
                                int* out_batch_ids, int* out_indexs, void* ws, size_t ws_bytes,
                                mccnn_stream_t stream);

/* SpatialConv -- spatial_conv.cc:24,159-321, spatial_conv.cu:24-325,796-871.
 * out[M, combin ? num_out_feats : num_in_feats].  Every output row is written
 * (no pre-zeroing needed).
 * state: optional device buffer of mccnn_spatial_conv_state_bytes() bytes (0 = this layer shape
 * keeps no state). The forward pass leaves there what it has computed anyway and the backward pass
 * needs again: one 16-byte record (delta, 1 / (pdf K)) per edge and, for combin layers with one
 * input feature, the per-centre sums of the factored algorithm; handing the same buffer to
 * mccnn_spatial_conv_bwd saves it its pre-pass over the edges (and, for those layers, one more).
 * NULL is always allowed (the reference op has no such output: SpatialConvGrad then recomputes). */
size_t mccnn_spatial_conv_state_bytes(int m, int e, int num_in_feats, int num_out_feats, int combin);
size_t mccnn_spatial_conv_fwd_workspace_bytes(int m, int e, int num_in_feats, int num_out_feats,
                                              int combin);
int mccnn_spatial_conv_fwd(const float* sorted_pts, const float* sorted_feats,
                           const int* sorted_batch_ids, const float* pdfs, const float* samples,
                           const int* start_idx, const int* packed, const float* aabb_min,
                           const float* aabb_max, const float* w1, const float* b1, const float* w2,
                           const float* b2, const float* w3, const float* b3, int n, int m, int e,
                           int num_in_feats, int num_out_feats, int combin, int batch_size,
                           float radius, int scale_inv, int avg, float* out, void* state,
                           void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* SpatialConvGrad -- spatial_conv.cc:56,323-519, spatial_conv.cu:327-792,873-966.
 * Gradients w.r.t. the features and the six MLP tensors only
 * (MCConvModuleSrc:74-81).  All seven outputs are fully written; gradients of
 * padded output neurons (nu >= neuronsOut), which the reference leaves
 * uninitialised (spatial_conv.cu:921,924), are zero. state: the buffer the forward call of the
 * SAME inputs filled, or NULL. start_t / perm_t: optional transposed neighbour list (see
 * mccnn_transpose_neighbors): depth-wise layers need it (built internally when NULL); combin layers
 * with 2..4 input features use it IF supplied -- their feature gradient is then gathered in a fixed
 * order instead of added with float atomics (bit-reproducible); NULL keeps the atomics. */
size_t mccnn_spatial_conv_bwd_workspace_bytes(int n, int m, int e, int num_in_feats,
                                              int num_out_feats, int combin);
int mccnn_spatial_conv_bwd(const float* sorted_pts, const float* sorted_feats,
                           const int* sorted_batch_ids, const float* pdfs, const float* samples,
                           const int* start_idx, const int* packed, const float* aabb_min,
                           const float* aabb_max, const float* w1, const float* b1, const float* w2,
                           const float* b2, const float* w3, const float* b3, const float* out_grad,
                           int n, int m, int e, int num_in_feats, int num_out_feats, int combin,
                           int batch_size, float radius, int scale_inv, int avg, const void* state,
                           const int* start_t, const int* perm_t, float* feat_grad, float* dw1, float* db1, float* dw2,
                           float* db2, float* dw3, float* db3, void* ws, size_t ws_bytes,
                           mccnn_stream_t stream);

/* bf16 FEATURE STORAGE for depth-wise layers (extension, BASELINE cfg3; the reference is f32-only).
 * Same operators as mccnn_spatial_conv_fwd / _bwd with combin == 0, but the gathered rows -- features
 * [n, num_feats], outputs [m, num_feats], out-gradients and feature gradients -- are stored as bf16
 * (two-byte elements, round-to-nearest-even on store, widened exactly on load); the kernel-MLP
 * tensors, every product and every accumulation stay f32. These layers are bound by the gather of
 * 4 * Fin bytes per edge and direction: bf16 rows halve that. Requirements: num_feats % 8
 * == 0, (num_feats + 7) / 8 <= 64 blocks, 16-byte aligned row buffers; otherwise MCCNN_E_SHAPE.
 * Workspace sizes: the f32 queries with num_in_feats = num_out_feats = num_feats, combin = 0. */
int mccnn_spatial_conv_fwd_bf16(const float* sorted_pts, const void* sorted_feats_bf16,
                                const int* sorted_batch_ids, const float* pdfs, const float* samples,
                                const int* start_idx, const int* packed, const float* aabb_min,
                                const float* aabb_max, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, int n, int m, int e,
                                int num_feats, int batch_size, float radius, int scale_inv, int avg,
                                void* out_bf16, void* ws, size_t ws_bytes, mccnn_stream_t stream);
int mccnn_spatial_conv_bwd_bf16(const float* sorted_pts, const void* sorted_feats_bf16,
                                const int* sorted_batch_ids, const float* pdfs, const float* samples,
                                const int* start_idx, const int* packed, const float* aabb_min,
                                const float* aabb_max, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3,
                                const void* out_grad_bf16, int n, int m, int e, int num_feats,
                                int batch_size, float radius, int scale_inv, int avg, const int* start_t,
                                const int* perm_t, void* feat_grad_bf16, float* dw1, float* db1, float* dw2,
                                float* db2, float* dw3, float* db3, void* ws, size_t ws_bytes,
                                mccnn_stream_t stream);

/* Transposed neighbour list (CSR by neighbour index j): start_t[n+1] and perm_t[e] = edge ids
 * grouped by j, ascending inside a row. Depth-wise SpatialConvGrad needs it to compute the
 * feature gradient without float atomics; pass the pair to mccnn_spatial_conv_bwd (start_t /
 * perm_t, both may be NULL -> built internally on every call) to amortise it over the layers
 * that share one neighbour list, as ConvolutionBuilder's caches do (MCConvBuilder.py:366-376). */
size_t mccnn_transpose_neighbors_workspace_bytes(int n, int e);
int mccnn_transpose_neighbors(const int* packed, int e, int n, int* start_t, int* perm_t, void* ws,
                              size_t ws_bytes, mccnn_stream_t stream);

/* DEVICE-SIDE POINT COUNTS (hierarchy construction without per-level host read-backs).
 * PointHierarchy (MCConvBuilder.py:101-128) chains sort -> Poisson sampling -> transform_indexs level
 * after level, and the reference reads the number of samples back to the host at every level
 * (poisson_sampling.cu:222) because it sizes the next level's launches. These variants take the
 * number of valid points from DEVICE memory (*n_dev <= n_cap; launches and buffers are sized by the
 * capacity n_cap), so a caller can run all levels back to back -- mccnn_poisson_sampling_count /_fill
 * already work from the cell table and take n_cap as n -- and read every level's size back ONCE at the
 * end. Same results as the host-count forms. sort_step2_dn moves geometry only (feature rows are
 * gathered once the sizes are known). */
int mccnn_sort_step1_dn(const float* pts, const int* batch_ids, const float* aabb_min,
                        const float* aabb_max, int n_cap, const int* n_dev, int batch_size,
                        int num_cells, int* keys, int* new_idx, void* ws, size_t ws_bytes,
                        mccnn_stream_t stream);
int mccnn_sort_step2_dn(const float* pts, const int* batch_ids, const int* keys, const int* new_idx,
                        int n_cap, const int* n_dev, int batch_size, int num_cells, float* out_pts,
                        int* out_batch_ids, int* cell_indexs, void* ws, size_t ws_bytes,
                        mccnn_stream_t stream);
/* One level of a point hierarchy in one call (extension; MCConvBuilder.py:101-128 per level): mccnn_sort_step1_dn +
 * mccnn_sort_step2_dn + mccnn_poisson_sampling_count / _fill + mccnn_transform_indexs_dn with the level's point count in
 * device memory (n_dev) and the sample count left there (s_dev, -1 when a wait of the single-launch Poisson form timed
 * out: the caller repeats the level op by op). Outputs are sized by n_cap. */
size_t mccnn_hierarchy_level_workspace_bytes(int n_cap, int batch_size, int num_cells);
int mccnn_hierarchy_level(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max,
                          int n_cap, const int* n_dev, int batch_size, int num_cells, float radius,
                          int scale_inv, int mode, int* new_idx, float* sorted_pts, int* sorted_batch_ids,
                          int* cell_indexs, float* out_pts, int* out_batch_ids, int* out_indexs,
                          int* transformed_indexs, int* s_dev, void* ws, size_t ws_bytes,
                          mccnn_stream_t stream);
/* SortPointsStep1 + SortPointsStep2 of the POINTS only in one call (extension: what a convolution builder needs of a
 * grid is the sorted points / batch ids, the cell table and index_new_pos -- feature rows are sorted where they are
 * consumed). Same outputs as the two ops; inv_idx (optional): the inverse permutation (sorted position -> input row). */
size_t mccnn_build_grid_workspace_bytes(int n, int batch_size, int num_cells);
int mccnn_build_grid(const float* pts, const int* batch_ids, const float* aabb_min, const float* aabb_max, int n,
                     int batch_size, int num_cells, int* new_idx, float* out_pts, int* out_batch_ids,
                     int* cell_indexs, int* inv_idx, void* ws, size_t ws_bytes, mccnn_stream_t stream);
int mccnn_transform_indexs_dn(const int* in_idx, int s_cap, const int* s_dev, const int* new_idx,
                              int n_cap, const int* n_dev, int* out_idx, void* ws, size_t ws_bytes,
                              mccnn_stream_t stream);

/* ROW-PER-LANE LAYOUTS of a neighbour list (extension; no counterpart in the reference, whose kernels walk the list
 * with 8 threads per (edge, block) and float atomics -- spatial_conv.cu:104-176,477-561).
 * A plan stores a CSR list in SELL-64-sigma form. Rows (centres for the forward plan, neighbour points for the
 * transposed plan) longer than 128 edges are cut into VIRTUAL rows of at most 128 edges; virtual rows are sorted by
 * length inside windows of 1024 (taken in the visiting order of the rows) and cut into slices of 64; slice s holds
 * (longest virtual row in s) x 64 slots, slot (it, lane) = edge #it of the lane's virtual row: a 16-byte record
 * (delta0, delta1, delta2, 1 / (pdf K)) and the index of the row at the other end of the edge, zero records as padding.
 * With it a lane owns a ROW: the sum over a centre's edges is a per-lane accumulator instead of a segmented wave scan,
 * and the backward pass finishes the feature-gradient row of a point in the same sweep that feeds the weight-gradient
 * sums (mccnn_spatial_conv_fwd_rows / _bwd_rows). One plan serves every layer over the same neighbour list, PDFs,
 * radius and avg flag.
 * Every size is a function of (rows, e) -- mccnn_rowplan_sizes: num_slices S, slot_capacity (a window's slices hold
 * at most 64 x 128 + its edges slots, so e + 8192 x windows bounds the total), scratch_rows (bound on the number of
 * virtual rows: pieces of cut rows leave their partial sums in one scratch row each) -- so a plan is laid out and
 * filled WITHOUT any host read-back.
 *   layout: plan_vrow[64 S] (row of (slice, lane), -1 = padding lane), plan_vcode[64 S] (virtual row id v of (slice,
 *           lane) when its row is cut, ~v when it is not), slice_off[S + 1] (first slot of every slice; slices beyond
 *           the last virtual row are empty), vpos_row[rows] (first virtual row id of every row). row_start: CSR offsets
 *           of the rows (start_idx with rows = m for the forward plan; start_t of mccnn_transpose_neighbors with
 *           rows = n for the transposed one; entry `rows` is not read, e closes the last row). order: optional
 *           cell-coherent visiting order of the rows (a permutation; only affects locality). Deterministic.
 *   fill:   permutes the per-edge records of mccnn_edge_records into slot order. rec: 16 bytes per slot, other: 4
 *           bytes per slot, both slot_capacity long. */
int mccnn_rowplan_sizes(int rows, int e, int* num_slices, long long* slot_capacity, long long* scratch_rows);
size_t mccnn_rowplan_workspace_bytes(int rows, int e);
int mccnn_rowplan_layout(const int* row_start, int rows, int e, const int* order, int* plan_vrow,
                         int* plan_vcode, int* slice_off, int* vpos_row, void* ws, size_t ws_bytes,
                         mccnn_stream_t stream);
/* Per-edge records (delta0, delta1, delta2, 1 / (pdf K)) in EDGE order, 16 bytes each -- the same bits the forward
 * pass leaves in its state buffer (mccnn_spatial_conv_state_bytes): written once per (list, PDFs, radius, avg) and
 * permuted into both plans by mccnn_rowplan_fill. */
int mccnn_edge_records(const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs,
                       const float* samples, const int* start_idx, const int* packed, const float* aabb_min,
                       const float* aabb_max, int n, int m, int e, int batch_size, float radius,
                       int scale_inv, int avg, void* rec_edges, mccnn_stream_t stream);
int mccnn_rowplan_fill(int transposed, const void* rec_edges, const int* packed, int rows, int e,
                       const int* row_start, const int* perm_t, const int* plan_vrow,
                       const int* plan_vcode, const int* slice_off, const int* vpos_row, void* rec,
                       int* other, mccnn_stream_t stream);

/* The whole plan in ONE call and ONE buffer (what a step of a network pays per neighbour list and direction is host
 * work: five calls and seven allocations cost more than the kernels of a coarse level's list).
 * mccnn_rowplan_buffer: byte offsets of {plan_vrow, plan_vcode, slice_off, vpos_row, plan_other, plan_rec} inside a
 * plan buffer of total_bytes for (rows, e), plus the three sizes of mccnn_rowplan_sizes.
 * mccnn_rowplan_build = [mccnn_transpose_neighbors] + mccnn_rowplan_layout + [mccnn_edge_records] +
 * mccnn_rowplan_fill into that buffer. transposed != 0: rows = the n points; start_t [n + 1] / perm_t [e] are written
 * first unless tlist_ready != 0. rec_edges [e x 16 bytes] is written unless rec_ready != 0 (it serves both plans of a
 * list). order: optional visiting order of the rows (forward plans). */
int mccnn_rowplan_buffer(int rows, int e, long long offsets[6], long long* total_bytes, int* num_slices,
                         long long* slot_capacity, long long* scratch_rows);
size_t mccnn_rowplan_build_workspace_bytes(int rows, int e, int transposed);
/* Sizes of the plan buffer and of the build's workspace that hold for EVERY edge count 0 .. e_cap (for a caller that
 * allocates before the list's true size is known). */
int mccnn_rowplan_bound(int rows, int e_cap, int transposed, long long* buffer_bytes, long long* ws_bytes);
/* 1 when mccnn_rowplan_build evaluates the records of a plan of (rows, e) inside its fill (small lists): rec_edges may then
 * be NULL and is neither read nor written. */
int mccnn_rowplan_inline_records(int rows, int e);
int mccnn_rowplan_build(int transposed, const float* sorted_pts, const int* sorted_batch_ids, const float* pdfs,
                        const float* samples, const int* start_idx, const int* packed, const float* aabb_min,
                        const float* aabb_max, int n, int m, int e, int batch_size, float radius, int scale_inv,
                        int avg, const int* order, void* rec_edges, int rec_ready, int* start_t, int* perm_t,
                        int tlist_ready, void* plan_buffer, void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* SpatialConv / SpatialConvGrad for DEPTH-WISE layers (combin == 0, num_feats % 8 == 0, 16-byte aligned rows) over a
 * row plan -- same results as mccnn_spatial_conv_fwd / _bwd (spatial_conv.cu:178-325,563-792) up to float summation
 * order. fwd_rows takes the FORWARD plan (rows = the m centres); bwd_rows takes the TRANSPOSED plan (rows = the n
 * points, start_t = its row offsets) and evaluates the kernel MLP once per (edge, block) for the feature gradient and
 * the six parameter gradients together (the edge-major form needs a second pass over the transposed list). bf16 != 0:
 * rows (features, outputs, out-gradients, feature gradients) stored as bf16 like mccnn_spatial_conv_*_bf16. scratch:
 * scratch_rows x num_feats floats (mccnn_rowplan_sizes). Every output row is written exactly once; no atomics;
 * bit-reproducible. feat_index (optional, NULL = none): sorted_feats (and feat_grad) hold the rows of the points in
 * another order -- row j of the sorted list is row feat_index[j] there (the grid's inverse permutation: the features of
 * the UNSORTED points are read where they lie and their gradient written back in that order, no sorted copy). */
int mccnn_spatial_conv_fwd_rows(const float* sorted_pts, const void* sorted_feats,
                                const int* sorted_batch_ids, const float* pdfs, const float* samples,
                                const int* start_idx, const int* packed, const float* aabb_min,
                                const float* aabb_max, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, int n, int m, int e,
                                int num_feats, int batch_size, float radius, int scale_inv, int avg,
                                int bf16, const int* plan_vrow, const int* plan_vcode,
                                const int* slice_off, const int* vpos_row, const void* plan_rec,
                                const int* plan_other, void* out, float* scratch, const int* feat_index,
                                mccnn_stream_t stream);
size_t mccnn_spatial_conv_bwd_rows_workspace_bytes(int n, int e, int num_feats);
int mccnn_spatial_conv_bwd_rows(const float* sorted_pts, const void* sorted_feats,
                                const int* sorted_batch_ids, const float* pdfs, const float* samples,
                                const int* start_idx, const int* packed, const float* aabb_min,
                                const float* aabb_max, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, const void* out_grad,
                                int n, int m, int e, int num_feats, int batch_size, float radius,
                                int scale_inv, int avg, int bf16, const int* start_t,
                                const int* plan_vrow, const int* plan_vcode, const int* slice_off,
                                const int* vpos_row, const void* plan_rec, const int* plan_other,
                                void* feat_grad, float* scratch, float* dw1, float* db1, float* dw2,
                                float* db2, float* dw3, float* db3, const int* feat_index, void* ws,
                                size_t ws_bytes, mccnn_stream_t stream);

/* NATIVE STEP EXECUTOR (extension; what ConvolutionBuilder.create_convolution, MCConvBuilder.py:336-427, does around
 * the ops): one call per convolution GEOMETRY and one per layer and direction instead of ~12 op calls, because a step of
 * a network is bound by the host issuing its launches, not by the kernels. Built on the entry points above --
 * identical launches, identical results.
 *
 * mccnn_geometry_t: the grid of the input level at the convolution radius (sort_points_step1/2 of the points), the
 * neighbour list of the output level's points in it (find_neighbors) and its PDFs (compute_pdf) -- one cache entry each
 * of cacheGrids_ / cacheNeighs_ / cachePDFs_ (MCConvBuilder.py:349-391) -- in ONE caller-provided device buffer of
 * mccnn_geometry_bytes(). The host object is created / destroyed by the library (no device memory behind it).
 * build: enqueues everything WITHOUT a host wait. e_capacity: rows of the neighbour list the buffer holds (the caller's
 * guess, e.g. the total of the last batch of this shape); the true total E is stored by the count pass into
 * *total_host, a PINNED host word owned by the caller that has to stay valid until the total has been read
 * (mccnn_geometry_edges, or the first mccnn_conv_* call, which wait for it). E > e_capacity: every mccnn_conv_* call
 * returns MCCNN_E_CAPACITY and the caller builds again with a larger buffer. grid_from: optional geometry over the same
 * points, boxes and cell count whose grid is shared instead of built again (keyGrid hit with another output level);
 * it has to outlive this one. use_pdf == 0: PDFs of 1 (MCConvBuilder.py:388-390). All input pointers are borrowed
 * until the geometry is destroyed or built again. */
#define MCCNN_E_CAPACITY (-6)   /* neighbour list longer than the e_capacity a geometry was built with */
typedef struct mccnn_geometry mccnn_geometry_t;
mccnn_geometry_t* mccnn_geometry_create(void);
void mccnn_geometry_destroy(mccnn_geometry_t* g);
size_t mccnn_geometry_bytes(int n, int m, int batch_size, int num_cells, int e_capacity, int with_grid);
int mccnn_geometry_build(mccnn_geometry_t* g, const float* pts, const int* batch_ids, int n, const float* centres,
                         const int* centre_batch_ids, int m, const float* aabb_min, const float* aabb_max,
                         int batch_size, int num_cells, float radius, int scale_inv, float window, int use_pdf,
                         int e_capacity, const mccnn_geometry_t* grid_from, void* buffer, size_t buffer_bytes,
                         int* total_host, mccnn_stream_t stream);

/* Several geometries of a network step in ONE call (extension): one launch per kernel KIND over all of them -- head clear,
 * keys + histogram, prefix sums of the cell counters, park, rank + move + cell tables, count pass, prefix sums of the counts,
 * fill pass, KDE: nine launches whatever `count` is (per geometry the chain of mccnn_geometry_build is nine as well) -- with
 * results identical to mccnn_geometry_build bit for bit. A request that shares the grid of another request names that
 * request's geometry in grid_from and comes AFTER it. */
typedef struct mccnn_geometry_request {
    mccnn_geometry_t* geometry;
    const float* pts; const int* batch_ids; int n;
    const float* centres; const int* centre_batch_ids; int m;
    const float* aabb_min; const float* aabb_max;
    int batch_size, num_cells;
    float radius; int scale_inv; float window; int use_pdf; int e_capacity;
    const mccnn_geometry_t* grid_from;
    void* buffer; size_t buffer_bytes;
    int* total_host;
} mccnn_geometry_request;
int mccnn_geometry_build_batch(const mccnn_geometry_request* requests, int count, mccnn_stream_t stream);
/* E, or -1 while the count pass has not retired (wait_us: 0 = look once, < 0 = wait, > 0 = wait at most that long). */
int mccnn_geometry_edges(mccnn_geometry_t* g, int wait_us);
/* out[0..7]: device addresses of sortPts [n,3], sortBatchs [n], cellIndexs [B,nc,nc,nc,2], index_new_pos [n], its
 * inverse [n], startIndexs [m], packedNeighs [e_capacity,2], pdfs [e_capacity]; out[8..13]: n, m, num_cells,
 * e_capacity, E (-1 = not read yet), batch_size; out[14]: device word holding E; out[15]: 1 = owns its grid. */
int mccnn_geometry_info(const mccnn_geometry_t* g, long long out[16]);
/* Further pieces a geometry keeps for the layers over it, each in a caller-provided buffer (256-byte aligned) that lives
 * as long as the geometry: what = 1 forward row plan, 2 transposed row plan, 4 transposed neighbour list, 8 per-edge
 * records. mccnn_conv_prepare says which are missing and how large they are. */
int mccnn_geometry_attach(mccnn_geometry_t* g, int what, void* buffer, size_t bytes);
/* Pieces built ahead of the layers that need them (the transposed list and the transposed row plan of a depth-wise
 * layer's backward pass on a side stream, under the forward passes): piece_bytes waits for E and gives the size of the
 * buffer to attach for ONE piece (what = 1, 2, 4 or 8) and the scratch its build needs; prebuild builds the attached
 * pieces named by the mask `what` (row plans need their records / transposed list attached as mccnn_conv_prepare would
 * ask; avg: the flag of the layers that will use the plans). */
int mccnn_geometry_piece_bytes(mccnn_geometry_t* g, int what, long long* bytes, long long* ws_bytes);
/* The same two sizes WITHOUT a geometry and without waiting: bounds for ONE piece over n points, m centres and any list
 * of at most e_cap edges -- for a caller that allocates the pieces when it queues the geometry's build (another thread
 * attaches and prebuilds them once E has arrived). */
int mccnn_geometry_piece_bound(int n, int m, int e_cap, int what, long long* bytes, long long* ws_bytes);
int mccnn_geometry_prebuild(mccnn_geometry_t* g, int what, int avg, void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* The pieces of several geometries at once (extension): what[k] = mask of the pieces of geoms[k] (their buffers attached with
 * mccnn_geometry_attach). One launch per kernel kind over all of them, in flushes of <= 12 small plans (single-workgroup
 * layout, records evaluated inline) and <= 8 large ones (layout, tile fill / bases + records + scatter), their transpositions
 * as batched chains; what a flush cannot take builds as mccnn_geometry_prebuild does, behind them. Results identical to
 * mccnn_geometry_prebuild per geometry (tests/test_gpu_native.py). Waits for the edge totals. */
size_t mccnn_geometry_prebuild_batch_ws_bytes(mccnn_geometry_t* const* geoms, const int* what, int count);
int mccnn_geometry_prebuild_batch(mccnn_geometry_t* const* geoms, const int* what, int count, int avg, void* ws, size_t ws_bytes,
                                  mccnn_stream_t stream);
/* One layer over a geometry: SpatialConv / SpatialConvGrad INCLUDING sort_features / its gradient
 * (MCConvModuleSrc:35-45,70-81). feats [n, num_in_feats] and feat_grad are rows of the UNSORTED points (f32, or bf16
 * with bf16 != 0: depth-wise layers, num_in_feats % 8 == 0); out [m, combin ? num_out_feats : num_in_feats].
 * flags: bit 0 = keep the forward state for the backward pass, bit 1 = deterministic feature gradient of combin layers
 * with 2..4 input features (gathered through the transposed list instead of float atomics).
 * prepare (forward or backward): waits for E, then reports need_mask / need_bytes[k] (pieces to attach first, bit k),
 * the scratch bytes of the call and -- forward -- the bytes of `saved`, which the caller keeps from forward to
 * backward (sorted feature rows + forward state). The kernel family is chosen as the Python op surface chooses it
 * (row-per-lane depth-wise kernels, factored one-feature kernels, edge streaming). */
int mccnn_conv_prepare(mccnn_geometry_t* g, const void* feats, int num_in_feats, int num_out_feats, int combin, int bf16,
                       int backward, int flags, int* need_mask, long long need_bytes[4], long long* ws_bytes,
                       long long* saved_bytes, int* edges);
int mccnn_conv_forward(mccnn_geometry_t* g, const void* feats, int num_in_feats, int num_out_feats, int combin, int avg,
                       int bf16, int flags, const float* w1, const float* b1, const float* w2, const float* b2,
                       const float* w3, const float* b3, void* out, void* saved, size_t saved_bytes, void* ws,
                       size_t ws_bytes, mccnn_stream_t stream);
int mccnn_conv_backward(mccnn_geometry_t* g, const void* feats, const void* saved, size_t saved_bytes,
                        const void* out_grad, int num_in_feats, int num_out_feats, int combin, int avg, int bf16,
                        int flags, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                        const float* b3, void* feat_grad, float* dw1, float* db1, float* dw2, float* db2, float* dw3,
                        float* db3, void* ws, size_t ws_bytes, mccnn_stream_t stream);

/* TEST HOOK, not part of the operator surface: selects the convolution implementation for A/B
 * parity tests (bit 0: VALU fallback kernels, bit 1: general MFMA kernels for one-input-feature
 * layers; 0 = product default). Returns the previous mask. Initial value: MCCNN_FORCE_VALU /
 * MCCNN_NO_F1 in the environment, read once. */
int mccnn_debug_conv_impl(int mask);

/* Diagnostics: number of kernel launches the library has issued in this process (all streams). bench.py prints the
 * difference over one step: below ~50k points a step is bound by launches, not by the kernels. */
long long mccnn_debug_launch_count(void);
/* Diagnostics: nanoseconds the host has spent so far inside the library WAITING for an edge total (mccnn_geometry_edges /
 * the first mccnn_conv_* call over a geometry). bench.py subtracts it from a step's host time: what is left is the
 * host's own work per step. */
long long mccnn_debug_wait_ns(void);
/* ... of the CALLING thread only when it says so: a helper thread that waits in the caller's place switches its own
 * accounting off (returns the previous setting; thread-local, default on). */
int mccnn_debug_wait_accounting(int on);
/* The calling THREAD's launches are background work from now on (on != 0) / no longer (on == 0): they run on a queue of
 * their own beside kernels a step waits for (ConvolutionBuilder.prefetch_geometry: the geometry of the next batch under
 * the convolutions of the current one). Kernels that would fill every wave slot hold back in that mode; results do not
 * change. Returns the previous setting. */
int mccnn_background_launches(int on);
/* TEST HOOK: combin layers with one input feature run their forward edge pass with four consecutive edges per lane
 * (one segmented scan per 256 edges) on lists of at least `edges` edges (default 2 000 000; shorter lists keep 64-edge
 * chunks: more waves to spread over the chip). 0 = always, INT_MAX = never. Same results up to float summation order.
 * Returns the previous threshold. */
int mccnn_debug_f1_x4_min_edges(int edges);
/* TEST HOOK: small problems (coarse hierarchy levels: a few thousand points / edges) run single-workgroup forms of
 * the grid build, the list transposition ... that replace 4 - 8 launches by one; on = 0 sends them through the
 * multi-launch kernels of the large problems instead (same results). Returns the previous setting. Initial value: on,
 * unless MCCNN_SMALL_OFF is set in the environment. */
int mccnn_debug_small_kernels(int on);

#ifdef __cplusplus
}
#endif
#endif /* MCCNN_H_ */
