/* librdmnet_hip.so -- C-ABI of the MI355X (gfx950) implementation of RDMNet's dense-matching
 * inference path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     unless the function's comment says so;
 *   - `ws`/`ws_bytes` is caller-owned scratch (query the size with the matching *_workspace_bytes);
 *     the library never allocates or frees caller memory and keeps no global state (re-entrant) -- the diagnostic
 *     counters of the lock-step scheduler (rdm_lockstep_stats*, marked DIAGNOSTIC below) are the one exception;
 *   - return value: 0 = ok, <0 = error (see rdm_last_error(), thread-local);
 *   - data-dependent overflows on the device are reported through a caller-provided int32 `status`
 *     word (device memory, must be zero before the call; non-zero afterwards = RDM_ERR_CAPACITY).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root).
 */
#ifndef RDMNET_HIP_H_
#define RDMNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDM_ABI_VERSION 2   /* 2: rdm_engine_result / rdm_data_dict / rdm_kpconv_profile grew, three entry points left (round 3);
                               parameter sets are reference counted (round 4) */

int rdm_abi_version(void);
/* sizeof of the structs that cross this boundary, so that a binding can verify its own layout against the library it
 * loaded: which = 0 rdm_engine_config, 1 rdm_engine_result, 2 rdm_tensor_view, 3 rdm_kpconv_profile, 4 rdm_data_dict;
 * anything else returns 0.                                                                                        */
size_t rdm_abi_struct_size(int which);
const char* rdm_last_error(void);

/* libstdc++ std::unordered_map growth schedule used by rdm_grid_subsample to reproduce the
 * reference's output order: rehash to `buckets[j]` happens when the map already holds `at[j]`
 * elements.  Host-only helper (no GPU needed).  Returns the number of entries written. */
int rdm_rehash_schedule(int64_t max_elems, int64_t* at_host, int64_t* buckets_host, int cap);

/* ---- a1: voxel-grid subsampling ------------------------------------------------------------
 * Replaces rdmnet.ext.grid_subsampling
 *   (geotransformer/extensions/pybind.cpp:13-17,
 *    geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62,
 *    geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75).
 * points [n_points,3] f32 stacked clouds, lengths [batch] i64.  Writes the barycentres of the
 * occupied voxels of each cloud, stacked, in the reference's order (libstdc++ unordered_map
 * iteration order) into out_points (capacity n_points rows) and the per-cloud counts into
 * out_lengths [batch] i64.  No synchronisation: the caller reads out_lengths when it needs the
 * row count.                                                                                  */
size_t rdm_grid_subsample_workspace_bytes(int64_t n_points, int batch);
int rdm_grid_subsample(const float* points, int64_t n_points, const int64_t* lengths, int batch,
                       float voxel_size, float* out_points, int64_t* out_lengths, void* ws,
                       size_t ws_bytes, void* stream);
/* The same with the kernel form chosen by the caller: 0 = by size (what rdm_grid_subsample does: from 16 384 stacked points
 * the phases before the hash-map order replay run as separate launches over many workgroups), 1 = one workgroup per cloud
 * for every phase, 2 = the multi-launch form.  Identical output bit for bit (tests/test_native_gpu.py).                  */
int rdm_grid_subsample_form(const float* points, int64_t n_points, const int64_t* lengths, int batch, float voxel_size,
                            float* out_points, int64_t* out_lengths, void* ws, size_t ws_bytes, void* stream, int form);

/* ---- a2: radius neighbours -----------------------------------------------------------------
 * Replaces rdmnet.ext.radius_neighbors
 *   (geotransformer/extensions/pybind.cpp:8-12,
 *    geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:5-68,
 *    geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91)
 * and the column truncation of geotransformer/modules/ops/radius_search.py:24-26.
 * For every query of cloud b: all support points of cloud b with fp32 d2 < radius*radius, ascending
 * by (d2, index), as GLOBAL support indices; unused slots hold n_s.
 *   out_idx    [n_q, width] i64 (row stride = width); may be NULL when width == 0 (count-only pass)
 *   out_counts [n_q] i32 untruncated neighbour counts (may be NULL)
 *   out_max    [1] i32, atomically max-ed with the largest count (may be NULL; zero it first)
 *   status     [1] i32, non-zero on an internal error (2: the grid was built for a smaller radius).  There is no
 *              neighbour-count limit: rows beyond the kernels' 1024-key buffer are produced in rounds of a radix
 *              select over the (d2, index) keys, like the reference (radius_neighbors_cpu.cpp:36-64) returns them all
 * The reference's output width is max(count); call once with width = 0 to obtain it, or pass the
 * neighbour limit directly (the first min(limit, max) columns are identical).                  */
size_t rdm_radius_neighbors_workspace_bytes(int64_t n_q, int64_t n_s, int batch);
int rdm_radius_neighbors(const float* q_points, int64_t n_q, const float* s_points, int64_t n_s,
                         const int64_t* q_lengths, const int64_t* s_lengths, int batch,
                         float radius, int width, int64_t* out_idx, int32_t* out_counts,
                         int32_t* out_max, int32_t* status, void* ws, size_t ws_bytes,
                         void* stream);

/* The same search in two steps, so that several query sets share one grid over a support cloud (the
 * collate searches every level three times with the same radius: self, from the coarser and from the
 * finer level).  `radius` of a query must not exceed the radius the grid was built with (status = 2). */
size_t rdm_radius_grid_workspace_bytes(int64_t n_s);
int rdm_radius_grid_build(const float* s_points, int64_t n_s, const int64_t* s_lengths, int batch, float radius,
                          void* grid_ws, size_t grid_ws_bytes, void* stream);
int rdm_radius_grid_query(void* grid_ws, size_t grid_ws_bytes, int64_t n_s, const float* q_points, int64_t n_q,
                          const int64_t* q_lengths, int batch, float radius, int width, int64_t* out_idx,
                          int32_t* out_counts, int32_t* out_max, int32_t* status, void* ws, size_t ws_bytes,
                          void* stream);

/* The grid's support records in cell order: n_s x {x, y, z, bit pattern of the int32 row index}.  Rows of
 * one cell are contiguous, which makes the 4th component a spatially coherent processing order. */
const float* rdm_radius_grid_records(void* grid_ws, size_t grid_ws_bytes, int64_t n_s);

/* Calibration of the neighbour limits (geotransformer/utils/data.py:195-220): hist[c] += #{i : counts[i] == c}
 * for c < hist_n, i.e. the reference's np.bincount(counts, minlength=hist_n)[:hist_n] accumulated over calls.
 * counts = out_counts of a width-0 (count-only) rdm_radius_neighbors call.  hist_n <= 1024. */
int rdm_neighbor_histogram(const int32_t* counts, int64_t n, int32_t* hist, int hist_n, void* stream);

/* ---- §8f rank 3: raw-scan preprocessing --------------------------------------------------------
 * Centroid voxel down-sampling of one raw scan: points[n, ld] f32 with `channels` >= 3 leading columns
 * (x, y, z, then e.g. intensity) -> out[m, ldo], m in *out_count (device int32).  Replaces
 * preporcess/downsample_pcd_kitti.py:21-36 (Open3D 0.11.2 voxel_down_sample(0.3) on points + colors):
 * index = floor((p - (min_bound - voxel/2)) / voxel) in float64, per-voxel means accumulated in float64.
 * Open3D is not part of the reference tree -> parity unpinned; voxels are emitted in first-occurrence
 * order (Open3D's hash-map order is unspecified).  status (device int32, caller zeroes): 1 = a point is
 * non-finite or more than 2^21 voxels from the minimum (point skipped).  n < 2^26.                    */
size_t rdm_voxel_downsample_workspace_bytes(int64_t n);
int rdm_voxel_downsample(const float* points, int64_t n, int64_t ld, int channels, double voxel, float* out,
                         int64_t ldo, int32_t* out_count, int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* ---- §8f rank 4: RANSAC pose from correspondences ------------------------------------------------
 * The reference's second evaluation mode (experiments/infer.py:75-82, eval.py:179-186 ->
 * geotransformer/utils/open3d.py:173-203: Open3D registration_ransac_based_on_correspondence, point-to-point,
 * ransac_n 4, 50 000 iterations, 0.3 m).  src_corr/ref_corr: device f32 [n_corr, 3].  Every iteration draws
 * ransac_n correspondences with replacement (counter-based hash of (seed, iteration, draw)), fits a rigid
 * transform (float64 Kabsch), scores it on all correspondences (float64); the winner is the iteration with
 * most inliers, then lowest inlier RMSE, then lowest index; its transform is returned without refit.
 * Outputs (device): transform f32[16] row-major 4x4 (identity if nothing fits), stats int32[2] =
 * {winning iteration or -1, its inliers}, inlier_rmse f32[1], optional hyp_inliers int32[num_iterations].
 * Open3D is not part of the reference tree -> parity unpinned (checked against oracle/preprocess.py).   */
size_t rdm_ransac_workspace_bytes(int num_iterations);
int rdm_ransac_correspondences(const float* src_corr, const float* ref_corr, int64_t n_corr, float distance_threshold,
                               int ransac_n, int num_iterations, uint64_t seed, float* transform, int32_t* stats,
                               float* inlier_rmse, int32_t* hyp_inliers, void* ws, size_t ws_bytes, void* stream);

/* ---- dense contraction ---------------------------------------------------------------------
 * C[b] = act((A[b] (m x k) * op(B[b])) / rowdiv[row] + bias[col]) in fp32 on the f32 MFMA.
 * trans_b = 0: B is [k, n] row-major (pre-transposed nn.Linear weights, KPConv weights viewed
 * [15*C_in, C_out]); trans_b = 1: B is [n, k] row-major.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.1).
 * Replaces torch.nn.Linear / torch.matmul / torch.einsum call sites of the path, e.g.
 * geotransformer/modules/kpconv/kpconv.py:107-110, modules/kpconv/modules.py:77,
 * experiments/model_infer.py:310-311.  k, lda, ldb must be multiples of 4 (zero padded), A and B
 * 16-byte aligned.  ws (rdm_gemm_workspace_bytes) enables deterministic split-K; may be NULL.   */
size_t rdm_gemm_workspace_bytes(int64_t m, int64_t n, int batches);
int rdm_gemm(const float* a, int64_t lda, int64_t stride_a, const float* b, int64_t ldb,
             int64_t stride_b, int trans_b, float* c, int64_t ldc, int64_t stride_c, int64_t m,
             int64_t n, int64_t k, int batches, const float* bias, const float* rowdiv, int act,
             void* ws, size_t ws_bytes, void* stream);
/* The same with the kernel form chosen by the caller (tests, A/B runs; also rdm_linear_group_norm_form, rdm_decoder_stage_form):
 * form 0 = the library's choice; 1 / 2 = the products the dispatch model gives to its 64 x 64 tile (weights as B, no batched
 * gathers) run on the WIDE form instead -- 128 x 128 x 32 tiles, fragment-ordered LDS images, one barrier per k-tile -- on
 * v_mfma_f32_32x32x2_f32 (1) or v_mfma_f32_16x16x4_f32 (2).  Every form returns the same bits: an output element is one fp32 fma
 * chain over ascending k inside the same split-K ranges whatever the tile or the MFMA shape (tools/mfma_order_probe.hip), and
 * the GroupNorm partials keep their 64-row blocks and combination order (tests/test_ops_gpu.py; docs/EXPERIMENTS.md 5h for why
 * the default stays the 64 x 64 tile).                                                                                        */
int rdm_gemm_form(const float* a, int64_t lda, int64_t stride_a, const float* b, int64_t ldb,
                  int64_t stride_b, int trans_b, float* c, int64_t ldc, int64_t stride_c, int64_t m,
                  int64_t n, int64_t k, int batches, const float* bias, const float* rowdiv, int act,
                  void* ws, size_t ws_bytes, int form, void* stream);
/* The tile and split-K factor the dispatch model chose for the calling thread's last rdm_gemm / fused Linear call:
 * out4_host = {tile rows, tile columns, k-tile depth, split-K factor} (diagnostic: profiles/r03_gemm_shapes.md).  */
int rdm_gemm_last_plan(int* out4_host);

/* ---- a4: KPConv neighbourhood aggregation ---------------------------------------------------
 * Replaces the gather half of KPConv.forward (geotransformer/modules/kpconv/kpconv.py:91-105,
 * 113-115): wf[m, k*c + ch] = sum_h max(0, 1 - |s[idx[m,h]] - q[m] - kp[k]| / sigma) * feats[idx[m,h], ch]
 * and nn[m] = max(1, #neighbours whose feature row sums to > 0) (as float).  Pad indices (>= n_s)
 * are the reference's shadow point/zero row.  s_positive[i] = (sum_c feats[i,c] > 0), see
 * rdm_row_positive / rdm_group_norm.  width (optional device int32) caps the row width like the
 * reference's `[:, :min(limit, max_count)]`.  c = 1 or any multiple of 32; any h (rows wider than the 128 slots a wavefront stages in LDS run in chunks).
 * The second half of the convolution is rdm_gemm(wf, W[15*c, c'], rowdiv = nn, bias).          */
int rdm_kpconv_gather(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                      const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                      const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                      const float* kernel_points, float sigma, float* wf, int64_t ldw, float* nn,
                      void* stream);
/* rdm_kpconv_fused: the WHOLE KPConv.forward (kpconv.py:79-122) in one kernel for the fine levels -- c_in = 1 (c_out
 * 64), 32 -> 32, 64 -> 64: out[m, c'] = (sum_k sum_c wf[m, k, c] W[k, c, c']) / nn[m] + bias[c'] with wf and nn as in
 * rdm_kpconv_gather; the [m, 15*c] intermediate stays in LDS.  w_packed: W [15, c_in, c_out] reordered by
 * rdm_kpconv_pack_weights (host arrays; rdm_kpconv_packed_floats floats).  gn_partial (optional): fp64 column sums /
 * sums of squares of the output, one partial row per workgroup: [rdm_kpconv_fused_partial_rows(m, c_in)][2][c_out] -- the
 * input of the GroupNorm that follows every KPConv.  rdm_kpconv_fused_group_norm = that convolution +
 * act(GroupNorm(.)) (modules.py:141-145, 205-207), workspace rdm_kpconv_fused_workspace_bytes.
 * order_records (optional, round 4): m x float4 {x, y, z, query row} in the cell order of the query level's search grid
 * (rdm_radius_grid_records).  For h <= 128 and c_in = 32, or c_in = 64 with queries and support on the same level (2 m > n_s),
 * a workgroup then takes 16 queries that are neighbours in space, stages the union of their support rows (feature row,
 * point, positive flag) ONCE in LDS and aggregates from there ("LDS-staged neighbour tiles"); null = the queries in row order
 * (spatially random for a level of the pyramid: nothing to share, the lock-step kernel runs).  The convolution
 * output does not depend on the order; the GroupNorm partials group the rows by workgroup, i.e. by that order.      */
int rdm_kpconv_fused_enabled(void);   /* 1 iff RDM_FUSED_KPCONV is set: engine and per-op path then use the fused kernel */
int rdm_kpconv_fused_supported(int64_t c_in, int64_t c_out);
int64_t rdm_kpconv_fused_partial_rows(int64_t m, int64_t c_in);
size_t rdm_kpconv_packed_floats(int64_t c_in, int64_t c_out);
int rdm_kpconv_pack_weights(const float* w_host, int64_t c_in, int64_t c_out, float* packed_host);
int rdm_kpconv_fused(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats,
                     int64_t c, int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi,
                     const int32_t* width, const float* kernel_points, float sigma, const float* w_packed,
                     const float* bias, int64_t c_out, float* out, int64_t ldo, double* gn_partial,
                     const float* order_records, void* stream);
/* The same with the kernel chosen by the caller (tests and A/B runs): form 0 = the library's choice (rdm_kpconv_fused), 1 = the
 * lock-step kernel (every (query, neighbour) row fetched from L2), 2 = the LDS-tile kernel where it applies (c_in = 32 / 64,
 * h <= 128).  Both give the same convolution output bit for bit and one partial row per 16 queries (form 2 groups the
 * queries of a row by `order_records`).                                                                              */
int rdm_kpconv_fused_form(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats,
                          int64_t c, int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi,
                          const int32_t* width, const float* kernel_points, float sigma, const float* w_packed,
                          const float* bias, int64_t c_out, float* out, int64_t ldo, double* gn_partial,
                          const float* order_records, int form, void* stream);
size_t rdm_kpconv_fused_workspace_bytes(int64_t m, int64_t c_in, int64_t c_out);
int rdm_kpconv_fused_group_norm(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                                const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                                const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                                const float* kernel_points, float sigma, const float* w_packed, const float* bias,
                                int64_t c_out, int groups, const float* gamma, const float* beta, float eps, int act,
                                float* conv_out, int64_t ld_conv, float* y, int64_t ldy, void* ws, size_t ws_bytes,
                                const float* order_records, void* stream);
/* Same as rdm_kpconv_gather, visiting the queries in the order given by `order_records` (m x float4 whose 4th component
 * holds the query row, e.g. rdm_radius_grid_records of the query level): neighbouring queries share
 * most of their neighbours, so the gathered lines are re-used from the CU's L1.  Results are identical. */
int rdm_kpconv_gather_ordered(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                              const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                              const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                              const float* kernel_points, float sigma, float* wf, int64_t ldw, float* nn,
                              const float* order_records, void* stream);
/* The same with the kernel form chosen by the caller (tests and A/B runs): 0 = the library's choice, 1 = one wavefront per
 * (query, 64-channel slice) fetching every neighbour row, 2 = the support rows of 16 cell-ordered queries staged once in LDS
 * (needs order_records, c a multiple of 64 >= 128, h <= 128; else form 1).  Same WF and nn bits in every form.            */
int rdm_kpconv_gather_form(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                              const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                              const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                              const float* kernel_points, float sigma, float* wf, int64_t ldw, float* nn,
                              const float* order_records, int form, void* stream);
int rdm_row_positive(const float* x, int64_t n, int64_t c, int64_t ld, uint8_t* out, void* stream);

/* ---- a5: block glue --------------------------------------------------------------------------
 * rdm_group_norm: y = act(GroupNorm(x) [+ residual]) with statistics over ALL n rows
 *   (geotransformer/modules/kpconv/modules.py:33-50, 53-83, 204-225); optionally also writes the
 *   positive-row flag of y.  rdm_layer_norm: y = act(LayerNorm(x [+ residual]))
 *   (torch.nn.LayerNorm call sites: modules/transformer/vanilla_transformer.py:79,101, output_layer.py:13,20, rdmnet/vote/vote.py:58-77).
 * rdm_gather_max: modules/kpconv/functional.py:54-67.  rdm_upsample_concat: functional.py:6-22 +
 *   the torch.cat of experiments/backbone.py:131-143; pad columns of y are zeroed.              */
size_t rdm_group_norm_workspace_bytes(int64_t n, int64_t c);
int rdm_group_norm(const float* x, int64_t n, int64_t c, int64_t ldx, int groups, const float* gamma,
                   const float* beta, float eps, const float* residual, int64_t ldr, int act, float* y,
                   int64_t ldy, uint8_t* positive, void* ws, size_t ws_bytes, void* stream);
/* The same with the launch structure chosen by the caller (tests, A/B runs): form 0 = the library's choice -- up to 4 096 rows with
 * whole 64-column slabs, finalize and apply are ONE launch (every workgroup recomputes the scale / shift of its slab from the few
 * dozen partial rows: one dependent launch less on a pair's critical path, same bits) --, form 1 = statistics, finalize and apply
 * as separate launches everywhere.  Applies to this call only.                                                                 */
int rdm_group_norm_form(const float* x, int64_t n, int64_t c, int64_t ldx, int groups, const float* gamma,
                        const float* beta, float eps, const float* residual, int64_t ldr, int act, float* y, int64_t ldy,
                        uint8_t* positive, void* ws, size_t ws_bytes, int form, void* stream);
/* rdm_linear_group_norm: y = act(GroupNorm(x W + bias [/ rowdiv]) [+ residual]) -- UnaryBlock / the
 * KPConv weight contraction + norm_conv (modules/kpconv/modules.py:53-83, 196-207): the GEMM epilogue
 * emits the GroupNorm statistics, saving a pass over the activations.  lin_out [m, n] is scratch for
 * the pre-norm activations.  Same operand rules as rdm_gemm (trans_b = 0).                       */
size_t rdm_linear_group_norm_workspace_bytes(int64_t m, int64_t n);
int rdm_linear_group_norm(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                          const float* rowdiv, int64_t m, int64_t n, int64_t k, int groups, const float* gamma,
                          const float* beta, float eps, const float* residual, int64_t ldr, int act, float* lin_out,
                          int64_t ld_lin, float* y, int64_t ldy, uint8_t* positive, void* ws, size_t ws_bytes,
                          void* stream);
int rdm_linear_group_norm_form(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                               const float* rowdiv, int64_t m, int64_t n, int64_t k, int groups, const float* gamma,
                               const float* beta, float eps, const float* residual, int64_t ldr, int act, float* lin_out,
                               int64_t ld_lin, float* y, int64_t ldy, uint8_t* positive, void* ws, size_t ws_bytes,
                               int form, void* stream);  /* form: see rdm_gemm_form */
/* rdm_patch_scores: the patch score matrices of the fine matching (experiments/model_infer.py:291-311: index_select of the
 * patch features + einsum('bnd,bmd->bnm') / sqrt(d)): scores[b, i, j] = <ref_feats[ref_idx[b, i]], src_feats[src_idx[b, j]]>
 * / rowdiv[i], ref_idx / src_idx [batch, side] int64 with the reference's padded gather (an index outside the tensor selects a
 * zero row).  scores [batch, side, side] contiguous.  The gathered [batch, side, d] tensors are never materialised. */
int rdm_patch_scores(const float* ref_feats, int64_t ld_ref, int64_t n_ref, const int64_t* ref_idx, const float* src_feats,
                     int64_t ld_src, int64_t n_src, const int64_t* src_idx, int64_t batch, int64_t side, int64_t d,
                     const float* rowdiv, float* scores, void* stream);
/* rdm_decoder_stage: one stage of the decoder (experiments/backbone.py:118-151; nearest_upsample =
 * geotransformer/modules/kpconv/functional.py:6-22, UnaryBlock / LastUnaryBlock = modules.py:53-101):
 *   y = act(GroupNorm([coarse[idx[:, 0]] | skip] W + bias))        gamma != NULL (lin_out: scratch for the pre-norm rows)
 *   lin_out = [coarse[idx[:, 0]] | skip] W + bias                  gamma == NULL
 * idx [m, ldi] is the upsampling table (column 0 used; an index outside [0, n_coarse) gives a zero row); W [pad4(c1+c2), n pad]
 * as rdm_gemm's B.  When c1 is a multiple of 32 the concatenated rows exist only inside the GEMM's operand tiles. */
size_t rdm_decoder_stage_workspace_bytes(int64_t m, int64_t n, int64_t k);
int rdm_decoder_stage(const float* coarse, int64_t n_coarse, int64_t c1, int64_t ld1, const int64_t* idx, int64_t ldi,
                      const float* skip, int64_t c2, int64_t ld2, int64_t m, const float* w, int64_t ldw, const float* bias,
                      int64_t n, int groups, const float* gamma, const float* beta, float eps, int act, float* lin_out,
                      int64_t ld_lin, float* y, int64_t ldy, void* ws, size_t ws_bytes, void* stream);
int rdm_decoder_stage_form(const float* coarse, int64_t n_coarse, int64_t c1, int64_t ld1, const int64_t* idx, int64_t ldi,
                           const float* skip, int64_t c2, int64_t ld2, int64_t m, const float* w, int64_t ldw, const float* bias,
                           int64_t n, int groups, const float* gamma, const float* beta, float eps, int act, float* lin_out,
                           int64_t ld_lin, float* y, int64_t ldy, void* ws, size_t ws_bytes, int form, void* stream);  /* form: see rdm_gemm_form */
int rdm_layer_norm(const float* x, int64_t n, int64_t c, int64_t ldx, const float* residual,
                   int64_t ldr, const float* gamma, const float* beta, float eps, int act, float* y,
                   int64_t ldy, void* stream);
/* rdm_linear_layer_norm: y = act(LayerNorm(x W + bias [+ residual])) in one launch for the transformer width
 * (n = 128, k % 16 == 0, W [128, k] = the nn.Linear weight as stored, row stride ldw >= k): the Linear + residual
 * LayerNorm pairs of the attention layers
 * (rdmnet/thdroformer/thdroformer.py:159-173, modules/transformer/vanilla_transformer.py:87-103, output_layer.py:13-21). */
int rdm_linear_layer_norm(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int64_t m, int64_t n,
                          int64_t k, const float* residual, int64_t ldr, const float* gamma, const float* beta, float eps,
                          int act, float* y, int64_t ldy, void* stream);
/* rdm_attention_tail: everything of an attention layer after softmax(QK^T)V in one launch:
 *   y = LayerNorm(hidden Wo^T + bo + x); z = relu(y W1^T + b1); out = LayerNorm(z W2^T + b2 + y)
 * (rdmnet/thdroformer/thdroformer.py:142-173 RPEAttentionLayer / :159-173, geotransformer/modules/transformer/
 * vanilla_transformer.py:69-103, output_layer.py:6-21 AttentionOutput).  d = 128 with a 256-wide FFN; weights as
 * nn.Linear stores them: wo [128,128], w1 [256,128], w2 [128,256], row strides ld_* (multiples of 4), 16-byte aligned.
 * hidden, x, out: [m, 128] rows.  Other widths: rdm_linear_layer_norm / rdm_gemm + rdm_layer_norm.                  */
int rdm_attention_tail(const float* hidden, int64_t ld_hidden, const float* x, int64_t ldx, int64_t m, int64_t d,
                       const float* wo, int64_t ld_wo, const float* bo, const float* gamma1, const float* beta1,
                       const float* w1, int64_t ld_w1, const float* b1, const float* w2, int64_t ld_w2, const float* b2,
                       const float* gamma2, const float* beta2, float eps, float* out, int64_t ld_out, void* stream);
/* The same on weights stored once in OPERAND order (round 6): rdm_attention_tail_pack_weights writes wo | w1 | w2 as float4
 * [(wavefront, step), lane] into `packed` (rdm_attention_tail_packed_floats() floats, 16-byte aligned), so that every weight load
 * instruction of the kernel reads one contiguous KB instead of 16 rows x 64 B (28 k -> 19.5 k clocks per workgroup: the kernel
 * is bound by the CU's L2 fill path).  Same values in the same lanes: the bits of rdm_attention_tail.                        */
size_t rdm_attention_tail_packed_floats(void);
int rdm_attention_tail_pack_weights(const float* wo, int64_t ld_wo, const float* w1, int64_t ld_w1, const float* w2, int64_t ld_w2,
                                    float* packed, void* stream);
int rdm_attention_tail_packed(const float* hidden, int64_t ld_hidden, const float* x, int64_t ldx, int64_t m, int64_t d,
                              const float* packed, const float* bo, const float* gamma1, const float* beta1, const float* b1,
                              const float* b2, const float* gamma2, const float* beta2, float eps, float* out, int64_t ld_out,
                              void* stream);
int rdm_gather_max(const float* x, int64_t n_s, int64_t c, int64_t ldx, const int64_t* idx, int64_t m,
                   int64_t h, int64_t ldi, const int32_t* width, float* y, int64_t ldy, void* stream);
/* rdm_gather_rows: y[i,:] = x[idx[i],:] on raw 32-bit words, out-of-range index -> zero row (the
 * padded index_select of geotransformer/modules/ops/index_select.py:4-31 and boolean-mask selects). */
int rdm_gather_rows(const void* x, int64_t n_src, int64_t words, int64_t ldx, const int64_t* idx, int64_t m,
                    void* y, int64_t ldy, void* stream);
int rdm_upsample_concat(const float* coarse, int64_t n_coarse, int64_t c1, int64_t ld1,
                        const int64_t* idx, int64_t ldi, const float* skip, int64_t c2, int64_t ld2,
                        int64_t m, float* y, int64_t ldy, void* stream);

/* ---- a7: 3DRoFormer attention ------------------------------------------------------------------
 * rdm_rope: in-place learned rotary embedding of q (and k when non-NULL): pair p of row r is rotated
 *   by theta = 2*pi*sigmoid(emb[r, p]) (rdmnet/thdroformer/thdroformer.py:56-85; emb has d_model/2
 *   columns = heads x 16).
 * rdm_attention: out = softmax(q k^T / sqrt(head_dim)) v per head, heads packed along the columns
 *   (thdroformer.py:20-40 with k=None, :112-139; geotransformer/modules/transformer/
 *   vanilla_transformer.py:51-66).  head_dim must be 32.                                        */
int rdm_rope(float* q, int64_t ldq, float* k, int64_t ldk, const float* emb, int64_t lde, int64_t n,
             int64_t d_model, void* stream);
int rdm_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                  float* out, int64_t ldo, int64_t n_q, int64_t n_k, int heads, int head_dim, void* stream);
/* Same with Q, K, V and the probabilities rounded to bf16 for the two contractions (bf16 MFMA); logits,
 * softmax and accumulation in fp32 (BASELINE.json configs[3]: "bf16 attention + fp32 SVD").  Tensors in HBM
 * stay fp32. */
int rdm_attention_bf16(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                  float* out, int64_t ldo, int64_t n_q, int64_t n_k, int heads, int head_dim, void* stream);
/* rdm_attention_self_pair: the self-attention of both stacked clouds in one launch -- rows [0, n0) attend to rows
 * [0, n0), rows [n0, n0 + n1) to rows [n0, n0 + n1) (rdmnet/thdroformer/thdroformer.py:225-236 applies the self layer to
 * ref and src separately); bf16 != 0 selects the bf16-operand variant.  Same results as two rdm_attention calls.   */
int rdm_attention_self_pair(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            float* out, int64_t ldo, int64_t n0, int64_t n1, int heads, int head_dim, int bf16,
                            void* stream);

/* ---- a8/a9 helpers ------------------------------------------------------------------------------
 * rdm_vote_shift: xyz + clamp(offset[:, :3], +-limit) (rdmnet/vote/vote.py:98-108).
 * rdm_sigmoid_column: clamp(sigmoid(x[:, 0]), 0, 1) of a strided column (experiments/model_infer.py:161-162).
 * rdm_l2_normalize: F.normalize(x, p=2, dim=1) (experiments/model_infer.py:248-249).              */
int rdm_vote_shift(const float* xyz, const float* offsets, int64_t ldo, int64_t n, float lx, float ly,
                   float lz, float* out, void* stream);
int rdm_sigmoid_column(const float* x, int64_t ldx, int64_t n, float* out, void* stream);
int rdm_l2_normalize(const float* x, int64_t ldx, int64_t n, int64_t c, float* y, int64_t ldy, void* stream);

/* ---- a10: NMS -----------------------------------------------------------------------------------
 * keep[i] = 1 iff no lower-index neighbour of row i is kept: the greedy index-order sweep of
 * NMS.forward (rdmnet/vote/vote.py:33-40) given the radius-neighbour rows of the shifted nodes.
 * rdm_compact_indices: order-preserving list of kept rows in [begin, end) and their count (the
 * boolean-mask selects of experiments/model_infer.py:221-229).                                   */
int rdm_nms(const int64_t* idx, int64_t n, int64_t h, int64_t ldi, const int32_t* width, uint8_t* keep,
            void* stream);
int rdm_compact_indices(const uint8_t* keep, int64_t begin, int64_t end, int32_t* order, int32_t* count,
                        void* stream);

/* ---- a11: point-to-node grouping -----------------------------------------------------------------
 * Replaces point_to_node_partition (geotransformer/modules/ops/pointcloud_partition.py:60-107) for
 * one cloud: every point joins its nearest node (first minimum of the reference's fp32 distance
 * formula), every node keeps its k nearest OWN points ascending by (distance, index); unused slots
 * hold n_points / mask 0.  status != 0 if a node owns more than 4096 points.                     */
size_t rdm_point_to_node_workspace_bytes(int64_t n_points, int64_t n_nodes);
int rdm_point_to_node(const float* points, int64_t n_points, const float* nodes, int64_t n_nodes, int k,
                      int64_t* knn_idx, uint8_t* knn_mask, uint8_t* node_mask, int32_t* status, void* ws,
                      size_t ws_bytes, void* stream);
/* rdm_point_to_node_pair: the same grouping for the two clouds of a pair (a: ref, b: src) with one set of launches;
 * workspace: rdm_point_to_node_workspace_bytes(n_a, m_a) + rdm_point_to_node_workspace_bytes(n_b, m_b).         */
int rdm_point_to_node_pair(const float* points_a, int64_t n_a, const float* nodes_a, int64_t m_a, const float* points_b,
                           int64_t n_b, const float* nodes_b, int64_t m_b, int k, int64_t* knn_idx_a, uint8_t* knn_mask_a,
                           uint8_t* node_mask_a, int64_t* knn_idx_b, uint8_t* knn_mask_b, uint8_t* node_mask_b,
                           int32_t* status, void* ws, size_t ws_bytes, void* stream);

/* ---- a12: coarse matching -------------------------------------------------------------------------
 * Replaces SuperPointMatching.forward (geotransformer/modules/geotransformer/superpoint_matching.py:
 * 14-61).  scores holds f_ref . f_src^T on entry (rdm_gemm, trans_b) and the dual-normalised
 * matching scores on return; the k best (descending, ties by flat index) are written as node
 * indices + scores, *out_count = min(k, #non-empty pairs).                                        */
size_t rdm_coarse_matching_workspace_bytes(int64_t m, int64_t n);
int rdm_coarse_matching(float* scores, int64_t m, int64_t n, int64_t ld, const uint8_t* ref_mask,
                        const uint8_t* src_mask, int dual_normalization, int k, int64_t* ref_idx,
                        int64_t* src_idx, float* out_scores, int32_t* out_count, void* ws, size_t ws_bytes,
                        void* stream);

/* rdm_coarse_matching_features: the same stage from the L2-normalised superpoint features themselves
 * (superpoint_matching.py:14-61 with pairwise_distance(normalized=True), ops/pairwise_distance.py:4-31): dot
 * products, exp, row / column sums and the dual normalisation are evaluated in fp64 and rounded once to the
 * fp32 score that is ranked.  The reference's top-k order hangs on relative score gaps down to 4e-7
 * (tests/golden/coarse_order_analysis.json); this entry reproduces the reference's indices exactly when fed the
 * reference's features, which the fp32 pipeline above (GEMM, then rdm_coarse_matching) cannot promise.
 * ref_feats [m, d] (row stride ld_ref), src_feats [n, d]; d <= 448.                                         */
size_t rdm_coarse_matching_features_workspace_bytes(int64_t m, int64_t n);
int rdm_coarse_matching_features(const float* ref_feats, int64_t ld_ref, int64_t m, const float* src_feats,
                                 int64_t ld_src, int64_t n, int64_t d, const uint8_t* ref_mask,
                                 const uint8_t* src_mask, int dual_normalization, int k, int64_t* ref_idx,
                                 int64_t* src_idx, float* out_scores, int32_t* out_count, void* ws,
                                 size_t ws_bytes, void* stream);

/* ---- a14: Sinkhorn ---------------------------------------------------------------------------------
 * Replaces LearnableLogOptimalTransport.forward (geotransformer/modules/sinkhorn/
 * learnable_sinkhorn.py:13-66): scores [batch, m, n], masks [batch, m] / [batch, n] (1 = valid),
 * alpha = dustbin score (device scalar), out [batch, m+1, n+1].  m, n <= 128.                     */
int rdm_sinkhorn(const float* scores, int64_t batch, int64_t m, int64_t n, const uint8_t* row_mask,
                 const uint8_t* col_mask, const float* alpha, int iters, float* out, void* stream);

/* ---- a15/a16: local-to-global registration ----------------------------------------------------------
 * Replaces LocalGlobalRegistration.forward (geotransformer/modules/geotransformer/
 * local_global_registration.py:204-243; k=1, dustbin, non-mutual) including weighted_procrustes
 * (geotransformer/modules/registration/procrustes.py:6-73) without the reference's host SVD.
 * Outputs have capacity batch*2*side rows; counts[0..2] = {n_correspondences, n_hypotheses, best}. */
size_t rdm_lgr_workspace_bytes(int64_t batch);
int rdm_lgr(const float* log_scores, const float* ref_knn_points, const float* src_knn_points,
            const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int64_t batch, int64_t side,
            float acceptance_radius, int correspondence_threshold, int num_refinement_steps, float* ref_corr,
            float* src_corr, float* corr_scores, float* transform, int32_t* counts, void* ws, size_t ws_bytes,
            void* stream);

/* ---- native orchestration: one call per scan pair ----------------------------------------------
 * rdm_engine_run = the collate of geotransformer/utils/data.py:13-77 + RDMNet.forward of
 * experiments/model_infer.py:109-354 as a fixed sequence of the kernels above on one stream, with
 * activations in an engine-owned device arena (the ~700 launches of a pair cost ~2 us each from
 * native code instead of ~15 us each from Python).  Parameters are handed over by their reference
 * state-dict names (weights/rdmnet.pth.tar -> state['model'], engine/base_tester.py:97-107).
 * An engine is bound to the device that is current when it is created and must be used by one host
 * thread at a time; create one engine per in-flight pair.                                        */
typedef struct rdm_engine rdm_engine;

typedef struct rdm_engine_config {
  int num_stages;             /* 5 */
  int kernel_size;            /* 15 */
  int group_norm;             /* 32 */
  float init_voxel_size;      /* 0.3 */
  float init_radius;          /* 4.25 * 0.3 */
  float init_sigma;           /* 2.0 * 0.3 */
  int neighbor_limits[5];
  int out_dim;                /* 256 */
  int num_heads;              /* 4 */
  int num_layers;             /* 4 (self,cross) pairs, transformer #1 */
  int num_layers2;            /* 4, transformer #2 */
  int vote_mlp_layers;        /* 2 */
  float vote_limit[3];        /* 3.0 m */
  float nms_radius;           /* 2.4 m */
  int points_in_patch;        /* 128 */
  int num_correspondences;    /* 256 */
  int dual_normalization;     /* 1 */
  int sinkhorn_iterations;    /* 100 */
  float acceptance_radius;    /* 0.6 m */
  int correspondence_threshold; /* 3 */
  int num_refinement_steps;   /* 5 */
  int use_vote;               /* 1; 0 = infer.py:119-120 (Mulran): superpoints = un-shifted coarse points with the
                                 first transformer's features (the reference leaves this case undefined) */
  int attention_bf16;         /* 0 = fp32 QK^T / PV; 1 = bf16 operands, fp32 softmax + accumulation */
  size_t arena_bytes;         /* 0 = default (3 GiB) */
} rdm_engine_config;

typedef struct rdm_engine_result {
  float transform[16];        /* estimated_transform, row-major 4x4, src -> ref (host copy) */
  int32_t n_correspondences, n_hypotheses, best_hypothesis;
  int64_t n_ref_nodes, n_src_nodes, n_node_correspondences;
  int64_t level_sizes[5];        /* stacked [ref; src] points per pyramid level */
  int64_t level_ref_sizes[5];    /* of which ref */
  const float* ref_corr_points;  /* device, [n_correspondences, 3]; valid until the next run */
  const float* src_corr_points;
  const float* corr_scores;
  const float* transform_dev;
  size_t arena_used;
  /* the same correspondences on the HOST (engine-owned pinned memory, written by the run's last kernel; valid until the
   * next call on this engine): what infer.py:70-101 reads after `.cpu()`.  n_host_correspondences == n_correspondences
   * (the buffer holds the path's upper bound, num_correspondences x 2 x points_in_patch).                             */
  const float* host_ref_corr_points;  /* [n, 3] */
  const float* host_src_corr_points;  /* [n, 3] */
  const float* host_corr_scores;      /* [n] */
  int32_t n_host_correspondences;
} rdm_engine_result;

typedef struct rdm_tensor_view {
  void* data;                 /* device pointer into the engine arena (valid until the next run) */
  int64_t rows, cols, ld;     /* ld in elements */
  int dtype;                  /* 0 = f32, 1 = i64, 2 = u8, 3 = i32 */
} rdm_tensor_view;

typedef struct rdm_kpconv_profile {   /* one KPConv layer of the last run (HIP events on the run's stream) */
  int64_t m, h, c_in, c_out, pooled_channels;
  float gather_ms;            /* the neighbourhood kernel alone: rdm_kpconv_gather, or rdm_kpconv_fused (gather + weight
                                 contraction in one kernel) when `fused` */
  float total_ms;             /* whole layer: + weight GEMM (two-kernel form) or + GroupNorm passes (fused form), + the
                                 shortcut max-pool of strided blocks */
  int32_t fused;              /* 1: the layer ran as ONE kernel (c_in = 1, 32, 64) */
  int32_t reserved;
} rdm_kpconv_profile;

int rdm_engine_create(const rdm_engine_config* cfg, rdm_engine** out);
void rdm_engine_destroy(rdm_engine* e);
int rdm_engine_set_param(rdm_engine* e, const char* name, const float* data_host, const int64_t* shape_host, int ndim);
int rdm_engine_finalize(rdm_engine* e);
/* Instead of set_param + finalize: `e` uses the prepared device parameters of `src` (finalized, SAME device, same model-shape
 * fields of the configuration -- else RDM_ERR_ARG) -- one copy of the weights for all the engines a process keeps in flight.
 * The parameter set is reference counted: the engines may be destroyed in any order (the device memory goes with the last
 * user), and rdm_engine_finalize of `src` afterwards gives `src` a fresh set while its sharers keep the one they have.   */
int rdm_engine_share_params(rdm_engine* e, const rdm_engine* src);
/* ref/src points: device f32 [n,3].  Synchronises `stream` (4 small read-backs of data-dependent sizes). */
int rdm_engine_run(rdm_engine* e, const float* ref_points, int64_t n_ref, const float* src_points, int64_t n_src,
                   rdm_engine_result* result_host, void* stream);

/* The reference's `data_dict` (experiments/model_infer.py:113-124, produced by registration_collate_fn_stack_mode,
 * geotransformer/utils/data.py:139-192) as device pointers: what RDMNet.forward(data_dict) reads.  Clouds are stacked
 * [ref; src]; index tables are int64 row-major with the stated row stride (the reference hands over `[:, :limit]`
 * views), pad index = number of support points.  `*_count` are optional device int32 scalars holding the effective
 * table width min(limit, max_count) when a table was allocated wider than that (this library's own collate without
 * the shape read-back); null = every column is valid, as in the reference's tensors.                              */
typedef struct rdm_data_dict {
  const float* features;  int64_t features_ld;       /* [n_points[0], 1] */
  const float* points[5];                            /* [n_points[i], 3] contiguous */
  const int64_t* lengths[5];                         /* device int64[2] per level */
  int64_t n_points[5], n_ref[5];                     /* host copies: stacked rows per level, of which ref */
  const int64_t* neighbors[5];   int64_t neighbors_width[5],   neighbors_ld[5];   const int32_t* neighbors_count[5];
  const int64_t* subsampling[4]; int64_t subsampling_width[4], subsampling_ld[4]; const int32_t* subsampling_count[4];
  const int64_t* upsampling[4];  int64_t upsampling_width[4],  upsampling_ld[4];  const int32_t* upsampling_count[4];
  /* optional: the {max count, status} words of the searches that built the tables (rdm_engine_collate's "search_flags",
   * device int32 [n_collate_status, 2]).  A non-zero status word makes rdm_engine_forward return RDM_ERR_CAPACITY at its
   * first read-back instead of computing on a broken table; null / 0 for tables from another collate.              */
  const int32_t* collate_status; int64_t n_collate_status;
} rdm_data_dict;

/* rdm_engine_collate = the collate alone (registration_collate_fn_stack_mode / precompute_data_stack_mode,
 * geotransformer/utils/data.py:13-77,139-192) as ONE native call: four subsamplings + 13 searches; the tables stay in the
 * engine's arena as stage tensors "points0..4", "lengths0..4" (i64 [1,2]), "neighbors0..4", "subsampling0..3",
 * "upsampling0..3" (i64 [n, limit]; effective widths in "search_flags", i32 [32,2] = {max count, status} per search in the
 * order self / sub / up per level) -- fetch them with rdm_engine_export.  result_host receives the level sizes.      */
int rdm_engine_collate(rdm_engine* e, const float* ref_points, int64_t n_ref, const float* src_points, int64_t n_src,
                       rdm_engine_result* result_host, void* stream);

/* rdm_engine_forward = RDMNet.forward(data_dict) alone (experiments/model_infer.py:109-354): the same native sequence
 * as rdm_engine_run after its collate, on tables the caller built -- with this library's collate
 * (rdmnet_amd.collate) or the reference's.  Same result structure, taps and waits as rdm_engine_run; bit-identical to
 * it when fed the tables rdm_engine_run builds itself.                                                              */
int rdm_engine_forward(rdm_engine* e, const rdm_data_dict* data, rdm_engine_result* result_host, void* stream);

/* Batched collate (round 5): the collates of `n_pairs` (1 .. 16) pairs -- precompute_data_stack_mode, data.py:13-77, once per
 * pair in the reference's DataLoader workers -- as ONE sequence of launches: every subsampling launch works on 2 * n_pairs
 * clouds, the search grids are built as (pair, level) items, the 12 searches a plain run needs per pair are flushed 16 per
 * launch, the level sizes of all pairs return in one read-back.  The pyramids stay in the engine's arena;
 * rdm_engine_forward_batched(e, k, ...) then runs RDMNet.forward (model_infer.py:109-354) of pair k on them: result structure,
 * waits and read-outs of rdm_engine_run, and the same bits (every pair's tables, points and query order are what its own
 * collate writes).  ref_points / src_points: host arrays of device pointers, float32 [n, 3] each; the clouds must stay valid
 * until the collate has run on `stream`.  Not with rdm_engine_keep_taps (a run that keeps its stage tensors builds the
 * reference's full tables).  Any other run on the engine discards the batch.                                          */
int rdm_engine_collate_batch(rdm_engine* e, int n_pairs, const float* const* ref_points, const int64_t* n_ref,
                             const float* const* src_points, const int64_t* n_src, void* stream);
int rdm_engine_forward_batched(rdm_engine* e, int k, rdm_engine_result* result_host, void* stream);

/* Lock step (round 5, experimental): rdm_engine_run of `n_pairs` (1 .. 8) pairs on as many engines -- one arena and one
 * result buffer each, the weights shared (rdm_engine_share_params) -- on ONE stream: the runs advance together on the calling
 * thread, the launches of the same kernel of all pairs go out as one grouped launch, the size read-backs of the pairs become
 * waits of the group.  What the reference does pair after pair (engine/single_tester.py:86-134) and this library otherwise does on
 * one stream per pair.  Every pair: the bits of rdm_engine_run on it alone.  collate_batched != 0: the collates of the group run
 * as one launch sequence on engines[0] (rdm_engine_collate_batch) before the forwards run in lock step (not when engines[0] keeps
 * its stage tensors).  A pair that exhausts its arena is run again on its own (the arena grows as in rdm_engine_run).        */
int rdm_engine_run_lockstep(rdm_engine* const* engines, int n_pairs, const float* const* ref_points, const int64_t* n_ref,
                            const float* const* src_points, const int64_t* n_src, rdm_engine_result* const* results,
                            int collate_batched, void* stream);
/* rdm_engine_forward of `n_pairs` (1 .. 8) callers' data_dicts on as many engines in lock step (round 6): the drop-in operator
 * API -- model(data_dict), experiments/model_infer.py:109-354 -- for several pairs on one stream; with rdm_engine_keep_taps every
 * engine holds the stage tensors of ITS pair afterwards (rdm_engine_export per engine).  Every pair: the bits of
 * rdm_engine_forward on it alone.                                                                                            */
int rdm_engine_forward_lockstep(rdm_engine* const* engines, int n_pairs, const rdm_data_dict* const* data,
                                rdm_engine_result* const* results, void* stream);
/* rdm_engine_collate of `n_pairs` (1 .. 8) pairs on as many engines in lock step: what the reference's DataLoader workers do pair
 * by pair (registration_collate_fn_stack_mode, geotransformer/utils/data.py:139-192); every engine then holds its pair's
 * pyramid and 13 tables as stage tensors, exactly as after rdm_engine_collate on the pair alone.                              */
int rdm_engine_collate_lockstep(rdm_engine* const* engines, int n_pairs, const float* const* ref_points, const int64_t* n_ref,
                                const float* const* src_points, const int64_t* n_src, rdm_engine_result* const* results, void* stream);
/* DIAGNOSTIC, process-global (not part of the stateless compute ABI): developer counters of the lock-step scheduler
 * (tools/lockstep_lab.py): out[0..5] = ns spent in lock-step runs, ns of them in
 * host waits, waits, grouped launches, records carried, runs (summed over threads; reset != 0 clears them).  In the lab build
 * (make lab), with RDM_LOCKSTEP_STATS in the environment, rdm_lockstep_stats_dump prints launches and records per kernel.    */
void rdm_lockstep_stats(long long* out, int reset);
/* DIAGNOSTIC: self-test of the lock-step scheduler on scripted stand-in launches (no GPU needed; tests/test_lockstep.py). */
int rdm_lockstep_selftest(int n_ctx, const int* script, int len, int* log, int cap, int* rcs);
void rdm_lockstep_stats_dump(void);
/* Stage intermediates by name (test/inspection aid): enable before a run, query after it. */
/* Per-KPConv-layer HIP-event timing of the last run; get_profile returns the number of layers. */
/* Re-allocates the engine's activation arena at `bytes`, growable (rdm_engine_config.arena_bytes != 0 fixes the size instead; the
 * default is 3 GiB, doubled and the pair re-run when a pair exhausts it).  For callers that know their largest pair or lock-step
 * group; the reference has no counterpart (torch's caching allocator).  Waits for the device; drops a collated batch.           */
int rdm_engine_reserve(rdm_engine* e, size_t bytes);
/* How rdm_engine_run waits for its stream at the size read-backs: sleep_us = 0 (default) uses hipStreamSynchronize,
 * which spins a host core per in-flight pair; sleep_us > 0 polls hipStreamQuery and sleeps that long in between (for
 * hosts whose CPU quota is smaller than ranks x pairs in flight). */
int rdm_engine_set_wait(rdm_engine* e, int sleep_us);
/* How many scan pairs the caller keeps in flight on this GPU (one engine and stream each; default 1).  A scheduling hint, results do
 * not depend on it: from 3 the tiled GEMM leaves half of a CU's registers and LDS to the other pairs' kernels (two workgroups
 * per CU instead of four: +3 % pairs/s at four in flight, -2 % with one pair alone, docs/EXPERIMENTS.md 5d).                            */
int rdm_engine_set_pairs_in_flight(rdm_engine* e, int n);
/* Latency mode.  With one pair in flight most of the GPU idles while chains of one-workgroup kernels run, so the engine runs the
 * wide, independent parts of a pair -- the first level's grid, neighbour search and encoder blocks beside the subsampling of the
 * deeper levels; the decoder beside the second transformer / grouping / coarse-matching chain -- on a side stream of its own
 * (created on first use) and joins them by host waits next to read-backs the run performs anyway.  Same kernels on the same
 * operands: results are bit-identical in every mode.  mode: 0 = never, 1 = when pairs_in_flight == 1 (default), 2 = always.
 * Not used when `stream` is the null stream; the first half only when `stream` is idle at the call.  The side stream must lie on
 * another hardware pipe than `stream` (two queues of one pipe are served in turns); the engine finds that out with a pair of 60 us
 * spin kernels the first time it sees a caller stream (a few hundred microseconds, once) and stays serial where no such stream exists.
 * The reference has no counterpart (its loop is synchronous: geotransformer/engine/single_tester.py:86-134).                    */
int rdm_engine_set_overlap(rdm_engine* e, int mode);
/* Per-KPConv-layer profile of the next runs: 0 = off, 1 = HIP events around every layer's neighbourhood kernel and around the
 * whole layer (rdm_engine_get_profile: sizes + milliseconds), 2 = the layers' sizes only, no events -- for the other pairs of a
 * lock-step group whose first engine records the events: the launches (and durations) are the group's.                     */
int rdm_engine_enable_profile(rdm_engine* e, int enable);
int rdm_engine_get_profile(rdm_engine* e, rdm_kpconv_profile* out, int cap);
int rdm_engine_keep_taps(rdm_engine* e, int enable);
int rdm_engine_get_tensor(rdm_engine* e, const char* name, rdm_tensor_view* out);
/* Copies n stage tensors of the last run into caller buffers (dst[i]: rows * ld * element size bytes of names[i]) with
 * one batched launch: how model(data_dict) fills the reference's output_dict (model_infer.py:133-334).           */
int rdm_engine_export(rdm_engine* e, int n, const char* const* names, void* const* dst, void* stream);
/* rdm_engine_get_tensor for n names at once (out[n]). */
int rdm_engine_describe(rdm_engine* e, int n, const char* const* names, rdm_tensor_view* out);
/* Plain device-to-device copy on `stream` (lets a host without a HIP binding read arena tensors). */
int rdm_copy_device(void* dst, const void* src, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RDMNET_HIP_H_ */
