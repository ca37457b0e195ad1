// Cross-translation-unit helpers of librdmnet_hip.so (not part of the C-ABI).
#pragma once
#include <cstddef>
#include <cstdint>

namespace rdm {

// GEMM with an optional GroupNorm-statistics epilogue.  When `gn_partial` is non-null and the GEMM
// does not run split-K, every block row writes per-column (sum, sum of squares) of ITS output rows to
// gn_partial[(block_row*2 + {0,1}) * n + col] (fp64) and *gn_blocks receives the number of block rows;
// otherwise *gn_blocks = 0 and the caller must compute the statistics itself.
int gemm_with_stats(const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc, int64_t m,
                    int64_t n, int64_t k, const float* bias, const float* rowdiv, void* ws, size_t ws_bytes,
                    double* gn_partial, int* gn_blocks, void* stream, int form = 0);

// gemm_with_stats with A = [coarse[idx[:, 0]] | skip] formed inside the kernel (the decoder's upsample + concatenation,
// backbone.py:118-151); returns 1 (nothing launched) when the shapes need the materialised concatenation instead.
void gemm_set_lds_pad(unsigned bytes);
int gemm_concat_with_stats(const float* coarse, int64_t ld1, int64_t c1, int64_t n_coarse, const int64_t* idx, int64_t ldi,
                           const float* skip, int64_t ld2, int64_t c2, const float* b, int64_t ldb, float* c, int64_t ldc,
                           int64_t m, int64_t n, const float* bias, int act, void* ws, size_t ws_bytes, double* gn_partial,
                           int* gn_blocks, void* stream, int form = 0);

// Two independent products C_i = A_i B_i + bias_i (rdm_gemm semantics, no activation) in one launch when both are
// transformer-sized; otherwise two rdm_gemm calls.
int gemm_pair(const float* a0, int64_t lda0, const float* b0, int64_t ldb0, float* c0, int64_t ldc0, int64_t m0, int64_t n0,
              int64_t k0, const float* bias0, const float* a1, int64_t lda1, const float* b1, int64_t ldb1, float* c1,
              int64_t ldc1, int64_t m1, int64_t n1, int64_t k1, const float* bias1, void* ws, size_t ws_bytes, void* stream);

// Upper bound of block rows gemm_with_stats can produce for m rows.
// (64-row tiles without split-K; the split-K reduce uses as few as 8 rows per block)
inline int64_t gemm_stats_max_blocks(int64_t m) { return (m + 7) / 8 + 1; }

// rdm_grid_subsample with the form chosen by the caller: mode 0 = by size (rdm_grid_subsample), 1 = the single-workgroup
// kernel, 2 = the multi-launch form (phases spread over the GPU, large clouds).  Same output bit for bit.
int grid_subsample_mode(const float* points, int64_t n_points, const int64_t* lengths, int batch, float voxel_size,
                        float* out_points, int64_t* out_lengths, void* ws, size_t ws_bytes, void* stream, int mode);

// GroupNorm given (optional) precomputed partials: nblk > 0 uses them, nblk == 0 computes them.  form: 0 = the library's choice,
// 1 = statistics, finalize and apply as separate launches everywhere (rdm_group_norm_form).
int group_norm_finish(const double* partial, int nblk, const float* x, int64_t n, int64_t c, int64_t ldx, int groups,
                      const float* gamma, const float* beta, float eps, const float* residual, int64_t ldr, int act,
                      float* y, int64_t ldy, uint8_t* positive, void* ws, size_t ws_bytes, void* stream, int form = 0);

// Radius searches batched: a call only records the search in `queue` (radius_redo_queue_bytes() bytes of host memory,
// reset once; up to 16 searches -- further ones run at once); radius_redo_flush launches TWO kernels for all recorded
// searches: the first pass, and the second pass (queries with more than 256 neighbours, redone with the large buffer).
// The grids and all arguments must stay valid until the flush.  redo_flags: n_q bytes of device memory
// that must stay valid until the flush (rdm_radius_grid_query keeps them in its scratch and runs both passes at once).
// Up to four rdm_gather_rows in one launch (independent gathers: none may read another's output).
int gather_rows_multi(int n, const void* const* x, const int64_t* n_src, const int64_t* words, const int64_t* ldx,
                      const int64_t* const* idx, const int64_t* m, void* const* y, const int64_t* ldy, void* stream);

// rdm_radius_grid_build for up to 8 support clouds (the levels of a pair) with one set of launches.
int radius_grid_build_multi(int n, const float* const* s_points, const int64_t* n_s, const int64_t* const* s_lengths, int batch,
                            const float* radius, void* const* grid_ws, const size_t* grid_ws_bytes, void* stream);

// A search "grid" of one cell per cloud (brute force: every query tests every point of its cloud) in ONE launch, for support sets
// of a few hundred points; same workspace layout and queries as rdm_radius_grid_build (any radius).
int radius_grid_build_trivial(const float* s_points, int64_t n_s, const int64_t* s_lengths, int batch, void* grid_ws,
                              size_t grid_ws_bytes, void* stream);

// rdm_compact_indices for the rows [0, n_ref) and [n_ref, n) with one launch (order + 0 / order + n_ref, counts[0 / 1]);
// optionally mirrors `mirror_words` int32 status words (which must contain `counts`) into mapped host memory.
int compact_indices_pair(const uint8_t* keep, int64_t n_ref, int64_t n, int32_t* order, int32_t* counts,
                         const int32_t* mirror_src, int32_t* mirror_dst, int mirror_words, void* stream);

// rdm_gather_max visiting the queries in the order of `order_records` (the cell-sorted float4 records of the query level's
// search grid, rdm_radius_grid_records; null = row order): same output, better L2 locality of the gathered rows.
// i32 (here and below): the neighbour table `idx` holds int32 elements instead of the C-ABI's int64 -- rdm_engine_run keeps the
// tables it builds AND consumes itself in 32 bits (half the bytes written by the searches and read by every KPConv layer and
// shortcut pool); every public entry point passes 0, the int64 layout of the reference.
int gather_max_ordered(const float* x, int64_t n_s, int64_t c, int64_t ldx, const int64_t* idx, int64_t m, int64_t h, int64_t ldi,
                       const int32_t* width, float* y, int64_t ldy, const float* order_records, int i32, void* stream);
// rdm_kpconv_gather_ordered / rdm_kpconv_fused_form on a table of either element width
int kpconv_gather_impl(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats, int64_t c,
                       int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                       const float* kernel_points, float sigma, float* wf, int64_t ldw, float* nn, const float* order_records,
                       int i32, void* stream, int form = 0);
// The aggregation of kpconv_gather_impl with the support rows of 16 cell-ordered queries staged once in LDS (kpconv_fused.hip:
// kpconv_tile_kernel<64, GATHER>, one 64-channel slice per workgroup): same WF / nn bits, fewer L2 -> CU line fills.
bool kpconv_tile_gather_applies(int64_t c, int64_t h, bool has_order);
int kpconv_tile_gather(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats, int64_t c,
                       int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                       const float* kernel_points, float sigma, float* wf, int64_t ldw, float* nn, const float* order_records,
                       int i32, void* stream);
int kpconv_fused_impl(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats, int64_t c,
                      int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                      const float* kernel_points, float sigma, const float* w_packed, const float* bias, int64_t c_out, float* out,
                      int64_t ldo, double* gn_partial, const float* order_records, int form, int i32, void* stream);

size_t radius_redo_queue_bytes();
void radius_redo_queue_reset(void* queue);
int radius_grid_query_deferred(void* grid_ws, size_t grid_ws_bytes, int64_t n_s, const float* q_points, int64_t n_q,
                               const int64_t* q_lengths, int batch, float radius, int width, int64_t* out_idx,
                               int32_t* out_counts, int32_t* out_max, int32_t* status, unsigned char* redo_flags, void* queue,
                               int i32, void* stream);
int radius_redo_flush(void* queue, void* stream);

}  // namespace rdm
