/* librdmnet_hip.so -- C-ABI of the MI355X (gfx950) implementation of RDMNet's dense-matching
 * inference path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     unless the function's comment says so;
 *   - `ws`/`ws_bytes` is caller-owned scratch (query the size with the matching *_workspace_bytes);
 *     the library never allocates or frees caller memory and keeps no global state (re-entrant);
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

#define RDM_ABI_VERSION 1

int rdm_abi_version(void);
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
 *   status     [1] i32, set non-zero if a query had more neighbours than the kernel's capacity
 * The reference's output width is max(count); call once with width = 0 to obtain it, or pass the
 * neighbour limit directly (the first min(limit, max) columns are identical).                  */
size_t rdm_radius_neighbors_workspace_bytes(int64_t n_q, int64_t n_s, int batch);
int rdm_radius_neighbors(const float* q_points, int64_t n_q, const float* s_points, int64_t n_s,
                         const int64_t* q_lengths, const int64_t* s_lengths, int batch,
                         float radius, int width, int64_t* out_idx, int32_t* out_counts,
                         int32_t* out_max, int32_t* status, void* ws, size_t ws_bytes,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RDMNET_HIP_H_ */
