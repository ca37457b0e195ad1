// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// C-ABI shim around the *reference's own* native sources (compiled where they lie under
// /root/reference; nothing is copied).  Built by oracle/Makefile into oracle/_ref/libref_ext.so.
// It exposes the two functions the reference exports through pybind
// (geotransformer/extensions/pybind.cpp:6-17) with plain pointers, so tests and
// bench.py's cpu_baseline leg can call the real nanoflann / unordered_map code.
//
// Wrapped reference entry points:
//   grid_subsampling_cpu   geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:50-75
//   radius_neighbors_cpu   geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cpu/grid_subsampling/grid_subsampling_cpu.h"
#include "cpu/radius_neighbors/radius_neighbors_cpu.h"

extern "C" {

// Returns the number of subsampled points; *out_points is malloc'ed (caller frees with ref_free).
int64_t ref_grid_subsampling(const float* points, int64_t n_points, const int64_t* lengths,
                             int64_t batch, float voxel, float** out_points, int64_t* out_lengths) {
  std::vector<PointXYZ> pts(reinterpret_cast<const PointXYZ*>(points),
                            reinterpret_cast<const PointXYZ*>(points) + n_points);
  std::vector<long> lens(lengths, lengths + batch);
  std::vector<PointXYZ> s_pts;
  std::vector<long> s_lens;
  grid_subsampling_cpu(pts, s_pts, lens, s_lens, voxel);
  for (int64_t b = 0; b < batch; ++b) out_lengths[b] = s_lens[b];
  float* buf = static_cast<float*>(std::malloc(sizeof(float) * 3 * (s_pts.size() + 1)));
  std::memcpy(buf, s_pts.data(), sizeof(float) * 3 * s_pts.size());
  *out_points = buf;
  return static_cast<int64_t>(s_pts.size());
}

// Returns max_count (row width); *out_idx is malloc'ed [nq, max_count] int64.
int64_t ref_radius_neighbors(const float* q, int64_t nq, const float* s, int64_t ns,
                             const int64_t* q_lengths, const int64_t* s_lengths, int64_t batch,
                             float radius, int64_t** out_idx) {
  std::vector<PointXYZ> qv(reinterpret_cast<const PointXYZ*>(q),
                           reinterpret_cast<const PointXYZ*>(q) + nq);
  std::vector<PointXYZ> sv(reinterpret_cast<const PointXYZ*>(s),
                           reinterpret_cast<const PointXYZ*>(s) + ns);
  std::vector<long> ql(q_lengths, q_lengths + batch);
  std::vector<long> sl(s_lengths, s_lengths + batch);
  std::vector<long> idx;
  radius_neighbors_cpu(qv, sv, ql, sl, idx, radius);
  int64_t width = nq > 0 ? static_cast<int64_t>(idx.size() / nq) : 0;
  int64_t* buf = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (idx.size() + 1)));
  std::memcpy(buf, idx.data(), sizeof(int64_t) * idx.size());
  *out_idx = buf;
  return width;
}

void ref_free(void* p) { std::free(p); }

}  // extern "C"
