// TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's two native operators.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product path (rdmnet_amd/) never does.
//
// Parity status: PINNED.  tests/test_oracle_native.py checks every function below against
//   (1) oracle/_ref/libref_ext.so = the reference's own C++ compiled from /root/reference, and
//   (2) golden vectors in tests/golden/ captured from the reference's pybind module.
//
// Build: `make -C oracle` (g++ -O2 -ffp-contract=off: the arithmetic below must not be fused).
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct Cell {
  int n = 0;
  float sx = 0.f, sy = 0.f, sz = 0.f;
};

// One cloud of the stacked batch.
// Follows geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-48:
//   :9-11  bounding box, origin = floor(min * (float)(1/v)) * v      (cloud.h:92-98 operator*, :104-106 floor)
//   :13-20 nX, nY = floor((max - origin) / v) + 1
//   :28-42 key = iX + nX*iY + nX*nY*iZ with i = floor((p - origin) / v) in fp32; running fp32 sum + count
//   :44-47 emit sum * (float)(1.0 / count) in std::unordered_map iteration order
void subsample_one(const float* p, int64_t n, float v, std::vector<float>& out) {
  if (n <= 0) return;
  float lo[3] = {p[0], p[1], p[2]}, hi[3] = {p[0], p[1], p[2]};
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      float x = p[3 * i + d];
      if (x < lo[d]) lo[d] = x;
      if (x > hi[d]) hi[d] = x;
    }
  const float inv = static_cast<float>(1. / v);  // the double quotient is narrowed by operator*(PointXYZ, float)
  float org[3];
  for (int d = 0; d < 3; ++d) org[d] = std::floor(lo[d] * inv) * v;
  const std::size_t nx = static_cast<std::size_t>(std::floor((hi[0] - org[0]) / v) + 1);
  const std::size_t ny = static_cast<std::size_t>(std::floor((hi[1] - org[1]) / v) + 1);

  std::unordered_map<std::size_t, Cell> cells;  // iteration order of libstdc++ IS the output order
  for (int64_t i = 0; i < n; ++i) {
    const std::size_t ix = static_cast<std::size_t>(std::floor((p[3 * i + 0] - org[0]) / v));
    const std::size_t iy = static_cast<std::size_t>(std::floor((p[3 * i + 1] - org[1]) / v));
    const std::size_t iz = static_cast<std::size_t>(std::floor((p[3 * i + 2] - org[2]) / v));
    Cell& c = cells[ix + nx * iy + nx * ny * iz];
    c.n += 1;
    c.sx += p[3 * i + 0];
    c.sy += p[3 * i + 1];
    c.sz += p[3 * i + 2];
  }
  for (auto& kv : cells) {
    const float w = static_cast<float>(1.0 / kv.second.n);
    out.push_back(kv.second.sx * w);
    out.push_back(kv.second.sy * w);
    out.push_back(kv.second.sz * w);
  }
}

}  // namespace

extern "C" {

// grid_subsampling (geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62,
// batch loop grid_subsampling_cpu.cpp:50-75).  *out_points is malloc'ed; free with oracle_free.
int64_t oracle_grid_subsampling(const float* points, int64_t n_points, const int64_t* lengths,
                                int64_t batch, float voxel, float** out_points,
                                int64_t* out_lengths) {
  (void)n_points;
  std::vector<float> out;
  int64_t start = 0;
  for (int64_t b = 0; b < batch; ++b) {
    std::size_t before = out.size();
    subsample_one(points + 3 * start, lengths[b], voxel, out);
    out_lengths[b] = static_cast<int64_t>((out.size() - before) / 3);
    start += lengths[b];
  }
  float* buf = static_cast<float*>(std::malloc(sizeof(float) * (out.size() + 3)));
  std::memcpy(buf, out.data(), sizeof(float) * out.size());
  *out_points = buf;
  return static_cast<int64_t>(out.size() / 3);
}

// radius_neighbors (geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:5-68,
// radius_neighbors_cpu.cpp:3-91).  Brute force restatement of the kd-tree search:
//   metric   nanoflann.hpp:435-441  d2 = ((dx*dx) + (dy*dy)) + (dz*dz), fp32, diff = query - support
//   accept   nanoflann.hpp:249-250  strict d2 < r2, r2 = radius*radius in fp32 (radius_neighbors_cpu.cpp:12)
//   order    nanoflann.hpp:1280-1289 ascending d2; ties are unordered there, canonical here = (d2, index)
//   layout   radius_neighbors_cpu.cpp:66-90 global support index, pad = total support count,
//            width = max count over ALL queries of the batch
// Returns the width; *out_idx is malloc'ed [nq, width] int64; counts (optional) gets per-query counts.
int64_t oracle_radius_neighbors(const float* q, int64_t nq, const float* s, int64_t ns,
                                const int64_t* q_lengths, const int64_t* s_lengths, int64_t batch,
                                float radius, int64_t** out_idx, int32_t* counts) {
  const float r2 = radius * radius;
  std::vector<std::vector<std::pair<float, int64_t>>> hits(static_cast<std::size_t>(nq));
  std::vector<int64_t> q_cloud(static_cast<std::size_t>(nq)), s_begin(batch + 1, 0);
  {
    int64_t qi = 0;
    for (int64_t b = 0; b < batch; ++b) {
      for (int64_t k = 0; k < q_lengths[b]; ++k) q_cloud[qi++] = b;
      s_begin[b + 1] = s_begin[b] + s_lengths[b];
    }
  }
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < nq; ++i) {
    const int64_t b = q_cloud[i];
    const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
    auto& h = hits[i];
    for (int64_t j = s_begin[b]; j < s_begin[b + 1]; ++j) {
      const float dx = qx - s[3 * j], dy = qy - s[3 * j + 1], dz = qz - s[3 * j + 2];
      float d2 = dx * dx;
      d2 = d2 + dy * dy;
      d2 = d2 + dz * dz;
      if (d2 < r2) h.emplace_back(d2, j);
    }
    std::sort(h.begin(), h.end());
  }
  std::size_t width = 0;
  for (auto& h : hits) width = std::max(width, h.size());
  int64_t* buf = static_cast<int64_t*>(std::malloc(sizeof(int64_t) * (nq * width + 1)));
  for (int64_t i = 0; i < nq; ++i) {
    if (counts) counts[i] = static_cast<int32_t>(hits[i].size());
    for (std::size_t c = 0; c < width; ++c)
      buf[i * width + c] = c < hits[i].size() ? hits[i][c].second : ns;
  }
  *out_idx = buf;
  return static_cast<int64_t>(width);
}

// Squared distances of the listed neighbours with the metric above (used by tests to canonicalise
// the kd-tree's arbitrary order among exact ties).
void oracle_neighbor_d2(const float* q, int64_t nq, const float* s, int64_t ns, const int64_t* idx,
                        int64_t width, float* d2_out) {
  for (int64_t i = 0; i < nq; ++i)
    for (int64_t c = 0; c < width; ++c) {
      const int64_t j = idx[i * width + c];
      if (j >= ns) {
        d2_out[i * width + c] = INFINITY;
        continue;
      }
      const float dx = q[3 * i] - s[3 * j], dy = q[3 * i + 1] - s[3 * j + 1],
                  dz = q[3 * i + 2] - s[3 * j + 2];
      float d2 = dx * dx;
      d2 = d2 + dy * dy;
      d2 = d2 + dz * dz;
      d2_out[i * width + c] = d2;
    }
}

void oracle_free(void* p) { std::free(p); }

}  // extern "C"
