// SURVEY.md §8f rank 3 -- raw-scan preprocessing: centroid voxel down-sampling of a ~120 k-point scan
// (xyz + intensity) to the 16-20 k-point clouds the path consumes.
//
// Replaces preporcess/downsample_pcd_kitti.py:21-36, i.e. Open3D 0.11.2's PointCloud::VoxelDownSample(0.3)
// on points and "colors" (the intensity replicated three times).  Open3D itself is NOT under /root/reference
// (requirements.txt:8), so parity is UNPINNED; the arithmetic restated here is Open3D's published one:
//   voxel_min_bound = min_bound - 0.5 * voxel      (float64)
//   index           = floor((p - voxel_min_bound) / voxel) per axis   (float64, points widened from fp32)
//   output          = per-voxel mean of points and of intensities, accumulated in float64
// Open3D emits voxels in the iteration order of its unordered_map<Vector3i> (unspecified); here the order is
// the FIRST-OCCURRENCE order of the voxels in the input, and sums run in ascending point order, so the
// result is a deterministic function of the input (bit-identical to oracle/preprocess.py).
//
// Multi-kernel, grid-wide phases (a raw scan is too large for the one-workgroup-per-cloud scheme of a1):
//   bbox -> keys + hash de-dup (first occurrence per voxel) -> scan(first flags) = voxel ranks ->
//   per-voxel counts -> scan(counts) -> fill per-voxel point lists -> sort each short list, sum in fp64.
#include "../../include/rdmnet_hip.h"
#include "common.h"

#pragma clang fp contract(off)

namespace {
using namespace rdm;

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kScanBlock = 1024;
constexpr int kScanItems = 4;  // per thread -> 4096 items per block

__device__ __forceinline__ unsigned ordered(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ __launch_bounds__(256) void vd_bbox_kernel(const float* __restrict__ pts, int64_t n, int ld,
                                                       unsigned* __restrict__ min_ord) {
  __shared__ unsigned red[4][3];
  unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x)
    for (int d = 0; d < 3; ++d) lo[d] = min(lo[d], ordered(pts[i * ld + d]));
  for (int d = 0; d < 3; ++d)
    for (int o = 32; o > 0; o >>= 1) lo[d] = min(lo[d], static_cast<unsigned>(__shfl_xor(static_cast<int>(lo[d]), o, 64)));
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) red[threadIdx.x >> 6][d] = lo[d];
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned m = min(min(red[0][threadIdx.x], red[1][threadIdx.x]), min(red[2][threadIdx.x], red[3][threadIdx.x]));
    atomicMin(min_ord + threadIdx.x, m);
  }
}

__global__ __launch_bounds__(256) void vd_key_kernel(const float* __restrict__ pts, int64_t n, int ld, double voxel,
                                                      const unsigned* __restrict__ min_ord,
                                                      unsigned long long* __restrict__ ht_keys,
                                                      unsigned* __restrict__ ht_first, unsigned ht_mask,
                                                      unsigned* __restrict__ pt_slot, int32_t* __restrict__ status) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  unsigned long long key = 0;
  for (int d = 0; d < 3; ++d) {
    const double lo = static_cast<double>(unordered(min_ord[d])) - voxel * 0.5;
    const double c = floor((static_cast<double>(pts[i * ld + d]) - lo) / voxel);
    if (!(c >= 0.0 && c < 2097152.0)) {  // 21 bits per axis
      atomicExch(status, 1);
      pt_slot[i] = 0xffffffffu;
      return;
    }
    key |= static_cast<unsigned long long>(c) << (21 * d);
  }
  unsigned slot = static_cast<unsigned>((key * 0x9E3779B97F4A7C15ull) >> 38) & ht_mask;
  while (true) {
    const unsigned long long prev = atomicCAS(&ht_keys[slot], kEmptyKey, key);
    if (prev == kEmptyKey || prev == key) break;
    slot = (slot + 1) & ht_mask;
  }
  atomicMin(&ht_first[slot], static_cast<unsigned>(i));
  pt_slot[i] = slot;
}

// ---- device-wide exclusive scan of int32 in three launches (block scan, scan of block totals, add) ----
__device__ __forceinline__ int block_exclusive(int v, int* lds, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  if (w == 0) {
    const int x = lane < kScanBlock / 64 ? lds[lane] : 0;
    int s = x;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(s, o, 64);
      if (lane >= o) s += t;
    }
    if (lane < kScanBlock / 64) lds[lane] = s - x;
    if (lane == kScanBlock / 64 - 1) lds[kScanBlock / 64] = s;
  }
  __syncthreads();
  const int r = inc - v + lds[w];
  total = lds[kScanBlock / 64];
  __syncthreads();
  return r;
}

// FLAGS = true: the scanned value of item i is (ht_first[pt_slot[i]] == i), computed on the fly
template <bool FLAGS>
__global__ __launch_bounds__(kScanBlock) void vd_scan_blocks_kernel(const int* __restrict__ in, const unsigned* __restrict__ ht_first,
                                                                     const unsigned* __restrict__ pt_slot, int64_t n,
                                                                     int* __restrict__ out, int* __restrict__ block_tot) {
  __shared__ int lds[kScanBlock / 64 + 2];
  const int64_t base = (blockIdx.x * static_cast<int64_t>(kScanBlock) + threadIdx.x) * kScanItems;
  int v[kScanItems], local = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    int x = 0;
    if (i < n) {
      if (FLAGS) {
        const unsigned s = pt_slot[i];
        x = (s != 0xffffffffu && ht_first[s] == static_cast<unsigned>(i)) ? 1 : 0;
      } else {
        x = in[i];
      }
    }
    v[k] = x;
    local += x;
  }
  int total;
  int pre = block_exclusive(local, lds, total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + k;
    if (i < n) out[i] = pre;
    pre += v[k];
  }
  if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlock) void vd_scan_totals_kernel(int* __restrict__ block_tot, int nblocks,
                                                                     int32_t* __restrict__ grand_total) {
  __shared__ int lds[kScanBlock / 64 + 2];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += kScanBlock) {
    const int i = b0 + threadIdx.x;
    const int x = i < nblocks ? block_tot[i] : 0;
    int total;
    const int pre = block_exclusive(x, lds, total);
    if (i < nblocks) block_tot[i] = carry + pre;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *grand_total = carry;
}

__global__ __launch_bounds__(256) void vd_scan_add_kernel(int* __restrict__ out, int64_t n, const int* __restrict__ block_off) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n) out[i] += block_off[i / (kScanBlock * kScanItems)];
}

// rank of each voxel = exclusive count of earlier first occurrences; zero the per-voxel counters
__global__ __launch_bounds__(256) void vd_rank_kernel(const unsigned* __restrict__ ht_first, const unsigned* __restrict__ pt_slot,
                                                       const int* __restrict__ scan, int64_t n, unsigned* __restrict__ ht_rank) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const unsigned s = pt_slot[i];
  if (s != 0xffffffffu && ht_first[s] == static_cast<unsigned>(i)) ht_rank[s] = static_cast<unsigned>(scan[i]);
}

__global__ __launch_bounds__(256) void vd_count_kernel(const unsigned* __restrict__ pt_slot, const unsigned* __restrict__ ht_rank,
                                                        int64_t n, int* __restrict__ ecnt) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const unsigned s = pt_slot[i];
  if (s != 0xffffffffu) atomicAdd(&ecnt[ht_rank[s]], 1);
}

__global__ __launch_bounds__(256) void vd_fill_kernel(const unsigned* __restrict__ pt_slot, const unsigned* __restrict__ ht_rank,
                                                       int64_t n, const int* __restrict__ ebase, int* __restrict__ efill,
                                                       int* __restrict__ list) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const unsigned s = pt_slot[i];
  if (s == 0xffffffffu) return;
  const int e = static_cast<int>(ht_rank[s]);
  list[ebase[e] + atomicAdd(&efill[e], 1)] = static_cast<int>(i);
}

// one thread per voxel: order its points, accumulate in float64 in that order, emit the means
__global__ __launch_bounds__(256) void vd_reduce_kernel(const float* __restrict__ pts, int ld, int channels,
                                                         const int32_t* __restrict__ m_dev, const int* __restrict__ ecnt,
                                                         const int* __restrict__ ebase, int* __restrict__ list,
                                                         float* __restrict__ out, int ldo) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= *m_dev) return;
  const int c = ecnt[e];
  int* L = list + ebase[e];
  for (int x = 1; x < c; ++x) {  // a 0.3 m voxel of a raw scan holds ~6 points
    const int val = L[x];
    int y = x - 1;
    while (y >= 0 && L[y] > val) {
      L[y + 1] = L[y];
      --y;
    }
    L[y + 1] = val;
  }
  for (int ch = 0; ch < channels; ++ch) {
    double s = 0.0;
    for (int x = 0; x < c; ++x) s += static_cast<double>(pts[static_cast<int64_t>(L[x]) * ld + ch]);
    out[static_cast<int64_t>(e) * ldo + ch] = static_cast<float>(s / static_cast<double>(c));
  }
}

struct VdViews {
  unsigned* min_ord;  // 3 (+ pad)
  int32_t* totals;    // [0] = number of voxels, [1] = scratch total
  unsigned long long* ht_keys;
  unsigned* ht_first;
  unsigned* ht_rank;
  unsigned* pt_slot;
  int* scan;
  int* ecnt;
  int* ebase;
  int* efill;
  int* list;
  int* block_tot;
  unsigned ht_cap;
};

bool carve(Arena& ar, int64_t n, VdViews* v) {
  const size_t N = static_cast<size_t>(n > 0 ? n : 1);
  unsigned cap = 1024;
  while (cap < 2 * N) cap <<= 1;
  v->ht_cap = cap;
  v->min_ord = ar.take<unsigned>(4);
  v->totals = ar.take<int32_t>(4);
  v->ht_keys = ar.take<unsigned long long>(cap);
  v->ht_first = ar.take<unsigned>(cap);
  v->ht_rank = ar.take<unsigned>(cap);
  v->pt_slot = ar.take<unsigned>(N);
  v->scan = ar.take<int>(N);
  v->ecnt = ar.take<int>(N);
  v->ebase = ar.take<int>(N);
  v->efill = ar.take<int>(N);
  v->list = ar.take<int>(N);
  v->block_tot = ar.take<int>(ceil_div<size_t>(N, kScanBlock * kScanItems) + 1);
  return ar.ok;
}

}  // namespace

extern "C" size_t rdm_voxel_downsample_workspace_bytes(int64_t n) {
  rdm::Arena ar(nullptr, 0);
  VdViews v;
  carve(ar, n, &v);
  return ar.off;
}

extern "C" int rdm_voxel_downsample(const float* points, int64_t n, int64_t ld, int channels, double voxel, float* out,
                                    int64_t ldo, int32_t* out_count, int32_t* status, void* ws, size_t ws_bytes,
                                    void* stream) {
  using namespace rdm;
  RDM_REQUIRE(out_count && status, "rdm_voxel_downsample: null pointer");
  RDM_REQUIRE(n >= 0 && n < (1ll << 26) && channels >= 3 && ld >= channels && ldo >= channels && voxel > 0.0,
              "rdm_voxel_downsample: bad arguments (n=%lld channels=%d ld=%lld ldo=%lld voxel=%g)", (long long)n, channels,
              (long long)ld, (long long)ldo, voxel);
  hipStream_t st = static_cast<hipStream_t>(stream);
  Arena ar(ws, ws_bytes);
  VdViews v;
  if (!carve(ar, n, &v)) {
    set_error("rdm_voxel_downsample: workspace too small (%zu < %zu bytes)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  RDM_HIP_CHECK(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
  if (n == 0) return RDM_OK;
  RDM_REQUIRE(points && out, "rdm_voxel_downsample: null points");
  const unsigned nb = static_cast<unsigned>(ceil_div<int64_t>(n, 256));
  const int sb = static_cast<int>(ceil_div<int64_t>(n, kScanBlock * kScanItems));
  RDM_HIP_CHECK(hipMemsetAsync(v.min_ord, 0xff, 4 * sizeof(unsigned), st));
  RDM_HIP_CHECK(hipMemsetAsync(v.ht_keys, 0xff, sizeof(unsigned long long) * v.ht_cap, st));
  RDM_HIP_CHECK(hipMemsetAsync(v.ht_first, 0xff, sizeof(unsigned) * v.ht_cap, st));
  RDM_HIP_CHECK(hipMemsetAsync(v.ecnt, 0, sizeof(int) * n, st));
  RDM_HIP_CHECK(hipMemsetAsync(v.efill, 0, sizeof(int) * n, st));
  hipLaunchKernelGGL(vd_bbox_kernel, dim3(std::min(nb, 1024u)), dim3(256), 0, st, points, n, static_cast<int>(ld), v.min_ord);
  hipLaunchKernelGGL(vd_key_kernel, dim3(nb), dim3(256), 0, st, points, n, static_cast<int>(ld), voxel,
                     v.min_ord, v.ht_keys, v.ht_first, v.ht_cap - 1, v.pt_slot, status);
  // voxel ranks in first-occurrence order
  hipLaunchKernelGGL(vd_scan_blocks_kernel<true>, dim3(sb), dim3(kScanBlock), 0, st, nullptr, v.ht_first, v.pt_slot, n, v.scan,
                     v.block_tot);
  hipLaunchKernelGGL(vd_scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, st, v.block_tot, sb, out_count);
  hipLaunchKernelGGL(vd_scan_add_kernel, dim3(nb), dim3(256), 0, st, v.scan, n, v.block_tot);
  hipLaunchKernelGGL(vd_rank_kernel, dim3(nb), dim3(256), 0, st, v.ht_first, v.pt_slot, v.scan, n, v.ht_rank);
  // per-voxel lists (the scan runs over n slots; slots >= #voxels hold zero counts)
  hipLaunchKernelGGL(vd_count_kernel, dim3(nb), dim3(256), 0, st, v.pt_slot, v.ht_rank, n, v.ecnt);
  hipLaunchKernelGGL(vd_scan_blocks_kernel<false>, dim3(sb), dim3(kScanBlock), 0, st, v.ecnt, nullptr, nullptr, n, v.ebase,
                     v.block_tot);
  hipLaunchKernelGGL(vd_scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, st, v.block_tot, sb, v.totals + 1);
  hipLaunchKernelGGL(vd_scan_add_kernel, dim3(nb), dim3(256), 0, st, v.ebase, n, v.block_tot);
  hipLaunchKernelGGL(vd_fill_kernel, dim3(nb), dim3(256), 0, st, v.pt_slot, v.ht_rank, n, v.ebase, v.efill, v.list);
  hipLaunchKernelGGL(vd_reduce_kernel, dim3(nb), dim3(256), 0, st, points, static_cast<int>(ld), channels, out_count, v.ecnt,
                     v.ebase, v.list, out, static_cast<int>(ldo));
  return launch_status("rdm_voxel_downsample");
}
