// Phase timing of grid_subsample_kernel (one workgroup per cloud): compiles grid_subsample.hip with shader-clock stamps and
// prints the clocks of P0 (bounding box + table reset), P1 (keys + de-duplication), P2 (first-occurrence ranks), P3-P6 (per-voxel
// sums), P7 (hash-map order replay), P8 (emit) for cloud 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRDM_GS_TIMING tools/gs_phase_lab.hip rdmnet_amd/csrc/capi.cpp -o tools/bin/gs_phase_lab
//   ./tools/bin/gs_phase_lab [points per cloud] [voxel]
#include "../rdmnet_amd/csrc/grid_subsample.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16000;
  const float voxel = argc > 2 ? atof(argv[2]) : 0.6f;
  // a KITTI-like ring: points on a ground plane and walls within 60 m
  std::vector<float> h(size_t(2) * n * 3);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return float(st % 1000003) / 1000003.f; };
  for (int i = 0; i < 2 * n; ++i) {
    const float r = 3.f + 57.f * rnd() * rnd(), t = 6.2831853f * rnd();
    h[3 * i] = r * cosf(t); h[3 * i + 1] = r * sinf(t); h[3 * i + 2] = rnd() < 0.7f ? -1.7f + 0.05f * rnd() : 3.f * rnd();
  }
  float *pts, *out; int64_t *len, *olen; void* ws;
  (void)hipMalloc(&pts, h.size() * 4); (void)hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, h.size() * 4);
  int64_t hl[2] = {n, n};
  (void)hipMalloc(&len, 16); (void)hipMemcpy(len, hl, 16, hipMemcpyHostToDevice); (void)hipMalloc(&olen, 16);
  const size_t wsb = rdm_grid_subsample_workspace_bytes(2 * n, 2);
  (void)hipMalloc(&ws, wsb);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    (void)hipEventRecord(e0, 0);
    const int rc = rdm_grid_subsample(pts, 2 * n, len, 2, voxel, out, olen, ws, wsb, nullptr);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[16]; (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(rdm_gs_clk), sizeof(c));
    int64_t ol[2]; (void)hipMemcpy(ol, olen, 16, hipMemcpyDeviceToHost);
    printf("rc=%d n=%d voxel=%.2f -> %lld voxels, %.1f us (all launches); clocks: P0 %llu P1 %llu P2 %llu P3-6 %llu P7 %llu P8 %llu | total %llu || P3-6: count %llu, scan %llu, fill %llu, sums %llu, crowded %llu\n", rc, n, voxel,
           (long long)ol[0], ms * 1e3, c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], c[5] - c[4], c[6] - c[5], c[6] - c[0], c[7] - c[3], c[8] - c[7], c[9] - c[8], c[10] - c[9], c[4] - c[10]);
  }
  return 0;
}
