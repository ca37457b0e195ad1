// Phase timing of rn_query_kernel<256> on a level-0-like self search: shader clocks per wavefront (= per query), averaged: prologue (meta, query point, cloud id), the 27 cell headers + prefix, the candidate scan, the
// register rank sort, the row write.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRDM_RN_TIMING tools/rn_phase_lab.hip rdmnet_amd/csrc/capi.cpp -o tools/bin/rn_phase_lab
#include "../rdmnet_amd/csrc/radius_neighbors.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16000;
  const float radius = argc > 2 ? atof(argv[2]) : 1.275f;
  const int width = argc > 3 ? atoi(argv[3]) : 38;
  std::vector<float> h(size_t(2) * n * 3);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return float(st % 1000003) / 1000003.f; };
  for (int i = 0; i < 2 * n; ++i) {  // voxel-subsampled ground + walls: about one point per 0.3 m cell
    const float r = 3.f + 57.f * sqrtf(rnd()), t = 6.2831853f * rnd();
    h[3 * i] = r * cosf(t); h[3 * i + 1] = r * sinf(t); h[3 * i + 2] = rnd() < 0.7f ? -1.7f + 0.05f * rnd() : 3.f * rnd();
  }
  float* pts; int64_t *len, *out; int32_t* flags; void *gws, *ws;
  (void)hipMalloc(&pts, h.size() * 4); (void)hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int64_t hl[2] = {n, n};
  (void)hipMalloc(&len, 16); (void)hipMemcpy(len, hl, 16, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, size_t(2) * n * width * 8); (void)hipMalloc(&flags, 64); (void)hipMemset(flags, 0, 64);
  const size_t gb = rdm_radius_grid_workspace_bytes(2 * n), wb = rdm_radius_neighbors_workspace_bytes(2 * n, 2 * n, 2);
  (void)hipMalloc(&gws, gb); (void)hipMalloc(&ws, wb);
  int rc = rdm_radius_grid_build(pts, 2 * n, len, 2, radius, gws, gb, nullptr);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    static unsigned* clk = nullptr;
    if (!clk) { (void)hipMalloc(&clk, size_t(2) * n * 8 * 4); (void)hipMemcpyToSymbol(HIP_SYMBOL(rdm_rn_clk), &clk, sizeof(clk)); }
    (void)hipMemset(clk, 0, size_t(2) * n * 8 * 4);
    (void)hipEventRecord(e0, 0);
    rc |= rdm_radius_grid_query(gws, gb, 2 * n, pts, 2 * n, len, 2, radius, width, out, nullptr, flags, flags + 1, ws, wb, nullptr);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> hc(size_t(2) * n * 8);
    (void)hipMemcpy(hc.data(), clk, hc.size() * 4, hipMemcpyDeviceToHost);
    double c[8] = {0};
    for (size_t i = 0; i < hc.size(); ++i) c[i % 8] += hc[i];
    int32_t f[2]; (void)hipMemcpy(f, flags, 8, hipMemcpyDeviceToHost);
    const double q = 2.0 * n;
    printf("rc=%d %d queries, max count %d, %.1f us; clocks per query: prologue %.0f, cell headers %.0f, scan %.0f, sort %.0f, write %.0f\n", rc, 2 * n, f[0],
           ms * 1e3, c[0] / q, c[1] / q, c[2] / q, c[3] / q, c[4] / q);
  }
  return 0;
}
