// Phase timing of gemm_wide_kernel (128 x 128 x 32 tiles): compiles gemm.hip with shader-clock stamps and prints, for workgroup
// (0,0,0), the clocks of prologue / main loop / epilogue, the clocks per k-tile (median, min, max) and the share of a tile spent
// at its barrier, plus the kernel time.  -DGW_ABLATE=<bits> removes parts of the loop (timing probes: wrong results).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRDM_GEMM_TIMING [-DGW_ABLATE=n] tools/gemm_wide_lab.hip rdmnet_amd/csrc/capi.cpp
//         rdmnet_amd/csrc/norm.hip rdmnet_amd/csrc/lockstep.cpp -o tools/bin/gemm_wide_lab;   ./tools/bin/gemm_wide_lab M K N splits [1 = 16x16x4 MFMA]
#include "../rdmnet_amd/csrc/gemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 15516, K = argc > 2 ? atoi(argv[2]) : 1536, N = argc > 3 ? atoi(argv[3]) : 512;
  const int splits = argc > 4 ? atoi(argv[4]) : 1;
  const int mi16 = argc > 5 ? atoi(argv[5]) : 0;  // 1: the v_mfma_f32_16x16x4_f32 variant
  auto dev = [](size_t n) { float* p; (void)hipMalloc(&p, n * 4); std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = float((i * 2654435761u) % 1000) / 1000.f - 0.5f; (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; };
  GemmArgs g;
  g.A = dev(size_t(M) * K); g.B = dev(size_t(K) * N); g.C = dev(size_t(M) * N); g.bias = dev(N); g.rowdiv = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = N; g.ldc = N; g.sa = g.sb = g.sc = 0; g.act = 0; g.splits = splits;
  g.part = splits > 1 ? dev(size_t(M) * N * splits) : nullptr; g.stats = nullptr;
  g.A2 = nullptr; g.aidx = nullptr; g.bidx = nullptr; g.lda2 = g.ldi = g.c1 = g.n_coarse = g.n_b = 0; g.xcd_tiles = 0;
  (void)hipMalloc(&g.clk, 400 * 8);
  std::vector<unsigned long long> h(400);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  dim3 grid((N + 127) / 128, (M + 127) / 128, splits);
  const int tiles = std::min(128, ((K + 31) / 32 + splits - 1) / splits);
  for (int it = 0; it < 4; ++it) {
    (void)hipMemset(g.clk, 0, 400 * 8);
    (void)hipEventRecord(e0, 0);
    if (mi16) hipLaunchKernelGGL((gemm_wide_kernel<false, true>), grid, dim3(256), 0, 0, g);
    else hipLaunchKernelGGL((gemm_wide_kernel<false>), grid, dim3(256), 0, 0, g);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), g.clk, 400 * 8, hipMemcpyDeviceToHost);
    std::vector<long long> per, bar;
    for (int t = 0; t + 1 < tiles; ++t) per.push_back((long long)(h[8 + t + 1] - h[8 + t]));
    for (int t = 0; t < tiles; ++t) bar.push_back((long long)(h[264 + t] - h[136 + t]));
    std::sort(per.begin(), per.end()); std::sort(bar.begin(), bar.end());
    const double flops = 2.0 * M * K * N;
    printf("GW_ABLATE=%d mi16=%d M=%d K=%d N=%d splits=%d blocks=%d run %d: %.1f us = %.1f TF; prologue %llu, main %llu (%d k-tiles), epilogue %llu clocks; per tile median %lld min %lld max %lld; "
           "at the barrier median %lld max %lld\n", GW_ABLATE, mi16, M, K, N, splits, grid.x * grid.y * grid.z, it, ms * 1e3, flops / ms / 1e9, h[1] - h[0], h[2] - h[1], tiles, h[3] - h[2],
           per.empty() ? 0 : per[per.size() / 2], per.empty() ? 0 : per.front(), per.empty() ? 0 : per.back(), bar[bar.size() / 2], bar.back());
  }
  return 0;
}
