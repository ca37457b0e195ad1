// Phase timing of gemm_kernel<64,64,2,2,32,false,2> (the path's main tile): compiles gemm.hip with shader-clock stamps and
// prints, for workgroup 0, the clocks of prologue (first tiles in flight -> LDS), main loop and epilogue, plus the kernel time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRDM_GEMM_TIMING tools/gemm_phase_lab.hip rdmnet_amd/csrc/capi.cpp
//         rdmnet_amd/csrc/norm.hip -o tools/bin/gemm_phase_lab;   ./tools/bin/gemm_phase_lab M K N splits [1 = with GroupNorm statistics and a row divisor]
#include "../rdmnet_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3879, K = argc > 2 ? atoi(argv[2]) : 1920, N = argc > 3 ? atoi(argv[3]) : 128;
  const int splits = argc > 4 ? atoi(argv[4]) : 1;
  const int with_stats = argc > 5 ? atoi(argv[5]) : 0;
  auto dev = [](size_t n) { float* p; (void)hipMalloc(&p, n * 4); std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = float((i * 2654435761u) % 1000) / 1000.f - 0.5f; (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; };
  GemmArgs g;
  g.A = dev(size_t(M) * K); g.B = dev(size_t(K) * N); g.C = dev(size_t(M) * N); g.bias = dev(N); g.rowdiv = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = N; g.ldc = N; g.sa = g.sb = g.sc = 0; g.act = 0; g.splits = splits;
  g.part = splits > 1 ? dev(size_t(M) * N * splits) : nullptr; g.stats = nullptr;
  g.A2 = nullptr; g.aidx = nullptr; g.bidx = nullptr; g.lda2 = g.ldi = g.c1 = g.n_coarse = g.n_b = 0; g.xcd_tiles = 0;
  if (with_stats) { double* st; (void)hipMalloc(&st, size_t((M + 63) / 64) * 2 * N * 8); g.stats = st; g.rowdiv = dev(M); }
  (void)hipMalloc(&g.clk, 64);
  unsigned long long h[8];
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  dim3 grid((N + 63) / 64, (M + 63) / 64, splits);
  for (int it = 0; it < 4; ++it) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((gemm_kernel<64, 64, 2, 2, 32, false, 2>), grid, dim3(256), 0, 0, g);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, g.clk, 64, hipMemcpyDeviceToHost);
    const int tiles = ((K + 31) / 32 + splits - 1) / splits;
    printf("M=%d K=%d N=%d splits=%d blocks=%d run %d: %.1f us; prologue %llu, main %llu (%d k-tiles, %llu per tile), epilogue %llu clocks (staged in LDS after %llu, stores issued after %llu)\n", M, K, N,
           splits, grid.x * grid.y * grid.z, it, ms * 1e3, h[1] - h[0], h[2] - h[1], tiles, (h[2] - h[1]) / tiles, h[3] - h[2], h[4] - h[2], h[5] - h[2]);
  }
  return 0;
}
