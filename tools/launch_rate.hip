// Aggregate kernel-launch rate of one process from T host threads, one stream each (is the multi-stream
// engine bound by the host's launch path?).  hipcc --offload-arch=gfx950 -O2 -o tools/bin/launch_rate tools/launch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(float* p) { if (p && threadIdx.x == 0 && blockIdx.x == 12345678) p[0] = 1.f; }
int main() {
  const int N = 20000;
  for (int T : {1, 2, 3, 4, 6, 8}) {
    std::vector<hipStream_t> st(T);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto work = [&](int t) {
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st[t], nullptr);
      hipStreamSynchronize(st[t]);
    };
    for (int t = 0; t < T; ++t) work(t);  // warm
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads %d: %.0f launches/s aggregate, %.2f us per launch per thread\n", T, T * N / s, s / N * 1e6);
    for (auto& x : st) hipStreamDestroy(x);
  }
  return 0;
}
