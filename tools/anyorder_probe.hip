// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  Two one-workgroup kernels that spin for a
// fixed number of shader clocks: back to back they take 2 T, overlapped T.   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/bin/anyorder_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long clocks, int* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < clocks) {}
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  int* d;
  hipMalloc(&d, 4096);
  hipStream_t st;
  hipStreamCreate(&st);
  const unsigned long long T = 20000000ull;  // ~ 0.2 s at 100 MHz memtime
  for (int mode = 0; mode < 3; ++mode) {
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 4; ++rep) {
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, 0, T, d);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, T, d + 64);
      if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, T, d + 128);
    }
    hipStreamSynchronize(st);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %d (%s): %.1f ms for 4 rounds\n", mode, mode == 0 ? "in order x2" : (mode == 1 ? "second any-order" : "second+third any-order"), ms);
  }
  return 0;
}
