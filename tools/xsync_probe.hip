// Cost of a fork/join between two HIP streams on gfx950: events against stream memory operations.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/xsync_probe tools/xsync_probe.hip ; gpurun -- tools/bin/xsync_probe
// Each iteration: a chain of K small kernels on the main stream; in the middle a fork to a side stream, one small kernel there,
// and a join back.  Reported: host-side time per iteration (the main stream is synchronised at its end) for
//   none   : no side stream (the side kernel runs on the main stream)
//   events : hipEventRecord + hipStreamWaitEvent, both directions
//   values : hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory, both directions
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ int g_spin = 0;  // > 0: every kernel also spins that many clock reads (the GPU then lags behind the host, as in a real run)
__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < g_spin) {}
}
int main(int argc, char** argv) {
  float* buf; CK(hipMalloc(&buf, 1 << 20));
  hipStream_t mainS, side; CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t f, j; CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  uint64_t* sig = nullptr;
  hipError_t se = hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 16, hipMallocSignalMemory);
  printf("signal memory: %s\n", hipGetErrorString(se));
  const int K = argc > 1 ? atoi(argv[1]) : 40, iters = argc > 2 ? atoi(argv[2]) : 300;
  unsigned seq = 0;
  { int spin = argc > 3 ? atoi(argv[3]) : 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_spin), &spin, sizeof(spin))); printf("spin %d x 10 ns per kernel\n", spin); }
  for (int mode = 0; mode < 5; ++mode) {
    if (mode == 2 && se != hipSuccess) continue;
    double best = 1e9, sum = 0;
    for (int it = 0; it < iters + 20; ++it) {
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < K / 2; ++k) tiny<<<4, 256, 0, mainS>>>(buf, 1024);
      if (mode == 0) {
        tiny<<<4, 256, 0, mainS>>>(buf + 4096, 1024);
      } else if (mode == 1) {
        CK(hipEventRecord(f, mainS)); CK(hipStreamWaitEvent(side, f, 0));
        tiny<<<4, 256, 0, side>>>(buf + 4096, 1024);
        for (int k = 0; k < 4; ++k) tiny<<<4, 256, 0, mainS>>>(buf, 1024);
        CK(hipEventRecord(j, side)); CK(hipStreamWaitEvent(mainS, j, 0));
      } else if (mode == 3 || mode == 4) {  // as 1 / 0, with two host synchronisations of the main stream between fork and join
        if (mode == 3) { CK(hipEventRecord(f, mainS)); CK(hipStreamWaitEvent(side, f, 0)); }
        tiny<<<4, 256, 0, mode == 3 ? side : mainS>>>(buf + 4096, 1024);
        for (int h = 0; h < 2; ++h) {
          for (int k = 0; k < 2; ++k) tiny<<<4, 256, 0, mainS>>>(buf, 1024);
          CK(hipStreamSynchronize(mainS));
        }
        if (mode == 3) { CK(hipEventRecord(j, side)); CK(hipStreamWaitEvent(mainS, j, 0)); }
      } else {
        ++seq;
        CK(hipStreamWriteValue32(mainS, sig, seq, 0)); CK(hipStreamWaitValue32(side, sig, seq, hipStreamWaitValueGte, 0xffffffffu));
        tiny<<<4, 256, 0, side>>>(buf + 4096, 1024);
        for (int k = 0; k < 4; ++k) tiny<<<4, 256, 0, mainS>>>(buf, 1024);
        CK(hipStreamWriteValue32(side, sig + 1, seq, 0)); CK(hipStreamWaitValue32(mainS, sig + 1, seq, hipStreamWaitValueGte, 0xffffffffu));
      }
      for (int k = 0; k < K / 2; ++k) tiny<<<4, 256, 0, mainS>>>(buf, 1024);
      CK(hipStreamSynchronize(mainS));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (it >= 20) { sum += us; if (us < best) best = us; }
    }
    printf("%s: mean %.1f us  best %.1f us per iteration (%d small kernels)\n", mode == 0 ? "none  " : mode == 1 ? "events" : mode == 2 ? "values" : mode == 3 ? "events + 2 host syncs" : "none + 2 host syncs", sum / iters, best, K + 1 + (mode ? 4 : 0));
  }
  return 0;
}
