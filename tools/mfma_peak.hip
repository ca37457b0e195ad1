// Peak-rate probe of the fp32 MFMA instructions on gfx950 (no memory traffic): how many TFLOP/s can a kernel
// made only of v_mfma_f32_32x32x2_f32 / 16x16x4_f32 reach, with 1..4 wavefronts per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[ACC];
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.f) out[0] = s;
}
template <int ACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[ACC];
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  if (s == 123.f) out[0] = s;
}
template <typename F>
double timeit(F f) {
  f();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e-3;
}
int main() {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000;
  for (int blocks_per_cu : {1, 2, 4}) {
    const int grid = 256 * blocks_per_cu;
    double t = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f); });
    double fl = (double)grid * 4 * iters * 8 * 4 * 4096.0;
    printf("32x32x2 f32, 4 acc, %d block(s)/CU: %.1f TFLOP/s\n", blocks_per_cu, fl / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f); });
    fl = (double)grid * 4 * iters * 8 * 1 * 4096.0;
    printf("32x32x2 f32, 1 acc (dependent chain), %d block(s)/CU: %.1f TFLOP/s\n", blocks_per_cu, fl / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f); });
    fl = (double)grid * 4 * iters * 8 * 4 * 2048.0;
    printf("16x16x4 f32, 4 acc, %d block(s)/CU: %.1f TFLOP/s\n", blocks_per_cu, fl / t / 1e12);
  }
  return 0;
}
