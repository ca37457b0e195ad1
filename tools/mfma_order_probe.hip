// Does v_mfma_f32_16x16x4_f32 accumulate like v_mfma_f32_32x32x2_f32 -- both as the fp32 fma chain over ascending k?  If so a
// GEMM built on either instruction returns the same bits.  One wavefront: C[16x16] = A[16xK] B[Kx16] three ways.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_order_probe.hip -o tools/bin/mfma_order_probe && ./tools/bin/mfma_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 256;
__global__ void probe(const float* A, const float* B, float* c16, float* c32, float* cf) {  // A [32][K], B [K][32]
  const int lane = threadIdx.x;
  {  // 16x16x4 on rows/cols 0..15
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int i = lane & 15, kb = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + kb], B[(k0 + kb) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) c16[(4 * kb + r) * 16 + i] = acc[r];
  }
  {  // 32x32x2 on rows/cols 0..31 (the 16x16 corner is compared)
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int i = lane & 31, kb = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + kb], B[(k0 + kb) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
      if (row < 16 && i < 16) c32[row * 16 + i] = acc[r];
    }
  }
  for (int e = lane; e < 256; e += 64) {  // fma chain
    const int row = e / 16, col = e % 16;
    float v = 0.f;
    for (int k = 0; k < K; ++k) v = __builtin_fmaf(A[row * K + k], B[k * 32 + col], v);
    cf[e] = v;
  }
}
int main() {
  std::vector<float> a(32 * K), b(K * 32);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& x : a) x = rnd() * 3.f;
  for (auto& x : b) x = rnd();
  float *A, *B, *c16, *c32, *cf;
  hipMalloc(&A, a.size() * 4); hipMalloc(&B, b.size() * 4); hipMalloc(&c16, 1024); hipMalloc(&c32, 1024); hipMalloc(&cf, 1024);
  hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, c16, c32, cf);
  float h16[256], h32[256], hf[256];
  hipMemcpy(h16, c16, 1024, hipMemcpyDeviceToHost); hipMemcpy(h32, c32, 1024, hipMemcpyDeviceToHost); hipMemcpy(hf, cf, 1024, hipMemcpyDeviceToHost);
  int d1 = 0, d2 = 0, d3 = 0;
  for (int i = 0; i < 256; ++i) { d1 += memcmp(&h16[i], &h32[i], 4) != 0; d2 += memcmp(&h32[i], &hf[i], 4) != 0; d3 += memcmp(&h16[i], &hf[i], 4) != 0; }
  printf("K=%d: 16x16x4 vs 32x32x2: %d of 256 differ; 32x32x2 vs fma chain: %d; 16x16x4 vs fma chain: %d  (sample %.9g %.9g %.9g)\n", K, d1, d2, d3, h16[5], h32[5], hf[5]);
  return 0;
}
