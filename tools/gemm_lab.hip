// Standalone fp32 MFMA GEMM lab (developer tool): C[M,N] = A[M,K] * Bt[N,K]^T, both operands K-contiguous.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_lab tools/gemm_lab.hip && ./tools/bin/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Variant V1: 128x128 tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x2 tiles.  LDS tiles [row][BK] with k
// contiguous (row stride BK+4 floats), filled with ds_write_b128 from float4 global loads, read with ds_read_b128:
// lane (r = lane&31, h = lane>>5) takes k = 8q + 4h .. +3 and uses them in 4 successive MFMA steps (the contraction
// order is permuted identically for A and B).
template <int BK, int MODE>
__global__ __launch_bounds__(256) void gemm_v1(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                int M, int N, int K, int splits, float* __restrict__ part) {
  constexpr int BM = 128, BN = 128, LD = BK + 4;
  __shared__ float As[2][BM][LD];
  __shared__ float Bs[2][BN][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int split = blockIdx.z;
  const int ktiles = (K + BK - 1) / BK;
  const int per = (ktiles + splits - 1) / splits;
  const int kt0 = split * per, kt1 = min(ktiles, kt0 + per);
  constexpr int F4 = BK / 4;                  // float4 per tile row
  constexpr int V = BM * F4 / 256;            // float4 per thread per operand
  float4 ra[V], rb[V];
  auto load = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / F4, kc = (idx % F4) * 4;
      const int gm = m0 + row, gn = n0 + row, gk = k0 + kc;
      if (MODE == 1 || MODE == 2) { ra[i] = make_float4(1, 2, 3, 4); rb[i] = make_float4(1, 1, 1, 1); continue; }
      ra[i] = (gm < M && gk < K) ? *reinterpret_cast<const float4*>(A + (long long)gm * K + gk) : make_float4(0, 0, 0, 0);
      rb[i] = (gn < N && gk < K) ? *reinterpret_cast<const float4*>(Bt + (long long)gn * K + gk) : make_float4(0, 0, 0, 0);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / F4, kc = (idx % F4) * 4;
      *reinterpret_cast<float4*>(&As[buf][row][kc]) = ra[i];
      *reinterpret_cast<float4*>(&Bs[buf][row][kc]) = rb[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (kt0 < kt1) { load(kt0); store(0); }
  __syncthreads();
  const int lr = lane & 31, lh = lane >> 5;
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) load(kt + 1);
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = MODE == 3 ? make_float4(1, 2, 3, (float)q) : *reinterpret_cast<const float4*>(&As[buf][wm * 64 + i * 32 + lr][8 * q + 4 * lh]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = MODE == 3 ? make_float4(1, 1, 1, (float)kt) : *reinterpret_cast<const float4*>(&Bs[buf][wn * 64 + j * 32 + lr][8 * q + 4 * lh]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;
            const float bv = s == 0 ? b[j].x : s == 1 ? b[j].y : s == 2 ? b[j].z : b[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
    }
    if (MODE != 2) {
      if (kt + 1 < kt1) store(buf ^ 1);
      __syncthreads();
    }
  }
  float* out = splits > 1 ? part + (long long)split * M * N : C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) out[(long long)row * N + col] = acc[i][j][r];
      }
  }
}

__global__ void reduce_k(const float* part, float* C, long long mn, int splits) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= mn) return;
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += part[s * mn + i];
  C[i] = v;
}

template <int MODE>
void bench_mode(const float* dA, const float* dB, float* dC, float* dP, int M, int N, int K, int sp) {
  auto run = [&]() {
    dim3 grid((N + 127) / 128, (M + 127) / 128, sp);
    hipLaunchKernelGGL((gemm_v1<32, MODE>), grid, dim3(256), 0, 0, dA, dB, dC, M, N, K, sp, dP);
  };
  for (int i = 0; i < 3; ++i) run();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000 / 20;
  printf("  mode %d (BK=32, splits %d, kernel only): %7.1f us  %6.1f TFLOP/s\n", MODE, sp, us, 2.0 * M * K * N / us / 1e6);
}

int main(int argc, char** argv) {
  struct Shape { int m, k, n, s; };
  std::vector<Shape> shapes = {{5289, 1536, 512, 3}, {13795, 768, 256, 1}, {3879, 1920, 128, 13}, {16384, 2048, 2048, 1}};
  for (auto sh : shapes) {
    const int M = sh.m, K = sh.k, N = sh.n;
    float *dA, *dB, *dC, *dP;
    hipMalloc(&dA, (size_t)M * K * 4); hipMalloc(&dB, (size_t)N * K * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMalloc(&dP, (size_t)M * N * 4 * 16);
    hipMemset(dA, 0, (size_t)M * K * 4); hipMemset(dB, 0, (size_t)N * K * 4);
    printf("M=%d K=%d N=%d\n", M, K, N);
    bench_mode<0>(dA, dB, dC, dP, M, N, K, sh.s);
    bench_mode<1>(dA, dB, dC, dP, M, N, K, sh.s);
    bench_mode<2>(dA, dB, dC, dP, M, N, K, sh.s);
    bench_mode<3>(dA, dB, dC, dP, M, N, K, sh.s);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dP);
  }
  return 0;
}
