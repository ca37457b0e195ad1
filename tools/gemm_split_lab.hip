// Lab: fp32-accurate GEMM on the bf16 matrix cores.  C[M,N] = A[M,K] B[K,N] with every operand split into
// three bf16 pieces (a = hi + mid + lo exactly: 3 x 8 significant bits) and six products (hi.hi, hi.mid, mid.hi, hi.lo,
// lo.hi, mid.mid; the three dropped ones are <= 2^-24 of |a||b| each) on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate:
// the six products cost 6/16 of the fp32 instruction's matrix-pipe time for the same K), fp32 accumulation -- hi.hi in one
// accumulator, the five corrections in a second one, added once at the end.
//   * B (the weights) is split ONCE on the host into planes stored k-contiguous per output column ([3][N][K] bf16), so a B
//     operand of the MFMA (8 consecutive k of one column) is one 16-byte load;
//   * A (activations) is split by the thread that stages it into LDS (8 consecutive k of one row: 2 float4 in, 3 x 16 B out);
//   * 64 x 64 tile, 4 wavefronts (2 x 2), BK = 32, LDS rows padded to 80 B (conflict-free ds_read_b128), double-buffered.
// Timed against rdm_gemm (fp32 MFMA, the product kernel) on the path's shapes; errors against an fp64 product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_split_lab.hip -Lrdmnet_amd -lrdmnet_hip -Wl,-rpath,'$ORIGIN/../../rdmnet_amd' -o tools/bin/gemm_split_lab
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/rdmnet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 64, BN = 64, BK = 32, LDS_ROW = 40;  // bf16 elements per LDS row (32 + 8 pad = 80 B)

struct SplitArgs {
  const float* A;       // [M, lda] fp32
  const uint16_t* BT;   // [3][N][ldk] bf16 planes (hi, mid, lo), k contiguous
  float* C;             // [M, ldc]
  int M, N, K, lda, ldk, ldc;
  long long plane;      // elements between planes of BT
};

// a = hi + mid + lo (bf16 each, round to nearest even), for two values at once
__device__ __forceinline__ void split2(f32x2 a, bf16x2& hi, bf16x2& mid, bf16x2& lo) {
  hi = __builtin_convertvector(a, bf16x2);
  const f32x2 r1 = a - __builtin_convertvector(hi, f32x2);
  mid = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(mid, f32x2);
  lo = __builtin_convertvector(r2, bf16x2);
}

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NPROD>  // 6 = the fp32-accurate form; 3 (hi.hi, hi.mid, mid.hi) and 1 (plain bf16) for the error table
__global__ __launch_bounds__(256) void gemm_split_kernel(SplitArgs g) {
  __shared__ __attribute__((aligned(16))) uint16_t As[2][3][BM][LDS_ROW];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[2][3][BN][LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int ktiles = g.K / BK;
  // staging roles: A: thread -> (row = tid / 4, 8 consecutive k at 8 * (tid % 4)); B: thread -> (col = tid / 4, same k chunk), 3 planes
  const int srow = tid >> 2, sk = (tid & 3) * 8;
  const float* a_src = g.A + static_cast<long long>(min(m0 + srow, g.M - 1)) * g.lda + sk;
  const uint16_t* b_src = g.BT + static_cast<long long>(min(n0 + srow, g.N - 1)) * g.ldk + sk;
  float4 ra[2];
  uint4 rb[3];
  auto load = [&](int kt) {
    const int k0 = min(kt, ktiles - 1) * BK;
    ra[0] = *reinterpret_cast<const float4*>(a_src + k0);
    ra[1] = *reinterpret_cast<const float4*>(a_src + k0 + 4);
#pragma unroll
    for (int p = 0; p < 3; ++p) rb[p] = *reinterpret_cast<const uint4*>(b_src + p * g.plane + k0);
  };
  auto store = [&](int buf) {
    bf16x2 h[4], m[4], l[4];
    split2(f32x2{ra[0].x, ra[0].y}, h[0], m[0], l[0]);
    split2(f32x2{ra[0].z, ra[0].w}, h[1], m[1], l[1]);
    split2(f32x2{ra[1].x, ra[1].y}, h[2], m[2], l[2]);
    split2(f32x2{ra[1].z, ra[1].w}, h[3], m[3], l[3]);
    auto pack = [](const bf16x2 (&v)[4]) {
      uint4 r;
      r.x = __builtin_bit_cast(unsigned, v[0]); r.y = __builtin_bit_cast(unsigned, v[1]);
      r.z = __builtin_bit_cast(unsigned, v[2]); r.w = __builtin_bit_cast(unsigned, v[3]);
      return r;
    };
    *reinterpret_cast<uint4*>(&As[buf][0][srow][sk]) = pack(h);
    *reinterpret_cast<uint4*>(&As[buf][1][srow][sk]) = pack(m);
    *reinterpret_cast<uint4*>(&As[buf][2][srow][sk]) = pack(l);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(&Bs[buf][p][srow][sk]) = rb[p];
  };
  f32x16 acc = {0}, cor = {0};
  load(0);
  store(0);
  load(1);
  lds_barrier();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[ks][p] = *reinterpret_cast<const bf16x8*>(&As[buf][p][wm * 32 + li][ks * 16 + lk * 8]);
        b[ks][p] = *reinterpret_cast<const bf16x8*>(&Bs[buf][p][wn * 32 + li][ks * 16 + lk * 8]);
      }
    store(buf ^ 1);   // tile kt + 1 (loaded one trip ago) into the other buffer
    load(kt + 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (NPROD >= 6) {
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][2], b[ks][0], cor, 0, 0, 0);  // lo . hi
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b[ks][2], cor, 0, 0, 0);  // hi . lo
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], b[ks][1], cor, 0, 0, 0);  // mid . mid
      }
      if (NPROD >= 3) {
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], b[ks][0], cor, 0, 0, 0);  // mid . hi
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b[ks][1], cor, 0, 0, 0);  // hi . mid
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b[ks][0], acc, 0, 0, 0);    // hi . hi
    }
    lds_barrier();
  }
  acc = acc + cor;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col = n0 + wn * 32 + li;
    if (row < g.M && col < g.N) g.C[static_cast<long long>(row) * g.ldc + col] = acc[r];
  }
}

// Variant 2: the B operand (pre-split weights, k-contiguous per column) goes straight from global memory (L2) into the MFMA
// registers -- 16 bytes per lane, plane and k-step -- so that only A passes through LDS: half the LDS traffic of the kernel above
// (whose 12 ds_read_b128 + 6 ds_write_b128 per k-tile cost more LDS-pipe time than its 12 MFMAs cost matrix-pipe time).
template <int NPROD>
__global__ __launch_bounds__(256) void gemm_split_direct_kernel(SplitArgs g) {
  __shared__ __attribute__((aligned(16))) uint16_t As[2][3][BM][LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int ktiles = g.K / BK;
  const int srow = tid >> 2, sk = (tid & 3) * 8;
  const float* a_src = g.A + static_cast<long long>(min(m0 + srow, g.M - 1)) * g.lda + sk;
  const uint16_t* b_src = g.BT + static_cast<long long>(min(n0 + wn * 32 + li, g.N - 1)) * g.ldk + lk * 8;
  float4 ra[2];
  bf16x8 rb[2][2][3];  // [stage][k-step][plane]
  auto load_a = [&](int kt) {
    const int k0 = min(kt, ktiles - 1) * BK;
    ra[0] = *reinterpret_cast<const float4*>(a_src + k0);
    ra[1] = *reinterpret_cast<const float4*>(a_src + k0 + 4);
  };
  auto load_b = [&](int st, int kt) {
    const int k0 = min(kt, ktiles - 1) * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) rb[st][ks][p] = *reinterpret_cast<const bf16x8*>(b_src + p * g.plane + k0 + ks * 16);
  };
  auto store_a = [&](int buf) {
    bf16x2 h[4], m[4], l[4];
    split2(f32x2{ra[0].x, ra[0].y}, h[0], m[0], l[0]);
    split2(f32x2{ra[0].z, ra[0].w}, h[1], m[1], l[1]);
    split2(f32x2{ra[1].x, ra[1].y}, h[2], m[2], l[2]);
    split2(f32x2{ra[1].z, ra[1].w}, h[3], m[3], l[3]);
    auto pack = [](const bf16x2 (&v)[4]) {
      uint4 r;
      r.x = __builtin_bit_cast(unsigned, v[0]); r.y = __builtin_bit_cast(unsigned, v[1]);
      r.z = __builtin_bit_cast(unsigned, v[2]); r.w = __builtin_bit_cast(unsigned, v[3]);
      return r;
    };
    *reinterpret_cast<uint4*>(&As[buf][0][srow][sk]) = pack(h);
    *reinterpret_cast<uint4*>(&As[buf][1][srow][sk]) = pack(m);
    *reinterpret_cast<uint4*>(&As[buf][2][srow][sk]) = pack(l);
  };
  f32x16 acc = {0}, cor = {0};
  load_a(0);
  load_b(0, 0);
  store_a(0);
  load_a(1);
  lds_barrier();
  auto step = [&](int kt, int st) {
    const int buf = kt & 1;
    bf16x8 a[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[ks][p] = *reinterpret_cast<const bf16x8*>(&As[buf][p][wm * 32 + li][ks * 16 + lk * 8]);
    store_a(buf ^ 1);
    load_a(kt + 2);
    load_b(st ^ 1, kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (NPROD >= 6) {
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][2], rb[st][ks][0], cor, 0, 0, 0);
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], rb[st][ks][2], cor, 0, 0, 0);
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], rb[st][ks][1], cor, 0, 0, 0);
      }
      if (NPROD >= 3) {
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], rb[st][ks][0], cor, 0, 0, 0);
        cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], rb[st][ks][1], cor, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], rb[st][ks][0], acc, 0, 0, 0);
    }
    lds_barrier();
  };
  for (int kt = 0; kt < ktiles; kt += 2) {
    step(kt, 0);
    if (kt + 1 < ktiles) step(kt + 1, 1);
  }
  acc = acc + cor;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col = n0 + wn * 32 + li;
    if (row < g.M && col < g.N) g.C[static_cast<long long>(row) * g.ldc + col] = acc[r];
  }
}

static uint16_t bf16_rn(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
static float bf16_f(uint16_t h) {
  uint32_t u = static_cast<uint32_t>(h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  struct Shape { int m, k, n; const char* what; };
  const Shape shapes[] = {{32000, 32, 128, "unary (fine level)"}, {10944, 960, 64, "KPConv 64 x 15 weights"}, {3904, 1920, 128, "KPConv 128"},
                          {3904, 128, 512, "unary2 expansion"}, {3904, 512, 128, "unary1 squeeze"}, {1344, 3840, 256, "KPConv 256"},
                          {576, 7680, 512, "KPConv 512"}, {3904, 1536, 512, "decoder3"}, {10944, 768, 256, "decoder2"}, {16384, 2048, 2048, "large"}};
  printf("| shape (M x K x N) | fp32 MFMA (rdm_gemm, split-K where its model wants it) us | split-bf16 x6, A and B through LDS, us | x6, B straight from L2, us | speed-up of the better | max err / max |C|: fp32 MFMA | x6 | x3 | x1 (plain bf16) |\n|---|---|---|---|---|---|---|---|---|\n");
  for (const Shape& s : shapes) {
    const int M = s.m, K = s.k, N = s.n;
    std::vector<float> hA(size_t(M) * K), hB(size_t(K) * N);
    unsigned long long st = 88172645463325252ull + M;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return float(int(st % 2000001) - 1000000) / 1000000.f; };
    for (auto& v : hA) v = rnd() * (1.f + 3.f * fabsf(rnd()));
    for (auto& v : hB) v = rnd() / sqrtf(float(K));
    std::vector<uint16_t> hBT(size_t(3) * N * K);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) {
        const float b = hB[size_t(k) * N + n];
        const uint16_t h = bf16_rn(b);
        const float r1 = b - bf16_f(h);
        const uint16_t m = bf16_rn(r1);
        const float r2 = r1 - bf16_f(m);
        hBT[(size_t(0) * N + n) * K + k] = h;
        hBT[(size_t(1) * N + n) * K + k] = m;
        hBT[(size_t(2) * N + n) * K + k] = bf16_rn(r2);
      }
    float *dA, *dB, *dC, *dC2;
    uint16_t* dBT;
    void* ws;
    const size_t wsb = rdm_gemm_workspace_bytes(M, N, 1);
    (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4); (void)hipMalloc(&dC, size_t(M) * N * 4);
    (void)hipMalloc(&dC2, size_t(M) * N * 4); (void)hipMalloc(&dBT, hBT.size() * 2); (void)hipMalloc(&ws, wsb);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dBT, hBT.data(), hBT.size() * 2, hipMemcpyHostToDevice);
    SplitArgs g{dA, dBT, dC2, M, N, K, K, K, N, static_cast<long long>(N) * K};
    const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time_it = [&](auto fn) {
      for (int i = 0; i < 3; ++i) fn();
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) fn();
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      return ms * 1e3f / 20;
    };
    if (argc > 1 && atoi(argv[1]) > 1) {
      // the regime the path runs in: S products of this shape in flight on S streams (the runtime puts them on S hardware pipes);
      // reported: wall time per product with S in flight (= 1 / aggregate rate), each form on its own buffers
      const int S = atoi(argv[1]);
      std::vector<hipStream_t> sts(S);
      std::vector<float*> cs(S);
      std::vector<void*> wss(S);
      for (int i = 0; i < S; ++i) {
        (void)hipStreamCreateWithFlags(&sts[i], hipStreamNonBlocking);
        (void)hipMalloc(&cs[i], size_t(M) * N * 4);
        (void)hipMalloc(&wss[i], wsb);
      }
      auto time_s = [&](auto fn) {
        for (int i = 0; i < 3 * S; ++i) fn(i % S);
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 40 * S; ++i) fn(i % S);
        (void)hipDeviceSynchronize();
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (40.0 * S);
      };
      const double a_ref = time_s([&](int i) { rdm_gemm(dA, K, 0, dB, N, 0, 0, cs[i], N, 0, M, N, K, 1, nullptr, nullptr, 0, wss[i], wsb, sts[i]); });
      const double a6 = time_s([&](int i) { SplitArgs gi = g; gi.C = cs[i]; hipLaunchKernelGGL(gemm_split_kernel<6>, grid, dim3(256), 0, sts[i], gi); });
      const double a6d = time_s([&](int i) { SplitArgs gi = g; gi.C = cs[i]; hipLaunchKernelGGL(gemm_split_direct_kernel<6>, grid, dim3(256), 0, sts[i], gi); });
      const double flop = 2.0 * M * K * N;
      printf("| %d x %d x %d (%s) | %d in flight: fp32 %.1f us (%.0f TF) | x6 LDS %.1f us (%.0f TF) | x6 direct %.1f us (%.0f TF) | %.2f |\n", M, K, N, s.what, S, a_ref,
             flop / a_ref / 1e6, a6, flop / a6 / 1e6, a6d, flop / a6d / 1e6, a_ref / fmin(a6, a6d));
      fflush(stdout);
      for (int i = 0; i < S; ++i) { (void)hipStreamDestroy(sts[i]); (void)hipFree(cs[i]); (void)hipFree(wss[i]); }
      (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dC2); (void)hipFree(dBT); (void)hipFree(ws);
      continue;
    }
    const float t_ref = time_it([&] { rdm_gemm(dA, K, 0, dB, N, 0, 0, dC, N, 0, M, N, K, 1, nullptr, nullptr, 0, ws, wsb, nullptr); });
    const float t6 = time_it([&] { hipLaunchKernelGGL(gemm_split_kernel<6>, grid, dim3(256), 0, 0, g); });
    const float t6d = time_it([&] { hipLaunchKernelGGL(gemm_split_direct_kernel<6>, grid, dim3(256), 0, 0, g); });
    {
      std::vector<float> chk(size_t(M) * N), ref6(size_t(M) * N);
      (void)hipMemcpy(chk.data(), dC2, chk.size() * 4, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(gemm_split_kernel<6>, grid, dim3(256), 0, 0, g);
      (void)hipMemcpy(ref6.data(), dC2, ref6.size() * 4, hipMemcpyDeviceToHost);
      if (memcmp(chk.data(), ref6.data(), chk.size() * 4) != 0) printf("(variant 2 differs from variant 1!)\n");
    }
    // errors on a sample of rows against fp64
    std::vector<float> hC(size_t(M) * N), hC6(size_t(M) * N), hC3(size_t(M) * N), hC1(size_t(M) * N);
    (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hC6.data(), dC2, hC6.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(gemm_split_kernel<3>, grid, dim3(256), 0, 0, g);
    (void)hipMemcpy(hC3.data(), dC2, hC3.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(gemm_split_kernel<1>, grid, dim3(256), 0, 0, g);
    (void)hipMemcpy(hC1.data(), dC2, hC1.size() * 4, hipMemcpyDeviceToHost);
    double e_ref = 0, e6 = 0, e3 = 0, e1v = 0, cmax = 0;
    for (int r = 0; r < M; r += (M > 512 ? M / 257 : 1))
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += double(hA[size_t(r) * K + k]) * double(hB[size_t(k) * N + n]);
        cmax = fmax(cmax, fabs(acc));
        e_ref = fmax(e_ref, fabs(hC[size_t(r) * N + n] - acc));
        e6 = fmax(e6, fabs(hC6[size_t(r) * N + n] - acc));
        e3 = fmax(e3, fabs(hC3[size_t(r) * N + n] - acc));
        e1v = fmax(e1v, fabs(hC1[size_t(r) * N + n] - acc));
      }
    printf("| %d x %d x %d (%s) | %.1f | %.1f | %.1f | %.2f | %.2e | %.2e | %.2e | %.2e |\n", M, K, N, s.what, t_ref, t6, t6d, t_ref / fminf(t6, t6d), e_ref / cmax,
           e6 / cmax, e3 / cmax, e1v / cmax);
    fflush(stdout);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dC2); (void)hipFree(dBT); (void)hipFree(ws);
  }
  return 0;
}
