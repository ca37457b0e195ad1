// a4 -- KPConv.forward as ONE kernel for the fine levels (C_in = 1, 32, 64): neighbourhood aggregation AND the
// kernel-weight contraction, without the [M, 15*C] intermediate in HBM.
//
// Reference: geotransformer/modules/kpconv/kpconv.py:79-122.  For every query m:
//   WF[m, k, c] = sum_h max(0, 1 - |s[idx[m,h]] - q[m] - kp[k]| / sigma) * feats[idx[m,h], c]      :91-105
//   out[m, c']  = (sum_k sum_c WF[m, k, c] W[k, c, c']) / max(1, #{h : sum_c feats[idx[m,h], c] > 0}) + bias[c']   :107-121
// The two-kernel form (kpconv.hip + gemm.hip) writes WF -- 15*C floats per query, 61 MB for the 32 k-point level at
// C = 32 -- and reads it back.  Here a workgroup aggregates 16 queries (one wavefront per query at a time, the 16x16x4
// MFMA formulation of kpconv.hip), parks the 16 x 15C block in LDS, and multiplies it by W on the same matrix cores:
// the A operand of v_mfma_f32_16x16x4_f32 is a 16-B LDS read (16 queries x 4 consecutive k per instruction), the B
// operand one coalesced 16-B global read per lane from a copy of W stored in operand order
// (rdm_kpconv_pack_weights); the K range is split over the wavefronts and the slices meet in LDS in a fixed order.
// The epilogue divides by the neighbour count, adds the bias, writes the [M, C'] output and accumulates the fp64
// column sums GroupNorm needs (same partial layout as the GEMM epilogue, gemm.hip).
//   C = 32: 16 wavefronts x 1 query, 2 column tiles x 8 K slices; 31 KB block + 32 KB staging -> 2 workgroups per CU
//   C = 64:  8 wavefronts x 2 queries, 4 column tiles x 2 K slices; 62 KB block + 16 KB staging -> 2 workgroups per CU
//   C_in = 1 (first layer, features == 1): no matrix core needed on either side; one wavefront per query, lane = output
//   channel, 16 wavefronts x 4 queries per workgroup.
// Measured (docs/EXPERIMENTS.md 5b / 5d, profiles/r02_pmc_fused_kpconv.md): HBM-side writes of these layers drop from 200 MB to 24 MB per
// scan pair at the same pairs/s; the engine's default for these layers since round 3.  The kernel's time is the sum of its
// phases' L2 -> CU traffic (neighbour lines + W re-read per 16 queries: tools/kpconv_bench.py, tools/lab/kpconv_fused_pc.hip
// for the producer / consumer variant that overlaps the phases and measured the same).
#include <atomic>
#include <cstdlib>

#include "../../include/rdmnet_hip.h"
#include "common.h"
#include "internal.h"
#include "lockstep.h"

namespace {

using namespace rdm;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKP = 15;
constexpr int kMaxH = 128;

// Influence of kernel point k on a neighbour at relative position rel (kpconv.py:96-99), d = rel - kp_k:
// max(0, 1 - |d| / sigma) with the hardware square root and a reciprocal multiply.  The fused multiply-adds are written out
// (and contraction is off inside) so that every kernel of this file rounds alike whatever surrounds the call: the form is the
// one the compiler had chosen for the round-2 / round-3 kernels -- fma(dy, dy, dx dx) + dz dz, then fma(-1/sigma, sqrt, 1).
__device__ __forceinline__ float kp_influence(float dx, float dy, float dz, float inv_sigma) {
#pragma clang fp contract(off)
  const float xx = dx * dx, zz = dz * dz;
  const float d2 = __builtin_fmaf(dy, dy, xx) + zz;
  return fmaxf(0.f, __builtin_fmaf(-inv_sigma, __builtin_amdgcn_sqrtf(d2), 1.f));
}

struct FusedArgs {
  const float* q_points;       // [M,3]
  const float* s_points;       // [Ns,3]
  const float* s_feats;        // [Ns, ldf]
  const unsigned char* s_pos;  // [Ns]
  const int64_t* idx;          // [M, ldi] (int32 elements when i32)
  int i32;
  const float* kp;             // [15,3]
  const int32_t* width;        // optional device int: effective row width
  const float* w;              // packed weights (see rdm_kpconv_pack_weights)
  const float* bias;           // [C']
  float* out;                  // [M, ldo]
  double* stats;               // [gridDim.x][2][C'] or null
  int M, Ns, H;
  int ldf, ldi, ldo;
  float sigma;
  // gather-only form of the tile kernel (c_in >= 128, one 64-channel slice per workgroup): `out` = WF [M, ldo] with the slice's
  // first channel already added, ctot = c_in (the stride between kernel points inside a WF row), nn [M] or null (slice 0 writes it)
  int ctot;
  float* nn;
};

// The same for two neighbours at once on the packed-fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma: two IEEE operations per
// lane and instruction, same roundings as the scalar form -- the aggregation of the LDS-tile kernel is bound by VALU issue).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 kp_influence2(f32x2 dx, f32x2 dy, f32x2 dz, float inv_sigma) {
#pragma clang fp contract(off)
  const f32x2 xx = dx * dx, zz = dz * dz;
  const f32x2 d2 = __builtin_elementwise_fma(dy, dy, xx) + zz;
  const f32x2 sq = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
  const f32x2 ninv = {-inv_sigma, -inv_sigma}, one = {1.f, 1.f};
  const f32x2 w = __builtin_elementwise_fma(ninv, sq, one);
  return f32x2{fmaxf(0.f, w.x), fmaxf(0.f, w.y)};
}

// ---- C_in in {32, 64}
template <int C, int QB, int NW, int ITERS>
__device__ __forceinline__ void kpconv_fused_kernel_body(const dim3 blockIdx, const dim3 gridDim, FusedArgs a) {
  (void)blockIdx; (void)gridDim;
  constexpr int VEC = C / 16;           // channels per lane and gather tile pass (channel = VEC*j + e)
  constexpr int NT = C / 16;            // output column tiles (C' = C)
  constexpr int RT = QB / 16;           // output row tiles
  constexpr int TILES = NT * RT;        // output tiles of 16 x 16
  constexpr int KS = NW / TILES;        // K split over wavefronts
  constexpr int K16 = kKP * C / 16;     // 16-deep contraction steps (30 / 60)
  constexpr int LDW = kKP * C + 4;      // LDS row stride of the aggregated block (floats)
  constexpr int QPW = QB / NW;          // queries per wavefront and iteration
  constexpr int PF = 4;                 // neighbour groups fetched per trip
  static_assert(TILES * KS == NW && QPW * NW == QB && (KS - 1) * TILES * 1024 <= NW * kMaxH * 16, "wavefront roles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* WF = smem;                                                        // [QB][LDW]
  float4* nb_all = reinterpret_cast<float4*>(smem + QB * LDW);             // [NW][kMaxH]: rel.xyz, w = support row
  float* nn_s = smem + QB * LDW + NW * kMaxH * 4;                          // [QB]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  float4* nb = nb_all + wave * kMaxH;
  const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f, kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
  const float inv_sigma = 1.0f / a.sigma;  // (hardware sqrt + reciprocal multiply as in kpconv.hip)
  int H = a.H;
  if (a.width) H = min(H, *a.width);
  // roles of the contraction phase
  const int tw = wave % TILES, kh = wave / TILES, rt = tw / NT, ct = tw % NT;
  const float bias_v = a.bias[16 * ct + j];
  double st_s = 0.0, st_ss = 0.0;  // GroupNorm column sums of this lane's column (rows 4g .. 4g+3 of every row tile it owns)

  for (int it = 0; it < ITERS; ++it) {
    const int q0 = (blockIdx.x * ITERS + it) * QB;
    if (q0 >= a.M) break;  // (uniform over the workgroup)
    // ------------------------------------------------------------ aggregation: QPW queries per wavefront
    for (int qq = 0; qq < QPW; ++qq) {
      const int ql = wave * QPW + qq, m = q0 + ql;
      if (m >= a.M) break;
      const float qx = a.q_points[3 * m], qy = a.q_points[3 * m + 1], qz = a.q_points[3 * m + 2];
      int positives = 0;
      f32x4 acc[VEC];
#pragma unroll
      for (int t = 0; t < VEC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int hc = 0; hc < H; hc += kMaxH) {  // chunks of the staging row (one for every KITTI limit), as in kpconv.hip
        const int Hc = min(H - hc, kMaxH);
        int Hq = 0;  // slots up to the last real neighbour (kpconv.hip: shadow slots contribute exact zeros)
        for (int hb = 0; hb < Hc; hb += 64) {
          const int h = hb + lane;
          const int64_t id = h < Hc ? ld_index(a.idx, static_cast<int64_t>(m) * a.ldi + hc + h, a.i32) : -1;
          const unsigned long long rm = __builtin_amdgcn_ballot_w64(id >= 0 && id < a.Ns);
          if (rm) Hq = hb + 64 - __builtin_clzll(rm);
          float4 v;
          if (id >= 0 && id < a.Ns) {
            v.x = a.s_points[3 * id] - qx;
            v.y = a.s_points[3 * id + 1] - qy;
            v.z = a.s_points[3 * id + 2] - qz;
            v.w = __int_as_float(static_cast<int>(id));
            positives += a.s_pos[id];
          } else {  // shadow neighbour: point at 1e6, zero features
            v.x = 1.0e6f - qx;
            v.y = 1.0e6f - qy;
            v.z = 1.0e6f - qz;
            v.w = __int_as_float(-1);
          }
          if (h < Hc) nb[h] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int h0 = 0; h0 < Hq; h0 += 4 * PF) {
          float w[PF];
          float f[PF][VEC];
#pragma unroll
          for (int p = 0; p < PF; ++p) {
            const int h = h0 + 4 * p + g;
            int id = -1;
            w[p] = 0.f;
            if (h < Hq) {
              const float4 v = nb[h];
              id = __float_as_int(v.w);
              w[p] = kp_influence(v.x - kx, v.y - ky, v.z - kz, inv_sigma);
              if (j >= kKP || id < 0) w[p] = 0.f;
            }
            const float* row = a.s_feats + static_cast<int64_t>(id < 0 ? 0 : id) * a.ldf + VEC * j;
            if (id >= 0) {
              if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(row);
                f[p][0] = t.x; f[p][1] = t.y; f[p][2] = t.z; f[p][3] = t.w;
              } else {
                const float2 t = *reinterpret_cast<const float2*>(row);
                f[p][0] = t.x; f[p][1] = t.y;
              }
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[p][e] = 0.f;
            }
          }
#pragma unroll
          for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p], f[p][e], acc[e], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // nb is rewritten by the next chunk / query
        __builtin_amdgcn_wave_barrier();
      }
      positives = wave_sum_i(positives);
      // park WF[ql, k, c]: accumulator row 4g + r = kernel point, lane j holds channels VEC*j .. VEC*j + VEC-1
      float* dst = WF + ql * LDW + VEC * j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * g + r;
        if (k >= kKP) continue;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + k * C) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        else *reinterpret_cast<float2*>(dst + k * C) = make_float2(acc[0][r], acc[1][r]);
      }
      if (lane == 0) nn_s[ql] = static_cast<float>(positives > 1 ? positives : 1);
    }
    __syncthreads();

    // ------------------------------------------------------------ contraction: out[QB, C'] = WF[QB, 15C] W[15C, C']
    // wavefront (rt, ct, kh): row tile rt, column tile ct, K steps [kh, kh+1) * K16 / KS.  Step s covers k = 16 s ..
    // 16 s + 15 in the order lane group g -> k = 16 s + 4 g + e (e = the e-th of four MFMAs): A is one 16-B LDS read,
    // B one 16-B global read of the packed weights.
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    {
      const float* arow = WF + (16 * rt + j) * LDW + 4 * g;
      const float4* wp = reinterpret_cast<const float4*>(a.w) + static_cast<int64_t>(ct) * 64 + lane;
      const int s_begin = kh * K16 / KS, s_end = (kh + 1) * K16 / KS;
#pragma unroll 4
      for (int s = s_begin; s < s_end; ++s) {
        const float4 bv = wp[static_cast<int64_t>(s) * NT * 64];
        const float4 av = *reinterpret_cast<const float4*>(arow + 16 * s);
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, o1, 0, 0, 0);
      }
    }
    f32x4 o = o0 + o1;
    // the K slices of a tile meet in the neighbour-staging area (idle since the barrier above; the barrier that ends the
    // iteration keeps the next aggregation from overwriting it early) and are added in slice order
    f32x4* red = reinterpret_cast<f32x4*>(nb_all);  // [(KS - 1) * TILES][64]
    if (kh > 0) red[((kh - 1) * TILES + tw) * 64 + lane] = o;
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int k = 1; k < KS; ++k) o = o + red[((k - 1) * TILES + tw) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * rt + 4 * g + r, m = q0 + ql;
        if (m < a.M) {
          float v = o[r] / nn_s[ql];
          v += bias_v;
          a.out[static_cast<int64_t>(m) * a.ldo + 16 * ct + j] = v;
          st_s += static_cast<double>(v);
          st_ss += static_cast<double>(v) * static_cast<double>(v);
        }
      }
    }
    __syncthreads();  // WF and nn_s are rewritten by the next iteration
  }
  if (a.stats) {  // column sums of this workgroup's rows: lane groups g, then row tiles, in a fixed order
    st_s = (st_s + __shfl_xor(st_s, 16, 64)) + (__shfl_xor(st_s, 32, 64) + __shfl_xor(st_s, 48, 64));
    st_ss = (st_ss + __shfl_xor(st_ss, 16, 64)) + (__shfl_xor(st_ss, 32, 64) + __shfl_xor(st_ss, 48, 64));
    double* ex = reinterpret_cast<double*>(smem);  // [TILES][16][2]
    if (kh == 0 && g == 0) {
      ex[(tw * 16 + j) * 2 + 0] = st_s;
      ex[(tw * 16 + j) * 2 + 1] = st_ss;
    }
    __syncthreads();
    if (threadIdx.x < C) {
      const int c = threadIdx.x, cti = c / 16, cj = c % 16;
      double s = 0.0, ss = 0.0;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        s += ex[((r * NT + cti) * 16 + cj) * 2 + 0];
        ss += ex[((r * NT + cti) * 16 + cj) * 2 + 1];
      }
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * C + c] = s;
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * C + c] = ss;
    }
  }
}
template <int C, int QB, int NW, int ITERS>
__global__ __launch_bounds__(64 * NW) void kpconv_fused_kernel(FusedArgs a) { kpconv_fused_kernel_body<C, QB, NW, ITERS>(blockIdx, gridDim, a); }


// ---- C_in in {32, 64}, neighbour rows of at most 128 slots: the support rows of a block of queries staged ONCE in LDS
//
// The lock-step kernel above fetches the feature row, the point and the positive flag of every (query, neighbour) pair
// from L2 -- three 128-B lines per pair, M x H_real of them -- although neighbouring queries share most of their
// neighbours (a query has ~37 real neighbours at the fine levels; 16 queries of one grid cell have 100-200 distinct ones).
// Here a workgroup takes 16 queries that are consecutive in the CELL ORDER of their level's search grid (`order`: the
// cell-sorted {x, y, z, row} records of rdm_radius_grid_records; null = row order), and
//   1. every wavefront reads its queries' index rows (coalesced) and enters the ids into an LDS hash set (atomicCAS,
//      1024 entries, 16 probes),
//   2. the occupied entries are numbered by ballots + a prefix over the wavefronts (slot < CAP; the rest stay "global"),
//   3. the distinct rows -- feature row, point, positive flag -- are fetched ONCE by all threads together (one round trip
//      with every load in flight, 16-B accesses, a row = 8 / 16 consecutive lanes) into the LDS tile,
//   4. the aggregation of kpconv_fused_kernel runs with its B operand (feature rows) and the points read from LDS through
//      the id -> slot map; neighbour order h, influence arithmetic and MFMA sequence are unchanged, so the block
//      WF[16, 15 C] holds the same bits; ids that found no slot (hash probes exhausted, more than CAP distinct rows in the
//      block) are fetched from global memory as before,
//   5. the tile is overwritten by the parked block and the contraction with W, normalisation, bias and GroupNorm partials
//      follow as above (same K split over the wavefronts as kpconv_fused_kernel<C>: the output is bit-identical to it).
//   C = 32: 8 wavefronts x 2 queries, tile of 240 rows (30 KB, under the 30.3 KB of the parked block) -> 46 KB, THREE workgroups
//           per CU (two of 16 wavefronts measured 65.6 against 59.4 us at the first level: the third workgroup's compute
//           phases fill more of the other two's latency phases)
//   C = 64: 8 wavefronts x 2 queries, tile of 240 rows (60 KB, the parked block needs 60.3 KB) -> 75 KB, two per CU
// What bounds it (tools/tile_lab.py, docs/EXPERIMENTS.md 5e): inside the aggregation the CU's four matrix pipes are ~80 % busy (fp32 MFMA
// at 64 flop / clk / SIMD) -- points and rows now come from LDS, L2 -> CU traffic of the layer drops by the re-use factor --
// while index rows, hash inserts, tile load and epilogue (half of a workgroup's residency) are serial latencies that two or
// three resident workgroups only partly overlap; the layer lands within 15 % of the lock-step kernel either way.
#ifdef RDM_TILE_TIMING
// tools/tile_lab.py: s_memtime stamps (100 MHz) of wavefront 0 at the phase boundaries of every workgroup
__device__ long long* rdm_tile_clk;  // [blocks][8]
#define TILE_T0() long long tl_t = __builtin_amdgcn_s_memtime(); int tl_k = 0; const long long tl_rt0 = wall_clock64()
#define TILE_PHASE() do { const long long now = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && rdm_tile_clk) rdm_tile_clk[tl_blk * 8 + tl_k] = now - tl_t; tl_t = now; ++tl_k; } while (0)
// per-wavefront stamps inside the aggregation phase: [blocks][16 waves][4] = sum prologue, trips, (trips count), -
__device__ long long* rdm_tile_wclk;
#define TILE_W(slot, val) do { if ((threadIdx.x & 63) == 0 && rdm_tile_wclk) rdm_tile_wclk[(tl_blk * 16 + (threadIdx.x >> 6)) * 8 + (slot)] = (val); } while (0)
#define TILE_NOW() __builtin_amdgcn_s_memtime()
#ifdef RDM_TILE_TRIP_STAMPS  // serialising stamps inside a trip (diagnosis only: they change the schedule)
#define TRIP_DECL() long long ts_a = 0, ts_b = 0, ts_c = 0, ts_d = 0, ts_t = 0
#define TRIP_BEGIN() ts_t = TILE_NOW()
#define TRIP_LDS(acc_) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long n_ = TILE_NOW(); acc_ += n_ - ts_t; ts_t = n_; } while (0)
#define TRIP_REG(acc_, v_) do { asm volatile("" ::"v"(v_)); asm volatile("s_nop 0" ::: "memory"); const long long n_ = TILE_NOW(); acc_ += n_ - ts_t; ts_t = n_; } while (0)
#define TRIP_END() do { TILE_W(4, ts_a); TILE_W(5, ts_b); TILE_W(6, ts_c); TILE_W(7, ts_d); } while (0)
#endif
#else
#define TILE_T0() do { } while (0)
#define TILE_PHASE() do { } while (0)
#define TILE_W(slot, val) do { } while (0)
#define TILE_NOW() 0ll
#endif
#ifndef RDM_TILE_TRIP_STAMPS
#define TRIP_DECL() do { } while (0)
#define TRIP_BEGIN() do { } while (0)
#define TRIP_LDS(acc_) do { } while (0)
#define TRIP_REG(acc_, v_) do { } while (0)
#define TRIP_END() do { } while (0)
#endif
constexpr int kTileQB = 16;       // queries per workgroup
constexpr int kTileHash = 1024;   // entries of the id -> slot hash
constexpr int kTileProbes = 16;

template <int C> struct TileCfg;
#ifndef RDM_TILE32_NW  // (lab builds: -DRDM_TILE32_NW=16 -DRDM_TILE32_CAP=384 = two workgroups of 16 wavefronts per CU, 65.6 against 59.4 us at the first level)
#define RDM_TILE32_NW 8
#define RDM_TILE32_CAP 240
#endif
template <> struct TileCfg<32> { static constexpr int NW = RDM_TILE32_NW, CAP = RDM_TILE32_CAP; };
template <> struct TileCfg<64> { static constexpr int NW = 8, CAP = 240; };

template <int C>
struct TileLds {
  static constexpr int CAP = TileCfg<C>::CAP;
  static constexpr int LDW = kKP * C + 4;
  // floats: the tile [CAP + 1][C] (row CAP = the shadow slot: zero features), then the parked block
  static constexpr int R0 = kTileQB * LDW > (CAP + 1) * C ? kTileQB * LDW : (CAP + 1) * C;
  static constexpr size_t off_pts = static_cast<size_t>(R0) * 4;                 // float4 [CAP + 1]: x, y, z, 1 (0: shadow slot)
  static constexpr size_t off_hkey = off_pts + static_cast<size_t>(CAP + 1) * 16;  // int [kTileHash]
  static constexpr size_t off_hval = off_hkey + kTileHash * 4;                   // short [kTileHash]
  static constexpr size_t off_sid = off_hval + kTileHash * 2;                    // int [CAP]: slot -> support row
  static constexpr size_t off_code = off_sid + static_cast<size_t>(CAP) * 4;     // short [kTileQB][kMaxH]
  static constexpr size_t off_ppos = off_code + kTileQB * kMaxH * 2;             // uint8 [CAP + 1 (+ pad)]: positive flags
  static constexpr size_t off_misc = off_ppos + ((static_cast<size_t>(CAP) + 1 + 15) / 16) * 16;  // nn_s [16], mrow [16], count
  static constexpr size_t bytes = off_misc + 256;
};

// (two workgroups of 16 / three of 8 wavefronts per CU at C = 32, two of 8 at C = 64: 8 / 6 / 4 wavefronts per SIMD)
// GATHER = true (round 5): the aggregation half alone, for the layers whose weight contraction stays a GEMM (c_in >= 128): a
// workgroup serves ONE 64-channel slice of its 16 queries' WF rows (a.s_feats / a.out point at the slice, gridDim.y slices) and
// writes them where kpconv_gather_kernel writes them -- same neighbour order, same influences, same MFMA sequence per accumulator:
// the same bits -- but fetches every distinct support row of the block once instead of once per (query, neighbour).
template <int C, bool GATHER = false>
__device__ __forceinline__ void kpconv_tile_kernel_body(const dim3 blockIdx, const dim3 gridDim, FusedArgs a, const float4* __restrict__ order, int xcd_ranges) {
  (void)blockIdx; (void)gridDim;
  if constexpr (GATHER) {  // the slice of this workgroup
    a.s_feats += static_cast<int64_t>(blockIdx.y) * C;
    a.out += static_cast<int64_t>(blockIdx.y) * C;
  }
  using L = TileLds<C>;
  constexpr int NW = TileCfg<C>::NW, CAP = TileCfg<C>::CAP, NTH = 64 * NW, QB = kTileQB, QPW = QB / NW;
  constexpr int VEC = C / 16, NT = C / 16, TILES = NT, K16 = kKP * C / 16, LDW = L::LDW, PF = 4;
  constexpr int KS = C == 32 ? 8 : 2;      // K slices of the contraction: kpconv_fused_kernel<C>'s (the same partial sums, added in the same order)
  constexpr int SPW = KS * TILES / NW;     // slices a wavefront contracts one after the other
  constexpr int LPR = C / 4;                               // 16-B lanes per feature row
  constexpr int ROW_PASSES = (CAP * LPR + NTH - 1) / NTH;  // tile-load trips of the whole workgroup
  static_assert(TILES * KS == NW * SPW && SPW >= 1 && QPW * NW == QB && static_cast<size_t>(KS - 1) * TILES * 1024 <= L::off_misc - L::off_pts &&
                    CAP <= NTH && kTileHash % NTH == 0 && CAP < 32768,
                "wavefront roles / reduction area / one point per thread / 16-bit slot codes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* R0 = reinterpret_cast<float*>(smem_raw);
  float4* pts = reinterpret_cast<float4*>(smem_raw + L::off_pts);
  int* hkey = reinterpret_cast<int*>(smem_raw + L::off_hkey);
  short* hval = reinterpret_cast<short*>(smem_raw + L::off_hval);
  int* sid = reinterpret_cast<int*>(smem_raw + L::off_sid);
  short* code = reinterpret_cast<short*>(smem_raw + L::off_code);
  float* nn_s = reinterpret_cast<float*>(smem_raw + L::off_misc);
  int* mrow = reinterpret_cast<int*>(smem_raw + L::off_misc + 64);
  int* n_keys = reinterpret_cast<int*>(smem_raw + L::off_misc + 128);
  unsigned char* ppos = smem_raw + L::off_ppos;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
  // Block ids are dealt to the XCDs in contiguous ranges (workgroup b runs on XCD b % 8 -- observed placement,
  // MI355X_MICROARCH.md: for speed only): neighbouring blocks of cell-ordered queries, whose support rows overlap, then meet in
  // ONE L2 instead of eight (round 5: 65.7 against 68.4 us at the first level in the lab forms 5 / 3 of a persistent rewrite that
  // was itself dropped -- static block assignment and software-pipelined index rows ran 57-80 us where this kernel runs 43-60,
  // docs/EXPERIMENTS.md round 5).  The partial rows stay indexed by the BLOCK: same bits whatever the mapping.
  const int per_xcd = gridDim.x / 8;
  const int blk = (xcd_ranges && static_cast<int>(blockIdx.x) < 8 * per_xcd) ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
  const int q0 = blk * QB;
  int H = a.H;
  if (a.width) H = min(H, *a.width);  // (<= kMaxH: checked by the host)
  const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f, kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
  const float inv_sigma = 1.0f / a.sigma;

#ifdef RDM_TILE_TIMING
  const int tl_blk = blk;
#endif
  TILE_T0();
  // ---------------------------------------------------------------- 0 / 1: index rows (in flight while the hash is emptied)
  int qm[QPW], qHq[QPW], qid[QPW][2], qpos[QPW][2];
  float qx[QPW], qy[QPW], qz[QPW];
  int64_t raw[QPW][2];
#pragma unroll
  for (int qq = 0; qq < QPW; ++qq) {
    const int ql = wave * QPW + qq, u = q0 + ql;
    int m = -1;
    qx[qq] = qy[qq] = qz[qq] = 0.f;
    if (u < a.M) {
      if (order) {
        const float4 rec = order[u];
        m = __float_as_int(rec.w);
        qx[qq] = rec.x; qy[qq] = rec.y; qz[qq] = rec.z;  // (the record carries the point's own bits)
      } else {
        m = u;
        qx[qq] = a.q_points[3 * m]; qy[qq] = a.q_points[3 * m + 1]; qz[qq] = a.q_points[3 * m + 2];
      }
    }
    qm[qq] = m;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int h = lane + 64 * k;
      raw[qq][k] = (m >= 0 && h < H) ? ld_index(a.idx, static_cast<int64_t>(m) * a.ldi + h, a.i32) : -1;
    }
    if (lane == 0) mrow[ql] = m;
  }
  for (int t = tid; t < kTileHash; t += NTH) hkey[t] = -1;
  if (tid == 0) *n_keys = 0;
  __syncthreads();
  TILE_PHASE();  // 0: index rows issued, hash emptied
  // distinct support rows: ids into the hash set
#pragma unroll
  for (int qq = 0; qq < QPW; ++qq) {
    int Hq = 0;  // slots up to the last real neighbour
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t id64 = raw[qq][k];
      const bool real = id64 >= 0 && id64 < a.Ns;
      const unsigned long long rm = __builtin_amdgcn_ballot_w64(real);
      if (rm) Hq = 64 * k + 64 - __builtin_clzll(rm);
      const int id = real ? static_cast<int>(id64) : -1;
      int pos = -1;
      bool entered = false;
      if (real) {
        unsigned p = (static_cast<unsigned>(id) * 2654435761u) >> 22;  // 10 bits
#pragma unroll 1  // (unrolled 16 times the loop's exec masks spill 34 / 52 scalar registers)
        for (int t = 0; t < kTileProbes; ++t) {
          const int old = atomicCAS(&hkey[p], -1, id);
          if (old == -1) entered = true;  // this lane entered the id: it also numbers it below
          if (old == -1 || old == id) {
            pos = static_cast<int>(p);
            break;
          }
          p = (p + 1) & (kTileHash - 1);
        }
      }
      // slot numbers of the ids this wavefront entered: one counter update per wavefront (arrival order; the output does
      // not depend on the numbering)
      const unsigned long long em = __builtin_amdgcn_ballot_w64(entered);
      if (em) {
        int base = 0;
        if (lane == __builtin_ctzll(em)) base = atomicAdd(n_keys, __builtin_popcountll(em));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(em));
        if (entered) {
          const int sl = base + __builtin_popcountll(em & ((1ull << lane) - 1ull));
          hval[pos] = static_cast<short>(sl < CAP ? sl : -1);
          if (sl < CAP) sid[sl] = id;
        }
      }
      qid[qq][k] = id;
      qpos[qq][k] = pos;
    }
    qHq[qq] = Hq;
  }
  __syncthreads();
  TILE_PHASE();  // 1: hash inserts + slot numbers

  const int n_slots = min(*n_keys, CAP);
  TILE_PHASE();  // 2: (slot numbering: folded into the inserts)

  // ---------------------------------------------------------------- 3: the tile, one round trip; slot codes of every (query, h)
  {
    float4 row[ROW_PASSES];
    const int c4 = 4 * (tid % LPR), r_first = tid / LPR;
    const int r_last = n_slots > 0 ? n_slots - 1 : 0;
#pragma unroll
    for (int ps = 0; ps < ROW_PASSES; ++ps) {  // (every load is issued -- rows beyond the tile re-read its last one -- then stored)
      const int rr = min(r_first + ps * (NTH / LPR), r_last);
      row[ps] = n_slots > 0 ? *reinterpret_cast<const float4*>(a.s_feats + static_cast<int64_t>(sid[rr]) * a.ldf + c4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < n_slots) {
      const int id = sid[tid];
      pt = make_float4(a.s_points[3 * id], a.s_points[3 * id + 1], a.s_points[3 * id + 2], 1.f);
      ppos[tid] = a.s_pos[id];
    }
    // codes: slot (0 .. CAP - 1); CAP = the shadow slot (a point at 1e6 with a zero feature row: kpconv.py:91-93, its
    // influence and its features are exact zeros) for the padding behind a row's neighbours; -2 = a real neighbour without a
    // slot (fetched from global memory below)
#pragma unroll
    for (int qq = 0; qq < QPW; ++qq)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int sl = qpos[qq][k] >= 0 ? static_cast<int>(hval[qpos[qq][k]]) : -1;
        const int cd = qid[qq][k] < 0 ? CAP : (sl >= 0 ? sl : -2);
        // layout [query][trip = h / 16][g = h % 4][p = (h % 16) / 4]: lane group g reads the four codes of a trip as one 8-byte word
        const int h = lane + 64 * k;
        code[(wave * QPW + qq) * kMaxH + (h & ~15) + 4 * (h & 3) + ((h & 15) >> 2)] = static_cast<short>(cd);
        qpos[qq][k] = cd;  // (from here on: the code)
      }
    if (tid < C / 4) *reinterpret_cast<float4*>(R0 + CAP * C + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) {
      pts[CAP] = make_float4(1.0e6f, 1.0e6f, 1.0e6f, 0.f);
      ppos[CAP] = 0;
    }
#pragma unroll
    for (int ps = 0; ps < ROW_PASSES; ++ps) {
      const int rr = r_first + ps * (NTH / LPR);
      if (rr < n_slots) *reinterpret_cast<float4*>(R0 + rr * C + c4) = row[ps];
    }
    if (tid < n_slots) pts[tid] = pt;
  }
  __syncthreads();
  TILE_PHASE();  // 3: tile load

  // ---------------------------------------------------------------- 4: aggregation (kpconv_fused_kernel's, operands from LDS)
  // A trip = 16 neighbours (four per lane group g): slot codes, then points and feature rows, all from LDS with no branch
  // and no select -- padding slots carry the shadow slot's code, whose influence and features are exact zeros, and lane
  // j = 15 (no kernel point) fills accumulator row 15, which is never parked.  A query that has a neighbour without a slot
  // (wavefront-uniform test, rare) takes the second loop, which patches those neighbours in from global memory.
  f32x4 acc[QPW][VEC];
  int positives[QPW];
#ifdef RDM_TILE_DEPHASE
  // the wavefronts leave the barrier together and would hit LDS, then the VALU, then the matrix pipe in lock-step, trip after
  // trip; staggered starts let one group's LDS reads run beside another's MFMAs
  if ((wave & 3) == 1) __builtin_amdgcn_s_sleep(RDM_TILE_DEPHASE);
  if ((wave & 3) == 2) __builtin_amdgcn_s_sleep(2 * RDM_TILE_DEPHASE);
  if ((wave & 3) == 3) __builtin_amdgcn_s_sleep(3 * RDM_TILE_DEPHASE);
#endif
#pragma unroll
  for (int qq = 0; qq < QPW; ++qq) {
#pragma unroll
    for (int t = 0; t < VEC; ++t) acc[qq][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ql = wave * QPW + qq, m = qm[qq];
    const long long tw0 = TILE_NOW();
    const bool slow = __builtin_amdgcn_ballot_w64(qpos[qq][0] < 0 || qpos[qq][1] < 0) != 0ull;  // (wavefront-uniform)
    int pc = static_cast<int>(ppos[max(qpos[qq][0], 0)]) * (qpos[qq][0] >= 0) + static_cast<int>(ppos[max(qpos[qq][1], 0)]) * (qpos[qq][1] >= 0);
    if (slow) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (qpos[qq][k] < 0) pc += a.s_pos[qid[qq][k]];
    }
    positives[qq] = wave_sum_i(pc);
    const int Hq = qHq[qq];
    const short* crow = code + ql * kMaxH + 4 * g;  // [trip][g][p]
    const long long tw1 = TILE_NOW();
    if (qq == 0) { TILE_W(0, tw1 - tw0); TILE_W(2, (long long)Hq); TILE_W(3, (long long)slow); }
    TRIP_DECL();
    if (!slow) {
      for (int h0 = 0; h0 < Hq; h0 += 4 * PF) {  // (h0 + 15 <= 127: the code row has 128 entries)
        TRIP_BEGIN();
        int c[PF];
        {
          const short4 cw = *reinterpret_cast<const short4*>(crow + h0);
          c[0] = cw.x; c[1] = cw.y; c[2] = cw.z; c[3] = cw.w;
        }
        TRIP_LDS(ts_a);
        float4 P[PF];
        float f[PF][VEC];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          P[p] = pts[c[p]];
          const float* row = R0 + c[p] * C + VEC * j;
          if constexpr (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(row);
            f[p][0] = t.x; f[p][1] = t.y; f[p][2] = t.z; f[p][3] = t.w;
          } else {
            const float2 t = *reinterpret_cast<const float2*>(row);
            f[p][0] = t.x; f[p][1] = t.y;
          }
        }
        TRIP_LDS(ts_b);
        float w[PF];
#pragma unroll
        for (int p = 0; p < PF; p += 2) {  // two neighbours per packed instruction
          const f32x2 q2x = {qx[qq], qx[qq]}, q2y = {qy[qq], qy[qq]}, q2z = {qz[qq], qz[qq]};
          const f32x2 k2x = {kx, kx}, k2y = {ky, ky}, k2z = {kz, kz};
          const f32x2 wx = kp_influence2((f32x2{P[p].x, P[p + 1].x} - q2x) - k2x, (f32x2{P[p].y, P[p + 1].y} - q2y) - k2y,
                                         (f32x2{P[p].z, P[p + 1].z} - q2z) - k2z, inv_sigma);
          // (x 1 for a row of the tile, x 0 for the shadow slot, whose influence is 0 anyway: the factor makes the 4th component
          // of the point live, so that the read stays one 16-byte LDS access instead of a 12-byte one at half the rate)
          w[p] = wx.x * P[p].w;
          w[p + 1] = wx.y * P[p + 1].w;
        }
        TRIP_REG(ts_c, w[PF - 1]);
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[qq][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p], f[p][e], acc[qq][e], 0, 0, 0);
        TRIP_REG(ts_d, acc[qq][VEC - 1][0]);
      }
      if (qq == 0) TRIP_END();
    } else {
      for (int h0 = 0; h0 < Hq; h0 += 4 * PF) {
        float w[PF];
        float f[PF][VEC];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          const int h = h0 + 4 * p + g;
          const int c = crow[h0 + p];
          if (c >= 0) {
            const float4 P = pts[c];
            w[p] = kp_influence((P.x - qx[qq]) - kx, (P.y - qy[qq]) - ky, (P.z - qz[qq]) - kz, inv_sigma);
            const float* row = R0 + c * C + VEC * j;
            if constexpr (VEC == 4) {
              const float4 t = *reinterpret_cast<const float4*>(row);
              f[p][0] = t.x; f[p][1] = t.y; f[p][2] = t.z; f[p][3] = t.w;
            } else {
              const float2 t = *reinterpret_cast<const float2*>(row);
              f[p][0] = t.x; f[p][1] = t.y;
            }
          } else {  // no slot: this neighbour's row comes from global memory
            const int id = static_cast<int>(ld_index(a.idx, static_cast<int64_t>(m) * a.ldi + h, a.i32));
            w[p] = kp_influence((a.s_points[3 * id] - qx[qq]) - kx, (a.s_points[3 * id + 1] - qy[qq]) - ky,
                                (a.s_points[3 * id + 2] - qz[qq]) - kz, inv_sigma);
            const float* row = a.s_feats + static_cast<int64_t>(id) * a.ldf + VEC * j;
            if constexpr (VEC == 4) {
              const float4 t = *reinterpret_cast<const float4*>(row);
              f[p][0] = t.x; f[p][1] = t.y; f[p][2] = t.z; f[p][3] = t.w;
            } else {
              const float2 t = *reinterpret_cast<const float2*>(row);
              f[p][0] = t.x; f[p][1] = t.y;
            }
          }
        }
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[qq][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[p], f[p][e], acc[qq][e], 0, 0, 0);
      }
    }
#ifdef RDM_TILE_TIMING
    if (qq == 0) { const float probe = acc[0][0][0]; asm volatile("" ::"v"(probe)); TILE_W(1, TILE_NOW() - tw1); }
#endif
  }
  if constexpr (GATHER) {  // WF[m, k, slice]: accumulator row 4 g + r = kernel point, lane j holds channels 4 j .. 4 j + 3 (kpconv.hip's store)
#pragma unroll
    for (int qq = 0; qq < QPW; ++qq) {
      const int m = qm[qq];
      if (m < 0) continue;
      float* dst = a.out + static_cast<int64_t>(m) * a.ldo + VEC * j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * g + r;
        if (k >= kKP) continue;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + k * a.ctot) = make_float4(acc[qq][0][r], acc[qq][1][r], acc[qq][2][r], acc[qq][3][r]);
        else *reinterpret_cast<float2*>(dst + k * a.ctot) = make_float2(acc[qq][0][r], acc[qq][1][r]);
      }
      if (lane == 0 && blockIdx.y == 0 && a.nn) a.nn[m] = static_cast<float>(positives[qq] > 1 ? positives[qq] : 1);
    }
    return;
  }
  __syncthreads();  // every wavefront is done with the tile: the parked block takes its place
  TILE_PHASE();  // 4: aggregation

#pragma unroll
  for (int qq = 0; qq < QPW; ++qq) {
    const int ql = wave * QPW + qq;
    float* dst = R0 + ql * LDW + VEC * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 4 * g + r;
      if (k >= kKP) continue;
      if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + k * C) = make_float4(acc[qq][0][r], acc[qq][1][r], acc[qq][2][r], acc[qq][3][r]);
      else *reinterpret_cast<float2*>(dst + k * C) = make_float2(acc[qq][0][r], acc[qq][1][r]);
    }
    if (lane == 0) nn_s[ql] = static_cast<float>(positives[qq] > 1 ? positives[qq] : 1);
  }
  __syncthreads();

  // ---------------------------------------------------------------- 5: out[16, C'] = WF[16, 15 C] W[15 C, C'] (as above, one row tile)
  const int ct = wave % TILES, khg = wave / TILES;
  const float bias_v = a.bias[16 * ct + j];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + L::off_pts);  // [(KS - 1) * TILES][64]: the staging areas are idle now
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sub = 0; sub < SPW; ++sub) {
    const int kh = khg * SPW + sub;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    const float* arow = R0 + j * LDW + 4 * g;
    const float4* wp = reinterpret_cast<const float4*>(a.w) + static_cast<int64_t>(ct) * 64 + lane;
    const int s_begin = kh * K16 / KS, s_end = (kh + 1) * K16 / KS;
#pragma unroll 4
    for (int s = s_begin; s < s_end; ++s) {
      const float4 bv = wp[static_cast<int64_t>(s) * NT * 64];
      const float4 av = *reinterpret_cast<const float4*>(arow + 16 * s);
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, o1, 0, 0, 0);
    }
    const f32x4 os = o0 + o1;
    if (kh == 0) o = os;
    else red[((kh - 1) * TILES + ct) * 64 + lane] = os;
  }
  __syncthreads();
  TILE_PHASE();  // 5: park + contraction
  double st_s = 0.0, st_ss = 0.0;
  if (khg == 0) {
#pragma unroll
    for (int k = 1; k < KS; ++k) o = o + red[((k - 1) * TILES + ct) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ql = 4 * g + r, m = mrow[ql];
      if (m >= 0) {
        float v = o[r] / nn_s[ql];
        v += bias_v;
        a.out[static_cast<int64_t>(m) * a.ldo + 16 * ct + j] = v;
        st_s += static_cast<double>(v);
        st_ss += static_cast<double>(v) * static_cast<double>(v);
      }
    }
  }
  if (a.stats) {  // column sums of this workgroup's rows: lane groups g in a fixed order
    st_s = (st_s + __shfl_xor(st_s, 16, 64)) + (__shfl_xor(st_s, 32, 64) + __shfl_xor(st_s, 48, 64));
    st_ss = (st_ss + __shfl_xor(st_ss, 16, 64)) + (__shfl_xor(st_ss, 32, 64) + __shfl_xor(st_ss, 48, 64));
    if (khg == 0 && g == 0) {
      a.stats[(static_cast<int64_t>(blk) * 2 + 0) * C + 16 * ct + j] = st_s;
      a.stats[(static_cast<int64_t>(blk) * 2 + 1) * C + 16 * ct + j] = st_ss;
    }
  }
  TILE_PHASE();  // 6: epilogue
#ifdef RDM_TILE_TIMING
  if (threadIdx.x == 0 && rdm_tile_clk) rdm_tile_clk[tl_blk * 8 + 7] = n_slots + ((wall_clock64() - tl_rt0) << 16);  // (100 MHz clock)
#endif
}
template <int C, bool GATHER = false>
__global__ __launch_bounds__(64 * TileCfg<C>::NW, (C == 32 && TileCfg<C>::NW == 8 ? 3 : 2) * TileCfg<C>::NW / 4) void kpconv_tile_kernel(FusedArgs a, const float4* __restrict__ order, int xcd_ranges) { kpconv_tile_kernel_body<C, GATHER>(blockIdx, gridDim, a, order, xcd_ranges); }


// (Round 5 measured a rewrite of this kernel -- a wavefront's queries software-pipelined (index row of query t + 2 and the two
// gathers of t + 1 in flight while t is evaluated), the positive count as a ballot of the gathered feature, the staged
// neighbours in four LDS planes feeding packed-fp32 instructions, ~6.5 VALU instructions per neighbour instead of 12: 22.8-23.2 us
// against 26.8-27.1 for 32 000 queries, 19.5 with {x, y, z, f} records gathered in one access -- and did NOT adopt it: its
// summation order moves the first layer's features by 1e-7, which is enough to make another local hypothesis win the
// registration of the two near-tie golden cases (9 m crop: margin 1; pair 0<->7: margin 0), i.e. to leave the set of poses the
// REFERENCE returns for them.  A 4 us kernel is not worth a weaker parity statement; docs/EXPERIMENTS.md 5f.)
// ---- C_in = 1: one wavefront per query, lane = output channel (C' = 64); QPW queries per wavefront
constexpr int kC1Out = 64, kC1Waves = 16, kC1Qpw = 4;  // 64 queries per workgroup, four per wavefront
__device__ __forceinline__ void kpconv_fused_c1_kernel_body(const dim3 blockIdx, const dim3 gridDim, FusedArgs a) {
  (void)blockIdx; (void)gridDim;
  __shared__ float4 nb_all[kC1Waves][kMaxH];  // rel.xyz, w = feature (0 for shadow neighbours)
  __shared__ double ex[kC1Waves][kC1Out][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  float4* nb = nb_all[wave];
  const float kx = j < kKP ? a.kp[3 * j] : 0.f, ky = j < kKP ? a.kp[3 * j + 1] : 0.f, kz = j < kKP ? a.kp[3 * j + 2] : 0.f;
  const float inv_sigma = 1.0f / a.sigma;
  int H = a.H;
  if (a.width) H = min(H, *a.width);
  float wcol[kKP];  // W[k][lane]
#pragma unroll
  for (int k = 0; k < kKP; ++k) wcol[k] = a.w[k * kC1Out + lane];
  const float bias_v = a.bias[lane];
  double st_s = 0.0, st_ss = 0.0;
  const int m0 = (blockIdx.x * kC1Waves + wave) * kC1Qpw;
  for (int qq = 0; qq < kC1Qpw; ++qq) {
    const int m = m0 + qq;
    if (m >= a.M) break;
    const float qx = a.q_points[3 * m], qy = a.q_points[3 * m + 1], qz = a.q_points[3 * m + 2];
    int positives = 0;
    float acc = 0.f;  // lane (g, j): kernel point j over neighbours g, g+4, ...
    for (int hc = 0; hc < H; hc += kMaxH) {  // chunks of the staging row
      const int Hc = min(H - hc, kMaxH);
      int Hq = 0;  // slots up to the last real neighbour
      for (int hb = 0; hb < Hc; hb += 64) {
        const int h = hb + lane;
        const int64_t id = h < Hc ? ld_index(a.idx, static_cast<int64_t>(m) * a.ldi + hc + h, a.i32) : -1;
        const unsigned long long rm = __builtin_amdgcn_ballot_w64(id >= 0 && id < a.Ns);
        if (rm) Hq = hb + 64 - __builtin_clzll(rm);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0 && id < a.Ns) {
          v.x = a.s_points[3 * id] - qx;
          v.y = a.s_points[3 * id + 1] - qy;
          v.z = a.s_points[3 * id + 2] - qz;
          v.w = a.s_feats[id * a.ldf];
          positives += a.s_pos[id];
        }
        if (h < Hc) nb[h] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int h = g; h < Hq; h += 4) {
        const float4 v = nb[h];
        const float dx = v.x - kx, dy = v.y - ky, dz = v.z - kz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        acc += fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(d2) * inv_sigma) * v.w;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // nb is rewritten by the next chunk / query
      __builtin_amdgcn_wave_barrier();
    }
    positives = wave_sum_i(positives);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);  // every lane (., j) now holds WF[k = j]
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < kKP; ++k)
      o = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), k)), wcol[k], o);
    float v = o / static_cast<float>(positives > 1 ? positives : 1);
    v += bias_v;
    a.out[static_cast<int64_t>(m) * a.ldo + lane] = v;
    st_s += static_cast<double>(v);
    st_ss += static_cast<double>(v) * static_cast<double>(v);
  }
  if (a.stats) {
    ex[wave][lane][0] = st_s;
    ex[wave][lane][1] = st_ss;
    __syncthreads();
    if (wave == 0) {
      double s = 0.0, ss = 0.0;
#pragma unroll
      for (int w = 0; w < kC1Waves; ++w) {  // fixed order
        s += ex[w][lane][0];
        ss += ex[w][lane][1];
      }
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * kC1Out + lane] = s;
      a.stats[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * kC1Out + lane] = ss;
    }
  }
}
__global__ __launch_bounds__(64 * kC1Waves) void kpconv_fused_c1_kernel(FusedArgs a) { kpconv_fused_c1_kernel_body(blockIdx, gridDim, a); }


template <int C, int QB, int NW>
constexpr size_t fused_lds_bytes() { return sizeof(float) * (static_cast<size_t>(QB) * (kKP * C + 4) + NW * kMaxH * 4 + QB); }

// C = 32: 16 wavefronts x 1 query, 31 KB block + 32 KB staging = 63 KB -> two workgroups (32 wavefronts) per CU, the
// occupancy of the stand-alone gather; C = 64: the 62 KB block leaves room for 8 wavefronts x 2 queries (78 KB, two per CU)
constexpr int kQb32 = 16, kNw32 = 16, kIters32 = 1, kQb64 = 16, kNw64 = 8, kIters64 = 1;  // (16 rows per workgroup in every form: one partial-row count)

}  // namespace

// Whether the engine and the per-op path route the fine levels (C_in = 1, 32, 64) through the fused kernel.  ON by
// default since round 3: the [M, 15*C] tensor between gather and weight product was a 13x write amplification of
// these layers (profiles/r02_pmc_fetch_write.md); RDM_FUSED_KPCONV=0 selects the two-kernel form for A/B runs.
extern "C" int rdm_kpconv_fused_enabled(void) {
  static const bool on = [] {
    const char* v = ::rdm::dev_knob("RDM_FUSED_KPCONV");
    return !(v != nullptr && v[0] == '0');
  }();
  return on ? 1 : 0;
}

extern "C" int rdm_kpconv_fused_supported(int64_t c_in, int64_t c_out) {
  return (c_in == 1 && c_out == kC1Out) || (c_in == 32 && c_out == 32) || (c_in == 64 && c_out == 64);
}

namespace {
// The LDS-tile kernel serves c_in = 32 / 64 with neighbour rows of at most 128 slots (every KITTI limit); wider rows and
// RDM_KPCONV_TILE=0 (lab build, A/B runs) take the lock-step kernel.
// form: 0 = the library's choice, 1 = the lock-step kernel, 2 = the LDS-tile kernel where it applies.  The choice (measured per
// shape on MI355X, tools/kpconv_bench.py): c_in = 32 always (59.4 / 28.6 us against 60.7 / 30.7 at the first level and its
// strided block); c_in = 64 when queries and support are the same level (44 against 51 us) -- a strided block's 16 queries (of
// the coarser level) reach 145 ... 240+ distinct rows, beyond the 240 the tile holds beside the parked block, and the rows
// that overflow are fetched the old way on top of the tile's fixed costs (23.7 against 18.3 us).  Without order records (queries
// in row order = the hash-map order of the subsampling: spatially random) there is nothing to share: the lock-step kernel.
bool use_tile(int64_t c_in, int64_t h, int64_t m, int64_t n_s, bool has_order, int form = 0) {
  static const bool on = [] {
    const char* v = ::rdm::dev_knob("RDM_KPCONV_TILE");
    return !(v != nullptr && v[0] == '0');
  }();
  if (!((c_in == 32 || c_in == 64) && h <= kMaxH)) return false;
  if (form != 0) return form >= 2;
  return on && has_order && (c_in == 32 || 2 * m > n_s);
}
int64_t rows_per_block(int64_t c_in) {  // (the same in every form: the caller sizes the partial array before the form is chosen)
  static_assert(kTileQB == kQb32 * kIters32 && kTileQB == kQb64 * kIters64, "one partial-row count for both forms");
  return c_in == 1 ? kC1Waves * kC1Qpw : kTileQB;
}
}  // namespace

// The gather-only tile form for kpconv_gather_impl (kpconv.hip): WF [m, ldw] and nn [m] exactly as kpconv_gather_kernel writes them.
// Applies to c a multiple of 64 (>= 128 in the backbone), rows of at most 128 slots, with order records.
bool rdm::kpconv_tile_gather_applies(int64_t c, int64_t h, bool has_order) {
  return has_order && c >= 128 && c % 64 == 0 && h <= kMaxH;
}
int rdm::kpconv_tile_gather(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats, int64_t c,
                            int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                            const float* kernel_points, float sigma, float* wf, int64_t ldw, float* nn, const float* order_records,
                            int i32, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(kpconv_tile_gather_applies(c, h, order_records != nullptr), "kpconv_tile_gather: unsupported shape");
  RDM_REQUIRE(ldf % 4 == 0 && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(s_feats) & 15) == 0 && (reinterpret_cast<uintptr_t>(wf) & 15) == 0,
              "kpconv_tile_gather: features / WF must be 16-byte aligned with row strides that are multiples of 4");
  if (m == 0) return RDM_OK;
  FusedArgs a{};
  a.q_points = q_points; a.s_points = s_points; a.s_feats = s_feats; a.s_pos = s_positive; a.idx = idx; a.i32 = i32 ? 1 : 0;
  a.kp = kernel_points; a.width = width; a.w = nullptr; a.bias = nullptr; a.out = wf; a.stats = nullptr;
  a.M = static_cast<int>(m); a.Ns = static_cast<int>(n_s); a.H = static_cast<int>(h);
  a.ldf = static_cast<int>(ldf); a.ldi = static_cast<int>(ldi); a.ldo = static_cast<int>(ldw); a.sigma = sigma;
  a.ctot = static_cast<int>(c); a.nn = nn;
  static std::atomic<uint64_t> gattr{0};
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_tile_kernel<64, true>), static_cast<int>(TileLds<64>::bytes), gattr));
  const dim3 grid(static_cast<unsigned>(ceil_div<int64_t>(m, kTileQB)), static_cast<unsigned>(c / 64));
  RDM_DUP_LOOP("gather")
  ::rdm::launch<kpconv_tile_kernel_body<64, true>, kpconv_tile_kernel<64, true>, 64 * TileCfg<64>::NW, 2 * TileCfg<64>::NW / 4>(grid, TileLds<64>::bytes, static_cast<hipStream_t>(stream), a,
                     reinterpret_cast<const float4*>(order_records), 1);
  return launch_status("kpconv_tile_kernel<64, gather>");
}

// Rows of the fp64 GroupNorm partial array [rows][2][c_out] a call with m queries and neighbour rows of h slots writes
// (one per workgroup).
extern "C" int64_t rdm_kpconv_fused_partial_rows(int64_t m, int64_t c_in) {
  return m <= 0 ? 0 : rdm::ceil_div<int64_t>(m, rows_per_block(c_in));
}

extern "C" size_t rdm_kpconv_packed_floats(int64_t c_in, int64_t c_out) {
  return c_in == 1 ? static_cast<size_t>(16) * c_out : static_cast<size_t>(kKP) * c_in * c_out;
}

// w [15, c_in, c_out] (the checkpoint layout, host) -> the B-operand order of the fused kernel (host):
//   c_in = 1: [16, c_out] (row 15 zero);  otherwise float4 records [s][column tile][lane = 16 g + j] holding
//   W[16 s + 4 g + e][16 tile + j], e = 0..3 (the row index of W is k * c_in + c).
extern "C" int rdm_kpconv_pack_weights(const float* w, int64_t c_in, int64_t c_out, float* packed) {
  using namespace rdm;
  RDM_REQUIRE(w && packed && rdm_kpconv_fused_supported(c_in, c_out), "rdm_kpconv_pack_weights: unsupported (%lld -> %lld)",
              (long long)c_in, (long long)c_out);
  if (c_in == 1) {
    for (int64_t k = 0; k < 16; ++k)
      for (int64_t c = 0; c < c_out; ++c) packed[k * c_out + c] = k < kKP ? w[k * c_out + c] : 0.f;
    return RDM_OK;
  }
  const int64_t nt = c_out / 16, k16 = kKP * c_in / 16;
  for (int64_t s = 0; s < k16; ++s)
    for (int64_t t = 0; t < nt; ++t)
      for (int64_t lane = 0; lane < 64; ++lane)
        for (int64_t e = 0; e < 4; ++e)
          packed[((s * nt + t) * 64 + lane) * 4 + e] = w[(16 * s + 4 * (lane >> 4) + e) * c_out + 16 * t + (lane & 15)];
  return RDM_OK;
}

extern "C" int rdm_kpconv_fused(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats,
                                int64_t c, int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h,
                                int64_t ldi, const int32_t* width, const float* kernel_points, float sigma,
                                const float* w_packed, const float* bias, int64_t c_out, float* out, int64_t ldo,
                                double* gn_partial, const float* order_records, void* stream) {
  return rdm_kpconv_fused_form(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points, sigma,
                               w_packed, bias, c_out, out, ldo, gn_partial, order_records, 0, stream);
}

extern "C" int rdm_kpconv_fused_form(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats,
                                     int64_t c, int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h,
                                     int64_t ldi, const int32_t* width, const float* kernel_points, float sigma,
                                     const float* w_packed, const float* bias, int64_t c_out, float* out, int64_t ldo,
                                     double* gn_partial, const float* order_records, int form, void* stream) {
  return rdm::kpconv_fused_impl(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points, sigma,
                                w_packed, bias, c_out, out, ldo, gn_partial, order_records, form, 0, stream);
}

int rdm::kpconv_fused_impl(const float* q_points, int64_t m, const float* s_points, int64_t n_s, const float* s_feats, int64_t c,
                           int64_t ldf, const uint8_t* s_positive, const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                           const float* kernel_points, float sigma, const float* w_packed, const float* bias, int64_t c_out,
                           float* out, int64_t ldo, double* gn_partial, const float* order_records, int form, int i32, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(form >= 0 && form <= 3, "rdm_kpconv_fused_form: form must be 0, 1 or 2 (3: lab variant of 2 without the XCD ranges)");
  RDM_REQUIRE(q_points && s_points && s_feats && s_positive && idx && kernel_points && w_packed && bias && out,
              "rdm_kpconv_fused: null pointer");
  RDM_REQUIRE(rdm_kpconv_fused_supported(c, c_out), "rdm_kpconv_fused: unsupported channel counts %lld -> %lld", (long long)c,
              (long long)c_out);
  RDM_REQUIRE(m >= 0 && n_s > 0 && h > 0 && ldo >= c_out, "rdm_kpconv_fused: bad sizes (h=%lld)", (long long)h);
  RDM_REQUIRE(c == 1 || (ldf % 4 == 0 && (reinterpret_cast<uintptr_t>(s_feats) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(w_packed) & 15) == 0),
              "rdm_kpconv_fused: features / packed weights must be 16-byte aligned with a row stride that is a multiple of 4");
  if (m == 0) return RDM_OK;
  FusedArgs a;
  a.q_points = q_points; a.s_points = s_points; a.s_feats = s_feats; a.s_pos = s_positive; a.idx = idx; a.i32 = i32 ? 1 : 0;
  a.kp = kernel_points;
  a.width = width; a.w = w_packed; a.bias = bias; a.out = out; a.stats = gn_partial;
  a.M = static_cast<int>(m); a.Ns = static_cast<int>(n_s); a.H = static_cast<int>(h);
  a.ldf = static_cast<int>(ldf); a.ldi = static_cast<int>(ldi); a.ldo = static_cast<int>(ldo); a.sigma = sigma;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>(rdm_kpconv_fused_partial_rows(m, c));
  RDM_DUP_LOOP("fused") {
  if (c == 1) {
    // (measured and dropped in round 4: the queries in cell order -- 28.2 against 26.8 us at the first level -- and the index rows
    // and gathers of a wavefront's four queries requested together -- 30.4 us; round 5: halving the instruction count and
    // pipelining the queries buys 15 %, four queries per pipeline stage lose it again: the address processing of the
    // scattered gathers bounds the kernel, see the note above kpconv_fused_c1_kernel)
    ::rdm::launch<kpconv_fused_c1_kernel_body, kpconv_fused_c1_kernel, 64 * kC1Waves>(dim3(blocks), 0, st, a);
    continue;
  }
  if (use_tile(c, h, m, n_s, order_records != nullptr, form)) {  // the support rows of 16 cell-ordered queries staged once in LDS
    static std::atomic<uint64_t> tattr32{0}, tattr64{0};
    const float4* order = reinterpret_cast<const float4*>(order_records);
    const int xcd_ranges = form == 3 ? 0 : 1;  // (form 3, lab: block ids in dispatch order, as in round 4)
    if (c == 32) {
      RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_tile_kernel<32>), static_cast<int>(TileLds<32>::bytes), tattr32));
      ::rdm::launch<kpconv_tile_kernel_body<32>, kpconv_tile_kernel<32>, 64 * TileCfg<32>::NW, (TileCfg<32>::NW == 8 ? 3 : 2) * TileCfg<32>::NW / 4>(dim3(blocks), TileLds<32>::bytes, st, a, order, xcd_ranges);
    } else {
      RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_tile_kernel<64>), static_cast<int>(TileLds<64>::bytes), tattr64));
      ::rdm::launch<kpconv_tile_kernel_body<64>, kpconv_tile_kernel<64>, 64 * TileCfg<64>::NW, 2 * TileCfg<64>::NW / 4>(dim3(blocks), TileLds<64>::bytes, st, a, order, xcd_ranges);
    }
    continue;
  }
  // (the C = 64 instance needs > 64 KB of dynamic LDS: the attribute is set once per device)
  static std::atomic<uint64_t> attr32{0}, attr64{0};
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_fused_kernel<32, kQb32, kNw32, kIters32>),
                                    static_cast<int>(fused_lds_bytes<32, kQb32, kNw32>()), attr32));
  RDM_HIP_CHECK(set_max_dynamic_lds(reinterpret_cast<const void*>(kpconv_fused_kernel<64, kQb64, kNw64, kIters64>),
                                    static_cast<int>(fused_lds_bytes<64, kQb64, kNw64>()), attr64));
  const size_t lds32 = fused_lds_bytes<32, kQb32, kNw32>(), lds64 = fused_lds_bytes<64, kQb64, kNw64>();
  if (c == 32)
    ::rdm::launch<kpconv_fused_kernel_body<32, kQb32, kNw32, kIters32>, kpconv_fused_kernel<32, kQb32, kNw32, kIters32>, 64 * kNw32>(dim3(blocks), lds32, st, a);
  else
    ::rdm::launch<kpconv_fused_kernel_body<64, kQb64, kNw64, kIters64>, kpconv_fused_kernel<64, kQb64, kNw64, kIters64>, 64 * kNw64>(dim3(blocks), lds64, st, a);
  }
  return launch_status("kpconv_fused_kernel");
}

extern "C" size_t rdm_kpconv_fused_workspace_bytes(int64_t m, int64_t c_in, int64_t c_out) {
  const size_t nblk = static_cast<size_t>(rdm_kpconv_fused_partial_rows(m > 0 ? m : 1, c_in));
  return rdm::align_up(nblk * 2 * c_out * sizeof(double)) + rdm_group_norm_workspace_bytes(m, c_out) + 256;
}

// KPConv + the GroupNorm (+ activation) that follows it in every block of the backbone (modules.py:141-145, 205-207):
// conv_out receives the convolution, y = act(GroupNorm(conv_out)).  The statistics come from the fused kernel's epilogue.
extern "C" int rdm_kpconv_fused_group_norm(const float* q_points, int64_t m, const float* s_points, int64_t n_s,
                                           const float* s_feats, int64_t c, int64_t ldf, const uint8_t* s_positive,
                                           const int64_t* idx, int64_t h, int64_t ldi, const int32_t* width,
                                           const float* kernel_points, float sigma, const float* w_packed, const float* bias,
                                           int64_t c_out, int groups, const float* gamma, const float* beta, float eps, int act,
                                           float* conv_out, int64_t ld_conv, float* y, int64_t ldy, void* ws, size_t ws_bytes,
                                           const float* order_records, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(gamma && beta && conv_out && y, "rdm_kpconv_fused_group_norm: null pointer");
  if (m == 0) return RDM_OK;
  Arena ar(ws, ws_bytes);
  const int nblk = static_cast<int>(rdm_kpconv_fused_partial_rows(m, c));
  double* partial = ar.take<double>(static_cast<size_t>(nblk) * 2 * c_out);
  const size_t gn_ws = rdm_group_norm_workspace_bytes(m, c_out);
  char* nws = ar.take<char>(gn_ws);
  if (!ar.ok) {
    set_error("rdm_kpconv_fused_group_norm: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  if (int e = rdm_kpconv_fused(q_points, m, s_points, n_s, s_feats, c, ldf, s_positive, idx, h, ldi, width, kernel_points, sigma,
                               w_packed, bias, c_out, conv_out, ld_conv, partial, order_records, stream))
    return e;
  return group_norm_finish(partial, nblk, conv_out, m, c_out, ld_conv, groups, gamma, beta, eps, nullptr, 0, act, y, ldy, nullptr,
                           nws, gn_ws, stream);
}

#ifdef RDM_TILE_TIMING
extern "C" int rdm_dbg_tile_timing(long long* buffer) {
  RDM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(rdm_tile_clk), &buffer, sizeof(buffer)));
  return rdm::RDM_OK;
}
extern "C" int rdm_dbg_tile_wave_timing(long long* buffer) {
  RDM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(rdm_tile_wclk), &buffer, sizeof(buffer)));
  return rdm::RDM_OK;
}
#endif
