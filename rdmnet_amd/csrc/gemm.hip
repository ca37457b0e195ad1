// fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TF peak).
//
//   C[b] = act( (A[b] (MxK) * op(B[b])) / rowdiv[m] + bias[n] )
//     op(B) = B            B stored [K, N] row-major   (trans_b = 0; pre-transposed weights)
//     op(B) = B^T          B stored [N, K] row-major   (trans_b = 1; activations, e.g. f_r * f_s^T)
//
// Used for every dense contraction of the path that is not an attention: Linear layers of the
// KPConv blocks (modules/kpconv/modules.py:53-101), the [M, 15*C] x [15*C, C'] kernel-weight
// contraction of KPConv (modules/kpconv/kpconv.py:107-110), transformer projections / FFN, the vote
// MLP, coarse-matching similarities and the batched patch score einsum (model_infer.py:310-311).
//
// Tiling: 256 threads = 4 wavefronts; block tile BM x BN x BK (BK = 16/32/64), LDS double-buffered, operands stored
// k-major in LDS so a 32x32x2 fragment read is two conflict-free 128-B rows.  Split-K (partials +
// fixed-order reduce, no atomics => deterministic) keeps the small-M / huge-K coarse levels busy.
// Requirements: lda, ldb, K multiples of 4 and 16-byte aligned bases (callers pad with zeros).
#include "../../include/rdmnet_hip.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "internal.h"
#include "lockstep.h"

namespace {

using namespace rdm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;    // [N] or null
  const float* rowdiv;  // [M] or null
  int M, N, K;
  int lda, ldb, ldc;
  long long sa, sb, sc;  // batch strides (elements)
  int act;               // 0 none, 1 relu, 2 leaky-relu(0.1)
  int splits;            // split-K factor (partials go to `part`)
  float* part;           // [batch*splits, M, N] when splits > 1
  double* stats;         // optional GroupNorm partials [grid.y][2][N] (non split-K only)
  // CAT kernels only -- A is the decoder's [nearest_upsample(coarse) | skip] (backbone.py:118-151, functional.py:6-22)
  // without materialising it: columns k < c1 of row m are coarse[aidx[m * ldi]] (row index out of range: zeros), the rest
  // skip[m]; A = coarse, lda its row stride; c1 a multiple of the k-tile depth
  const float* A2;
  const int64_t* aidx;
  int lda2, ldi, c1, n_coarse;
  // CAT + TRANS_B ("patch scores", model_infer.py:291-311): row m of batch b of A is A[aidx[b * M + m]] and row n of B is
  // B[bidx[b * N + n]] (an index outside [0, n_coarse) / [0, n_b): a zero row, as the reference's padded gather gives); tiles
  // whose rows or columns are all shadow rows are written as zeros without touching the features
  const int64_t* bidx;
  int n_b;
  int xcd_tiles;  // 1: output tiles re-mapped so that an XCD (workgroup id % 8) owns whole row tiles with all their column tiles
#ifdef RDM_GEMM_TIMING
  unsigned long long* clk;  // tools/gemm_phase_lab.hip: shader-clock stamps of workgroup (0,0,0), thread 0
#endif
};
#ifdef RDM_GEMM_TIMING
#define GEMM_STAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g.clk[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEMM_STAMP(k) do { } while (0)
#endif
#ifdef RDM_GEMM_TIMING  // (wide form: stamps of workgroup (0,0,0), wavefront 0; per-tile stamps for the first 128 k-tiles)
#define GEMM_WIDE_STAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (k) < 392) g.clk[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEMM_WIDE_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return v > 0.f ? v : 0.1f * v;
  return v;
}

// PF = global->register prefetch depth in K tiles (the loop barriers order LDS only, so the stages really stay in
// flight): the skinny products of this path run a handful of blocks per CU and a block covers part of the load
// latency itself.
template <int BM, int BN, int WM, int WN, int BK, bool TRANS_B, int PF, bool CAT = false>
__device__ __forceinline__ void gemm_kernel_body(const dim3 blockIdx, const dim3 gridDim, GemmArgs g) {
  (void)blockIdx; (void)gridDim;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0, "bad tile");
  // transposed-staged tiles use an odd row stride (scalar LDS writes of one k-column hit distinct
  // banks); the row-major B tile is written as float4 and keeps a 16-byte-aligned stride
  constexpr int LDA_S = BM + 1, LDB_S = TRANS_B ? BN + 1 : BN + 4;
  // one LDS buffer: the A/B tiles during the K loop, then the row-major staging tile of the epilogue, then the
  // exchange of the GroupNorm partial sums
  constexpr int CR = BM < 64 ? BM : 64;   // output rows staged per epilogue pass
  constexpr int LDC_S = BN + 4;
  constexpr int TPR = BN / 4;             // threads per output row (one float4 each)
  constexpr int RPI = 256 / TPR;          // rows per store iteration
  constexpr int kBytesAB = (2 * BK * LDA_S + 2 * BK * LDB_S) * 4, kBytesC = CR * LDC_S * 4, kBytesStat = RPI * BN * 2 * 8;
  constexpr int kBytes = kBytesAB > kBytesC ? (kBytesAB > kBytesStat ? kBytesAB : kBytesStat) : (kBytesC > kBytesStat ? kBytesC : kBytesStat);
  static_assert((2 * BK * LDA_S * 4) % 16 == 0 && CR % RPI == 0, "tile layout");
  __shared__ __attribute__((aligned(16))) char smem[kBytes];
  float (*As)[BK][LDA_S] = reinterpret_cast<float (*)[BK][LDA_S]>(smem);
  float (*Bs)[BK][LDB_S] = reinterpret_cast<float (*)[BK][LDB_S]>(smem + 2 * BK * LDA_S * 4);
  float (*Cs)[LDC_S] = reinterpret_cast<float (*)[LDC_S]>(smem);
  double (*stat_red)[1][2] = reinterpret_cast<double (*)[1][2]>(smem);    // [RPI * BN][1][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tile_x = blockIdx.x, tile_y = blockIdx.y;
  if (g.xcd_tiles) {  // (workgroups are dispatched x-fastest, round-robin over the 8 XCDs)
    const int nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nblk / 8, rr = nblk % 8, xcd = lin % 8, within = lin / 8;
    const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + within;
    tile_x = logical % gridDim.x;
    tile_y = logical / gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  GEMM_STAMP(0);
  const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
  const float* A = g.A + batch * g.sa;
  const float* B = g.B + batch * g.sb;

  // K range of this split (multiples of BK)
  const int ktiles = (g.K + BK - 1) / BK;
  const int per = (ktiles + g.splits - 1) / g.splits;
  const int kt0 = split * per, kt1 = min(ktiles, kt0 + per);

  constexpr int A_N4 = BM * BK / 4;                 // float4 in the A tile
  constexpr int A_V = (A_N4 + 255) / 256;           // float4 per thread for the A tile
  constexpr int KC4 = BK / 4;                       // float4 per tile row
  constexpr int B_N4 = BK * BN / 4;                 // float4 in the B tile
  constexpr int B_V = (B_N4 + 255) / 256;           // float4 per thread for the B tile
  static_assert(PF == 1 || PF == 2, "one or two register stages");
  // named stages (an array of stages indexed in a loop ended up in scratch memory)
  float4 ra0[A_V], rb0[B_V], ra1[PF == 2 ? A_V : 1], rb1[PF == 2 ? B_V : 1];
  // CAT: the gathered coarse row of each A row this thread stages (fixed over the K loop), -1 = shadow row (zeros)
  long long cat_row[CAT ? A_V : 1], cat_brow[CAT && TRANS_B ? B_V : 1];
  if constexpr (CAT && !TRANS_B) {
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      const int row = (tid + i * 256) / (BK / 4);
      const long long id = g.aidx[static_cast<long long>(min(m0 + row, g.M - 1)) * g.ldi];
      cat_row[i] = (id >= 0 && id < g.n_coarse) ? id : -1;
    }
  }
  if constexpr (CAT && TRANS_B) {
    A = g.A;  // gathered rows: no batch stride
    B = g.B;
    int any_a = 0, any_b = 0;
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      const int row = (tid + i * 256) / (BK / 4);
      const long long id = (m0 + row < g.M && (A_N4 % 256 == 0 || tid + i * 256 < A_N4)) ? g.aidx[static_cast<long long>(batch) * g.M + m0 + row] : -1;
      cat_row[i] = (id >= 0 && id < g.n_coarse) ? id : -1;
      any_a |= cat_row[i] >= 0;
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
      const int row = (tid + i * 256) / (BK / 4);
      const long long id = (n0 + row < g.N && (B_N4 % 256 == 0 || tid + i * 256 < B_N4)) ? g.bidx[static_cast<long long>(batch) * g.N + n0 + row] : -1;
      cat_brow[i] = (id >= 0 && id < g.n_b) ? id : -1;
      any_b |= cat_brow[i] >= 0;
    }
    any_a = __syncthreads_or(any_a);
    any_b = __syncthreads_or(any_b);
    if (!any_a || !any_b) {  // workgroup-uniform: a tile of zero rows (or zero columns) is zero
      float* Cz = g.C + batch * g.sc;
      for (int e = tid; e < BM * BN; e += 256) {
        const int r = m0 + e / BN, cc = n0 + e % BN;
        if (r < g.M && cc < g.N) Cz[static_cast<long long>(r) * g.ldc + cc] = 0.f;
      }
      return;
    }
  }

  // Loads are unconditional from clamped (always valid) addresses, the out-of-range parts are zeroed when the stage is
  // written to LDS, and the K loop issues and stores a stage on every trip: no control flow in the loop body, so that
  // (a) the compiler waits for the stage it needs (vmcnt(N)) and not for all outstanding loads, and (b) the whole
  // trip is one scheduling region in which the address arithmetic, the loads and the LDS stores of the next tile can
  // be placed BETWEEN the MFMAs of this one.  (K, lda, ldb are multiples of 4.)
  // (pieces [p0, p1) of the A_V + B_V float4 a thread moves per tile: the K loop spreads them over the shadows of its MFMAs)
  auto load_tiles = [&](float4 (&ra)[A_V], float4 (&rb)[B_V], int kt, int p0 = 0, int p1 = 1 << 20) {
    const int k0 = min(kt, kt1 - 1) * BK;  // past the end: the last tile again (stored to a buffer nobody reads)
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      if (i < p0 || i >= p1) continue;
      const int idx = tid + i * 256;
      const int row = idx / KC4, kc = (idx % KC4) * 4;
      if constexpr (CAT && TRANS_B) {
        ra[i] = *reinterpret_cast<const float4*>(A + max(cat_row[i], 0ll) * g.lda + min(k0 + kc, g.K - 4));
      } else if constexpr (CAT) {  // (a k-tile lies on one side of c1: the choice is workgroup-uniform)
        const float* src = k0 < g.c1 ? A + max(cat_row[i], 0ll) * g.lda + (k0 + kc)
                                     : g.A2 + static_cast<long long>(min(m0 + row, g.M - 1)) * g.lda2 + min(k0 - g.c1 + kc, g.K - g.c1 - 4);
        ra[i] = *reinterpret_cast<const float4*>(src);
      } else {
      ra[i] = *reinterpret_cast<const float4*>(A + static_cast<long long>(min(m0 + row, g.M - 1)) * g.lda + min(k0 + kc, g.K - 4));
      }
    }
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
      if (A_V + i < p0 || A_V + i >= p1) continue;
      const int idx = tid + i * 256;
      if (TRANS_B) {
        const int row = idx / KC4, kc = (idx % KC4) * 4;
        if constexpr (CAT) {
          rb[i] = *reinterpret_cast<const float4*>(B + max(cat_brow[i], 0ll) * g.ldb + min(k0 + kc, g.K - 4));
        } else {
        rb[i] = *reinterpret_cast<const float4*>(B + static_cast<long long>(min(n0 + row, g.N - 1)) * g.ldb + min(k0 + kc, g.K - 4));
        }
      } else {
        const int k = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
        rb[i] = *reinterpret_cast<const float4*>(B + static_cast<long long>(min(k0 + k, g.K - 1)) * g.ldb + min(n0 + n4, g.ldb - 4));
      }
    }
  };
  // (component-wise select: `ok ? v : zero4` on two float4 lvalues is lowered as a select of ADDRESSES through scratch)
  auto masked = [](const float4& v, bool ok) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); };
  auto store_tiles = [&](const float4 (&ra)[A_V], const float4 (&rb)[B_V], int buf, int kt, int p0 = 0, int p1 = 1 << 20) {
    const int k0 = min(kt, kt1 - 1) * BK;
#pragma unroll
    for (int i = 0; i < A_V; ++i) {
      if (i < p0 || i >= p1) continue;
      const int idx = tid + i * 256;
      const int row = idx / KC4, kc = (idx % KC4) * 4;
      if (A_N4 % 256 != 0 && idx >= A_N4) continue;
      bool a_ok = m0 + row < g.M && k0 + kc < g.K;
      if constexpr (CAT && TRANS_B) a_ok = a_ok && cat_row[i] >= 0;
      else if constexpr (CAT) a_ok = a_ok && (k0 >= g.c1 || cat_row[i] >= 0);
      const float4 v = masked(ra[i], a_ok);
      As[buf][kc + 0][row] = v.x;
      As[buf][kc + 1][row] = v.y;
      As[buf][kc + 2][row] = v.z;
      As[buf][kc + 3][row] = v.w;
    }
    if (TRANS_B) {
#pragma unroll
      for (int i = 0; i < B_V; ++i) {
        if (A_V + i < p0 || A_V + i >= p1) continue;
        const int idx = tid + i * 256;
        const int row = idx / KC4, kc = (idx % KC4) * 4;
        if (B_N4 % 256 != 0 && idx >= B_N4) continue;
        bool b_ok = n0 + row < g.N && k0 + kc < g.K;
        if constexpr (CAT) b_ok = b_ok && cat_brow[i] >= 0;
        const float4 v = masked(rb[i], b_ok);
        Bs[buf][kc + 0][row] = v.x;
        Bs[buf][kc + 1][row] = v.y;
        Bs[buf][kc + 2][row] = v.z;
        Bs[buf][kc + 3][row] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < B_V; ++i) {
        if (A_V + i < p0 || A_V + i >= p1) continue;
        const int idx = tid + i * 256;
        const int k = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
        if (B_N4 % 256 != 0 && idx >= B_N4) continue;
        // ldb and the column are multiples of 4, so a quad is inside the row or outside it (pad columns of B hold zeros)
        *reinterpret_cast<float4*>(&Bs[buf][k][n4]) = masked(rb[i], k0 + k < g.K && n0 + n4 < g.ldb);
      }
    }
  };

  // epilogue operands of this thread (bias of its 4 columns, divisor of its rows), requested before the K loop: at the
  // end they would each cost a full memory round trip with nothing left to hide it
  const bool partial = g.splits > 1;
  const int c4 = (tid % TPR) * 4, rsub = tid / TPR;
  const int gcol = n0 + c4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  float rdv[BM / RPI];
  if (!partial && g.bias)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = gcol + e < g.N ? g.bias[gcol + e] : 0.f;
#pragma unroll
  for (int q = 0; q < BM / RPI; ++q) rdv[q] = (!partial && g.rowdiv) ? g.rowdiv[min(m0 + q * RPI + rsub, g.M - 1)] : 1.f;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lk = lane >> 5, li = lane & 31;
  // One k-tile in two halves (round 5).  The operands of a half live in registers one half-step ahead: while the MFMAs of the
  // tile's FIRST half run, its second half is read from LDS (and the next tile goes from the register stage into the other LDS
  // buffer, the tile after it from global memory into the free stage); after the barrier that publishes that buffer, the first
  // half of the NEXT tile is read while the MFMAs of this tile's second half run.  A wavefront therefore never waits for an LDS
  // read in front of its MFMAs (the round-4 form read all 32 operands at the top of the tile: ~250 clocks per 1024 of MFMA work
  // in which the wavefront issued nothing; matrix pipe 65 % busy at full occupancy against the vendor library's 93 %,
  // tools/gemm_pmc_probe.py).  Every accumulator still receives its MFMAs in ascending k: the same bits.
  constexpr int HK = BK / 2, HS = HK / 2;  // k per half, MFMA k-steps per half
  float af_lo[HS][FM], bf_lo[HS][FN], af_hi[HS][FM], bf_hi[HS][FN];
  // operands of k-step s (two k) of a half: FM + FN LDS reads
  auto read_kstep = [&](float (&af)[HS][FM], float (&bf)[HS][FN], int buf, int kbase, int s) {
#pragma unroll
    for (int i = 0; i < FM; ++i) af[s][i] = As[buf][kbase + 2 * s + lk][wm * TM + i * 32 + li];
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[s][j] = Bs[buf][kbase + 2 * s + lk][wn * TN + j * 32 + li];
  };
  auto read_half = [&](float (&af)[HS][FM], float (&bf)[HS][FN], int buf, int kbase) {
#pragma unroll
    for (int s = 0; s < HS; ++s) read_kstep(af, bf, buf, kbase, s);
  };
  // A wavefront issues in order and a dependent MFMA holds its stream for 64 clocks: everything else of a half-tile has to stand
  // BETWEEN the MFMAs in program order to run in their shadows.  The compiler's scheduler does not keep it there on its own (and
  // scheduling-group hints held for one of the two unrolled trips only: in the other, ~60 VALU / memory instructions ran after
  // the last MFMA of the half with the matrix pipe idle), so the half is written as NCH chunks -- one MFMA, then that chunk's
  // share of the LDS reads, global loads and LDS stores -- fenced by scheduling barriers.
  constexpr int NCH = HS * FM * FN, NP = A_V + B_V;
  auto step = [&](auto& fa, auto& fb, const auto& na, const auto& nb, int kt, int buf) {
    // ---- first half: operands already in af_lo / bf_lo; the second half's arrive meanwhile
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int s = c / (FM * FN), i = (c / FN) % FM, j = c % FN;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_lo[s][i], bf_lo[s][j], acc[i][j], 0, 0, 0);
      if (c % (FM * FN) == 0) read_kstep(af_hi, bf_hi, buf, HK, s);
      // this chunk's global loads of tile kt + PF (into the free register stage: pieces in the first chunks) ...
      load_tiles(fa, fb, kt + PF, NCH >= NP ? (c < NP ? c : NP) : (c * NP) / NCH, NCH >= NP ? (c < NP ? c + 1 : NP) : ((c + 1) * NP) / NCH);
      // ... and LDS stores of tile kt + 1 (from the other stage, pieces in the last chunks; on the last trip a copy of the last
      // tile, never multiplied)
      store_tiles(na, nb, buf ^ 1, kt + 1, NCH >= NP ? (c >= NCH - NP ? c - (NCH - NP) : NP) : (c * NP) / NCH,
                  NCH >= NP ? (c >= NCH - NP ? c - (NCH - NP) + 1 : NP) : ((c + 1) * NP) / NCH);
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();  // LDS only: the register prefetch stages stay in flight
    __builtin_amdgcn_sched_barrier(0);
    // ---- second half; the next tile's first half arrives meanwhile
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int s = c / (FM * FN), i = (c / FN) % FM, j = c % FN;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_hi[s][i], bf_hi[s][j], acc[i][j], 0, 0, 0);
      if (c % (FM * FN) == 0) read_kstep(af_lo, bf_lo, buf ^ 1, 0, s);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (kt0 < kt1) {  // (an empty K range -- more splits than k-tiles -- leaves the accumulators at zero)
    load_tiles(ra0, rb0, kt0);
    if constexpr (PF == 2) load_tiles(ra1, rb1, kt0 + 1);
    store_tiles(ra0, rb0, 0, kt0);
  }
  lds_barrier();
  GEMM_STAMP(1);
  if (kt0 < kt1) {
    read_half(af_lo, bf_lo, 0, 0);
    if constexpr (PF == 1) {
      for (int kt = kt0; kt < kt1; ++kt) step(ra0, rb0, ra0, rb0, kt, (kt - kt0) & 1);
    } else {
      for (int kt = kt0; kt < kt1; kt += 2) {
        step(ra0, rb0, ra1, rb1, kt, 0);
        if (kt + 1 < kt1) step(ra1, rb1, ra0, rb0, kt + 1, 1);
      }
    }
  }

  // epilogue: the accumulators go through LDS into row-major order and leave as float4 rows -- 4 (at most 16)
  // dwordx4 stores per thread instead of 16-64 dword stores; the store ISSUE rate, not bandwidth, is what the
  // direct form was bound by (6-11 k clocks per workgroup, tools/gemm_phase_lab.hip)
  GEMM_STAMP(2);
  const bool stats = g.stats != nullptr && !partial;
  const bool has_rd = !partial && g.rowdiv != nullptr;
  const int act = g.act;
  float* C = partial ? g.part + static_cast<long long>(blockIdx.z) * g.M * g.N : g.C + batch * g.sc;
  const int ldc = partial ? g.N : g.ldc;
  const bool vec_ok = (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
  double cs[4] = {0.0, 0.0, 0.0, 0.0}, css[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int pass = 0; pass < BM / CR; ++pass) {
    // (the K loop ended with a barrier: every wavefront is done with the A/B tiles)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int rbase = wm * TM + i * 32 - pass * CR;  // this fragment's first row within the pass
      if (rbase < 0 || rbase >= CR) continue;          // wavefront-uniform
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) Cs[rbase + (r & 3) + 8 * (r >> 2) + 4 * lk][wn * TN + j * 32 + li] = acc[i][j][r];
    }
    lds_barrier();
    GEMM_STAMP(4);
#pragma unroll
    for (int it = 0; it < CR / RPI; ++it) {
      const int lrow = it * RPI + rsub;
      const int row = m0 + pass * CR + lrow;
      if (row < g.M && gcol < g.N) {
        const float4 t = *reinterpret_cast<const float4*>(&Cs[lrow][c4]);
        float v[4] = {t.x, t.y, t.z, t.w};
        if (has_rd) {  // block-uniform
          const float rd = rdv[pass * (CR / RPI) + it];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / rd;
        }
        if (!partial) {
          // bias + activation without branches: act 1 -> max(v, 0), act 2 -> v > 0 ? v : 0.1 v (apply_act's results)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = v[e] + bv[e];
            const float neg = act == 2 ? 0.1f * u : 0.f;
            v[e] = (act != 0 && !(u > 0.f)) ? neg : u;
          }
        }
        float* dst = C + static_cast<long long>(row) * ldc + gcol;
        if (vec_ok && gcol + 3 < g.N) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gcol + e < g.N) dst[e] = v[e];
        }
        if (stats) {  // block-uniform.  Columns past N hold act(0 + 0) = 0 (zero pad columns of B, no bias): no test needed
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cs[e] += v[e];
            css[e] += static_cast<double>(v[e]) * v[e];
          }
        }
      }
    }
    GEMM_STAMP(5);
    lds_barrier();  // Cs is rewritten by the next pass / the statistics exchange (LDS only: the stores stay in flight)
  }
  if (stats) {  // GroupNorm statistics of this block's rows: RPI row groups per column, fixed combination order
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      stat_red[rsub * BN + c4 + e][0][0] = cs[e];
      stat_red[rsub * BN + c4 + e][0][1] = css[e];
    }
    lds_barrier();
    for (int cl = tid; cl < BN; cl += 256) {
      const int col = n0 + cl;
      if (col >= g.N) continue;
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < RPI; ++w) {
        a += stat_red[w * BN + cl][0][0];
        b += stat_red[w * BN + cl][0][1];
      }
      g.stats[(static_cast<long long>(tile_y) * 2 + 0) * g.N + col] = a;
      g.stats[(static_cast<long long>(tile_y) * 2 + 1) * g.N + col] = b;
    }
  }
  GEMM_STAMP(3);
}
template <int BM, int BN, int WM, int WN, int BK, bool TRANS_B, int PF, bool CAT = false>
// (64-row / 128 x 32 tiles: four workgroups per CU = four wavefronts per SIMD, 128 registers each)
__global__ __launch_bounds__(256, (BM * BN <= 64 * 64 ? 4 : 1)) void gemm_kernel(GemmArgs g) { gemm_kernel_body<BM, BN, WM, WN, BK, TRANS_B, PF, CAT>(blockIdx, gridDim, g); }


// ---------------------------------------------------------------------------------------------------------------------
// The WIDE form (round 6): 128 x 128 x 32 tiles for the products with a weight matrix as B ([K, N] row-major), 2 x 2
// wavefronts of 64 x 64 each -- FOUR 32x32 accumulators per wavefront -- and two workgroups per CU.  What it changes against
// gemm_kernel<64, 64> (docs/EXPERIMENTS.md 5g: matrix pipe 65 % busy where the vendor library's kernel has 93 %):
//   * operand bytes per flop through the CU's L2 port and through LDS are halved (a 64 x 64 tile at the fp32 MFMA rate asks
//     for 16 B/clk per CU from L2, about what a CU gets);
//   * the LDS images are stored in FRAGMENT order -- plane (k-group g of 8 k, k parity lk) x row x 4 k-steps -- so that the
//     operands of FOUR k-steps of a 32-row fragment are ONE ds_read_b128 (16 per k-tile and wavefront instead of 64
//     ds_read_b32 for the same 64 x 64 of output), written as ds_write_b64 (A: a row's float4 of k holds two k-steps of either
//     parity) and ds_write_b128 (B: a thread loads the four rows k = 8 g + lk + 2 j of its four columns and stores one
//     column's four k-steps per instruction); XOR swizzles (A: row ^ 2 g, B: n ^ ((n >> 3) & 3)) keep reads and writes
//     conflict-free in their lane groups (MI355X_MICROARCH.md, LDS);
//   * ONE workgroup barrier per k-tile (64 MFMAs per wavefront), placed before the tile's last k-group: the next tile's first
//     operands are read behind it, under that group's MFMAs.
// Same arithmetic as the 64 x 64 tile on every output element: v_mfma_f32_32x32x2_f32 over the k pairs (2 s, 2 s + 1) in
// ascending s, the same split-K ranges (multiples of 32), the same epilogue -- and GroupNorm column partials per 64-ROW block
// in the 64 x 64 tile's combination order (16 row classes mod 16, fp64), so the products that move to this form keep their bits.
template <bool CAT, bool MI16 = false>
__device__ __forceinline__ void gemm_wide_body(const dim3 blockIdx, const dim3 gridDim, GemmArgs g) {
  (void)gridDim;
  constexpr int BM = 128, BN = 128, BK = 32;
  constexpr int IMG = 8 * BM * 4;  // floats of one operand image: 8 planes x 128 rows x 4 k-steps (BM == BN)
  constexpr int CR = 64, LDC_S = BN + 4, TPR = BN / 4, RPI = 256 / TPR;  // epilogue: 64 rows per pass, 32 threads per row, 8 rows per iteration
  static_assert(BM == BN && RPI == 8 && CR * LDC_S * 4 <= 4 * IMG * 4 && 16 * BN * 2 * 8 <= 4 * IMG * 4, "wide tile layout");
  __shared__ __attribute__((aligned(16))) char smem[4 * IMG * 4];  // 64 KB: [A0 | A1 | B0 | B1], then the epilogue's staging
  float* As = reinterpret_cast<float*>(smem);
  float* Bs = As + 2 * IMG;
  float (*Cs)[LDC_S] = reinterpret_cast<float (*)[LDC_S]>(smem);
  double (*stat_red)[BN][2] = reinterpret_cast<double (*)[BN][2]>(smem);  // [16 row classes][BN][sum, sum of squares]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // MI16: the same products on v_mfma_f32_16x16x4_f32 -- like 32x32x2 an fp32 fma chain over ascending k (tools/mfma_order_probe.hip: the
  // two instructions and the fmaf chain agree bit for bit), with a quarter of the accumulator registers moved per flop
  constexpr int NF = MI16 ? 4 : 2;    // fragments per wavefront and operand (16 or 32 rows each)
  constexpr int FR = MI16 ? 16 : 32;  // rows of a fragment
  constexpr int KG = MI16 ? 16 : 8;   // k per group of four k-steps (= one ds_read_b128 per fragment)
  constexpr int NG = BK / KG;         // k-groups per tile
  constexpr int KL = MI16 ? 4 : 2;    // lane groups along k of one MFMA
  const int lk = MI16 ? lane >> 4 : lane >> 5, li = MI16 ? lane & 15 : lane & 31;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
  const float* A = g.A + (CAT ? 0 : batch * g.sa);
  const float* B = g.B + batch * g.sb;
  const int ktiles = (g.K + BK - 1) / BK;
  const int per = (ktiles + g.splits - 1) / g.splits;
  const int kt0 = split * per, kt1 = min(ktiles, kt0 + per);
  GEMM_WIDE_STAMP(0);

  // ---- staging: a thread moves 4 float4 of A (rows a_row + 32 i, k = 4 a_q ..) and 4 float4 of B (rows 8 b_g + b_lk + 2 j, columns b_n4 ..)
  const int a_q = tid & 7, a_row = tid >> 3;
  const int b_p = tid >> 5, b_n4 = (tid & 31) * 4;
  const int b_k = MI16 ? 16 * (b_p >> 2) + (b_p & 3) : 8 * (b_p >> 1) + (b_p & 1);  // first of the thread's four rows (stride KL)
  float4 ra[4], rb[4];
  const float* a_ptr[4];
  long long cat_row[CAT ? 4 : 1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + a_row + 32 * i, g.M - 1);
    if constexpr (CAT) {
      const long long id = g.aidx[static_cast<long long>(row) * g.ldi];
      cat_row[i] = (id >= 0 && id < g.n_coarse) ? id : -1;
      a_ptr[i] = A + max(cat_row[i], 0ll) * g.lda;
    } else {
      a_ptr[i] = A + static_cast<long long>(row) * g.lda;
    }
  }
  const float* a2_ptr[CAT ? 4 : 1];
  if constexpr (CAT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a2_ptr[i] = g.A2 + static_cast<long long>(min(m0 + a_row + 32 * i, g.M - 1)) * g.lda2;
  }
  const float* b_ptr = B + min(n0 + b_n4, g.ldb - 4);
  const bool b_col_ok = n0 + b_n4 < g.ldb;
  auto masked = [](const float4& v, bool ok) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); };
  auto load_a = [&](int kt, int i) {
    const int k0 = min(kt, kt1 - 1) * BK;  // past the end: the last tile again (stored to a buffer nobody multiplies)
    if constexpr (CAT) {  // (a k-tile lies on one side of c1: workgroup-uniform)
      ra[i] = *reinterpret_cast<const float4*>(k0 < g.c1 ? a_ptr[i] + (k0 + 4 * a_q) : a2_ptr[i] + min(k0 - g.c1 + 4 * a_q, g.K - g.c1 - 4));
    } else {
      ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + min(k0 + 4 * a_q, g.K - 4));
    }
  };
  auto load_b = [&](int kt, int j) {
    const int k0 = min(kt, kt1 - 1) * BK;
    rb[j] = *reinterpret_cast<const float4*>(b_ptr + static_cast<long long>(min(k0 + b_k + KL * j, g.K - 1)) * g.ldb);
  };
  // A piece i: the float4 (k = 4 q .. 4 q + 3 of one row) is k-steps j, j + 1 of parity 0 (x, z) and of parity 1 (y, w) of k-group q / 2
  auto store_a = [&](int buf, int kt, int i) {
    const int k0 = min(kt, kt1 - 1) * BK;
    const int row = a_row + 32 * i;
    bool ok = m0 + row < g.M && k0 + 4 * a_q < g.K;
    if constexpr (CAT) ok = ok && (k0 >= g.c1 || cat_row[i] >= 0);
    const float4 v = masked(ra[i], ok);
    if constexpr (MI16) {  // k = 4 q + e: k-step q % 4 of lane group e of k-group q / 4 -> four planes, one float each
      float* p = As + buf * IMG + (((a_q >> 2) * 4 * BM + row) * 4 + (a_q & 3));
      p[0] = v.x; p[BM * 4] = v.y; p[2 * BM * 4] = v.z; p[3 * BM * 4] = v.w;
    } else {
      const int g2 = a_q & 6;  // 2 x the k-group
      float* p = As + buf * IMG + ((g2 * BM + (row ^ g2)) * 4 + (a_q & 1) * 2);
      *reinterpret_cast<float2*>(p) = make_float2(v.x, v.z);
      *reinterpret_cast<float2*>(p + BM * 4) = make_float2(v.y, v.w);
    }
  };
  // B piece c: column b_n4 + c of the thread's four rows = that column's four k-steps of plane b_p
  auto store_b = [&](int buf, int kt, int c) {
    const int k0 = min(kt, kt1 - 1) * BK;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4& r = rb[j];
      const float e = c == 0 ? r.x : (c == 1 ? r.y : (c == 2 ? r.z : r.w));
      v[j] = (b_col_ok && k0 + b_k + KL * j < g.K) ? e : 0.f;
    }
    const int n = b_n4 + c;
    *reinterpret_cast<float4*>(Bs + buf * IMG + (b_p * BN + (n ^ ((n >> 3) & 3))) * 4) = make_float4(v[0], v[1], v[2], v[3]);
  };

  const bool partial = g.splits > 1;
  const int c4 = (tid % TPR) * 4, rsub = tid / TPR;
  const int gcol = n0 + c4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (!partial && g.bias)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = gcol + e < g.N ? g.bias[gcol + e] : 0.f;

  using Acc = typename std::conditional<MI16, f32x4_t, f32x16>::type;
  constexpr int AR = MI16 ? 4 : 16;
  Acc acc[NF][NF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;

  // ---- fragment reads: k-group gq of buffer buf -> four k-steps of the two A and the two B fragments of this wavefront
  int a_frag[NF], b_frag[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    a_frag[i] = wm * 64 + i * FR + li;
    const int n = wn * 64 + i * FR + li;
    b_frag[i] = n ^ ((n >> 3) & 3);
  }
  float af[2][NF][4], bf[2][NF][4];  // [register set][fragment][k-step]
  auto read_group = [&](int set, int buf, int gq) {
    const float* ab = As + buf * IMG + (KL * gq + lk) * BM * 4;
    const float* bb = Bs + buf * IMG + (KL * gq + lk) * BN * 4;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const float4 t = *reinterpret_cast<const float4*>(ab + ((MI16 ? a_frag[i] : a_frag[i] ^ (2 * gq)) * 4));
      af[set][i][0] = t.x; af[set][i][1] = t.y; af[set][i][2] = t.z; af[set][i][3] = t.w;
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(bb + b_frag[j] * 4);
      bf[set][j][0] = t.x; bf[set][j][1] = t.y; bf[set][j][2] = t.z; bf[set][j][3] = t.w;
    }
  };
  // One k-group = 16 MFMAs (4 k-steps x 2 x 2 fragments; every accumulator takes its k-steps in ascending order), written as
  // chunks of one MFMA + one piece of the tile's other work, fenced by scheduling barriers (a wavefront issues in order:
  // everything else has to stand BETWEEN the MFMAs to run in their shadows -- gemm_kernel_body).
  auto group = [&](int set, auto&& piece) {
#pragma unroll
    for (int c = 0; c < 4 * NF * NF; ++c) {
      const int t = c / (NF * NF), i = (c / NF) % NF, j = c % NF;
      if constexpr (MI16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[set][i][t], bf[set][j][t], acc[i][j], 0, 0, 0);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][t], bf[set][j][t], acc[i][j], 0, 0, 0);
      piece(c);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#ifndef GW_ABLATE
#define GW_ABLATE 0  // tools/gemm_wide_lab.hip: 1 = no barrier in the loop, 2 = no global loads, 4 = no LDS stores (timing probes, wrong results)
#endif
  auto tile = [&](int kt, int buf) {
    GEMM_WIDE_STAMP(8 + (kt - kt0));
    if constexpr (MI16) {  // two k-groups of 64 MFMAs (32 clocks each): stores and the other group's operands under the first, barrier, loads under the second
      group(0, [&](int c) {
        if (c == 0) read_group(1, buf, 1);
        if (!(GW_ABLATE & 4)) {
          if (c >= 2 && c <= 16 && (c & 3) == 2) store_a(buf ^ 1, kt + 1, (c - 2) >> 2);
          if (c >= 20 && c <= 32 && (c & 3) == 0) store_b(buf ^ 1, kt + 1, (c - 20) >> 2);
        }
      });
      GEMM_WIDE_STAMP(136 + (kt - kt0));
      if (!(GW_ABLATE & 1)) lds_barrier();
      GEMM_WIDE_STAMP(264 + (kt - kt0));
      __builtin_amdgcn_sched_barrier(0);
      group(1, [&](int c) {
        if (c == 0) read_group(0, buf ^ 1, 0);
        if (!(GW_ABLATE & 2)) {
          if (c >= 2 && c <= 16 && (c & 3) == 2) load_a(kt + 2, (c - 2) >> 2);
          if (c >= 20 && c <= 32 && (c & 3) == 0) load_b(kt + 2, (c - 20) >> 2);
        }
      });
      return;
    }
    // group 0: operands in set 0; group 1's arrive; tile kt + 1 goes from the register stage into the other buffer
    group(0, [&](int c) {
      if (c == 0) read_group(1, buf, 1);
      if (!(GW_ABLATE & 4)) {
        if (c >= 1 && c <= 4) store_a(buf ^ 1, kt + 1, c - 1);
        if (c >= 5 && c <= 8) store_b(buf ^ 1, kt + 1, c - 5);
      }
    });
    // group 1: group 2's operands arrive; tile kt + 2 is requested from global memory
    group(1, [&](int c) {
      if (c == 0) read_group(0, buf, 2);
      if (!(GW_ABLATE & 2)) {
        if (c >= 1 && c <= 4) load_a(kt + 2, c - 1);
        if (c >= 5 && c <= 8) load_b(kt + 2, c - 5);
      }
    });
    group(0, [&](int c) {
      if (c == 0) read_group(1, buf, 3);
    });
    GEMM_WIDE_STAMP(136 + (kt - kt0));
    if (!(GW_ABLATE & 1)) lds_barrier();  // every wavefront has read this tile (its last operands are in registers) and written the next one
    GEMM_WIDE_STAMP(264 + (kt - kt0));
    __builtin_amdgcn_sched_barrier(0);
    group(1, [&](int c) {
      if (c == 0) read_group(0, buf ^ 1, 0);
    });
  };
  if (kt0 < kt1) {  // (an empty K range -- more splits than k-tiles -- leaves the accumulators at zero)
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(kt0, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b(kt0, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_a(0, kt0, i);
#pragma unroll
    for (int c = 0; c < 4; ++c) store_b(0, kt0, c);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(kt0 + 1, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b(kt0 + 1, j);
  }
  lds_barrier();
  GEMM_WIDE_STAMP(1);
  if (kt0 < kt1) {
    read_group(0, 0, 0);
    for (int kt = kt0; kt < kt1; ++kt) tile(kt, (kt - kt0) & 1);
  }
  lds_barrier();  // (the last tile's look-ahead reads are done: the staging tile may overwrite the images)
  GEMM_WIDE_STAMP(2);

  // ---- epilogue: two passes of 64 rows through LDS into row-major float4 stores; GroupNorm partials per 64-row block
  const bool stats = g.stats != nullptr && !partial;
  const bool has_rd = !partial && g.rowdiv != nullptr;
  const int act = g.act;
  float* C = partial ? g.part + static_cast<long long>(blockIdx.z) * g.M * g.N : g.C + batch * g.sc;
  const int ldc = partial ? g.N : g.ldc;
  const bool vec_ok = (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (m0 + pass * CR >= g.M) break;  // (workgroup-uniform: a 64-row block past the last row -- the 64 x 64 grid has no such block)
    float rdv[CR / RPI];
#pragma unroll
    for (int it = 0; it < CR / RPI; ++it) rdv[it] = has_rd ? g.rowdiv[min(m0 + pass * CR + it * RPI + rsub, g.M - 1)] : 1.f;
    if (wm == pass) {  // wavefront-uniform: the two wavefront rows own one pass each
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
          for (int r = 0; r < AR; ++r) {
            if constexpr (MI16) Cs[i * 16 + 4 * lk + r][wn * 64 + j * 16 + li] = acc[i][j][r];
            else Cs[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk][wn * 64 + j * 32 + li] = acc[i][j][r];
          }
    }
    lds_barrier();
    // row class of the 64 x 64 tile's epilogue: its thread rsub16 = row % 16 sums the rows rsub16, rsub16 + 16, .. in that order
    double cs[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, css[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int it = 0; it < CR / RPI; ++it) {
      const int lrow = it * RPI + rsub;
      const int row = m0 + pass * CR + lrow;
      if (row < g.M && gcol < g.N) {
        const float4 t = *reinterpret_cast<const float4*>(&Cs[lrow][c4]);
        float v[4] = {t.x, t.y, t.z, t.w};
        if (has_rd) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / rdv[it];
        }
        if (!partial) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = v[e] + bv[e];
            const float neg = act == 2 ? 0.1f * u : 0.f;
            v[e] = (act != 0 && !(u > 0.f)) ? neg : u;
          }
        }
        float* dst = C + static_cast<long long>(row) * ldc + gcol;
        if (vec_ok && gcol + 3 < g.N) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gcol + e < g.N) dst[e] = v[e];
        }
        if (stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cs[it & 1][e] += v[e];
            css[it & 1][e] += static_cast<double>(v[e]) * v[e];
          }
        }
      }
    }
    lds_barrier();  // Cs is rewritten by the statistics exchange / the next pass
    if (stats) {
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          stat_red[par * RPI + rsub][c4 + e][0] = cs[par][e];
          stat_red[par * RPI + rsub][c4 + e][1] = css[par][e];
        }
      lds_barrier();
      if (tid < BN && n0 + tid < g.N) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
          a += stat_red[w][tid][0];
          b += stat_red[w][tid][1];
        }
        const long long blk = static_cast<long long>(blockIdx.y) * 2 + pass;
        g.stats[(blk * 2 + 0) * g.N + n0 + tid] = a;
        g.stats[(blk * 2 + 1) * g.N + n0 + tid] = b;
      }
      lds_barrier();
    }
  }
  GEMM_WIDE_STAMP(3);
}
template <bool CAT, bool MI16 = false>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(GemmArgs g) { gemm_wide_body<CAT, MI16>(blockIdx, gridDim, g); }

// Latency-oriented kernel for the transformer-sized products (M up to ~1k rows, K a multiple of 16):
// one workgroup = ONE 32 x 32 output tile, its four wavefronts split K four ways, operands go straight
// from global memory to the MFMA registers (each lane reads 16 contiguous k of its A row; B rows are
// 128-B coalesced), and the four partial tiles are added through LDS in a fixed order.  The dependent
// MFMA chain per wavefront is K/8 instructions instead of K/2, which is what bounds a 350 x 128 x 128
// projection, not bandwidth.
__device__ __forceinline__ void gemm_small_body(const dim3 blockIdx, const GemmArgs& g, float (*red)[32][33]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int kq = g.K / 4;                 // this wavefront's K range (multiple of 4)
  const int kbeg = wave * kq;
  const int row = min(m0 + li, g.M - 1);
  const int col = n0 + li;
  const bool col_ok = col < g.ldb;        // pad columns of B are zero
  const float* arow = g.A + static_cast<long long>(row) * g.lda;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // chunks of 32 k: lane half lk covers k = kbeg + c + lk*16 + t, t = 0..15
  for (int c = 0; c < kq; c += 32) {
    float a[16], b[16];
    const int kb = kbeg + c + lk * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c + lk * 16 + 4 * q < kq) v = *reinterpret_cast<const float4*>(arow + kb + 4 * q);
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
      b[t] = (col_ok && c + lk * 16 + t < kq) ? g.B[static_cast<long long>(kb + t) * g.ldb + col] : 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lk][li] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, rr = e >> 5, cc = e & 31;
    const int orow = m0 + rr, ocol = n0 + cc;
    if (orow < g.M && ocol < g.N) {
      float v = ((red[0][rr][cc] + red[1][rr][cc]) + red[2][rr][cc]) + red[3][rr][cc];
      if (g.bias) v += g.bias[ocol];
      g.C[static_cast<long long>(orow) * g.ldc + ocol] = apply_act(v, g.act);
    }
  }
}

__device__ __forceinline__ void gemm_small_entry(const dim3 blockIdx, const dim3 gridDim, GemmArgs g) {
  (void)gridDim;
  __shared__ float red[4][32][33];
  gemm_small_body(blockIdx, g, red);
}
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g) { gemm_small_entry(blockIdx, gridDim, g); }
// two independent products in one launch (blockIdx.z): the q and the k|v projection of a cross-attention layer
__device__ __forceinline__ void gemm_small_pair_entry(const dim3 blockIdx, const dim3 gridDim, GemmArgs g0, GemmArgs g1) {
  (void)gridDim;
  __shared__ float red[4][32][33];
  const GemmArgs& g = blockIdx.z ? g1 : g0;
  if (static_cast<int>(blockIdx.y) * 32 >= g.M || static_cast<int>(blockIdx.x) * 32 >= g.N) return;  // whole workgroup
  gemm_small_body(blockIdx, g, red);
}
__global__ __launch_bounds__(256) void gemm_small_pair_kernel(GemmArgs g0, GemmArgs g1) { gemm_small_pair_entry(blockIdx, gridDim, g0, g1); }

// y = act(LayerNorm(x W + bias + residual)) for the transformer width (N = 128) in ONE launch: a workgroup owns
// 16 complete rows (wavefront w the columns 32w..32w+31 over the whole K on the 16x16x4 MFMA, operands straight from
// global memory with a permuted contraction: lane group kb takes k = 16*step + 4*kb + t; W in its checkpoint layout
// [128, K]), so the row statistics are a
// 16-lane shuffle + one LDS exchange away.  Replaces a gemm_small + layernorm launch pair (48 per scan pair).
struct LinLnArgs {
  const float *A, *B, *bias, *res, *gamma, *beta;
  float* out;
  int M, K, lda, ldb, ldr, ldo, act;
  float eps;
};
__device__ __forceinline__ void linear_ln128_body(const dim3 blockIdx, const dim3 gridDim, LinLnArgs a) {
  (void)gridDim;
  __shared__ float red[2][4][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kb = lane >> 4;
  const int m0 = blockIdx.x * 16;
  // contraction order: step s, lane group kb takes k = 16 s + 4 kb + t -- the four lane groups of a row read one
  // contiguous 64-byte line per step (a lane-group-major split of K touches 64 different lines per load instead of 16)
  const float* arow = a.A + static_cast<long long>(min(m0 + i, a.M - 1)) * a.lda + 4 * kb;
  // W is the nn.Linear weight as stored, [128, K] with k contiguous: the B operand of lane (column, kb) is a
  // float4 of 4 consecutive k, exactly like the A operand
  const float* brow0 = a.B + static_cast<long long>(32 * w + i) * a.ldb + 4 * kb;
  const float* brow1 = brow0 + 16ll * a.ldb;
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < a.K; s0 += 16) {
    const float4 av = *reinterpret_cast<const float4*>(arow + s0);
    const float4 bv0 = *reinterpret_cast<const float4*>(brow0 + s0);
    const float4 bv1 = *reinterpret_cast<const float4*>(brow1 + s0);
    const float b0[4] = {bv0.x, bv0.y, bv0.z, bv0.w}, b1[4] = {bv1.x, bv1.y, bv1.z, bv1.w};
    const float at[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], b0[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], b1[t], acc1, 0, 0, 0);
    }
  }
  // lane (i, g = kb) holds rows 4g + r of columns 32w + i and 32w + 16 + i
  const int g = kb, c0 = 32 * w + i, c1 = c0 + 16;
  const float bias0 = a.bias ? a.bias[c0] : 0.f, bias1 = a.bias ? a.bias[c1] : 0.f;
  float v0[4], v1[4], ps[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(m0 + 4 * g + r, a.M - 1);
    v0[r] = acc0[r] + bias0;
    v1[r] = acc1[r] + bias1;
    if (a.res) {
      v0[r] += a.res[static_cast<long long>(row) * a.ldr + c0];
      v1[r] += a.res[static_cast<long long>(row) * a.ldr + c1];
    }
    ps[r] = v0[r] + v1[r];
  }
  auto rows_reduce = [&](float (&p)[4], int which) -> void {  // p[r] -> sum over the 128 columns of row 4g + r
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) p[r] += __shfl_xor(p[r], o, 64);
    if (i == 0)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[which][w][4 * g + r] = p[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
      p[r] = ((red[which][0][4 * g + r] + red[which][1][4 * g + r]) + red[which][2][4 * g + r]) + red[which][3][4 * g + r];
  };
  rows_reduce(ps, 0);
  float mean[4], sq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mean[r] = ps[r] / 128.f;
    const float d0 = v0[r] - mean[r], d1 = v1[r] - mean[r];
    sq[r] = d0 * d0 + d1 * d1;
  }
  rows_reduce(sq, 1);
  const float g0 = a.gamma[c0], g1 = a.gamma[c1], be0 = a.beta[c0], be1 = a.beta[c1];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * g + r;
    if (row >= a.M) continue;
    const float rstd = 1.0f / __fsqrt_rn(sq[r] / 128.f + a.eps);
    float o0 = (v0[r] - mean[r]) * rstd * g0 + be0, o1 = (v1[r] - mean[r]) * rstd * g1 + be1;
    if (a.act == 1) {
      o0 = o0 > 0.f ? o0 : 0.f;
      o1 = o1 > 0.f ? o1 : 0.f;
    }
    a.out[static_cast<long long>(row) * a.ldo + c0] = o0;
    a.out[static_cast<long long>(row) * a.ldo + c1] = o1;
  }
}
__global__ __launch_bounds__(256) void linear_ln128_kernel(LinLnArgs a) { linear_ln128_body(blockIdx, gridDim, a); }

// Everything of an attention layer after softmax(QK^T)V in ONE launch (thdroformer.py:142-173,
// vanilla_transformer.py:69-103, output_layer.py:6-21), at the transformer width 128 with a 256-wide FFN:
//   y   = LayerNorm(hid Wo^T + bo + x)
//   z   = relu(y W1^T + b1)
//   out = LayerNorm(z W2^T + b2 + y)
// All three products are row-local, so a workgroup owns 16 complete rows through the whole chain: 8 wavefronts, each a
// 16-column slice of the 128-wide products (32 of the 256-wide one) on the 16x16x4 MFMA with the permuted contraction
// of linear_ln128_kernel.  Every weight operand a lane needs (48 float4) is requested before the first MFMA -- one
// L2 latency for the kernel instead of one per product; what bounds the kernel is the ~20 B/clk a CU gets from L2 for
// its 320 KB of weights, so the barriers order LDS only (lds_barrier) and the later products' weights keep streaming
// under the earlier products -- and y, z travel through LDS.  Replaces three launches.
struct TailArgs {
  const float *hid, *x, *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
  float* out;
  int M, ldh, ldx, ldo, ldwo, ldw1, ldw2;
  float eps;
  // round 6: the three weight matrices in OPERAND order (rdm_attention_tail_pack_weights): float4 [(wavefront, step), lane] so that
  // every weight load instruction of a wavefront reads ONE contiguous KB (8 full lines) instead of 16 rows x 64 B; same values in
  // the same lanes: same bits.  wo at 0, w1 at 4096, w2 at 12288 (float4 units); null = the checkpoint layout (wo / w1 / w2).
  const float4* packed;
#ifdef RDM_TAIL_TIMING
  unsigned long long* clk;  // tools/tail_lab.hip: shader-clock stamps of workgroup 0, wavefront 0
#endif
};
#ifdef RDM_TAIL_TIMING
#define TAIL_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) a.clk[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TAIL_STAMP(k) do { } while (0)
#endif
template <bool PACKED>
__device__ __forceinline__ void attention_tail128_body(const dim3 blockIdx, const dim3 gridDim, TailArgs a) {
  (void)gridDim;
  __shared__ __attribute__((aligned(16))) float ys[16][132];
  __shared__ __attribute__((aligned(16))) float zs[16][260];
  __shared__ float red[4][8][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kb = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int c = 16 * w + i;  // this lane's column of the 128-wide products
  TAIL_STAMP(0);
  // ---- all global operands up front -------------------------------------------------------------------------------
  // (contraction order of every product: step s, lane group kb takes k = 16 s + 4 kb + t, so the four lane groups
  // of a weight row read one contiguous 64-byte line per step)
  // A CU draws ~20 B/clk from L2 whatever the pattern (tools/tail_lab.hip: 393 KB in 20 k clocks, one workgroup or 44),
  // and a wavefront stalls at a load the texture path cannot accept yet: the weight loads are therefore ISSUED in
  // slices between the MFMA steps of the previous product (sched_barrier keeps the compiler from regrouping them).
  float4 bw0[8], bw1[2][8], bw2[16];
  // small operands first: loads return in order, and the first LayerNorm needs these
  float xres[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) xres[r] = a.x[static_cast<long long>(min(m0 + 4 * kb + r, a.M - 1)) * a.ldx + c];
  const float bo = a.bo ? a.bo[c] : 0.f, b2 = a.b2 ? a.b2[c] : 0.f;
  const float b10 = a.b1 ? a.b1[32 * w + i] : 0.f, b11 = a.b1 ? a.b1[32 * w + 16 + i] : 0.f;
  const float g1 = a.g1[c], be1 = a.be1[c], g2 = a.g2[c], be2 = a.be2[c];
  const float* p1a = a.w1 + static_cast<long long>(32 * w + i) * a.ldw1 + 4 * kb;
  const float* p1b = p1a + 16ll * a.ldw1;
  const float* p2 = a.w2 + static_cast<long long>(c) * a.ldw2 + 4 * kb;
  {
    // the 16 x 128 attention rows, shared by all wavefronts: one coalesced float4 per thread into LDS (zs is free until
    // the FFN product)
    const int hr = tid >> 5, hc = (tid & 31) * 4;
    const float4 hv = *reinterpret_cast<const float4*>(a.hid + static_cast<long long>(min(m0 + hr, a.M - 1)) * a.ldh + hc);
    const float* p0 = a.wo + static_cast<long long>(c) * a.ldwo + 4 * kb;
#pragma unroll
    for (int s = 0; s < 8; ++s) bw0[s] = PACKED ? a.packed[(w * 8 + s) * 64 + lane] : *reinterpret_cast<const float4*>(p0 + 16 * s);
    *reinterpret_cast<float4*>(&zs[hr][hc]) = hv;
  }
  __builtin_amdgcn_sched_barrier(0);

  // rows 4kb + r of this lane's column: LayerNorm over the 128 columns (16 lanes x 8 wavefronts), two passes
  auto layer_norm = [&](float (&v)[4], int slot, float gam, float bet) {
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = v[r];
    float mean[4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = row16_sum(p[r]);
      if (i == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[slot + pass][w][4 * kb + r] = p[r];
      lds_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) t += red[slot + pass][ww][4 * kb + r];  // fixed order
        if (pass == 0) {
          mean[r] = t / 128.f;
          const float d = v[r] - mean[r];
          p[r] = d * d;
        } else {
          const float rstd = 1.0f / __fsqrt_rn(t / 128.f + a.eps);
          v[r] = (v[r] - mean[r]) * rstd * gam + bet;
        }
      }
    }
  };

  // ---- y = LayerNorm(hid Wo^T + bo + x) ---------------------------------------------------------------------------
  lds_barrier();
  TAIL_STAMP(1);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    bw1[0][s] = PACKED ? a.packed[4096 + ((w * 2 + 0) * 8 + s) * 64 + lane] : *reinterpret_cast<const float4*>(p1a + 16 * s);
    bw1[1][s] = PACKED ? a.packed[4096 + ((w * 2 + 1) * 8 + s) * 64 + lane] : *reinterpret_cast<const float4*>(p1b + 16 * s);
    const float4 ha = *reinterpret_cast<const float4*>(&zs[i][16 * s + 4 * kb]);
    const float at[4] = {ha.x, ha.y, ha.z, ha.w}, bt[4] = {bw0[s].x, bw0[s].y, bw0[s].z, bw0[s].w};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], bt[t], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) bw2[s] = PACKED ? a.packed[12288 + (w * 16 + s) * 64 + lane] : *reinterpret_cast<const float4*>(p2 + 16 * s);
  __builtin_amdgcn_sched_barrier(0);
  float y[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = acc[r] + bo + xres[r];
  TAIL_STAMP(2);
  layer_norm(y, 0, g1, be1);
  TAIL_STAMP(3);
#pragma unroll
  for (int r = 0; r < 4; ++r) ys[4 * kb + r][c] = y[r];
  lds_barrier();

  // ---- z = relu(y W1^T + b1): columns 32w + 16t + i ---------------------------------------------------------------
  TAIL_STAMP(4);
  f32x4_t z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    bw2[8 + s] = PACKED ? a.packed[12288 + (w * 16 + 8 + s) * 64 + lane] : *reinterpret_cast<const float4*>(p2 + 16 * (8 + s));
    const float4 ya = *reinterpret_cast<const float4*>(&ys[i][16 * s + 4 * kb]);
    const float at[4] = {ya.x, ya.y, ya.z, ya.w};
    const float u0[4] = {bw1[0][s].x, bw1[0][s].y, bw1[0][s].z, bw1[0][s].w};
    const float u1[4] = {bw1[1][s].x, bw1[1][s].y, bw1[1][s].z, bw1[1][s].w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], u0[t], z0, 0, 0, 0);
      z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], u1[t], z1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float e0 = z0[r] + b10, e1 = z1[r] + b11;
    zs[4 * kb + r][32 * w + i] = e0 > 0.f ? e0 : 0.f;
    zs[4 * kb + r][32 * w + 16 + i] = e1 > 0.f ? e1 : 0.f;
  }
  lds_barrier();

  // ---- out = LayerNorm(z W2^T + b2 + y) ---------------------------------------------------------------------------
  TAIL_STAMP(5);
  acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float4 za = *reinterpret_cast<const float4*>(&zs[i][16 * s + 4 * kb]);
    const float at[4] = {za.x, za.y, za.z, za.w}, bt[4] = {bw2[s].x, bw2[s].y, bw2[s].z, bw2[s].w};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at[t], bt[t], acc, 0, 0, 0);
  }
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = acc[r] + b2 + y[r];
  TAIL_STAMP(6);
  layer_norm(o, 2, g2, be2);
  TAIL_STAMP(7);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * kb + r;
    if (row < a.M) a.out[static_cast<long long>(row) * a.ldo + c] = o[r];
  }
}
template <bool PACKED>
__global__ __launch_bounds__(512) void attention_tail128_kernel(TailArgs a) { attention_tail128_body<PACKED>(blockIdx, gridDim, a); }

// The tail's weights in operand order (TailArgs::packed): one thread per float4.
__global__ void attention_tail_pack_kernel(const float* wo, int ldwo, const float* w1, int ldw1, const float* w2, int ldw2, float4* out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 20480) return;
  const int lane = e & 63, i = lane & 15, kb = lane >> 4;
  const float* src;
  if (e < 4096) {
    const int w = e >> 9, s = (e >> 6) & 7;
    src = wo + static_cast<long long>(16 * w + i) * ldwo + 16 * s + 4 * kb;
  } else if (e < 12288) {
    const int t = e - 4096, w = t >> 10, h = (t >> 9) & 1, s = (t >> 6) & 7;
    src = w1 + static_cast<long long>(32 * w + 16 * h + i) * ldw1 + 16 * s + 4 * kb;
  } else {
    const int t = e - 12288, w = t >> 10, s = (t >> 6) & 15;
    src = w2 + static_cast<long long>(16 * w + i) * ldw2 + 16 * s + 4 * kb;
  }
  out[e] = *reinterpret_cast<const float4*>(src);
}

__device__ __forceinline__ void splitk_reduce_kernel_body(const dim3 blockIdx, const dim3 gridDim, GemmArgs g, int batches) {
  (void)blockIdx; (void)gridDim;
  const long long total = static_cast<long long>(batches) * g.M * g.N;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int n = static_cast<int>(i % g.N);
    const int m = static_cast<int>((i / g.N) % g.M);
    const int b = static_cast<int>(i / (static_cast<long long>(g.M) * g.N));
    const float* p = g.part + (static_cast<long long>(b) * g.splits) * g.M * g.N +
                     static_cast<long long>(m) * g.N + n;
    float v = 0.f;
    for (int s = 0; s < g.splits; ++s) v += p[static_cast<long long>(s) * g.M * g.N];  // fixed order
    if (g.rowdiv) v = v / g.rowdiv[m];
    if (g.bias) v += g.bias[n];
    g.C[b * g.sc + static_cast<long long>(m) * g.ldc + n] = apply_act(v, g.act);
  }
}
__global__ void splitk_reduce_kernel(GemmArgs g, int batches) { splitk_reduce_kernel_body(blockIdx, gridDim, g, batches); }


// Split-K reduce that also emits the GroupNorm column partials of its row block (layout of gn_partial_kernel
// in norm.hip, whose separate pass it replaces): grid = (row blocks, column chunks of 256); cw = min(N, 256)
// columns x (256 / cw) row lanes, kStatRowsPerLane rows per lane -- few rows per thread, because every output
// element costs `splits` dependent-latency loads and the matrices are small (M in the hundreds).
constexpr int kStatRowsPerLane = 8;
inline int stat_rows_per_block(long long n) { return kStatRowsPerLane * (256 / static_cast<int>(n < 256 ? n : 256)); }
__device__ __forceinline__ void splitk_reduce_stats_kernel_body(const dim3 blockIdx, const dim3 gridDim, GemmArgs g, double* stats) {
  (void)blockIdx; (void)gridDim;
  __shared__ double red[2][256];
  const int cw = g.N < 256 ? g.N : 256;
  const int lanes = 256 / cw;
  const int r0 = blockIdx.x * (kStatRowsPerLane * lanes);
  const int r1 = min(g.M, r0 + kStatRowsPerLane * lanes);
  const int col_in = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int col = blockIdx.y * 256 + col_in;
  double s = 0.0, ss = 0.0;
  if (rl < lanes && col < g.N) {
    const float bv = g.bias ? g.bias[col] : 0.f;
    const long long plane = static_cast<long long>(g.M) * g.N;
    // the kStatRowsPerLane rows of this thread are summed side by side: independent loads in flight per split
    float acc[kStatRowsPerLane];
    const float* p[kStatRowsPerLane];
#pragma unroll
    for (int u = 0; u < kStatRowsPerLane; ++u) {
      const int r = min(r0 + rl + u * lanes, g.M - 1);
      p[u] = g.part + static_cast<long long>(r) * g.N + col;
      acc[u] = 0.f;
    }
    for (int k = 0; k < g.splits; ++k) {  // fixed order per element
#pragma unroll
      for (int u = 0; u < kStatRowsPerLane; ++u) acc[u] += p[u][k * plane];
    }
#pragma unroll
    for (int u = 0; u < kStatRowsPerLane; ++u) {
      const int r = r0 + rl + u * lanes;
      if (r >= r1) break;
      float v = acc[u];
      if (g.rowdiv) v = v / g.rowdiv[r];
      v = apply_act(v + bv, g.act);
      g.C[static_cast<long long>(r) * g.ldc + col] = v;
      const double d = v;
      s += d;
      ss += d * d;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < cw && col < g.N) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < lanes; ++k) {
      a += red[0][k * cw + threadIdx.x];
      b += red[1][k * cw + threadIdx.x];
    }
    stats[(static_cast<long long>(blockIdx.x) * 2 + 0) * g.N + col] = a;
    stats[(static_cast<long long>(blockIdx.x) * 2 + 1) * g.N + col] = b;
  }
}
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(GemmArgs g, double* stats) { splitk_reduce_stats_kernel_body(blockIdx, gridDim, g, stats); }


thread_local unsigned g_lds_pad = 0;  // see rdm::gemm_set_lds_pad

template <int BM, int BN, int WM, int WN, int BK, int PF = 1>
void launch(const GemmArgs& g, int batches, bool trans_b, hipStream_t st) {
  dim3 grid(ceil_div(g.N, BN), ceil_div(g.M, BM), batches * g.splits);
  // unused dynamic LDS per workgroup = fewer GEMM workgroups per CU (rdm::gemm_set_lds_pad; lab build: RDM_GEMM_LDS_PAD=<bytes>)
  static const int pad_env = [] { const char* v = ::rdm::dev_knob("RDM_GEMM_LDS_PAD"); return v ? atoi(v) : -1; }();
  const unsigned pad = pad_env >= 0 ? static_cast<unsigned>(pad_env) : g_lds_pad;
  if constexpr (BM == 64 && BN == 64 && BK == 32 && PF == 2) {
    if (g.aidx && g.bidx) {  // gathered rows on both sides (patch scores)
      ::rdm::launch<gemm_kernel_body<BM, BN, WM, WN, BK, true, PF, true>, gemm_kernel<BM, BN, WM, WN, BK, true, PF, true>, 256, (BM * BN <= 64 * 64 ? 4 : 1)>(grid, pad, st, g);
      return;
    }
    if (g.aidx) {  // virtual [upsample | skip] A operand
      ::rdm::launch<gemm_kernel_body<BM, BN, WM, WN, BK, false, PF, true>, gemm_kernel<BM, BN, WM, WN, BK, false, PF, true>, 256, (BM * BN <= 64 * 64 ? 4 : 1)>(grid, pad, st, g);
      return;
    }
  }
  if (trans_b)
    ::rdm::launch<gemm_kernel_body<BM, BN, WM, WN, BK, true, PF>, gemm_kernel<BM, BN, WM, WN, BK, true, PF>, 256, (BM * BN <= 64 * 64 ? 4 : 1)>(grid, pad, st, g);
  else
    ::rdm::launch<gemm_kernel_body<BM, BN, WM, WN, BK, false, PF>, gemm_kernel<BM, BN, WM, WN, BK, false, PF>, 256, (BM * BN <= 64 * 64 ? 4 : 1)>(grid, pad, st, g);
}

// the wide form (gemm_wide_kernel): B = weights [K, N], one product or a batch with strides, optionally the decoder's virtual A
void launch_wide(const GemmArgs& g, int batches, hipStream_t st, bool mi16_form) {
  dim3 grid(ceil_div(g.N, 128), ceil_div(g.M, 128), batches * g.splits);
  static const bool mi16 = ::rdm::dev_knob("RDM_GEMM_WIDE_MI16") != nullptr;  // developer knob (A/B): the 16x16x4 MFMA, same bits
  if (mi16 || mi16_form) {
    if (g.aidx) ::rdm::launch<gemm_wide_body<true, true>, gemm_wide_kernel<true, true>, 256, 2>(grid, 0, st, g);
    else ::rdm::launch<gemm_wide_body<false, true>, gemm_wide_kernel<false, true>, 256, 2>(grid, 0, st, g);
    return;
  }
  if (g.aidx) ::rdm::launch<gemm_wide_body<true>, gemm_wide_kernel<true>, 256, 2>(grid, 0, st, g);
  else ::rdm::launch<gemm_wide_body<false>, gemm_wide_kernel<false>, 256, 2>(grid, 0, st, g);
}

}  // namespace

// Residency of the tiled GEMM for the calling thread's launches: `bytes` of unused dynamic LDS per workgroup.  A 64 x 64 tile
// holds 34 KB and 128 VGPRs x 4 wavefronts, so four of its workgroups fill a CU; when several scan pairs share the GPU, 20 KB
// of padding (two workgroups per CU) leaves registers and LDS for the other pairs' kernels: +3 % pairs/s at four in flight,
// -2 % with one (docs/EXPERIMENTS.md 5d).  The engine sets it from rdm_engine_set_pairs_in_flight; results do not depend on it.
void rdm::gemm_set_lds_pad(unsigned bytes) { g_lds_pad = bytes; }

extern "C" size_t rdm_gemm_workspace_bytes(int64_t m, int64_t n, int batches) {
  // split-K partials: at most 16 copies of the output, capped at 64 MB (the dispatch model never asks for more than
  // the workspace it is given).
  const size_t by_shape = static_cast<size_t>(m) * static_cast<size_t>(n) * 16;
  const size_t by_tiles = static_cast<size_t>(1024) * 128 * 128;
  return align_up(std::min(by_shape, by_tiles) * sizeof(float) * static_cast<size_t>(batches > 0 ? batches : 1));
}

namespace {

thread_local int g_last_plan[4] = {0, 0, 0, 0};  // tile rows, tile columns, k-tile depth, split-K factor of the last dispatch

// form (tests, A/B runs; rdm_gemm_form): 0 = the library's choice; 1 / 2 = the products the model gives to the 64 x 64 tile run on the
// wide form instead (128 x 128 x 32 tiles; 1: v_mfma_f32_32x32x2_f32, 2: v_mfma_f32_16x16x4_f32) -- same split factor, same bits.
int gemm_dispatch(GemmArgs g, int batches, bool trans_b, void* ws, size_t ws_bytes, int* stat_blocks, hipStream_t st, int form = 0) {
  const long long m = g.M, n = g.N, k = g.K;
  const char* tune_env = ::rdm::dev_knob("RDM_GEMM_TUNE");  // developer knob, see below; any value also bypasses the small kernel
  if (batches == 1 && !trans_b && !g.rowdiv && !g.stats && !g.aidx && m <= 1536 && k % 16 == 0 && k >= 64 && k <= 1024 &&
      m * n <= 1536 * 512 && !(tune_env && tune_env[0] != '0')) {
    if (stat_blocks) *stat_blocks = 0;
    g_last_plan[0] = 32; g_last_plan[1] = 32; g_last_plan[2] = static_cast<int>(k / 4); g_last_plan[3] = 4;  // K over four wavefronts
    RDM_DUP_LOOP("gemmsmall")
    ::rdm::launch<gemm_small_entry, gemm_small_kernel, 256>(dim3(ceil_div<long long>(n, 32), ceil_div<long long>(m, 32)), 0, st, g);
    return launch_status("gemm_small_kernel");
  }
  // Tile shape and split-K factor from a small cost model fitted to tools/gemm_sweep_graph.py (HIP-graph-replayed
  // timings of every product shape of the path): a block costs a fixed prologue + epilogue plus a time per 64 of K;
  // blocks run in rounds of (CUs x resident blocks); split-K adds the partial traffic and one more launch.
  enum Tile { T128, T64, T128x32 };
  // per 64 of K a block that shares its CU with r - 1 others needs max(lat * (1 + alpha (r - 1)), MFMA time * r);
  // constants refitted by tools/gemm_model_fit.py on the sweep in tools/data/ (782 us for the path's 36 shapes against
  // 769 us for the per-shape optimum)
  struct Cand { Tile tile; int bm, bn; double fixed_us, lat_k64_us, mfma_k64_us; int resident; double alpha; };
  static const Cand cands[3] = {{T128, 128, 128, 4.0, 2.6, 3.4, 3, 0.0}, {T64, 64, 64, 1.5, 2.2, 0.86, 4, 0.2},
                                {T128x32, 128, 32, 3.0, 1.3, 0.86, 3, 0.0}};
  static const int split_set[8] = {1, 2, 3, 4, 6, 8, 12, 16};
  // CUs the model plans for.  (RDM_GEMM_CUS, developer knob: with four pairs in flight a product effectively owns a
  // quarter of the chip, tools/exp_cumask.sh.)
  static const long long model_cus = [] {
    const char* v = ::rdm::dev_knob("RDM_GEMM_CUS");
    const long long n = v ? atoll(v) : 256;
    return n >= 8 && n <= 256 ? n : 256ll;
  }();
  Tile tile = T64;
  int best_s = 1;
  {
    double best = 1e30;
    const size_t ws_cap = ws ? ws_bytes : 0;
    for (const Cand& c : cands) {
      if (g.aidx && c.tile != T64) continue;
      if (c.tile == T128x32 && n > 64) continue;
      if (c.tile != T128x32 && n <= 32) continue;  // a 64-wide tile would be half empty
      if (c.tile == T128 && n < 128) continue;
      const long long tiles = ceil_div<long long>(m, c.bm) * ceil_div<long long>(n, c.bn) * batches;
      const int max_s = k >= 256 ? static_cast<int>(std::min<long long>(16, k / 128)) : 1;
      for (int sp : split_set) {
        if (sp > max_s) break;
        if (sp > 1 && static_cast<size_t>(m) * n * sp * batches * sizeof(float) > ws_cap) break;
        const long long blocks = tiles * sp;
        const double k_per = static_cast<double>(k) / sp / 64.0;
        const long long per_round = model_cus * c.resident;
        const long long full = blocks / per_round, rem = blocks % per_round;
        auto per_k64 = [&](double r) { return std::max(c.lat_k64_us * (1.0 + c.alpha * (r - 1.0)), c.mfma_k64_us * r); };
        double t = full * (c.fixed_us + k_per * per_k64(c.resident));
        if (rem) t += c.fixed_us + k_per * per_k64(static_cast<double>(ceil_div<long long>(rem, model_cus)));
        if (sp > 1) t += 3.0 + static_cast<double>(m) * n * sp * 8.0 / 1.5e6;  // reduce launch + partial write/read
        if (t < best) {
          best = t;
          tile = c.tile;
          best_s = sp;
        }
      }
    }
  }
  // developer knob for tuning runs (tools/gemm_sweep*.py): RDM_GEMM_TUNE="<tile 1..3>,<splits>" overrides the model
  int force_splits = 0, exp_tile = 0;
  if (const char* tune = ::rdm::dev_knob("RDM_GEMM_TUNE")) {
    int t = 0, sp = 0;
    if (sscanf(tune, "%d,%d", &t, &sp) >= 1) {
      if (t == 1) tile = T128;
      else if (t == 2) tile = T64;
      else if (t == 3) tile = T128x32;
      else if (t >= 4 && t <= 7) exp_tile = t;  // experimental tiles, tuning runs only (see the launch below)
      force_splits = sp;
      if (t >= 1 && t <= 7 && sp == 0) best_s = 1;
    }
  }
  // developer experiment: RDM_GEMM_BIG="<min M>[,<tile 4..7>]" runs the un-split products with at least that many rows and
  // n >= 128 on a larger tile (default 6 = 128x128x32) -- fewer operand bytes per flop through the CU's L2 port, which is what
  // co-limits the 64x64 tile when several pairs share the GPU
  static const int big_min_m = [] { const char* v = ::rdm::dev_knob("RDM_GEMM_BIG"); return v ? atoi(v) : 0; }();
  static const int big_tile = [] { const char* v = ::rdm::dev_knob("RDM_GEMM_BIG"); const char* c = v ? strchr(v, ',') : nullptr; return c ? atoi(c + 1) : 6; }();
  static const bool big_split = ::rdm::dev_knob("RDM_GEMM_BIG_SPLIT") != nullptr;  // (lab: the split-K products too -- same split factor, same bits)
  if (big_min_m > 0 && exp_tile == 0 && m >= big_min_m && n >= 128 && (best_s == 1 || big_split) && force_splits == 0 && batches == 1)
    exp_tile = big_tile;
  static const bool xcd_env = ::rdm::dev_knob("RDM_GEMM_XCD") != nullptr;  // developer knob (A/B)
  g.xcd_tiles = (xcd_env && ceil_div<long long>(n, 64) > 1 && ceil_div<long long>(m, 64) * ceil_div<long long>(n, 64) >= 64) ? 1 : 0;
  if (g.aidx) {  // the concatenating / gathering operands exist for the 64x64x32 tile only
    tile = T64;
    exp_tile = 0;
    if (g.bidx) best_s = 1, force_splits = 0;  // (256 batches fill the chip; the zero-tile shortcut writes C directly)
  }
  // The wide form takes the 64 x 64 tile's products (same split factor, same 64-row statistics blocks: the same bits) when the
  // output is at least `wide_min_n` columns wide and has enough rows to be worth 128-row tiles.
  static const int wide_min_m = [] { const char* v = ::rdm::dev_knob("RDM_GEMM_WIDE_MIN_M"); return v ? atoi(v) : 256; }();
  static const int wide_min_n = [] { const char* v = ::rdm::dev_knob("RDM_GEMM_WIDE_MIN_N"); return v ? atoi(v) : 128; }();
  // Measured (round 6, docs/EXPERIMENTS.md 5h): bit-identical, half the LDS instructions and no bank conflicts, 94 % of the MFMA rate
  // per k-tile with two workgroups per CU -- and no faster than the 64 x 64 tile, alone or in the lock-step schedule (639-641 against
  // 638-643 pairs/s): off unless the lab build asks for it (RDM_GEMM_WIDE=1).
  static const bool wide_knob = ::rdm::dev_knob("RDM_GEMM_WIDE") != nullptr;
  const bool wide_off = !wide_knob && form == 0;
  static const bool wide_force = ::rdm::dev_knob("RDM_GEMM_WIDE_FORCE") != nullptr;  // developer knob (probes): whatever tile the model chose
  if (wide_force && !trans_b && !g.bidx && exp_tile == 0) tile = T64;
  const bool wide = !wide_off && tile == T64 && exp_tile == 0 && !trans_b && !g.bidx && (form != 0 || (m >= wide_min_m && n >= wide_min_n));
  int bm = tile == T64 ? 64 : 128, bn = tile == T128 ? 128 : (tile == T64 ? 64 : 32);
  if (wide) { bm = 128; bn = 128; }
  if (exp_tile == 4) { bm = 128; bn = 64; }
  if (exp_tile == 5) { bm = 64; bn = 128; }
  if (exp_tile == 6) { bm = 128; bn = 128; }
  if (exp_tile == 7) { bm = 256; bn = 64; }
  {
    int sp = force_splits > 0 ? force_splits : best_s;
    const size_t need = static_cast<size_t>(m) * n * sp * batches * sizeof(float);
    if (sp > 1 && ws && ws_bytes >= need) {
      g.splits = sp;
      g.part = static_cast<float*>(ws);
    }
  }
  double* reduce_stats = nullptr;  // split-K: the statistics come from the reduce pass (64-row blocks)
  if (g.splits > 1) {
    if (batches == 1 && g.N % 4 == 0) reduce_stats = g.stats;
    g.stats = nullptr;
  }
  if (stat_blocks)  // (the wide form writes the 64 x 64 tile's 64-row blocks)
    *stat_blocks = g.stats ? static_cast<int>(ceil_div<long long>(m, wide ? 64 : bm))
                           : (reduce_stats ? static_cast<int>(ceil_div<long long>(m, stat_rows_per_block(n))) : 0);
  g_last_plan[0] = bm; g_last_plan[1] = bn; g_last_plan[3] = g.splits;
  g_last_plan[2] = wide ? 32 : (tile == T128 ? 16 : ((tile == T64 ? (k >= 48 || g.aidx) : k >= 32) ? 32 : 16));
  // k-tile depth: deep tiles for the latency-bound small configurations (a 350 x 128 x 128 projection
  // is two 64-deep steps instead of eight 16-deep ones), shallow where K itself is tiny
  RDM_DUP_LOOP("gemm") {
  if (wide) launch_wide(g, batches, st, form == 2);
  else
#ifdef RDM_DEV_KNOBS  // the experimental tiles exist in the lab build only (RDM_GEMM_TUNE=4..7, RDM_GEMM_BIG)
  if (exp_tile == 4) launch<128, 64, 2, 2, 32, 2>(g, batches, trans_b, st);
  else if (exp_tile == 5) launch<64, 128, 2, 2, 32, 2>(g, batches, trans_b, st);
  else if (exp_tile == 6) launch<128, 128, 2, 2, 32, 2>(g, batches, trans_b, st);
  else if (exp_tile == 7) launch<256, 64, 4, 1, 32, 2>(g, batches, trans_b, st);
  else
#endif
  switch (tile) {
    case T128: launch<128, 128, 2, 2, 16>(g, batches, trans_b, st); break;  // 32-deep measured slower (LDS halves residency)
    case T64:
      // 32-deep k-tiles: 34 KB of LDS per block -> 4 blocks per CU (64-deep: 2); measured +2 % with 4 pairs in flight
      // (two k-tiles of register prefetch: -3 % over the path's shapes; four: no further gain -- what bounds these tiles
      // is the ~20 B/clk a CU draws from L2, see tools/tail_lab.hip, not the latency of one load)
      // (gathered / concatenated operands exist in the 32-deep instantiation only; its loads clamp k to K - 4)
      static const bool bk16 = ::rdm::dev_knob("RDM_GEMM_BK16") != nullptr;  // developer knob (A/B): shallow k-tiles, twice the resident workgroups
      if ((k >= 48 && !bk16) || g.aidx) launch<64, 64, 2, 2, 32, 2>(g, batches, trans_b, st);
      else launch<64, 64, 2, 2, 16>(g, batches, trans_b, st);
      break;
    case T128x32:
      if (k >= 32) launch<128, 32, 4, 1, 32, 2>(g, batches, trans_b, st);
      else launch<128, 32, 4, 1, 16>(g, batches, trans_b, st);
      break;
  }
  }
  if (int e = launch_status("gemm_kernel")) return e;
  if (g.splits > 1 && reduce_stats) {
    RDM_DUP_LOOP("splitk")
    ::rdm::launch<splitk_reduce_stats_kernel_body, splitk_reduce_stats_kernel, 256>(dim3(ceil_div<long long>(m, stat_rows_per_block(n)), ceil_div<long long>(n, 256)), 0, st, g, reduce_stats);
    return launch_status("splitk_reduce_stats_kernel");
  }
  if (g.splits > 1) {
    const long long total = static_cast<long long>(batches) * m * n;
    const int blocks = static_cast<int>(std::min<long long>(ceil_div<long long>(total, 256), 2048));
    ::rdm::launch<splitk_reduce_kernel_body, splitk_reduce_kernel, 256>(dim3(blocks), 0, st, g, batches);
    return launch_status("splitk_reduce_kernel");
  }
  return RDM_OK;
}

}  // namespace

int rdm::gemm_with_stats(const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc, int64_t m,
                         int64_t n, int64_t k, const float* bias, const float* rowdiv, void* ws, size_t ws_bytes,
                         double* gn_partial, int* gn_blocks, void* stream, int form) {
  GemmArgs g;
  g.A = a; g.B = b; g.C = c; g.bias = bias; g.rowdiv = rowdiv;
  g.M = static_cast<int>(m); g.N = static_cast<int>(n); g.K = static_cast<int>(k);
  g.lda = static_cast<int>(lda); g.ldb = static_cast<int>(ldb); g.ldc = static_cast<int>(ldc);
  g.sa = g.sb = g.sc = 0;
  g.act = 0; g.splits = 1; g.part = nullptr; g.stats = gn_partial;
  g.A2 = nullptr; g.aidx = nullptr; g.lda2 = g.ldi = g.c1 = g.n_coarse = 0; g.bidx = nullptr; g.n_b = 0;
  return gemm_dispatch(g, 1, false, ws, ws_bytes, gn_blocks, static_cast<hipStream_t>(stream), form);
}

// C = [nearest_upsample(coarse)[idx[:, 0]] | skip] B + bias (decoder, backbone.py:118-151) without materialising the
// concatenation: the GEMM's A tiles come from the two sources directly.  c1 (columns of coarse) must be a multiple of 32
// and c1 + c2 the padded K of B; otherwise returns 1 and the caller concatenates (rdm_upsample_concat) as before.
int rdm::gemm_concat_with_stats(const float* coarse, int64_t ld1, int64_t c1, int64_t n_coarse, const int64_t* idx, int64_t ldi,
                                const float* skip, int64_t ld2, int64_t c2, const float* b, int64_t ldb, float* c, int64_t ldc,
                                int64_t m, int64_t n, const float* bias, int act, void* ws, size_t ws_bytes, double* gn_partial,
                                int* gn_blocks, void* stream, int form) {
  if (c1 % 32 != 0 || c2 % 4 != 0 || c2 < 4 || ld1 % 4 != 0 || ld2 % 4 != 0 || ldb % 4 != 0 || m <= 0 ||
      ((reinterpret_cast<uintptr_t>(coarse) | reinterpret_cast<uintptr_t>(skip) | reinterpret_cast<uintptr_t>(b)) & 15) != 0)
    return 1;
  GemmArgs g;
  g.A = coarse; g.B = b; g.C = c; g.bias = bias; g.rowdiv = nullptr;
  g.M = static_cast<int>(m); g.N = static_cast<int>(n); g.K = static_cast<int>(c1 + c2);
  g.lda = static_cast<int>(ld1); g.ldb = static_cast<int>(ldb); g.ldc = static_cast<int>(ldc);
  g.sa = g.sb = g.sc = 0;
  g.act = act; g.splits = 1; g.part = nullptr; g.stats = gn_partial;
  g.A2 = skip; g.aidx = idx; g.lda2 = static_cast<int>(ld2); g.ldi = static_cast<int>(ldi); g.c1 = static_cast<int>(c1);
  g.n_coarse = static_cast<int>(n_coarse);
  g.bidx = nullptr; g.n_b = 0;
  return gemm_dispatch(g, 1, false, ws, ws_bytes, gn_blocks, static_cast<hipStream_t>(stream), form);
}

namespace {
bool small_kernel_ok(long long m, long long n, long long k) {
  return m <= 1536 && k % 16 == 0 && k >= 64 && k <= 1024 && m * n <= 1536 * 512 && !::rdm::dev_knob("RDM_GEMM_TUNE");
}
}  // namespace

// Two bias-only products C_i = A_i B_i + bias_i in one launch when both fit the 32x32 K-split kernel (else two launches).
int rdm::gemm_pair(const float* a0, int64_t lda0, const float* b0, int64_t ldb0, float* c0, int64_t ldc0, int64_t m0, int64_t n0,
                   int64_t k0, const float* bias0, const float* a1, int64_t lda1, const float* b1, int64_t ldb1, float* c1,
                   int64_t ldc1, int64_t m1, int64_t n1, int64_t k1, const float* bias1, void* ws, size_t ws_bytes, void* stream) {
  if (m0 > 0 && m1 > 0 && small_kernel_ok(m0, n0, k0) && small_kernel_ok(m1, n1, k1) && lda0 % 4 == 0 && lda1 % 4 == 0 &&
      ldb0 % 4 == 0 && ldb1 % 4 == 0 && ((reinterpret_cast<uintptr_t>(a0) | reinterpret_cast<uintptr_t>(a1)) & 15) == 0) {
    GemmArgs g[2];
    const float* A[2] = {a0, a1}; const float* B[2] = {b0, b1}; float* C[2] = {c0, c1}; const float* bias[2] = {bias0, bias1};
    const int64_t M[2] = {m0, m1}, N[2] = {n0, n1}, K[2] = {k0, k1}, LA[2] = {lda0, lda1}, LB[2] = {ldb0, ldb1}, LC[2] = {ldc0, ldc1};
    for (int i = 0; i < 2; ++i) {
      g[i].A = A[i]; g[i].B = B[i]; g[i].C = C[i]; g[i].bias = bias[i]; g[i].rowdiv = nullptr;
      g[i].M = static_cast<int>(M[i]); g[i].N = static_cast<int>(N[i]); g[i].K = static_cast<int>(K[i]);
      g[i].lda = static_cast<int>(LA[i]); g[i].ldb = static_cast<int>(LB[i]); g[i].ldc = static_cast<int>(LC[i]);
      g[i].sa = g[i].sb = g[i].sc = 0; g[i].act = 0; g[i].splits = 1; g[i].part = nullptr; g[i].stats = nullptr;
      g[i].A2 = nullptr; g[i].aidx = nullptr; g[i].lda2 = g[i].ldi = g[i].c1 = g[i].n_coarse = 0; g[i].bidx = nullptr; g[i].n_b = 0;
    }
    const long long gx = ceil_div<long long>(std::max(n0, n1), 32), gy = ceil_div<long long>(std::max(m0, m1), 32);
    ::rdm::launch<gemm_small_pair_entry, gemm_small_pair_kernel, 256>(dim3(gx, gy, 2), 0, static_cast<hipStream_t>(stream), g[0], g[1]);
    return launch_status("gemm_small_pair_kernel");
  }
  if (m0 > 0)
    if (int e = rdm_gemm(a0, lda0, 0, b0, ldb0, 0, 0, c0, ldc0, 0, m0, n0, k0, 1, bias0, nullptr, 0, ws, ws_bytes, stream)) return e;
  if (m1 > 0) return rdm_gemm(a1, lda1, 0, b1, ldb1, 0, 0, c1, ldc1, 0, m1, n1, k1, 1, bias1, nullptr, 0, ws, ws_bytes, stream);
  return RDM_OK;
}

// What the dispatch model chose for the calling thread's last product: {tile rows, tile columns, k-tile depth, split-K factor}
// ({32, 32, K/4, 4} = the transformer-width kernel whose four wavefronts split K).  For profiles/r03_gemm_shapes.md.
extern "C" int rdm_gemm_last_plan(int* out4_host) {
  using namespace rdm;
  RDM_REQUIRE(out4_host, "rdm_gemm_last_plan: null pointer");
  for (int i = 0; i < 4; ++i) out4_host[i] = g_last_plan[i];
  return RDM_OK;
}

extern "C" int rdm_gemm(const float* a, int64_t lda, int64_t stride_a, const float* b, int64_t ldb,
                        int64_t stride_b, int trans_b, float* c, int64_t ldc, int64_t stride_c,
                        int64_t m, int64_t n, int64_t k, int batches, const float* bias,
                        const float* rowdiv, int act, void* ws, size_t ws_bytes, void* stream) {
  return rdm_gemm_form(a, lda, stride_a, b, ldb, stride_b, trans_b, c, ldc, stride_c, m, n, k, batches, bias, rowdiv, act, ws, ws_bytes, 0, stream);
}

extern "C" int rdm_gemm_form(const float* a, int64_t lda, int64_t stride_a, const float* b, int64_t ldb,
                             int64_t stride_b, int trans_b, float* c, int64_t ldc, int64_t stride_c,
                             int64_t m, int64_t n, int64_t k, int batches, const float* bias,
                             const float* rowdiv, int act, void* ws, size_t ws_bytes, int form, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(form >= 0 && form <= 2, "rdm_gemm_form: unknown form %d", form);
  RDM_REQUIRE(a && b && c, "rdm_gemm: null pointer");
  RDM_REQUIRE(m >= 0 && n >= 0 && k >= 0 && batches >= 1, "rdm_gemm: bad sizes");
  if (m == 0 || n == 0) return RDM_OK;
  RDM_REQUIRE(k % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0,
              "rdm_gemm: K (%lld), lda (%lld), ldb (%lld) must be multiples of 4 (pad with zeros)",
              (long long)k, (long long)lda, (long long)ldb);
  RDM_REQUIRE((reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0,
              "rdm_gemm: A and B must be 16-byte aligned");
  RDM_REQUIRE(act >= 0 && act <= 2, "rdm_gemm: unknown activation %d", act);
  GemmArgs g;
  g.A = a; g.B = b; g.C = c; g.bias = bias; g.rowdiv = rowdiv;
  g.M = static_cast<int>(m); g.N = static_cast<int>(n); g.K = static_cast<int>(k);
  g.lda = static_cast<int>(lda); g.ldb = static_cast<int>(ldb); g.ldc = static_cast<int>(ldc);
  g.sa = stride_a; g.sb = stride_b; g.sc = stride_c;
  g.act = act; g.splits = 1; g.part = nullptr; g.stats = nullptr;
  g.A2 = nullptr; g.aidx = nullptr; g.lda2 = g.ldi = g.c1 = g.n_coarse = 0; g.bidx = nullptr; g.n_b = 0;
  return gemm_dispatch(g, batches, trans_b != 0, ws, ws_bytes, nullptr, static_cast<hipStream_t>(stream), form);
}

// y = act(GroupNorm(x W + b [/ rowdiv]) [+ residual]): the Linear/KPConv-weight GEMM writes its
// GroupNorm statistics from its own epilogue, so the normalisation costs one tiny finalize launch and
// one fused apply pass.  lin_out [m, n] receives the pre-norm activations (scratch for the caller).
extern "C" size_t rdm_linear_group_norm_workspace_bytes(int64_t m, int64_t n) {
  return rdm_gemm_workspace_bytes(m, n, 1) + rdm::align_up(static_cast<size_t>(rdm::gemm_stats_max_blocks(m)) * 2 * n * sizeof(double)) +
         rdm_group_norm_workspace_bytes(m, n) + 1024;
}

extern "C" int rdm_linear_group_norm(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                     const float* rowdiv, int64_t m, int64_t n, int64_t k, int groups,
                                     const float* gamma, const float* beta, float eps, const float* residual,
                                     int64_t ldr, int act, float* lin_out, int64_t ld_lin, float* y, int64_t ldy,
                                     uint8_t* positive, void* ws, size_t ws_bytes, void* stream) {
  return rdm_linear_group_norm_form(x, ldx, w, ldw, bias, rowdiv, m, n, k, groups, gamma, beta, eps, residual, ldr, act, lin_out, ld_lin, y, ldy,
                                    positive, ws, ws_bytes, 0, stream);
}

extern "C" int rdm_linear_group_norm_form(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                                          const float* rowdiv, int64_t m, int64_t n, int64_t k, int groups,
                                          const float* gamma, const float* beta, float eps, const float* residual,
                                          int64_t ldr, int act, float* lin_out, int64_t ld_lin, float* y, int64_t ldy,
                                          uint8_t* positive, void* ws, size_t ws_bytes, int form, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(form >= 0 && form <= 2, "rdm_linear_group_norm_form: unknown form %d", form);
  RDM_REQUIRE(x && w && gamma && beta && lin_out && y, "rdm_linear_group_norm: null pointer");
  RDM_REQUIRE(k % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0, "rdm_linear_group_norm: K, ldx, ldw must be multiples of 4");
  if (m == 0) return RDM_OK;
  Arena ar(ws, ws_bytes);
  const size_t gemm_ws = rdm_gemm_workspace_bytes(m, n, 1);
  char* gws = ar.take<char>(gemm_ws);
  double* partial = ar.take<double>(static_cast<size_t>(gemm_stats_max_blocks(m)) * 2 * n);
  const size_t gn_ws = rdm_group_norm_workspace_bytes(m, n);
  char* nws = ar.take<char>(gn_ws);
  if (!ar.ok) {
    set_error("rdm_linear_group_norm: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  int nblk = 0;
  if (int e = gemm_with_stats(x, ldx, w, ldw, lin_out, ld_lin, m, n, k, bias, rowdiv, gws, gemm_ws, partial, &nblk, stream, form))
    return e;
  return group_norm_finish(partial, nblk, lin_out, m, n, ld_lin, groups, gamma, beta, eps, residual, ldr, act, y, ldy,
                           positive, nws, gn_ws, stream);
}

// Patch score matrices (experiments/model_infer.py:291-311): scores[b, i, j] = <ref_feats[ref_idx[b, i]], src_feats[src_idx[b, j]]>
// / divisor[i] with the reference's padded gather (index outside the tensor: zero row) folded into the operand loads -- the
// [B, K, D] patch feature tensors (2 x 33 MB) are never written, and tiles made of shadow rows only are stored as zeros.
extern "C" int rdm_patch_scores(const float* ref_feats, int64_t ld_ref, int64_t n_ref, const int64_t* ref_idx,
                                const float* src_feats, int64_t ld_src, int64_t n_src, const int64_t* src_idx, int64_t batch,
                                int64_t side, int64_t d, const float* rowdiv, float* scores, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(ref_feats && src_feats && ref_idx && src_idx && scores, "rdm_patch_scores: null pointer");
  RDM_REQUIRE(batch >= 0 && side > 0 && d > 0 && d % 4 == 0 && ld_ref % 4 == 0 && ld_src % 4 == 0 && n_ref > 0 && n_src > 0,
              "rdm_patch_scores: bad sizes");
  RDM_REQUIRE(((reinterpret_cast<uintptr_t>(ref_feats) | reinterpret_cast<uintptr_t>(src_feats)) & 15) == 0,
              "rdm_patch_scores: features must be 16-byte aligned");
  if (batch == 0) return RDM_OK;
  GemmArgs g;
  g.A = ref_feats; g.B = src_feats; g.C = scores; g.bias = nullptr; g.rowdiv = rowdiv;
  g.M = static_cast<int>(side); g.N = static_cast<int>(side); g.K = static_cast<int>(d);
  g.lda = static_cast<int>(ld_ref); g.ldb = static_cast<int>(ld_src); g.ldc = static_cast<int>(side);
  g.sa = g.sb = 0; g.sc = side * side;
  g.act = 0; g.splits = 1; g.part = nullptr; g.stats = nullptr;
  g.A2 = nullptr; g.aidx = ref_idx; g.lda2 = g.ldi = g.c1 = 0; g.n_coarse = static_cast<int>(n_ref);
  g.bidx = src_idx; g.n_b = static_cast<int>(n_src);
  return gemm_dispatch(g, static_cast<int>(batch), true, nullptr, 0, nullptr, static_cast<hipStream_t>(stream));
}

// Decoder stage (experiments/backbone.py:118-151): y = act(GroupNorm([nearest_upsample(coarse) | skip] W + b)), or the plain
// Linear into lin_out when gamma is null (decoder2).  The concatenated rows are formed inside the GEMM's A-tile loads when
// c1 is a multiple of 32 (gemm_concat_with_stats); other widths (the 257-column coarse tensor of decoder4) go through
// rdm_upsample_concat into the workspace first.
extern "C" size_t rdm_decoder_stage_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  return rdm_linear_group_norm_workspace_bytes(m, n) + rdm::align_up(static_cast<size_t>(m > 0 ? m : 1) * ((k + 3) / 4 * 4) * sizeof(float));
}

extern "C" int rdm_decoder_stage(const float* coarse, int64_t n_coarse, int64_t c1, int64_t ld1, const int64_t* idx, int64_t ldi,
                                 const float* skip, int64_t c2, int64_t ld2, int64_t m, const float* w, int64_t ldw,
                                 const float* bias, int64_t n, int groups, const float* gamma, const float* beta, float eps,
                                 int act, float* lin_out, int64_t ld_lin, float* y, int64_t ldy, void* ws, size_t ws_bytes,
                                 void* stream) {
  return rdm_decoder_stage_form(coarse, n_coarse, c1, ld1, idx, ldi, skip, c2, ld2, m, w, ldw, bias, n, groups, gamma, beta, eps, act, lin_out, ld_lin,
                                y, ldy, ws, ws_bytes, 0, stream);
}

extern "C" int rdm_decoder_stage_form(const float* coarse, int64_t n_coarse, int64_t c1, int64_t ld1, const int64_t* idx, int64_t ldi,
                                      const float* skip, int64_t c2, int64_t ld2, int64_t m, const float* w, int64_t ldw,
                                      const float* bias, int64_t n, int groups, const float* gamma, const float* beta, float eps,
                                      int act, float* lin_out, int64_t ld_lin, float* y, int64_t ldy, void* ws, size_t ws_bytes,
                                      int form, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(form >= 0 && form <= 2, "rdm_decoder_stage_form: unknown form %d", form);
  RDM_REQUIRE(coarse && idx && skip && w && lin_out && (!gamma || (beta && y)), "rdm_decoder_stage: null pointer");
  RDM_REQUIRE(c1 > 0 && c2 > 0 && n > 0 && m >= 0 && ldw % 4 == 0, "rdm_decoder_stage: bad sizes");
  if (m == 0) return RDM_OK;
  const int64_t k = c1 + c2, kpad = (k + 3) / 4 * 4;
  static const bool no_virtual = ::rdm::dev_knob("RDM_NO_VIRTUAL_CONCAT") != nullptr;  // developer knob: always materialise
  Arena ar(ws, ws_bytes);
  const size_t gemm_ws = rdm_gemm_workspace_bytes(m, n, 1);
  char* gws = ar.take<char>(gemm_ws);
  double* partial = ar.take<double>(static_cast<size_t>(gemm_stats_max_blocks(m)) * 2 * n);
  const size_t gn_ws = rdm_group_norm_workspace_bytes(m, n);
  char* nws = ar.take<char>(gn_ws);
  float* cat = ar.take<float>(static_cast<size_t>(m) * kpad);
  if (!ar.ok) {
    set_error("rdm_decoder_stage: workspace too small (%zu < %zu)", ws_bytes, ar.off);
    return RDM_ERR_WORKSPACE;
  }
  int nblk = 0;
  int rc = 1;
  if (!no_virtual && k == kpad)
    rc = gemm_concat_with_stats(coarse, ld1, c1, n_coarse, idx, ldi, skip, ld2, c2, w, ldw, lin_out, ld_lin, m, n, bias, 0, gws,
                                gemm_ws, gamma ? partial : nullptr, &nblk, stream, form);
  if (rc == 1) {
    if (int e = rdm_upsample_concat(coarse, n_coarse, c1, ld1, idx, ldi, skip, c2, ld2, m, cat, kpad, stream)) return e;
    rc = gemm_with_stats(cat, kpad, w, ldw, lin_out, ld_lin, m, n, kpad, bias, nullptr, gws, gemm_ws, gamma ? partial : nullptr,
                         &nblk, stream, form);
  }
  if (rc != 0 || !gamma) return rc;
  return group_norm_finish(partial, nblk, lin_out, m, n, ld_lin, groups, gamma, beta, eps, nullptr, 0, act, y, ldy, nullptr, nws,
                           gn_ws, stream);
}

// y = act(LayerNorm(x W^T + bias [+ residual]) * gamma + beta) with W [128, k] (nn.Linear weight as stored): the
// Linear + residual LayerNorm pairs of the attention layers (thdroformer.py:159-173, vanilla_transformer.py:87-103,
// output_layer.py:13-21) as one launch.  n must be 128 and k a multiple of 16; other widths use rdm_gemm + rdm_layer_norm.
extern "C" int rdm_linear_layer_norm(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int64_t m,
                                     int64_t n, int64_t k, const float* residual, int64_t ldr, const float* gamma,
                                     const float* beta, float eps, int act, float* y, int64_t ldy, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(x && w && gamma && beta && y, "rdm_linear_layer_norm: null pointer");
  RDM_REQUIRE(n == 128 && k >= 16 && k % 16 == 0 && m >= 0 && ldx % 4 == 0 && ldw >= k && ldw % 4 == 0 && act >= 0 && act <= 1,
              "rdm_linear_layer_norm: supports n = 128, k multiple of 16 (n=%lld k=%lld)", (long long)n, (long long)k);
  RDM_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0,
              "rdm_linear_layer_norm: x and w must be 16-byte aligned");
  if (m == 0) return RDM_OK;
  LinLnArgs a;
  a.A = x; a.B = w; a.bias = bias; a.res = residual; a.gamma = gamma; a.beta = beta; a.out = y;
  a.M = static_cast<int>(m); a.K = static_cast<int>(k); a.lda = static_cast<int>(ldx); a.ldb = static_cast<int>(ldw);
  a.ldr = static_cast<int>(ldr); a.ldo = static_cast<int>(ldy); a.act = act; a.eps = eps;
  ::rdm::launch<linear_ln128_body, linear_ln128_kernel, 256>(dim3(static_cast<unsigned>(ceil_div<int64_t>(m, 16))), 0, static_cast<hipStream_t>(stream), a);
  return launch_status("linear_ln128_kernel");
}

// The tail of an attention layer (output projection + residual LayerNorm + FFN + residual LayerNorm) in one launch;
// see attention_tail128_kernel.  Weights in checkpoint layout (k contiguous): wo [128,128], w1 [256,128], w2 [128,256].
extern "C" int rdm_attention_tail(const float* hidden, int64_t ld_hidden, const float* x, int64_t ldx, int64_t m, int64_t d,
                                       const float* wo, int64_t ld_wo, const float* bo, const float* gamma1, const float* beta1,
                                       const float* w1, int64_t ld_w1, const float* b1, const float* w2, int64_t ld_w2,
                                       const float* b2, const float* gamma2, const float* beta2, float eps, float* out,
                                       int64_t ld_out, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(hidden && x && wo && w1 && w2 && gamma1 && beta1 && gamma2 && beta2 && out, "rdm_attention_tail: null pointer");
  RDM_REQUIRE(d == 128 && m >= 0, "rdm_attention_tail: supports d = 128 with a 256-wide FFN (d=%lld)", (long long)d);
  RDM_REQUIRE(ld_hidden % 4 == 0 && ld_wo % 4 == 0 && ld_w1 % 4 == 0 && ld_w2 % 4 == 0 && ld_wo >= 128 && ld_w1 >= 128 &&
                  ld_w2 >= 256 && ld_hidden >= 128 && ldx >= 128 && ld_out >= 128,
              "rdm_attention_tail: bad leading dimensions");
  RDM_REQUIRE(((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(wo) | reinterpret_cast<uintptr_t>(w1) |
                reinterpret_cast<uintptr_t>(w2)) & 15) == 0,
              "rdm_attention_tail: hidden and the weights must be 16-byte aligned");
  if (m == 0) return RDM_OK;
  TailArgs a;
  a.hid = hidden; a.x = x; a.wo = wo; a.bo = bo; a.g1 = gamma1; a.be1 = beta1; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
  a.g2 = gamma2; a.be2 = beta2; a.out = out; a.M = static_cast<int>(m); a.ldh = static_cast<int>(ld_hidden);
  a.ldx = static_cast<int>(ldx); a.ldo = static_cast<int>(ld_out); a.ldwo = static_cast<int>(ld_wo);
  a.ldw1 = static_cast<int>(ld_w1); a.ldw2 = static_cast<int>(ld_w2); a.eps = eps;
  a.packed = nullptr;
  RDM_DUP_LOOP("tail")
  ::rdm::launch<attention_tail128_body<false>, attention_tail128_kernel<false>, 512>(dim3(static_cast<unsigned>(ceil_div<int64_t>(m, 16))), 0, static_cast<hipStream_t>(stream), a);
  return launch_status("attention_tail128_kernel");
}

// The same on weights in operand order (rdm_attention_tail_pack_weights): every weight load of a wavefront is one contiguous KB.
extern "C" size_t rdm_attention_tail_packed_floats(void) { return 81920; }

extern "C" int rdm_attention_tail_pack_weights(const float* wo, int64_t ld_wo, const float* w1, int64_t ld_w1, const float* w2,
                                               int64_t ld_w2, float* packed, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(wo && w1 && w2 && packed, "rdm_attention_tail_pack_weights: null pointer");
  RDM_REQUIRE(ld_wo % 4 == 0 && ld_w1 % 4 == 0 && ld_w2 % 4 == 0 && ld_wo >= 128 && ld_w1 >= 128 && ld_w2 >= 256,
              "rdm_attention_tail_pack_weights: bad leading dimensions");
  RDM_REQUIRE(((reinterpret_cast<uintptr_t>(wo) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
                reinterpret_cast<uintptr_t>(packed)) & 15) == 0, "rdm_attention_tail_pack_weights: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(attention_tail_pack_kernel, dim3(80), dim3(256), 0, static_cast<hipStream_t>(stream), wo, static_cast<int>(ld_wo), w1,
                     static_cast<int>(ld_w1), w2, static_cast<int>(ld_w2), reinterpret_cast<float4*>(packed));
  return launch_status("attention_tail_pack_kernel");
}

extern "C" int rdm_attention_tail_packed(const float* hidden, int64_t ld_hidden, const float* x, int64_t ldx, int64_t m, int64_t d,
                                         const float* packed, const float* bo, const float* gamma1, const float* beta1,
                                         const float* b1, const float* b2, const float* gamma2, const float* beta2, float eps,
                                         float* out, int64_t ld_out, void* stream) {
  using namespace rdm;
  RDM_REQUIRE(hidden && x && packed && gamma1 && beta1 && gamma2 && beta2 && out, "rdm_attention_tail_packed: null pointer");
  RDM_REQUIRE(d == 128 && m >= 0, "rdm_attention_tail_packed: supports d = 128 with a 256-wide FFN (d=%lld)", (long long)d);
  RDM_REQUIRE(ld_hidden % 4 == 0 && ld_hidden >= 128 && ldx >= 128 && ld_out >= 128, "rdm_attention_tail_packed: bad leading dimensions");
  RDM_REQUIRE(((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(packed)) & 15) == 0,
              "rdm_attention_tail_packed: hidden and the packed weights must be 16-byte aligned");
  if (m == 0) return RDM_OK;
  TailArgs a;
  a.hid = hidden; a.x = x; a.wo = nullptr; a.bo = bo; a.g1 = gamma1; a.be1 = beta1; a.w1 = nullptr; a.b1 = b1; a.w2 = nullptr; a.b2 = b2;
  a.g2 = gamma2; a.be2 = beta2; a.out = out; a.M = static_cast<int>(m); a.ldh = static_cast<int>(ld_hidden);
  a.ldx = static_cast<int>(ldx); a.ldo = static_cast<int>(ld_out); a.ldwo = a.ldw1 = a.ldw2 = 0; a.eps = eps;
  a.packed = reinterpret_cast<const float4*>(packed);
  RDM_DUP_LOOP("tail")
  ::rdm::launch<attention_tail128_body<true>, attention_tail128_kernel<true>, 512>(dim3(static_cast<unsigned>(ceil_div<int64_t>(m, 16))), 0, static_cast<hipStream_t>(stream), a);
  return launch_status("attention_tail128_kernel");
}

