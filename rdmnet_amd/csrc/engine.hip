// Native orchestration of the whole path: one call per scan pair.
//
// rdm_engine_run = collate (geotransformer/utils/data.py:13-77) + RDMNet.forward
// (experiments/model_infer.py:109-354) expressed as a sequence of this library's own C-ABI kernels
// on one HIP stream, with activations bump-allocated from an engine-owned device arena.  It exists
// because the path is hundreds of short launches per pair (~700 at first, ~340 now): issued from Python they cost ~15 us each, issued
// from here ~2 us.  The op sequence is identical to rdmnet_amd/model.py (the per-op mirror used by
// the stage tests), so both produce bit-identical results.
//
// Host synchronisations per pair (data-dependent sizes): subsampled level sizes, NMS survivor
// counts, number of patch correspondences, number of point correspondences.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rdmnet_hip.h"
#include <ctime>

#include "common.h"
#include "internal.h"
#include "lockstep.h"

namespace {

using namespace rdm;

struct Mat {  // row-major view
  float* p = nullptr;
  int64_t rows = 0, cols = 0, ld = 0;
  Mat cols_from(int64_t c0, int64_t n) const { return Mat{p + c0, rows, n, ld}; }
  Mat rows_from(int64_t r0, int64_t n) const { return Mat{p + r0 * ld, n, cols, ld}; }
};

struct HostParam {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

struct Linear {  // B operand [K_pad, N_pad] + bias
  float* b = nullptr;
  float* wt = nullptr;  // [out, kpad] (checkpoint layout, k contiguous) for the fused transformer kernels, out % 128 == 0 only
  float* bias = nullptr;
  float* packed = nullptr;  // KPConv weights in the operand order of the fused kernel (c_in = 1, 32, 64), else null
  int64_t in = 0, out = 0, kpad = 0, ldb = 0;
};

inline int64_t pad4(int64_t n) { return (n + 3) / 4 * 4; }

// The prepared device parameters (weights in the kernels' layouts, ~250 MB).  Engines hold them through a shared_ptr:
// rdm_engine_share_params hands the SAME set to another engine, and the device memory is released when the last engine
// that uses it is destroyed or re-finalized -- in whatever order the caller destroys them.
struct ParamSet {
  std::map<std::string, Linear> lin;
  std::map<std::string, float*> vec;
  std::vector<void*> owned;  // device allocations
  int device = -1;
  ParamSet() = default;
  ParamSet(const ParamSet&) = delete;
  ParamSet& operator=(const ParamSet&) = delete;
  ~ParamSet() {
    for (void* p : owned) (void)hipFree(p);
  }
};

struct Level {
  float* pts = nullptr;
  int64_t n = 0;
  int64_t* lengths = nullptr;  // device [2]
  int64_t n_ref = 0;
};
struct Table {
  int64_t* idx = nullptr;     // int32 elements when i32 (tables the engine builds AND consumes itself in a plain run)
  bool i32 = false;
  int64_t rows = 0, width = 0;
  int64_t ld = 0;            // row stride; 0 = width (tables built by the engine's own collate)
  int32_t* flags = nullptr;  // device [2]: max_count, status (optional: null = every column is valid)
  int64_t stride() const { return ld ? ld : width; }
};

struct Grid {  // a level's search grid (rdm_radius_grid_build): workspace + the support size it was built for
  void* ws = nullptr;
  size_t bytes = 0;
  int64_t n_s = 0;
};
// What the collate of one pair leaves for its forward (geotransformer/utils/data.py:13-77): the pyramid, the tables the engine
// consumes itself (int32 elements), the level grids (their cell-sorted records order the KPConv queries) and the status words.
struct PairPyramid {
  Level lv[5];
  Table nb[5], sub[4], up[4];
  Grid grids[5];
  int32_t* flags = nullptr;  // device [64]
  int calls = 0;             // searches recorded in `flags`
};

}  // namespace

struct rdm_engine {
  rdm_engine_config cfg;
  int device = -1;            // the device that was current at rdm_engine_create
  std::map<std::string, HostParam> host;
  std::shared_ptr<ParamSet> params = std::make_shared<ParamSet>();
  char* arena = nullptr;
  size_t arena_cap = 0, arena_off = 0;
  bool arena_exhausted = false, arena_fixed = false;  // fixed: the caller chose arena_bytes, never regrown
  void* pinned = nullptr;     // small host staging buffer for the size read-backs (mapped: kernels write it)
  void* pinned_dev = nullptr; // its device address
  int64_t host_corr_cap = 0;  // correspondences the mapped buffer holds behind its first 4 KB
  int wait_sleep_us = 0;      // 0: hipStreamSynchronize (spins a core); > 0: poll hipStreamQuery and sleep in between
  bool finalized = false;
  std::map<std::string, rdm_tensor_view> taps;
  bool keep_taps = false;
  bool collate_only = false;  // rdm_engine_collate: stop after the pyramid and its searches
  int pairs_in_flight = 1;    // rdm_engine_set_pairs_in_flight: how many pairs share the GPU (>= 3: GEMM residency capped)
  // Latency mode (rdm_engine_set_overlap): with ONE pair in flight most of the GPU idles while a chain of one-workgroup kernels
  // runs, so the engine runs the wide, independent parts of a pair (the first level's search + blocks, the decoder) on a side
  // stream of its own and joins them where their results are needed.  Same kernels, same operands: same bits.
  int overlap_mode = 1;       // 0: never, 1: when pairs_in_flight == 1, 2: always
  hipStream_t side = nullptr;
  std::map<hipStream_t, bool> side_ok;  // caller stream -> the side stream runs beside it (see ensure_side)
  // Batched collate (rdm_engine_collate_batch): the pyramids of B pairs built by ONE sequence of launches -- 2 B clouds per
  // subsampling launch, (pair, level) items per grid build, 16 searches per query launch -- stay in the arena below `arena_base`;
  // rdm_engine_forward_batched(k) runs pair k's forward above it.  Same kernels on the same operands per pair: same bits.
  std::vector<PairPyramid> batch;
  std::vector<int64_t> batch_n_ref, batch_n_src;
  size_t arena_base = 0;
  bool profile = false;
  bool profile_shapes_only = false;  // (enable = 2: the layer records without events -- the other pairs of a profiled lock-step group)
  std::vector<hipEvent_t> events;      // 3 per KPConv layer: before gather, between, after GEMM
  std::vector<rdm_kpconv_profile> prof;  // filled at the end of a run
  int prof_layers = 0;

  template <typename T>
  T* alloc(size_t count) {
    size_t bytes = align_up(count * sizeof(T) + 16);
    if (arena_off + bytes > arena_cap) return nullptr;
    T* r = reinterpret_cast<T*>(arena + arena_off);
    arena_off += bytes;
    return r;
  }
  Mat mat(int64_t rows, int64_t cols) {
    Mat m;
    m.rows = rows; m.cols = cols; m.ld = pad4(cols);
    m.p = alloc<float>(static_cast<size_t>(rows > 0 ? rows : 1) * m.ld);
    return m;
  }
};

namespace {

#define ENG_CHECK(...)             \
  do {                             \
    int _rc = (__VA_ARGS__);       \
    if (_rc != 0) return _rc;      \
  } while (0)
#define ENG_ALLOC(ptr)                                             \
  do {                                                             \
    if ((ptr) == nullptr) {                                        \
      set_error("rdm_engine: activation arena exhausted (%zu B)", e->arena_cap); \
      e->arena_exhausted = true;                                   \
      return RDM_ERR_WORKSPACE;                                    \
    }                                                              \
  } while (0)

struct Run {  // per-call context
  rdm_engine* e;
  hipStream_t st;
  void* ws;
  size_t ws_bytes;
  int groups;
};

void tap(Run& r, const char* name, const void* p, int64_t rows, int64_t cols, int64_t ld, int dtype) {
  if (!r.e->keep_taps) return;
  rdm_tensor_view v;
  v.data = const_cast<void*>(p); v.rows = rows; v.cols = cols; v.ld = ld; v.dtype = dtype;
  r.e->taps[name] = v;
}
void tap(Run& r, const char* name, const Mat& m) { tap(r, name, m.p, m.rows, m.cols, m.ld, 0); }

int linear(Run& r, const std::string& name, const Mat& x, Mat& y, int act = 0, bool alloc_out = true) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  auto it = e->params->lin.find(name);
  if (it == e->params->lin.end()) {
    set_error("rdm_engine: missing parameter %s", name.c_str());
    return RDM_ERR_ARG;
  }
  const Linear& L = it->second;
  if (alloc_out) {
    y = e->mat(x.rows, L.out);
    ENG_ALLOC(y.p);
  }
  return rdm_gemm(x.p, x.ld, 0, L.b, L.ldb, 0, 0, y.p, y.ld, 0, x.rows, L.out, L.kpad, 1, L.bias, nullptr, act, r.ws,
                  r.ws_bytes, r.st);
}

// two independent Linear layers as one launch when both are transformer-sized (gemm_pair)
int linear_pair(Run& r, const std::string& name0, const Mat& x0, Mat& y0, const std::string& name1, const Mat& x1, Mat& y1) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  auto i0 = e->params->lin.find(name0), i1 = e->params->lin.find(name1);
  if (i0 == e->params->lin.end() || i1 == e->params->lin.end()) {
    set_error("rdm_engine: missing parameter %s", (i0 == e->params->lin.end() ? name0 : name1).c_str());
    return RDM_ERR_ARG;
  }
  const Linear &L0 = i0->second, &L1 = i1->second;
  y0 = e->mat(x0.rows, L0.out);
  y1 = e->mat(x1.rows, L1.out);
  ENG_ALLOC(y0.p); ENG_ALLOC(y1.p);
  return gemm_pair(x0.p, x0.ld, L0.b, L0.ldb, y0.p, y0.ld, x0.rows, L0.out, L0.kpad, L0.bias, x1.p, x1.ld, L1.b, L1.ldb, y1.p,
                   y1.ld, x1.rows, L1.out, L1.kpad, L1.bias, r.ws, r.ws_bytes, r.st);
}

float* vecp(Run& r, const std::string& name) {
  auto it = r.e->params->vec.find(name);
  return it == r.e->params->vec.end() ? nullptr : it->second;
}

int group_norm(Run& r, const std::string& name, const Mat& x, Mat& y, int act, const Mat* res, uint8_t* positive) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  y = e->mat(x.rows, x.cols);
  ENG_ALLOC(y.p);
  float* g = vecp(r, name + ".norm.weight");
  float* b = vecp(r, name + ".norm.bias");
  if (!g || !b) {
    set_error("rdm_engine: missing parameter %s.norm.*", name.c_str());
    return RDM_ERR_ARG;
  }
  return rdm_group_norm(x.p, x.rows, x.cols, x.ld, r.groups, g, b, 1e-5f, res ? res->p : nullptr, res ? res->ld : 0, act,
                        y.p, y.ld, positive, r.ws, r.ws_bytes, r.st);
}

// Linear + GroupNorm (+ residual, activation): statistics come out of the GEMM epilogue
int unary(Run& r, const std::string& name, const Mat& x, Mat& y, int act, const Mat* res, uint8_t* positive) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  auto it = e->params->lin.find(name + ".mlp");
  if (it == e->params->lin.end()) {
    set_error("rdm_engine: missing parameter %s.mlp", name.c_str());
    return RDM_ERR_ARG;
  }
  const Linear& L = it->second;
  Mat t = e->mat(x.rows, L.out);
  y = e->mat(x.rows, L.out);
  ENG_ALLOC(t.p); ENG_ALLOC(y.p);
  float* g = vecp(r, name + ".norm.norm.weight");
  float* b = vecp(r, name + ".norm.norm.bias");
  if (!g || !b) {
    set_error("rdm_engine: missing parameter %s.norm.norm.*", name.c_str());
    return RDM_ERR_ARG;
  }
  return rdm_linear_group_norm(x.p, x.ld, L.b, L.ldb, L.bias, nullptr, x.rows, L.out, L.kpad, r.groups, g, b, 1e-5f,
                               res ? res->p : nullptr, res ? res->ld : 0, act, t.p, t.ld, y.p, y.ld, positive, r.ws,
                               r.ws_bytes, r.st);
}

// Decoder stage (backbone.py:118-151): Linear [+ GroupNorm + LeakyReLU] of [nearest_upsample(coarse) | skip]; the GEMM forms the
// concatenated rows in its A-tile loads when the widths allow it (rdm_decoder_stage) -- the [M, c1 + c2] tensor (33 MB at the
// finest decoder level) is then neither written nor re-read.
int decoder_stage(Run& r, const std::string& lin_name, const std::string* norm_name, const Mat& coarse, const int64_t* up_idx,
                  int64_t up_ld, const Mat& skip, int64_t m, Mat& y) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  auto it = e->params->lin.find(lin_name);
  if (it == e->params->lin.end()) {
    set_error("rdm_engine: missing parameter %s", lin_name.c_str());
    return RDM_ERR_ARG;
  }
  const Linear& L = it->second;
  RDM_REQUIRE(coarse.cols + skip.cols == L.in, "rdm_engine: %s expects %lld input columns", lin_name.c_str(), (long long)L.in);
  Mat t = e->mat(m, L.out);
  ENG_ALLOC(t.p);
  float *g = nullptr, *b = nullptr;
  if (norm_name) {
    y = e->mat(m, L.out);
    ENG_ALLOC(y.p);
    g = vecp(r, *norm_name + ".weight");
    b = vecp(r, *norm_name + ".bias");
    if (!g || !b) {
      set_error("rdm_engine: missing parameter %s.*", norm_name->c_str());
      return RDM_ERR_ARG;
    }
  } else {
    y = t;
  }
  const size_t need = rdm_decoder_stage_workspace_bytes(m, L.out, L.in);
  void* ws = r.ws;
  size_t ws_bytes = r.ws_bytes;
  if (need > ws_bytes) {
    ws = e->alloc<char>(need);
    ENG_ALLOC(ws);
    ws_bytes = need;
  }
  return rdm_decoder_stage(coarse.p, coarse.rows, coarse.cols, coarse.ld, up_idx, up_ld, skip.p, skip.cols, skip.ld, m, L.b, L.ldb,
                           L.bias, L.out, r.groups, g, b, 1e-5f, 2, t.p, t.ld, norm_name ? y.p : nullptr, norm_name ? y.ld : 0, ws,
                           ws_bytes, r.st);
}

int layer_norm(Run& r, const std::string& name, const Mat& x, const Mat* res, int act, Mat& y, bool alloc_out = true) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  if (alloc_out) {
    y = e->mat(x.rows, x.cols);
    ENG_ALLOC(y.p);
  }
  return rdm_layer_norm(x.p, x.rows, x.cols, x.ld, res ? res->p : nullptr, res ? res->ld : 0, vecp(r, name + ".weight"),
                        vecp(r, name + ".bias"), 1e-5f, act, y.p, y.ld, r.st);
}

// A KPConv layer's profile event k (rdm_engine_enable_profile).  In a lock-step group the launches around it are deferred: the
// event is too (lockstep_event: recorded right before this pair's next recorded launch goes out -- after the grouped launch the
// layer's kernel became, which serves every pair of the group: the record's durations are the GROUP's).
int layer_event(rdm_engine* e, int k, hipStream_t st) {
  if (e->profile_shapes_only) return RDM_OK;
  if (lockstep_active()) {
    lockstep_event(e->events[k], st);
    return RDM_OK;
  }
  RDM_HIP_CHECK(hipEventRecord(e->events[k], st));
  return RDM_OK;
}

int kpconv(Run& r, const std::string& name, const Mat& x, const uint8_t* x_pos, const Level& q, const Level& s,
           const Table& t, float sigma, const std::string& norm_name, Mat& y, const float* order,
           const Mat* pool_src = nullptr, Mat* pool_out = nullptr) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  auto it = e->params->lin.find(name + ".weights");
  if (it == e->params->lin.end()) {
    set_error("rdm_engine: missing parameter %s.weights", name.c_str());
    return RDM_ERR_ARG;
  }
  const Linear& W = it->second;
  const int64_t cin = x.cols, kdim = cin == 1 ? 16 : 15 * cin;
  const int i32 = t.i32 ? 1 : 0;  // the convolution kernels and the shortcut pool read `t` in its own element width
  if (W.packed && rdm_kpconv_fused_enabled()) {
    // fine levels (c_in = 1, 32, 64): the whole convolution is one kernel, the [M, 15 C] block never leaves the CU
    float* gam = vecp(r, norm_name + ".norm.weight");
    float* bet = vecp(r, norm_name + ".norm.bias");
    if (!gam || !bet) {
      set_error("rdm_engine: missing parameter %s.norm.*", norm_name.c_str());
      return RDM_ERR_ARG;
    }
    Mat conv = e->mat(q.n, W.out);
    y = e->mat(q.n, W.out);
    ENG_ALLOC(conv.p); ENG_ALLOC(y.p);
    RDM_REQUIRE(rdm_kpconv_fused_workspace_bytes(q.n, cin, W.out) <= r.ws_bytes, "rdm_engine: scratch too small");
    const int li = e->prof_layers;
    const bool prof = e->profile && (e->profile_shapes_only ? li < 16 : 3 * li + 2 < static_cast<int>(e->events.size()));
    if (prof) ENG_CHECK(layer_event(e, 3 * li, r.st));
    // (rdm_kpconv_fused_group_norm's two halves, so that the layer events bracket the convolution kernel alone)
    const int nblk = static_cast<int>(rdm_kpconv_fused_partial_rows(q.n, cin));
    double* gn_partial = static_cast<double*>(r.ws);
    const size_t stat_bytes = align_up(static_cast<size_t>(nblk) * 2 * W.out * sizeof(double));
    ENG_CHECK(kpconv_fused_impl(q.pts, q.n, s.pts, s.n, x.p, cin, x.ld, x_pos, t.idx, t.width, t.stride(), t.flags,
                                vecp(r, name + ".kernel_points"), sigma, W.packed, W.bias, W.out, conv.p, conv.ld, gn_partial, order, 0,
                                i32, r.st));
    if (prof) ENG_CHECK(layer_event(e, 3 * li + 1, r.st));
    ENG_CHECK(group_norm_finish(gn_partial, nblk, conv.p, q.n, W.out, conv.ld, r.groups, gam, bet, 1e-5f, nullptr, 0, 2, y.p, y.ld,
                                nullptr, static_cast<char*>(r.ws) + stat_bytes, r.ws_bytes - stat_bytes, r.st));
    if (pool_src) {
      *pool_out = e->mat(q.n, pool_src->cols);
      ENG_ALLOC(pool_out->p);
      ENG_CHECK(gather_max_ordered(pool_src->p, pool_src->rows, pool_src->cols, pool_src->ld, t.idx, q.n, t.width, t.stride(),
                                   t.flags, pool_out->p, pool_out->ld, order, i32, r.st));
    }
    if (prof) {
      ENG_CHECK(layer_event(e, 3 * li + 2, r.st));
      rdm_kpconv_profile p;
      p.m = q.n; p.h = t.width; p.c_in = cin; p.c_out = W.out; p.pooled_channels = pool_src ? pool_src->cols : 0;
      p.gather_ms = p.total_ms = 0.f;
      p.fused = 1; p.reserved = 0;
      e->prof.push_back(p);
      e->prof_layers++;
    }
    (void)order;
    return RDM_OK;
  }
  Mat wf = e->mat(q.n, kdim);
  ENG_ALLOC(wf.p);
  float* nn = e->alloc<float>(q.n > 0 ? q.n : 1);
  ENG_ALLOC(nn);
  const int li = e->prof_layers;
  const bool prof = e->profile && (e->profile_shapes_only ? li < 16 : 3 * li + 2 < static_cast<int>(e->events.size()));
  if (prof) ENG_CHECK(layer_event(e, 3 * li, r.st));
  ENG_CHECK(kpconv_gather_impl(q.pts, q.n, s.pts, s.n, x.p, cin, x.ld, x_pos, t.idx, t.width, t.stride(), t.flags,
                               vecp(r, name + ".kernel_points"), sigma, wf.p, wf.ld, nn, order, i32, r.st));
  if (prof) ENG_CHECK(layer_event(e, 3 * li + 1, r.st));
  Mat conv = e->mat(q.n, W.out);
  ENG_ALLOC(conv.p);
  // scratch split: [GEMM split-K partials | GroupNorm partials | GroupNorm finish]
  const size_t gemm_ws = rdm_gemm_workspace_bytes(q.n, W.out, 1);
  const size_t stat_bytes = align_up(static_cast<size_t>(gemm_stats_max_blocks(q.n)) * 2 * W.out * sizeof(double));
  RDM_REQUIRE(gemm_ws + stat_bytes + rdm_group_norm_workspace_bytes(q.n, W.out) <= r.ws_bytes, "rdm_engine: scratch too small");
  double* gn_partial = reinterpret_cast<double*>(static_cast<char*>(r.ws) + gemm_ws);
  int gn_blocks = 0;
  ENG_CHECK(gemm_with_stats(wf.p, wf.ld, W.b, W.ldb, conv.p, conv.ld, q.n, W.out, W.kpad, W.bias, nn, r.ws, gemm_ws,
                            gn_partial, &gn_blocks, r.st));
  if (pool_src) {  // strided block: the shortcut max-pool over the same neighbour table (functional.py:54-67)
    *pool_out = e->mat(q.n, pool_src->cols);
    ENG_ALLOC(pool_out->p);
    ENG_CHECK(gather_max_ordered(pool_src->p, pool_src->rows, pool_src->cols, pool_src->ld, t.idx, q.n, t.width, t.stride(),
                                   t.flags, pool_out->p, pool_out->ld, order, i32, r.st));
  }
  if (prof) {
    ENG_CHECK(layer_event(e, 3 * li + 2, r.st));
    rdm_kpconv_profile p;
    p.m = q.n; p.h = t.width; p.c_in = cin; p.c_out = W.out; p.pooled_channels = pool_src ? pool_src->cols : 0;
    p.gather_ms = p.total_ms = 0.f;
    p.fused = 0; p.reserved = 0;
    e->prof.push_back(p);
    e->prof_layers++;
  }
  // GroupNorm + LeakyReLU of the convolution output (modules.py:141-145, 205-207)
  y = e->mat(q.n, W.out);
  ENG_ALLOC(y.p);
  float* gam = vecp(r, norm_name + ".norm.weight");
  float* bet = vecp(r, norm_name + ".norm.bias");
  if (!gam || !bet) {
    set_error("rdm_engine: missing parameter %s.norm.*", norm_name.c_str());
    return RDM_ERR_ARG;
  }
  return group_norm_finish(gn_partial, gn_blocks, conv.p, q.n, W.out, conv.ld, r.groups, gam, bet, 1e-5f, nullptr, 0, 2, y.p,
                           y.ld, nullptr, static_cast<char*>(r.ws) + gemm_ws + stat_bytes, r.ws_bytes - gemm_ws - stat_bytes,
                           r.st);
}

// Everything of an attention layer after softmax(QK^T)V: output projection, residual LayerNorm, FFN,
// residual LayerNorm (thdroformer.py:142-173 / vanilla_transformer.py:69-103, output_layer.py:6-21).
// `out` is a pre-allocated view (rows of the stacked [ref; src] state).
// LayerNorm(Linear(x) + residual): one fused launch at the transformer width (rdm_linear_layer_norm), else two
int linear_ln(Run& r, const std::string& lin, const std::string& norm, const Mat& x, const Mat& res, Mat& y, bool alloc_out) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  auto it = e->params->lin.find(lin);
  if (it == e->params->lin.end()) {
    set_error("rdm_engine: missing parameter %s", lin.c_str());
    return RDM_ERR_ARG;
  }
  const Linear& L = it->second;
  if (L.wt && L.out == 128) {
    if (alloc_out) {
      y = e->mat(x.rows, L.out);
      ENG_ALLOC(y.p);
    }
    return rdm_linear_layer_norm(x.p, x.ld, L.wt, L.kpad, L.bias, x.rows, L.out, L.kpad, res.p, res.ld, vecp(r, norm + ".weight"),
                                 vecp(r, norm + ".bias"), 1e-5f, 0, y.p, y.ld, r.st);
  }
  Mat h;
  ENG_CHECK(linear(r, lin, x, h));
  return layer_norm(r, norm, h, &res, 0, y, alloc_out);
}

int attention_tail(Run& r, const std::string& p, const Mat& hid, const Mat& x, Mat out) {
  lockstep_align();  // (lockstep.h: the pairs of a lock-step group start every layer together)
  rdm_engine* e = r.e;
  auto lo = e->params->lin.find(p + ".attention.linear"), l1 = e->params->lin.find(p + ".output.expand"), l2 = e->params->lin.find(p + ".output.squeeze");
  if (lo != e->params->lin.end() && l1 != e->params->lin.end() && l2 != e->params->lin.end() && lo->second.wt && l1->second.wt && l2->second.wt &&
      lo->second.out == 128 && lo->second.kpad == 128 && l1->second.out == 256 && l1->second.kpad == 128 &&
      l2->second.out == 128 && l2->second.kpad == 256) {
    const Linear &Lo = lo->second, &L1 = l1->second, &L2 = l2->second;  // the whole tail in one launch
    auto pk = e->params->vec.find(p + ".__tail_packed");
    static const bool unpacked = ::rdm::dev_knob("RDM_TAIL_UNPACKED") != nullptr;  // developer knob (A/B): the checkpoint layout
    if (pk != e->params->vec.end() && !unpacked)
      return rdm_attention_tail_packed(hid.p, hid.ld, x.p, x.ld, hid.rows, 128, pk->second, Lo.bias, vecp(r, p + ".attention.norm.weight"),
                                       vecp(r, p + ".attention.norm.bias"), L1.bias, L2.bias, vecp(r, p + ".output.norm.weight"),
                                       vecp(r, p + ".output.norm.bias"), 1e-5f, out.p, out.ld, r.st);
    return rdm_attention_tail(hid.p, hid.ld, x.p, x.ld, hid.rows, 128, Lo.wt, Lo.kpad, Lo.bias, vecp(r, p + ".attention.norm.weight"),
                              vecp(r, p + ".attention.norm.bias"), L1.wt, L1.kpad, L1.bias, L2.wt, L2.kpad, L2.bias,
                              vecp(r, p + ".output.norm.weight"), vecp(r, p + ".output.norm.bias"), 1e-5f, out.p, out.ld, r.st);
  }
  Mat y, z1;
  ENG_CHECK(linear_ln(r, p + ".attention.linear", p + ".attention.norm", hid, x, y, true));
  ENG_CHECK(linear(r, p + ".output.expand", y, z1, 1));
  return linear_ln(r, p + ".output.squeeze", p + ".output.norm", z1, y, out, false);
}

// rdmnet/thdroformer/thdroformer.py:266-347 on the STACKED [ref; src] rows: every op whose weights
// are shared by both clouds (embedding, in/out projections, and in self layers the q|k|v projection,
// rotary embedding and the whole tail) runs once on all rows; only the attention itself is per cloud.
// Cross layers keep the reference's order: src attends to the UPDATED ref features (:244-245).
// `after_two` (optional) is called once two layers have been enqueued (latency mode: the host then has enough of a lead on the
// caller's stream to enqueue the side stream's launches without stalling it).
int thdroformer(Run& r, const std::string& name, const Mat& pts4, const Mat& x, int64_t n0, int num_layers, Mat out,
                const std::function<int()>* after_two = nullptr) {
  rdm_engine* e = r.e;
  const int heads = e->cfg.num_heads;
  const int64_t N = x.rows, n1 = N - n0;
  Mat emb, f;
  ENG_CHECK(linear(r, name + ".embedding.proj", pts4, emb));
  ENG_CHECK(linear(r, name + ".in_proj", x, f));
  const int64_t d = f.cols;
  const int hd = static_cast<int>(d / heads);
  const auto attend = e->cfg.attention_bf16 ? rdm_attention_bf16 : rdm_attention;
  for (int i = 0; i < 2 * num_layers; ++i) {
    const std::string p = name + ".transformer.layers." + std::to_string(i);
    Mat fnew = e->mat(N, d), hid = e->mat(N, d);
    ENG_ALLOC(fnew.p); ENG_ALLOC(hid.p);
    if (i % 2 == 0) {
      Mat qkv;
      ENG_CHECK(linear(r, p + ".qkv", f, qkv));
      Mat q = qkv.cols_from(0, d), k = qkv.cols_from(d, d), v = qkv.cols_from(2 * d, d);
      ENG_CHECK(rdm_rope(q.p, q.ld, k.p, k.ld, emb.p, emb.ld, N, d, r.st));
      ENG_CHECK(rdm_attention_self_pair(q.p, q.ld, k.p, k.ld, v.p, v.ld, hid.p, hid.ld, n0, n1, heads, hd,
                                        e->cfg.attention_bf16 ? 1 : 0, r.st));  // both clouds, one launch
      ENG_CHECK(attention_tail(r, p, hid, f, fnew));
    } else {
      Mat q, kv1, kv0;
      ENG_CHECK(linear_pair(r, p + ".q", f, q, p + ".kv", f.rows_from(n0, n1), kv1));  // independent: one launch
      ENG_CHECK(attend(q.p, q.ld, kv1.p, kv1.ld, kv1.p + d, kv1.ld, hid.p, hid.ld, n0, n1, heads, hd, r.st));
      ENG_CHECK(attention_tail(r, p, hid.rows_from(0, n0), f.rows_from(0, n0), fnew.rows_from(0, n0)));
      ENG_CHECK(linear(r, p + ".kv", fnew.rows_from(0, n0), kv0));
      ENG_CHECK(attend(q.p + n0 * q.ld, q.ld, kv0.p, kv0.ld, kv0.p + d, kv0.ld, hid.p + n0 * hid.ld, hid.ld, n1, n0, heads,
                              hd, r.st));
      ENG_CHECK(attention_tail(r, p, hid.rows_from(n0, n1), f.rows_from(n0, n1), fnew.rows_from(n0, n1)));
    }
    f = fnew;
    if (after_two && i == 1) ENG_CHECK((*after_two)());
  }
  return linear(r, name + ".out_proj", f, out, 0, false);
}

// [n,3] -> [n,4] zero padded
__device__ __forceinline__ void pad_points_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* p, int64_t n, float* out) {
  (void)blockIdx; (void)gridDim;
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  out[4 * i] = p[3 * i];
  out[4 * i + 1] = p[3 * i + 1];
  out[4 * i + 2] = p[3 * i + 2];
  out[4 * i + 3] = 0.f;
}
__global__ void pad_points_kernel(const float* p, int64_t n, float* out) { pad_points_kernel_body(blockIdx, gridDim, p, n, out); }

__global__ void fill_kernel(float* p, int64_t n, float v) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) p[i] = v;
}
// features = 1 (dataset.py:187-188) and the "row sum > 0" flag the first KPConv needs (kpconv.py:113-114), one launch
__device__ __forceinline__ void unit_features_kernel_body(const dim3 blockIdx, const dim3 gridDim, float* x, int64_t n, int64_t ld, uint8_t* positive) {
  (void)blockIdx; (void)gridDim;
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  for (int64_t c = 0; c < ld; ++c) x[i * ld + c] = 1.0f;  // (the pad columns as the separate fill wrote them)
  positive[i] = 1;
}
__global__ void unit_features_kernel(float* x, int64_t n, int64_t ld, uint8_t* positive) { unit_features_kernel_body(blockIdx, gridDim, x, n, ld, positive); }

// The NMS survivors' rows (vote.py:36-40 boolean-mask selects; model_infer.py:206-216): nodes, their zero-padded [n,4]
// copy for the positional Linear, features and the (n2p, n2n) score pairs -- one launch instead of seven gathers.
struct SelectNodesArgs {
  const int32_t* order;   // compacted indices: ref survivors from 0, src survivors from nc_ref
  int nc_ref, m_r, m_n;
  const float *xyz, *feats, *n2p, *n2n;
  int d, ldf, ldo;
  float *nodes, *nodes4, *out_feats, *scores;
};
__device__ __forceinline__ void select_nodes_kernel_body(const dim3 blockIdx, const dim3 gridDim, SelectNodesArgs a) {
  (void)blockIdx; (void)gridDim;
  const int j = blockIdx.x;
  const int src = j < a.m_r ? a.order[j] : a.order[a.nc_ref + (j - a.m_r)];
  const int t = threadIdx.x;
  if (t < 3) {
    const float v = a.xyz[3 * src + t];
    a.nodes[3 * j + t] = v;
    a.nodes4[4 * j + t] = v;
  } else if (t == 3) {
    a.nodes4[4 * j + 3] = 0.f;
  } else if (t == 4) {
    if (a.n2p) a.scores[2 * j] = a.n2p[src];
  } else if (t == 5) {
    if (a.n2n) a.scores[2 * j + 1] = a.n2n[src];
  }
  for (int c = t; c < a.ldo; c += 64) a.out_feats[static_cast<int64_t>(j) * a.ldo + c] = c < a.d ? a.feats[static_cast<int64_t>(src) * a.ldf + c] : 0.f;
}
__global__ __launch_bounds__(64) void select_nodes_kernel(SelectNodesArgs a) { select_nodes_kernel_body(blockIdx, gridDim, a); }

__device__ __forceinline__ void concat_points_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* a, int64_t na, const float* b, int64_t nb, float* out, int64_t* lengths) {
  (void)blockIdx; (void)gridDim;
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i == 0) {
    lengths[0] = na;
    lengths[1] = nb;
  }
  if (i < 3 * na) out[i] = a[i];
  else if (i < 3 * (na + nb)) out[i] = b[i - 3 * na];
}
__global__ void concat_points_kernel(const float* a, int64_t na, const float* b, int64_t nb, float* out, int64_t* lengths) { concat_points_kernel_body(blockIdx, gridDim, a, na, b, nb, out, lengths); }


// The host half of a pair's result in one launch: pose + counters (19 words) and the first n correspondences
// ([ref points 3n | src points 3n | scores n]) go straight into the mapped pinned buffer.
__device__ __forceinline__ void export_result_kernel_body(const dim3 blockIdx, const dim3 gridDim, const float* T, const float* rc, const float* sc, const float* cs,
                                                            uint32_t* head, float* corr, int cap) {
  (void)blockIdx; (void)gridDim;
  const int n = min(reinterpret_cast<const int32_t*>(T)[16], cap);
  const int tid = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
  if (tid < 19) head[tid] = reinterpret_cast<const uint32_t*>(T)[tid];
  for (int i = tid; i < 7 * n; i += nt) corr[i] = i < 3 * n ? rc[i] : (i < 6 * n ? sc[i - 3 * n] : cs[i - 6 * n]);
}
__global__ __launch_bounds__(256) void export_result_kernel(const float* T, const float* rc, const float* sc, const float* cs,
                                                            uint32_t* head, float* corr, int cap) { export_result_kernel_body(blockIdx, gridDim, T, rc, sc, cs, head, corr, cap); }


// dst = OR of the status words of n {max count, status} pairs (one wavefront)
__device__ __forceinline__ void or_status_kernel_body(const dim3 blockIdx, const dim3 gridDim, const int32_t* pairs, int n, int32_t* dst) {
  (void)blockIdx; (void)gridDim;
  int v = 0;
  for (int i = threadIdx.x; i < n; i += 64) v |= pairs[2 * i + 1];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  if (threadIdx.x == 0) *dst = v;
}
__global__ __launch_bounds__(64) void or_status_kernel(const int32_t* pairs, int n, int32_t* dst) { or_status_kernel_body(blockIdx, gridDim, pairs, n, dst); }


template <auto Body, auto Kernel, typename... A>
int launch1d(const char* what, int64_t n, hipStream_t st, A... args) {
  if (n <= 0) return RDM_OK;
  ::rdm::launch<Body, Kernel, 256>(dim3(static_cast<unsigned>(ceil_div<int64_t>(n, 256))), 0, st, args...);
  return launch_status(what);
}

// Wait for the engine's stream at a size read-back.  The runtime's synchronize busy-waits: with several pairs in
// flight per GPU and 8 ranks per node that is 32 spinning cores; hosts with a small CPU quota poll instead.
int wait_on(rdm_engine* e, hipStream_t st) {
  if (lockstep_active()) {  // a pair of a lock-step group: parked until every pair of the group waits; the group waits once
    lockstep_sync();
    return RDM_OK;
  }
  if (e->wait_sleep_us <= 0) {
    RDM_HIP_CHECK(hipStreamSynchronize(st));
    return RDM_OK;
  }
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return RDM_OK;
    if (q != hipErrorNotReady) {
      set_error("hipStreamQuery failed: %s", hipGetErrorString(q));
      return RDM_ERR_HIP;
    }
    timespec ts{0, static_cast<long>(e->wait_sleep_us) * 1000};
    nanosleep(&ts, nullptr);
  }
}
int wait_stream(Run& r) { return wait_on(r.e, r.st); }

// The engine's side stream (latency mode).  Measured and dropped (tools/overlap_probe.py): a CU mask that keeps the side stream's
// wide launches off every n-th CU, and a second masked stream for the chains beside it -- the one-workgroup kernels slow down
// by a third whenever the rest of the GPU is busy, whichever CUs they run on.
// The side stream has to sit on another hardware pipe than the caller's stream: the runtime hands its hardware queues to the chip's
// four compute pipes in creation order, and two queues of one pipe are served in turns -- their kernels never run side by side
// (a process with four worker streams puts the sixth queue on the pipe of the second: measured, tools/overlap_probe.py).  There
// is no query for that, so the engine tries it: two 60 us spin kernels, one per stream, take 60 us together or 120.
__global__ void spin_kernel(long long ticks) {  // 100 MHz wall clock; the pass count bounds it should the clock ever stand still
  const long long t0 = wall_clock64();
  for (int pass = 0; pass < (1 << 20) && wall_clock64() - t0 < ticks; ++pass) __builtin_amdgcn_s_sleep(8);
}
bool runs_beside(hipStream_t a, hipStream_t b) {
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 6000ll);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 6000ll);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
    best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  return best < 100.0;  // (60 us + launch and wake-up when concurrent, 120+ when serialised)
}
// -> whether latency mode can be used with `st` as the caller's stream (decided once per caller stream and engine)
int ensure_side(rdm_engine* e, hipStream_t st, bool* usable) {
  auto known = e->side_ok.find(st);
  if (known != e->side_ok.end()) {
    *usable = known->second;
    return RDM_OK;
  }
  if (e->side_ok.size() >= 64) e->side_ok.clear();  // (a caller that keeps creating streams)
  if (e->side && runs_beside(st, e->side)) {
    *usable = e->side_ok[st] = true;
    return RDM_OK;
  }
  bool have_user = false;  // is the present side stream already serving another caller stream?
  for (const auto& kv : e->side_ok) have_user |= kv.second;
  if (have_user) {
    *usable = e->side_ok[st] = false;
    return RDM_OK;
  }
  std::vector<hipStream_t> rejected;
  if (e->side) rejected.push_back(e->side);
  e->side = nullptr;
  for (int attempt = 0; attempt < 6 && !e->side; ++attempt) {
    hipStream_t cand = nullptr;
    RDM_HIP_CHECK(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
    if (runs_beside(st, cand)) e->side = cand;
    else rejected.push_back(cand);
  }
  for (hipStream_t r : rejected) (void)hipStreamDestroy(r);  // (only now: a destroyed stream's queue would be handed out again)
  *usable = e->side_ok[st] = e->side != nullptr;
  return RDM_OK;
}
// A run that returns early (an error between fork and join) must not leave side-stream work behind: the next run reuses the arena.
struct SideGuard {
  rdm_engine* e;
  int pending = 0;
  ~SideGuard() {
    if (pending > 0 && e->side) (void)hipStreamSynchronize(e->side);
  }
};

int d2h(Run& r, const void* dev, size_t bytes, void* host_dst) {
  // a kernel stores straight into the mapped pinned buffer (no runtime copy operation on the stream)
  copy_words(dev, r.e->pinned_dev, static_cast<int>((bytes + 3) / 4), r.st);
  ENG_CHECK(wait_stream(r));
  std::memcpy(host_dst, r.e->pinned, bytes);
  return RDM_OK;
}

}  // namespace

extern "C" int rdm_engine_create(const rdm_engine_config* cfg, rdm_engine** out) {
  RDM_REQUIRE(cfg && out, "rdm_engine_create: null pointer");
  RDM_REQUIRE(cfg->num_stages == 5 && cfg->kernel_size == 15, "rdm_engine_create: only 5 stages / 15 kernel points");
  int device = -1;
  RDM_HIP_CHECK(hipGetDevice(&device));
  rdm_engine* e = new rdm_engine();
  e->cfg = *cfg;
  e->device = device;
  e->params->device = device;
  e->arena_cap = cfg->arena_bytes ? cfg->arena_bytes : (size_t(3) << 30);
  e->arena_fixed = cfg->arena_bytes != 0;
  hipError_t err = hipMalloc(reinterpret_cast<void**>(&e->arena), e->arena_cap);
  if (err != hipSuccess) {
    set_error("rdm_engine_create: hipMalloc(%zu) failed: %s", e->arena_cap, hipGetErrorString(err));
    delete e;
    return RDM_ERR_HIP;
  }
  // mapped pinned memory: 4 KB for the size / pose read-backs, then the correspondences of a run ([3n | 3n | n] floats, at
  // most num_correspondences x 2 x points_in_patch of them: local_global_registration.py:145-202) -- the host half of a
  // pair's result (transform + correspondences on the host) without a copy operation on the stream
  e->host_corr_cap = static_cast<int64_t>(std::max(cfg->num_correspondences, 1)) * 2 * std::max(cfg->points_in_patch, 1);
  err = hipHostMalloc(&e->pinned, 4096 + static_cast<size_t>(e->host_corr_cap) * 7 * sizeof(float), hipHostMallocMapped);
  if (err == hipSuccess) err = hipHostGetDevicePointer(&e->pinned_dev, e->pinned, 0);
  if (err != hipSuccess) {
    set_error("rdm_engine_create: hipHostMalloc failed: %s", hipGetErrorString(err));
    (void)hipFree(e->arena);
    delete e;
    return RDM_ERR_HIP;
  }
  *out = e;
  return RDM_OK;
}

extern "C" void rdm_engine_destroy(rdm_engine* e) {
  if (!e) return;
  e->params.reset();  // (the device parameters go with the LAST engine that uses them)
  for (auto& ev : e->events) (void)hipEventDestroy(ev);
  if (e->side) {
    (void)hipStreamSynchronize(e->side);
    (void)hipStreamDestroy(e->side);
  }
  if (e->arena) (void)hipFree(e->arena);
  if (e->pinned) (void)hipHostFree(e->pinned);
  delete e;
}

extern "C" int rdm_engine_set_param(rdm_engine* e, const char* name, const float* data_host, const int64_t* shape_host,
                                    int ndim) {
  RDM_REQUIRE(e && name && data_host && ndim >= 0 && ndim <= 4, "rdm_engine_set_param: bad arguments");
  HostParam p;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    p.shape.push_back(shape_host[i]);
    n *= shape_host[i];
  }
  p.data.assign(data_host, data_host + n);
  e->host[name] = std::move(p);
  e->finalized = false;
  return RDM_OK;
}

namespace {
int upload(rdm_engine* e, const std::vector<float>& h, float** dev) {
  void* p = nullptr;
  RDM_HIP_CHECK(hipMalloc(&p, (h.size() + 4) * sizeof(float)));
  RDM_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  e->params->owned.push_back(p);
  *dev = static_cast<float*>(p);
  return RDM_OK;
}
// nn.Linear [out,in] (+ optional further [out_i,in] blocks stacked along out) -> B [pad4(in), pad4(out)]
int make_linear(rdm_engine* e, const std::string& key, const std::vector<const HostParam*>& ws,
                const std::vector<const HostParam*>& bs) {
  const int64_t in = ws[0]->shape[1];
  int64_t out = 0;
  for (auto* w : ws) out += w->shape[0];
  Linear L;
  L.in = in; L.out = out; L.kpad = pad4(in); L.ldb = pad4(out);
  std::vector<float> b(static_cast<size_t>(L.kpad) * L.ldb, 0.f), bias;
  int64_t o0 = 0;
  for (auto* w : ws) {
    for (int64_t o = 0; o < w->shape[0]; ++o)
      for (int64_t i = 0; i < in; ++i) b[i * L.ldb + o0 + o] = w->data[o * in + i];
    o0 += w->shape[0];
  }
  for (auto* bb : bs) bias.insert(bias.end(), bb->data.begin(), bb->data.end());
  ENG_CHECK(upload(e, b, &L.b));
  ENG_CHECK(upload(e, bias, &L.bias));
  if (out % 128 == 0 && out <= 512 && L.kpad % 16 == 0) {  // checkpoint layout ([out, in], stacked blocks below each other)
    std::vector<float> wt(static_cast<size_t>(out) * L.kpad, 0.f);
    int64_t r0 = 0;
    for (auto* w : ws) {
      for (int64_t o = 0; o < w->shape[0]; ++o)
        for (int64_t i = 0; i < in; ++i) wt[(r0 + o) * L.kpad + i] = w->data[o * in + i];
      r0 += w->shape[0];
    }
    ENG_CHECK(upload(e, wt, &L.wt));
  }
  e->params->lin[key] = L;
  return RDM_OK;
}
bool ends_with(const std::string& s, const char* suf) {
  const size_t n = std::strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
}  // namespace

extern "C" int rdm_engine_finalize(rdm_engine* e) {
  RDM_REQUIRE(e, "rdm_engine_finalize: null engine");
  // a fresh set: engines that share the previous one (rdm_engine_share_params) keep it alive and unchanged
  e->params = std::make_shared<ParamSet>();
  e->params->device = e->device;
  for (auto& kv : e->host) {
    const std::string& name = kv.first;
    const HostParam& p = kv.second;
    if (ends_with(name, "KPConv.weights")) {
      const int64_t k = p.shape[0], cin = p.shape[1], cout = p.shape[2];
      Linear L;
      L.in = k * cin; L.out = cout; L.kpad = pad4(cin == 1 ? 16 : k * cin); L.ldb = pad4(cout);
      std::vector<float> b(static_cast<size_t>(L.kpad) * L.ldb, 0.f);
      for (int64_t r = 0; r < k * cin; ++r)
        for (int64_t c = 0; c < cout; ++c) b[r * L.ldb + c] = p.data[r * cout + c];
      ENG_CHECK(upload(e, b, &L.b));
      auto bi = e->host.find(name.substr(0, name.size() - 8) + ".bias");
      RDM_REQUIRE(bi != e->host.end(), "rdm_engine_finalize: %s has no bias", name.c_str());
      ENG_CHECK(upload(e, bi->second.data, &L.bias));
      if (rdm_kpconv_fused_supported(cin, cout)) {  // fine levels: gather + weight contraction in one kernel
        std::vector<float> pk(rdm_kpconv_packed_floats(cin, cout));
        ENG_CHECK(rdm_kpconv_pack_weights(p.data.data(), cin, cout, pk.data()));
        ENG_CHECK(upload(e, pk, &L.packed));
      }
      e->params->lin[name] = L;
    } else if (ends_with(name, ".weight") && p.shape.size() == 2) {
      const std::string base = name.substr(0, name.size() - 7);
      auto bi = e->host.find(base + ".bias");
      RDM_REQUIRE(bi != e->host.end(), "rdm_engine_finalize: %s has no bias", name.c_str());
      ENG_CHECK(make_linear(e, base, {&p}, {&bi->second}));
      if (ends_with(base, ".attention.attention.proj_q")) {  // fused q|k|v and k|v projections
        const std::string a = base.substr(0, base.size() - 7);                    // ...attention.attention
        const std::string layer = a.substr(0, a.size() - std::strlen(".attention.attention"));
        auto W = [&](const char* n) { return &e->host.at(a + "." + n + ".weight"); };
        auto B = [&](const char* n) { return &e->host.at(a + "." + n + ".bias"); };
        ENG_CHECK(make_linear(e, layer + ".qkv", {W("proj_q"), W("proj_k"), W("proj_v")}, {B("proj_q"), B("proj_k"), B("proj_v")}));
        ENG_CHECK(make_linear(e, layer + ".q", {W("proj_q")}, {B("proj_q")}));
        ENG_CHECK(make_linear(e, layer + ".kv", {W("proj_k"), W("proj_v")}, {B("proj_k"), B("proj_v")}));
      }
    } else if (!(ends_with(name, ".bias") && e->host.count(name.substr(0, name.size() - 5) + ".weight") &&
                 e->host.at(name.substr(0, name.size() - 5) + ".weight").shape.size() == 2)) {
      float* d = nullptr;
      ENG_CHECK(upload(e, p.data, &d));
      e->params->vec[name] = d;
    }
  }
  {
    float* d = nullptr;
    ENG_CHECK(upload(e, std::vector<float>(static_cast<size_t>(e->cfg.points_in_patch), std::sqrt(static_cast<float>(e->cfg.out_dim))), &d));
    e->params->vec["__sqrt_out_dim"] = d;
  }
  // the attention tails' weights in operand order (rdm_attention_tail_pack_weights: every weight load of the kernel one contiguous KB)
  {
    std::vector<std::string> layers;
    for (auto& kv : e->params->lin)
      if (ends_with(kv.first, ".attention.linear")) layers.push_back(kv.first.substr(0, kv.first.size() - std::strlen(".attention.linear")));
    for (const std::string& p : layers) {
      auto lo = e->params->lin.find(p + ".attention.linear"), l1 = e->params->lin.find(p + ".output.expand"), l2 = e->params->lin.find(p + ".output.squeeze");
      if (l1 == e->params->lin.end() || l2 == e->params->lin.end() || !lo->second.wt || !l1->second.wt || !l2->second.wt ||
          lo->second.out != 128 || lo->second.kpad != 128 || l1->second.out != 256 || l1->second.kpad != 128 || l2->second.out != 128 || l2->second.kpad != 256)
        continue;
      float* d = nullptr;
      RDM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), rdm_attention_tail_packed_floats() * sizeof(float)));
      e->params->owned.push_back(d);
      ENG_CHECK(rdm_attention_tail_pack_weights(lo->second.wt, lo->second.kpad, l1->second.wt, l1->second.kpad, l2->second.wt, l2->second.kpad, d, nullptr));
      e->params->vec[p + ".__tail_packed"] = d;
    }
    RDM_HIP_CHECK(hipDeviceSynchronize());
  }
  e->finalized = true;
  return RDM_OK;
}

// The prepared parameters of `src` (device copies of the weights in the kernels' layouts, ~250 MB) serve `e` too: engines of
// one process that keep several pairs in flight hold ONE copy instead of one each.  `src` must be finalized, live on the same
// device and outlive `e`; e's own parameters (if any) are released.
extern "C" int rdm_engine_share_params(rdm_engine* e, const rdm_engine* src) {
  RDM_REQUIRE(e && src && e != src, "rdm_engine_share_params: bad arguments");
  RDM_REQUIRE(src->finalized, "rdm_engine_share_params: the source engine is not finalized");
  RDM_REQUIRE(e->device == src->device && src->params->device == src->device,
              "rdm_engine_share_params: the engines live on different devices (%d vs %d)", e->device, src->device);
  // the prepared layouts depend on the model-shape fields of the configuration (packed KPConv / Linear shapes, the
  // points_in_patch entries of the einsum divisor): they must agree
  const rdm_engine_config &a = e->cfg, &b = src->cfg;
  RDM_REQUIRE(a.num_stages == b.num_stages && a.kernel_size == b.kernel_size && a.group_norm == b.group_norm &&
                  a.out_dim == b.out_dim && a.num_heads == b.num_heads && a.num_layers == b.num_layers &&
                  a.num_layers2 == b.num_layers2 && a.vote_mlp_layers == b.vote_mlp_layers &&
                  a.points_in_patch == b.points_in_patch,
              "rdm_engine_share_params: the engines were created with different model shapes");
  e->host.clear();
  e->params = src->params;  // (its previous set is released here unless other engines use it)
  e->finalized = true;
  return RDM_OK;
}

extern "C" int rdm_engine_enable_profile(rdm_engine* e, int enable) {
  RDM_REQUIRE(e, "rdm_engine_enable_profile: null engine");
  RDM_REQUIRE(enable >= 0 && enable <= 2, "rdm_engine_enable_profile: 0 (off), 1 (events) or 2 (layer shapes only)");
  e->profile = enable != 0;
  e->profile_shapes_only = enable == 2;
  if (enable == 1 && e->events.empty()) {
    e->events.resize(3 * 16);
    for (auto& ev : e->events) RDM_HIP_CHECK(hipEventCreate(&ev));
  }
  return RDM_OK;
}

// The activation arena re-allocated at `bytes` and left GROWABLE (rdm_engine_config.arena_bytes fixes the size instead): a caller
// that knows its largest pair -- or the largest lock-step group it will collate on this engine -- sizes the arena once and never
// meets the grow-and-rerun of the first large pair; a small value makes that path testable.  Any collated batch is dropped.
extern "C" int rdm_engine_reserve(rdm_engine* e, size_t bytes) {
  RDM_REQUIRE(e && bytes >= (size_t(1) << 20), "rdm_engine_reserve: at least 1 MiB");
  RDM_HIP_CHECK(hipDeviceSynchronize());  // (runs of this engine may be in flight on any stream)
  e->batch.clear();
  e->arena_base = 0;
  RDM_HIP_CHECK(hipFree(e->arena));
  e->arena = nullptr;
  e->arena_cap = bytes;
  e->arena_fixed = false;
  const hipError_t err = hipMalloc(reinterpret_cast<void**>(&e->arena), e->arena_cap);
  if (err != hipSuccess) {
    e->arena_cap = 0;
    set_error("rdm_engine_reserve: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    return RDM_ERR_HIP;
  }
  return RDM_OK;
}

extern "C" int rdm_engine_set_wait(rdm_engine* e, int sleep_us) {
  RDM_REQUIRE(e && sleep_us >= 0, "rdm_engine_set_wait: bad arguments");
  e->wait_sleep_us = sleep_us;
  return RDM_OK;
}

extern "C" int rdm_engine_set_pairs_in_flight(rdm_engine* e, int n) {
  RDM_REQUIRE(e && n >= 1, "rdm_engine_set_pairs_in_flight: bad arguments");
  e->pairs_in_flight = n;
  return RDM_OK;
}

extern "C" int rdm_engine_set_overlap(rdm_engine* e, int mode) {
  RDM_REQUIRE(e && mode >= 0 && mode <= 2, "rdm_engine_set_overlap: bad arguments");
  e->overlap_mode = mode;
  return RDM_OK;
}

extern "C" int rdm_engine_get_profile(rdm_engine* e, rdm_kpconv_profile* out, int cap) {
  RDM_REQUIRE(e && out && cap >= 0, "rdm_engine_get_profile: bad arguments");
  const int n = std::min<int>(cap, static_cast<int>(e->prof.size()));
  for (int i = 0; i < n; ++i) out[i] = e->prof[i];
  return static_cast<int>(e->prof.size());
}

extern "C" int rdm_engine_keep_taps(rdm_engine* e, int enable) {
  RDM_REQUIRE(e, "rdm_engine_keep_taps: null engine");
  e->keep_taps = enable != 0;
  return RDM_OK;
}

extern "C" int rdm_engine_get_tensor(rdm_engine* e, const char* name, rdm_tensor_view* out) {
  RDM_REQUIRE(e && name && out, "rdm_engine_get_tensor: null pointer");
  auto it = e->taps.find(name);
  if (it == e->taps.end()) {
    set_error("rdm_engine_get_tensor: no tensor named %s (call rdm_engine_keep_taps first)", name);
    return RDM_ERR_ARG;
  }
  *out = it->second;
  return RDM_OK;
}

static int engine_run_once(rdm_engine* e, const float* ref_points, int64_t n_ref, const float* src_points, int64_t n_src,
                           const rdm_data_dict* dd, rdm_engine_result* res, void* stream, const PairPyramid* pre = nullptr);

// The default 3 GiB arena covers pairs of ~2 x 25 k points; denser input (raw scans, KITTI-360-sized clouds)
// grows it: on exhaustion the stream is drained, the arena doubled (288 GB of HBM leave room) and the pair
// re-run.  An arena size chosen by the caller (rdm_engine_config.arena_bytes) is never changed.
static int engine_run_growing(rdm_engine* e, const float* ref_points, int64_t n_ref, const float* src_points, int64_t n_src,
                              const rdm_data_dict* dd, rdm_engine_result* res, void* stream) {
  // with three or more pairs sharing the GPU the tiled GEMM keeps two workgroups per CU (gemm.hip: gemm_set_lds_pad)
  struct PadGuard {
    explicit PadGuard(unsigned b) { rdm::gemm_set_lds_pad(b); }
    ~PadGuard() { rdm::gemm_set_lds_pad(0); }
  } pad_guard(e->pairs_in_flight >= 3 ? 20480u : 0u);
  for (;;) {
    e->arena_exhausted = false;
    const int rc = engine_run_once(e, ref_points, n_ref, src_points, n_src, dd, res, stream);
    if (rc != RDM_ERR_WORKSPACE || !e->arena_exhausted || e->arena_fixed || e->arena_cap >= (size_t(96) << 30)) return rc;
    RDM_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    RDM_HIP_CHECK(hipFree(e->arena));
    e->arena = nullptr;
    e->arena_cap *= 2;
    const hipError_t err = hipMalloc(reinterpret_cast<void**>(&e->arena), e->arena_cap);
    if (err != hipSuccess) {
      e->arena_cap = 0;
      set_error("rdm_engine_run: growing the activation arena failed: %s", hipGetErrorString(err));
      return RDM_ERR_HIP;
    }
  }
}

extern "C" int rdm_engine_run(rdm_engine* e, const float* ref_points, int64_t n_ref, const float* src_points,
                              int64_t n_src, rdm_engine_result* res, void* stream) {
  RDM_REQUIRE(e && ref_points && src_points && res, "rdm_engine_run: null pointer");
  RDM_REQUIRE(e->finalized, "rdm_engine_run: call rdm_engine_finalize first");
  RDM_REQUIRE(n_ref > 0 && n_src > 0, "rdm_engine_run: empty cloud");
  return engine_run_growing(e, ref_points, n_ref, src_points, n_src, nullptr, res, stream);
}

// ---------------------------------------------------------------- batched collate
// The collate of B pairs (data.py:13-77, B times) as ONE sequence of launches.  The collate is the launch-bound part of a pair
// (four chains of one-workgroup-per-cloud subsampling kernels, seven small launches per grid build, two query launches): with
// four pairs in flight it costs 0.25 ms of a pair's 1.9 ms (profiles/r04_marginal_cost.txt), and the batching lab measured the
// class at 0.43x per pair when four pairs share its launches (profiles/r05_batch_lab.md).  Here
//   * every subsampling launch works on 2 B clouds (one workgroup per cloud: the latency of a pair's chain now serves B),
//   * the grids are built as (pair, level) items, eight per set of launches -- each item is exactly the grid of the pair's own
//     collate (its own bounding box, two clouds), so records and tables come out as in a single-pair run,
//   * the 12 searches of every pair are recorded into one queue and flushed 16 at a time (two launches per flush),
//   * the level sizes of all pairs return in one read-back.
// What a pair's forward reads is bit for bit what its own collate would have written (integer tables, the same points, the same
// cell-sorted records): rdm_engine_forward_batched(k) then equals rdm_engine_run on pair k.
static int collate_batch_once(rdm_engine* e, int B, const float* const* refs, const int64_t* n_refs, const float* const* srcs,
                              const int64_t* n_srcs, hipStream_t st) {
  const rdm_engine_config& c = e->cfg;
  e->arena_off = 0;
  e->arena_base = 0;
  e->batch.clear();
  e->taps.clear();
  Run r;
  r.e = e; r.st = st; r.groups = c.group_norm;
  int64_t n_tot = 0;
  for (int p = 0; p < B; ++p) n_tot += n_refs[p] + n_srcs[p];
  const int nc = 2 * B;  // clouds
  r.ws_bytes = std::max<size_t>(rdm_grid_subsample_workspace_bytes(n_tot, nc), size_t(16) << 20);
  r.ws = e->alloc<char>(r.ws_bytes);
  ENG_ALLOC(r.ws);
  // level 0: the clouds stacked [ref_0 src_0 ref_1 src_1 ...]
  float* pts[5];
  int64_t* len[5];
  pts[0] = e->alloc<float>(3 * n_tot);
  len[0] = e->alloc<int64_t>(nc);
  ENG_ALLOC(pts[0]); ENG_ALLOC(len[0]);
  {
    int64_t base = 0;
    for (int p = 0; p < B; ++p) {
      ENG_CHECK(launch1d<concat_points_kernel_body, concat_points_kernel>("concat_points", 3 * (n_refs[p] + n_srcs[p]), st, refs[p], n_refs[p], srcs[p], n_srcs[p],
                         pts[0] + 3 * base, len[0] + 2 * p));
      base += n_refs[p] + n_srcs[p];
    }
  }
  int64_t* all_len = e->alloc<int64_t>(static_cast<size_t>(4) * nc);  // levels 1..4 x clouds, contiguous for one read-back
  ENG_ALLOC(all_len);
  static const int gs_multi_levels = [] { const char* v = ::rdm::dev_knob("RDM_GS_MULTI_LEVELS"); return v ? atoi(v) : 1; }();
  float voxel = c.init_voxel_size;
  for (int i = 1; i < 5; ++i) {
    voxel *= 2.f;  // data.py:23-28
    pts[i] = e->alloc<float>(3 * n_tot);  // (capacity: the level sizes are not known on the host yet; the kernels walk `lengths`)
    len[i] = all_len + static_cast<size_t>(i - 1) * nc;
    ENG_ALLOC(pts[i]);
    ENG_CHECK(grid_subsample_mode(pts[i - 1], n_tot, len[i - 1], nc, voxel, pts[i], len[i], r.ws, r.ws_bytes, st,
                                  i <= gs_multi_levels ? 2 : 1));
  }
  std::vector<int64_t> host_len(static_cast<size_t>(4) * nc);
  RDM_REQUIRE(host_len.size() * sizeof(int64_t) <= 4096, "rdm_engine_collate_batch: too many pairs for the read-back buffer");
  ENG_CHECK(d2h(r, all_len, host_len.size() * sizeof(int64_t), host_len.data()));

  e->batch.resize(B);
  int32_t* flags_all = e->alloc<int32_t>(static_cast<size_t>(64) * B);
  ENG_ALLOC(flags_all);
  fill_words<int32_t>(flags_all, static_cast<int64_t>(64) * B, 0, st);
  // the pyramids: slices of the stacked level arrays
  {
    std::vector<int64_t> off(5, 0);
    for (int p = 0; p < B; ++p) {
      PairPyramid& py = e->batch[p];
      py.flags = flags_all + 64 * p;
      for (int i = 0; i < 5; ++i) {
        const int64_t a = i == 0 ? n_refs[p] : host_len[static_cast<size_t>(i - 1) * nc + 2 * p];
        const int64_t b = i == 0 ? n_srcs[p] : host_len[static_cast<size_t>(i - 1) * nc + 2 * p + 1];
        py.lv[i].pts = pts[i] + 3 * off[i];
        py.lv[i].lengths = len[i] + 2 * p;
        py.lv[i].n = a + b;
        py.lv[i].n_ref = a;
        off[i] += a + b;
      }
    }
  }
  // the (pair, level) grids, eight items per set of launches
  {
    const float* gp[8];
    int64_t gn[8];
    const int64_t* gl[8];
    float gr[8];
    void* gw[8];
    size_t gb[8];
    int k = 0;
    auto flush = [&]() -> int {
      if (k > 0) ENG_CHECK(radius_grid_build_multi(k, gp, gn, gl, 2, gr, gw, gb, st));
      k = 0;
      return RDM_OK;
    };
    for (int p = 0; p < B; ++p) {
      float rad = c.init_radius;
      for (int i = 0; i < 5; ++i, rad *= 2.f) {
        Grid& g = e->batch[p].grids[i];
        g.n_s = e->batch[p].lv[i].n;
        g.bytes = rdm_radius_grid_workspace_bytes(g.n_s);
        g.ws = e->alloc<char>(g.bytes);
        ENG_ALLOC(g.ws);
        gp[k] = e->batch[p].lv[i].pts; gn[k] = g.n_s; gl[k] = e->batch[p].lv[i].lengths; gr[k] = rad; gw[k] = g.ws; gb[k] = g.bytes;
        if (++k == 8) ENG_CHECK(flush());
      }
    }
    ENG_CHECK(flush());
  }
  // the searches of a plain run (engine_run_once: 12 per pair, int32 tables, one column of the up-sampling tables 1..3)
  {
    std::vector<char> queue(radius_redo_queue_bytes());
    radius_redo_queue_reset(queue.data());
    int queued = 0;
    for (int p = 0; p < B; ++p) {
      PairPyramid& py = e->batch[p];
      auto search = [&](const Level& q, const Grid& g, float rad, int limit, Table& t, bool i32) -> int {
        if (queued == 16) {  // (the queue's capacity: radius_neighbors.hip kRedoMax)
          ENG_CHECK(radius_redo_flush(queue.data(), st));
          queued = 0;
        }
        t.rows = q.n; t.width = limit; t.flags = py.flags + 2 * py.calls++;
        t.i32 = i32;
        t.idx = reinterpret_cast<int64_t*>(e->alloc<char>(static_cast<size_t>(q.n > 0 ? q.n : 1) * limit * (i32 ? 4 : 8)));
        ENG_ALLOC(t.idx);
        unsigned char* redo_flags = e->alloc<unsigned char>(static_cast<size_t>(q.n > 0 ? q.n : 1));
        ENG_ALLOC(redo_flags);
        ++queued;
        return radius_grid_query_deferred(g.ws, g.bytes, g.n_s, q.pts, q.n, q.lengths, 2, rad, limit, t.idx, nullptr, t.flags,
                                          t.flags + 1, redo_flags, queue.data(), i32 ? 1 : 0, st);
      };
      float radius = c.init_radius;
      for (int i = 0; i < 5; ++i) {  // (the order of engine_run_once: the status rows are numbered by it)
        ENG_CHECK(search(py.lv[i], py.grids[i], radius, c.neighbor_limits[i], py.nb[i], true));
        if (i < 4) ENG_CHECK(search(py.lv[i + 1], py.grids[i], radius, c.neighbor_limits[i], py.sub[i], true));
        if (i > 1) ENG_CHECK(search(py.lv[i - 1], py.grids[i], radius, 1, py.up[i - 1], false));
        radius *= 2.f;
      }
    }
    ENG_CHECK(radius_redo_flush(queue.data(), st));
  }
  e->arena_base = e->arena_off;
  e->batch_n_ref.assign(n_refs, n_refs + B);
  e->batch_n_src.assign(n_srcs, n_srcs + B);
  return RDM_OK;
}

extern "C" int rdm_engine_collate_batch(rdm_engine* e, int n_pairs, const float* const* ref_points, const int64_t* n_ref,
                                        const float* const* src_points, const int64_t* n_src, void* stream) {
  RDM_REQUIRE(e && ref_points && n_ref && src_points && n_src, "rdm_engine_collate_batch: null pointer");
  RDM_REQUIRE(e->finalized, "rdm_engine_collate_batch: call rdm_engine_finalize first");
  RDM_REQUIRE(n_pairs >= 1 && n_pairs <= 16, "rdm_engine_collate_batch: 1 .. 16 pairs");
  RDM_REQUIRE(!e->keep_taps, "rdm_engine_collate_batch: a run that keeps its stage tensors builds the reference's full tables pair by pair");
  for (int p = 0; p < n_pairs; ++p)
    RDM_REQUIRE(ref_points[p] && src_points[p] && n_ref[p] > 0 && n_src[p] > 0, "rdm_engine_collate_batch: pair %d is empty", p);
  for (;;) {  // (arena growth as in rdm_engine_run: on exhaustion the arena is doubled and the batch collated again)
    e->arena_exhausted = false;
    const int rc = collate_batch_once(e, n_pairs, ref_points, n_ref, src_points, n_src, static_cast<hipStream_t>(stream));
    if (rc != RDM_OK) {
      e->batch.clear();
      e->arena_base = 0;
    }
    if (rc != RDM_ERR_WORKSPACE || !e->arena_exhausted || e->arena_fixed || e->arena_cap >= (size_t(96) << 30)) return rc;
    RDM_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    RDM_HIP_CHECK(hipFree(e->arena));
    e->arena = nullptr;
    e->arena_cap *= 2;
    const hipError_t err = hipMalloc(reinterpret_cast<void**>(&e->arena), e->arena_cap);
    if (err != hipSuccess) {
      e->arena_cap = 0;
      set_error("rdm_engine_collate_batch: growing the activation arena failed: %s", hipGetErrorString(err));
      return RDM_ERR_HIP;
    }
  }
}

// RDMNet.forward of pair `k` of the collated batch; the result structure, rdm_engine_transform-style read-outs and the host
// views are those of rdm_engine_run and stay valid until the next call on this engine.  RDM_ERR_WORKSPACE: the arena is too
// small for this pair's forward above the batch's pyramids (the caller collates a smaller batch, or runs the pair alone:
// rdm_engine_run grows the arena).
extern "C" int rdm_engine_forward_batched(rdm_engine* e, int k, rdm_engine_result* res, void* stream) {
  RDM_REQUIRE(e && res, "rdm_engine_forward_batched: null pointer");
  RDM_REQUIRE(k >= 0 && k < static_cast<int>(e->batch.size()), "rdm_engine_forward_batched: no collated pair %d (rdm_engine_collate_batch first)", k);
  struct PadGuard {
    explicit PadGuard(unsigned b) { rdm::gemm_set_lds_pad(b); }
    ~PadGuard() { rdm::gemm_set_lds_pad(0); }
  } pad_guard(e->pairs_in_flight >= 3 ? 20480u : 0u);
  e->arena_exhausted = false;
  return engine_run_once(e, nullptr, e->batch_n_ref[k], nullptr, e->batch_n_src[k], nullptr, res, stream, &e->batch[k]);
}

// ---------------------------------------------------------------- lock step: several pairs, one stream, grouped launches
// rdm_engine_run of n pairs on n engines (one arena and one result buffer each, the weights shared) on ONE stream, in lock step
// (lockstep.h): the runs are contexts of the calling thread; launches of converted kernels are recorded and the same kernel of
// all pairs goes out as one grouped launch; the four size read-backs of the pairs become four waits of the group.  Every pair gets
// the bits of rdm_engine_run on it alone (same kernel bodies, arguments and grids).
namespace {
struct LockstepJob {
  rdm_engine* const* engines;
  const float* const* refs;
  const int64_t* n_refs;
  const float* const* srcs;
  const int64_t* n_srcs;
  rdm_engine_result* const* results;
  void* stream;
  bool collated;  // the pyramids of all pairs lie in engines[0]'s arena (rdm_engine_collate_batch)
  const rdm_data_dict* const* dds;  // rdm_engine_forward_lockstep: the callers' data_dicts (the forwards alone), else null
  bool collate_only = false;        // rdm_engine_collate_lockstep: the collates alone (full tables kept as stage tensors)
};
int lockstep_pair(int k, void* user) {
  const LockstepJob& j = *static_cast<const LockstepJob*>(user);
  rdm_engine* e = j.engines[k];
  e->arena_exhausted = false;
  if (j.collate_only) {  // (as rdm_engine_collate: the stage tensors of THIS call are kept whatever rdm_engine_keep_taps says)
    const bool keep_before = e->keep_taps;
    e->keep_taps = true;
    e->collate_only = true;
    const int rc = engine_run_once(e, j.refs[k], j.n_refs[k], j.srcs[k], j.n_srcs[k], nullptr, j.results[k], j.stream);
    e->collate_only = false;
    e->keep_taps = keep_before;
    return rc;
  }
  if (j.dds) {
    const rdm_data_dict* dd = j.dds[k];
    return engine_run_once(e, nullptr, dd->n_ref[0], nullptr, dd->n_points[0] - dd->n_ref[0], dd, j.results[k], j.stream);
  }
  if (j.collated) {  // (engines 1.. read the pyramid in engines[0]'s arena and keep their activations in their own)
    if (k > 0) {
      e->batch.clear();
      e->arena_base = 0;
    }
    return engine_run_once(e, nullptr, j.n_refs[k], nullptr, j.n_srcs[k], nullptr, j.results[k], j.stream, &j.engines[0]->batch[k]);
  }
  return engine_run_once(e, j.refs[k], j.n_refs[k], j.srcs[k], j.n_srcs[k], nullptr, j.results[k], j.stream);
}
int lockstep_wait(hipStream_t st, void* user) {
  rdm_engine* e = static_cast<rdm_engine*>(user);
  if (e->wait_sleep_us <= 0) {
    RDM_HIP_CHECK(hipStreamSynchronize(st));
    return RDM_OK;
  }
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return RDM_OK;
    if (q != hipErrorNotReady) {
      set_error("hipStreamQuery failed: %s", hipGetErrorString(q));
      return RDM_ERR_HIP;
    }
    timespec ts{0, static_cast<long>(e->wait_sleep_us) * 1000};
    nanosleep(&ts, nullptr);
  }
}
}  // namespace

extern "C" int rdm_engine_run_lockstep(rdm_engine* const* engines, int n_pairs, const float* const* ref_points, const int64_t* n_ref,
                                       const float* const* src_points, const int64_t* n_src, rdm_engine_result* const* results,
                                       int collate_batched, void* stream) {
  RDM_REQUIRE(engines && ref_points && n_ref && src_points && n_src && results, "rdm_engine_run_lockstep: null pointer");
  RDM_REQUIRE(n_pairs >= 1 && n_pairs <= kGroupMax, "rdm_engine_run_lockstep: 1 .. %d pairs", kGroupMax);
  for (int k = 0; k < n_pairs; ++k) {
    RDM_REQUIRE(engines[k] && results[k] && ref_points[k] && src_points[k] && n_ref[k] > 0 && n_src[k] > 0,
                "rdm_engine_run_lockstep: pair %d is incomplete", k);
    RDM_REQUIRE(engines[k]->finalized, "rdm_engine_run_lockstep: engine %d is not finalized", k);
    for (int j = 0; j < k; ++j) RDM_REQUIRE(engines[j] != engines[k], "rdm_engine_run_lockstep: every pair needs an engine of its own");
  }
  struct PadGuard {
    explicit PadGuard(unsigned b) { rdm::gemm_set_lds_pad(b); }
    ~PadGuard() { rdm::gemm_set_lds_pad(0); }
  } pad_guard(0u);  // (no GEMM residency cap: a grouped launch is wide enough to want all four workgroups per CU -- 633 against 627 pairs/s at 4 x 4)
  // the collates of all pairs as one launch sequence on engines[0] (exact, tests/test_engine_gpu.py), then the forwards in lock step
  bool any_taps = false;  // (an engine that keeps its stage tensors builds the reference's full tables: it collates its own pair)
  for (int k = 0; k < n_pairs; ++k) any_taps |= engines[k]->keep_taps;
  const bool collated = collate_batched != 0 && n_pairs > 1 && !any_taps;
  if (collated) ENG_CHECK(rdm_engine_collate_batch(engines[0], n_pairs, ref_points, n_ref, src_points, n_src, stream));
  LockstepJob job{engines, ref_points, n_ref, src_points, n_src, results, stream, collated, nullptr};
  int rcs[kGroupMax] = {};
  const int wrc = lockstep_run(n_pairs, lockstep_pair, &job, static_cast<hipStream_t>(stream), lockstep_wait, engines[0], rcs);
  if (wrc == -1) {
    set_error("rdm_engine_run_lockstep: called from inside a lock-step group (a host thread runs one group at a time)");
    return RDM_ERR_ARG;
  }
  if (wrc == -3) {
    set_error("rdm_engine_run_lockstep: no memory for the context stacks");
    return RDM_ERR_HIP;
  }
  if (wrc != 0) return wrc;  // (the group's host wait failed: the error text is the wait's)
  // a pair that exhausted its arena runs again on its own: rdm_engine_run grows the arena (engines[0] last: that rewrites the
  // arena the other pairs' pyramids lie in -- they are done by then)
  for (int k = n_pairs - 1; k >= 0; --k) {
    if (rcs[k] == RDM_ERR_WORKSPACE && engines[k]->arena_exhausted && !engines[k]->arena_fixed)
      rcs[k] = engine_run_growing(engines[k], ref_points[k], n_ref[k], src_points[k], n_src[k], nullptr, results[k], stream);
  }
  for (int k = 0; k < n_pairs; ++k)
    if (rcs[k] != RDM_OK) return rcs[k];
  return RDM_OK;
}

// RDMNet.forward (model_infer.py:109-354) of n callers' data_dicts on n engines in lock step: rdm_engine_forward per pair, the
// launches grouped as in rdm_engine_run_lockstep.  Every pair: the bits (and stage tensors, with rdm_engine_keep_taps) of
// rdm_engine_forward on it alone.
static int check_data_dict(const rdm_data_dict* dd, const char* who) {
  for (int i = 0; i < 5; ++i) {
    RDM_REQUIRE(dd->points[i] && dd->lengths[i] && dd->neighbors[i] && dd->n_points[i] > 1 && dd->n_ref[i] > 0 &&
                    dd->n_ref[i] < dd->n_points[i] && dd->neighbors_width[i] > 0 && dd->neighbors_ld[i] >= dd->neighbors_width[i],
                "%s: level %d of the data_dict is incomplete", who, i);
    if (i < 4)
      RDM_REQUIRE(dd->subsampling[i] && dd->upsampling[i] && dd->subsampling_width[i] > 0 && dd->upsampling_width[i] > 0 &&
                      dd->subsampling_ld[i] >= dd->subsampling_width[i] && dd->upsampling_ld[i] >= dd->upsampling_width[i],
                  "%s: level %d of the data_dict is incomplete", who, i);
  }
  RDM_REQUIRE(dd->features && dd->features_ld >= 1, "%s: features missing", who);
  return RDM_OK;
}

extern "C" int rdm_engine_forward_lockstep(rdm_engine* const* engines, int n_pairs, const rdm_data_dict* const* data,
                                           rdm_engine_result* const* results, void* stream) {
  RDM_REQUIRE(engines && data && results, "rdm_engine_forward_lockstep: null pointer");
  RDM_REQUIRE(n_pairs >= 1 && n_pairs <= kGroupMax, "rdm_engine_forward_lockstep: 1 .. %d pairs", kGroupMax);
  for (int k = 0; k < n_pairs; ++k) {
    RDM_REQUIRE(engines[k] && results[k] && data[k], "rdm_engine_forward_lockstep: pair %d is incomplete", k);
    RDM_REQUIRE(engines[k]->finalized, "rdm_engine_forward_lockstep: engine %d is not finalized", k);
    for (int j = 0; j < k; ++j) RDM_REQUIRE(engines[j] != engines[k], "rdm_engine_forward_lockstep: every pair needs an engine of its own");
    if (int rc = check_data_dict(data[k], "rdm_engine_forward_lockstep")) return rc;
  }
  struct PadGuard {
    explicit PadGuard(unsigned b) { rdm::gemm_set_lds_pad(b); }
    ~PadGuard() { rdm::gemm_set_lds_pad(0); }
  } pad_guard(0u);  // (as rdm_engine_run_lockstep)
  LockstepJob job{engines, nullptr, nullptr, nullptr, nullptr, results, stream, false, data};
  int rcs[kGroupMax] = {};
  const int wrc = lockstep_run(n_pairs, lockstep_pair, &job, static_cast<hipStream_t>(stream), lockstep_wait, engines[0], rcs);
  if (wrc == -1) {
    set_error("rdm_engine_forward_lockstep: called from inside a lock-step group (a host thread runs one group at a time)");
    return RDM_ERR_ARG;
  }
  if (wrc == -3) {
    set_error("rdm_engine_forward_lockstep: no memory for the context stacks");
    return RDM_ERR_HIP;
  }
  if (wrc != 0) return wrc;
  for (int k = n_pairs - 1; k >= 0; --k) {  // a pair that exhausted its arena runs again on its own (the arena grows)
    if (rcs[k] == RDM_ERR_WORKSPACE && engines[k]->arena_exhausted && !engines[k]->arena_fixed)
      rcs[k] = engine_run_growing(engines[k], nullptr, data[k]->n_ref[0], nullptr, data[k]->n_points[0] - data[k]->n_ref[0], data[k],
                                  results[k], stream);
  }
  for (int k = 0; k < n_pairs; ++k)
    if (rcs[k] != RDM_OK) return rcs[k];
  return RDM_OK;
}

// The collates of n pairs (rdm_engine_collate each: full tables at the reference's widths, kept as stage tensors of the pair's
// engine) on n engines in lock step.
extern "C" int rdm_engine_collate_lockstep(rdm_engine* const* engines, int n_pairs, const float* const* ref_points, const int64_t* n_ref,
                                           const float* const* src_points, const int64_t* n_src, rdm_engine_result* const* results,
                                           void* stream) {
  RDM_REQUIRE(engines && ref_points && n_ref && src_points && n_src && results, "rdm_engine_collate_lockstep: null pointer");
  RDM_REQUIRE(n_pairs >= 1 && n_pairs <= kGroupMax, "rdm_engine_collate_lockstep: 1 .. %d pairs", kGroupMax);
  for (int k = 0; k < n_pairs; ++k) {
    RDM_REQUIRE(engines[k] && results[k] && ref_points[k] && src_points[k] && n_ref[k] > 0 && n_src[k] > 0,
                "rdm_engine_collate_lockstep: pair %d is incomplete", k);
    RDM_REQUIRE(engines[k]->finalized, "rdm_engine_collate_lockstep: engine %d is not finalized", k);
    for (int j = 0; j < k; ++j) RDM_REQUIRE(engines[j] != engines[k], "rdm_engine_collate_lockstep: every pair needs an engine of its own");
  }
  LockstepJob job{engines, ref_points, n_ref, src_points, n_src, results, stream, false, nullptr, true};
  int rcs[kGroupMax] = {};
  const int wrc = lockstep_run(n_pairs, lockstep_pair, &job, static_cast<hipStream_t>(stream), lockstep_wait, engines[0], rcs);
  if (wrc == -1) {
    set_error("rdm_engine_collate_lockstep: called from inside a lock-step group (a host thread runs one group at a time)");
    return RDM_ERR_ARG;
  }
  if (wrc == -3) {
    set_error("rdm_engine_collate_lockstep: no memory for the context stacks");
    return RDM_ERR_HIP;
  }
  if (wrc != 0) return wrc;
  for (int k = n_pairs - 1; k >= 0; --k) {  // a pair that exhausted its arena collates again on its own (the arena grows)
    if (rcs[k] == RDM_ERR_WORKSPACE && engines[k]->arena_exhausted && !engines[k]->arena_fixed)
      rcs[k] = rdm_engine_collate(engines[k], ref_points[k], n_ref[k], src_points[k], n_src[k], results[k], stream);
  }
  for (int k = 0; k < n_pairs; ++k)
    if (rcs[k] != RDM_OK) return rcs[k];
  return RDM_OK;
}

// The collate alone (geotransformer/utils/data.py:13-77 on two clouds): the pyramid and its 13 searches stay in the engine's
// arena as stage tensors ("points0".."points4", "lengths0".., "neighbors0".., "subsampling0".."subsampling3",
// "upsampling0".., "search_flags") for rdm_engine_export; level sizes in result_host.  Stage tensors are kept for this call
// whatever rdm_engine_keep_taps says (and the setting is restored afterwards).
extern "C" int rdm_engine_collate(rdm_engine* e, const float* ref_points, int64_t n_ref, const float* src_points,
                                  int64_t n_src, rdm_engine_result* res, void* stream) {
  RDM_REQUIRE(e && ref_points && src_points && res, "rdm_engine_collate: null pointer");
  RDM_REQUIRE(n_ref > 0 && n_src > 0, "rdm_engine_collate: empty cloud");
  RDM_REQUIRE(e->finalized, "rdm_engine_collate: call rdm_engine_finalize first");
  const bool keep_before = e->keep_taps;  // the stage tensors of THIS call are kept; later runs keep what they kept before
  e->keep_taps = true;
  e->collate_only = true;
  const int rc = engine_run_growing(e, ref_points, n_ref, src_points, n_src, nullptr, res, stream);
  e->collate_only = false;
  e->keep_taps = keep_before;
  return rc;
}

// RDMNet.forward alone (experiments/model_infer.py:109-354) on a data_dict the caller collated -- with this library's
// collate or with the reference's (geotransformer/utils/data.py:139-192).
extern "C" int rdm_engine_forward(rdm_engine* e, const rdm_data_dict* dd, rdm_engine_result* res, void* stream) {
  RDM_REQUIRE(e && dd && res, "rdm_engine_forward: null pointer");
  RDM_REQUIRE(e->finalized, "rdm_engine_forward: call rdm_engine_finalize first");
  if (int rc = check_data_dict(dd, "rdm_engine_forward")) return rc;
  return engine_run_growing(e, nullptr, dd->n_ref[0], nullptr, dd->n_points[0] - dd->n_ref[0], dd, res, stream);
}

static int engine_run_once(rdm_engine* e, const float* ref_points, int64_t n_ref, const float* src_points, int64_t n_src,
                           const rdm_data_dict* dd, rdm_engine_result* res, void* stream, const PairPyramid* pre) {
  const rdm_engine_config& c = e->cfg;
  e->arena_off = pre ? e->arena_base : 0;  // (a pair of a collated batch: its pyramid lies below arena_base)
  if (!pre) {
    e->batch.clear();  // (the arena is rewritten: a collated batch is gone)
    e->arena_base = 0;
  }
  e->taps.clear();
  e->prof.clear();
  e->prof_layers = 0;
  Run r;
  r.e = e; r.st = static_cast<hipStream_t>(stream); r.groups = c.group_norm;
  const int64_t n0 = n_ref + n_src;
  // latency mode (see rdm_engine::overlap_mode); never on the legacy null stream, which every other blocking stream serialises with
  bool overlap = !pre && !lockstep_active() && !e->collate_only && r.st != nullptr && (e->overlap_mode == 2 || (e->overlap_mode == 1 && e->pairs_in_flight == 1));
  if (overlap) ENG_CHECK(ensure_side(e, r.st, &overlap));
  SideGuard side_guard{e};
  // kernel scratch: the largest consumers are the grid-subsample tables and split-K partials
  r.ws_bytes = pre ? size_t(96) << 20
                   : std::max<size_t>(rdm_grid_subsample_workspace_bytes(n0, 2),
                                      std::max<size_t>(rdm_radius_neighbors_workspace_bytes(n0, n0, 2), size_t(96) << 20));
  r.ws = e->alloc<char>(r.ws_bytes);
  ENG_ALLOC(r.ws);
  std::memset(res, 0, sizeof(*res));

  Level lv[5];
  Table nb[5], sub[4], up[4];
  int32_t* flags = pre ? pre->flags : e->alloc<int32_t>(64);
  ENG_ALLOC(flags);
  // Latency mode, first half: the caller's stream runs the subsampling of levels 1-4 (a chain of one- and two-workgroup kernels,
  // ~0.4 ms) while the side stream builds the first level's grid, searches its neighbours and runs the encoder's first two
  // blocks (wide launches that need nothing but level 0).  No device-side dependency: taken only when the caller's stream is
  // idle at the call (the inputs are then complete), and joined by a host wait next to the read-back of the level sizes.
  bool overlap_l0 = false;
  if (overlap && !dd) {  // (a hand-off event the caller just made the stream wait for -- dataset.PairStager -- takes it a few us)
    const auto t_poll = std::chrono::steady_clock::now();
    do {
      overlap_l0 = hipStreamQuery(r.st) == hipSuccess;
    } while (!overlap_l0 && std::chrono::steady_clock::now() - t_poll < std::chrono::microseconds(40));
  }
  Run rs = r;  // the side stream's launches of the first level
  if (overlap_l0) {
    rs.st = e->side;
    rs.ws_bytes = std::max<size_t>(size_t(96) << 20, rdm_linear_group_norm_workspace_bytes(n0, 128));
    rs.ws = e->alloc<char>(rs.ws_bytes);
    ENG_ALLOC(rs.ws);
  }
  if (!pre) fill_words<int32_t>(flags, 64, 0, overlap_l0 ? rs.st : r.st);  // (the first searches write their status rows on that stream)
  int call = pre ? pre->calls : 0;
  Grid grids[5] = {};
  std::vector<char> redo_queue(radius_redo_queue_bytes());
  radius_redo_queue_reset(redo_queue.data());
  auto build_grid = [&](const Level& s, float rad, Grid& g) -> int {
    g.n_s = s.n;
    g.bytes = rdm_radius_grid_workspace_bytes(s.n);
    g.ws = e->alloc<char>(g.bytes);
    ENG_ALLOC(g.ws);
    return rdm_radius_grid_build(s.pts, s.n, s.lengths, 2, rad, g.ws, g.bytes, r.st);
  };
  auto search = [&](const Level& q, const Grid& g, float rad, int limit, Table& t, bool i32 = false, void* queue = nullptr,
                    hipStream_t qst = nullptr) -> int {
    t.rows = q.n; t.width = limit; t.flags = flags + 2 * call++;
    t.i32 = i32;
    t.idx = reinterpret_cast<int64_t*>(e->alloc<char>(static_cast<size_t>(q.n > 0 ? q.n : 1) * limit * (i32 ? 4 : 8)));
    ENG_ALLOC(t.idx);
    // the large-buffer second pass of all 14 searches is one launch after the loop (radius_redo_flush)
    unsigned char* redo_flags = e->alloc<unsigned char>(static_cast<size_t>(q.n > 0 ? q.n : 1));
    ENG_ALLOC(redo_flags);
    return radius_grid_query_deferred(g.ws, g.bytes, g.n_s, q.pts, q.n, q.lengths, 2, rad, limit, t.idx, nullptr, t.flags,
                                      t.flags + 1, redo_flags, queue ? queue : redo_queue.data(), i32 ? 1 : 0, qst ? qst : r.st);
  };
  // the five level grids with one set of launches (radius r_i = 2^i r_0): they serve the searches of the collate and, through
  // their cell-sorted records, the spatial query order of the KPConv kernels and the shortcut pools
  auto build_level_grids = [&](int first = 0, int last = 5, hipStream_t gst = nullptr) -> int {  // levels [first, last)
    const float* gp[5];
    int64_t gn[5];
    const int64_t* gl[5];
    float gr[5];
    void* gw[5];
    size_t gb[5];
    float rad = c.init_radius;
    int k = 0;
    for (int i = 0; i < last; ++i, rad *= 2.f) {
      if (i < first) continue;
      grids[i].n_s = lv[i].n;
      grids[i].bytes = rdm_radius_grid_workspace_bytes(lv[i].n);
      grids[i].ws = e->alloc<char>(grids[i].bytes);
      ENG_ALLOC(grids[i].ws);
      gp[k] = lv[i].pts; gn[k] = lv[i].n; gl[k] = lv[i].lengths; gr[k] = rad; gw[k] = grids[i].ws; gb[k] = grids[i].bytes;
      ++k;
    }
    RDM_DUP_LOOP("rnbuild")
    ENG_CHECK(radius_grid_build_multi(k, gp, gn, gl, 2, gr, gw, gb, gst ? gst : r.st));
    return RDM_OK;
  };
  // ---------------------------------------------------------------- encoder (backbone.py:72-107), as two callables: the
  // latency mode runs the input and the first two blocks on the side stream while the deeper levels are still being subsampled
  Mat x;
  uint8_t* x_pos = nullptr;
  Mat feats[5];
  int fi = 0;
  auto encoder_input = [&](Run& rr) -> int {
    if (dd) {  // data_dict['features'] (model_infer.py:113), [N0, 1]
      x.p = const_cast<float*>(dd->features); x.rows = n0; x.cols = 1; x.ld = dd->features_ld;
    } else {
      x = e->mat(n0, 1);
      ENG_ALLOC(x.p);
    }
    x_pos = e->alloc<uint8_t>(n0);
    ENG_ALLOC(x_pos);
    if (dd) ENG_CHECK(rdm_row_positive(x.p, n0, 1, x.ld, x_pos, rr.st));
    else ENG_CHECK(launch1d<unit_features_kernel_body, unit_features_kernel>("unit_features", n0, rr.st, x.p, n0, x.ld, x_pos));
    return RDM_OK;
  };
  auto encoder_blocks = [&](Run& rr, int b0, int b1) -> int {
    const char* names[14] = {"encoder1_1", "encoder1_2", "encoder2_1", "encoder2_2", "encoder2_3", "encoder3_1", "encoder3_2",
                             "encoder3_3", "encoder4_1", "encoder4_2", "encoder4_3", "encoder5_1", "encoder5_2", "encoder5_3"};
    const int level[14] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4};
    const bool strided[14] = {false, false, true, false, false, true, false, false, true, false, false, true, false, false};
    for (int b = b0; b < b1; ++b) {
      const std::string name = std::string("encoder.") + names[b];
      const int lvl = level[b];
      const Level& s = lv[lvl];
      const Level& q = strided[b] ? lv[lvl + 1] : lv[lvl];
      const Table& t = strided[b] ? sub[lvl] : nb[lvl];
      const float sigma = c.init_sigma * static_cast<float>(1 << lvl);
      // visit the queries in the cell order of their level's search grid: neighbouring queries share most
      // neighbours, so gathered lines are re-used from L1 (results do not depend on the order)
      const Grid& qg = grids[strided[b] ? lvl + 1 : lvl];
      const float* order = qg.ws ? rdm_radius_grid_records(qg.ws, qg.bytes, qg.n_s) : nullptr;
      Mat y;
      if (b == 0) {
        ENG_CHECK(kpconv(rr, name + ".KPConv", x, x_pos, q, s, t, sigma, name + ".norm", y, order));
      } else {
        Mat h = x;
        uint8_t* h_pos = e->alloc<uint8_t>(x.rows > 0 ? x.rows : 1);
        ENG_ALLOC(h_pos);
        if (e->params->lin.count(name + ".unary1.mlp")) {
          ENG_CHECK(unary(rr, name + ".unary1", x, h, 2, nullptr, h_pos));
        } else {
          ENG_CHECK(rdm_row_positive(x.p, x.rows, x.cols, x.ld, h_pos, rr.st));
        }
        Mat cn;
        Mat sc = x;
        ENG_CHECK(kpconv(rr, name + ".KPConv", h, h_pos, q, s, t, sigma, name + ".norm_conv", cn, order,
                         strided[b] ? &x : nullptr, strided[b] ? &sc : nullptr));
        if (e->params->lin.count(name + ".unary_shortcut.mlp")) {
          Mat s2;
          ENG_CHECK(unary(rr, name + ".unary_shortcut", sc, s2, 0, nullptr, nullptr));
          sc = s2;
        }
        ENG_CHECK(unary(rr, name + ".unary2", cn, y, 2, &sc, nullptr));
      }
      x = y;
      tap(r, name.c_str(), x);
      if (b == 1 || b == 4 || b == 7 || b == 10 || b == 13) feats[fi++] = x;
    }
    return RDM_OK;
  };
  if (pre) {
    // ------------------------------------------------------------ a pair of a collated batch (rdm_engine_collate_batch)
    for (int i = 0; i < 5; ++i) {
      lv[i] = pre->lv[i];
      nb[i] = pre->nb[i];
      grids[i] = pre->grids[i];
      res->level_sizes[i] = lv[i].n;
      res->level_ref_sizes[i] = lv[i].n_ref;
      if (i < 4) {
        sub[i] = pre->sub[i];
        up[i] = pre->up[i];
      }
    }
  } else if (dd) {
    // ------------------------------------------------------------ the caller's data_dict (model_infer.py:113-131)
    for (int i = 0; i < 5; ++i) {
      lv[i].pts = const_cast<float*>(dd->points[i]);
      lv[i].n = dd->n_points[i];
      lv[i].n_ref = dd->n_ref[i];
      lv[i].lengths = const_cast<int64_t*>(dd->lengths[i]);
      res->level_sizes[i] = lv[i].n;
      res->level_ref_sizes[i] = lv[i].n_ref;
      nb[i].idx = const_cast<int64_t*>(dd->neighbors[i]);
      nb[i].rows = lv[i].n; nb[i].width = dd->neighbors_width[i]; nb[i].ld = dd->neighbors_ld[i];
      nb[i].flags = const_cast<int32_t*>(dd->neighbors_count[i]);
    }
    for (int i = 0; i < 4; ++i) {
      sub[i].idx = const_cast<int64_t*>(dd->subsampling[i]);
      sub[i].rows = lv[i + 1].n; sub[i].width = dd->subsampling_width[i]; sub[i].ld = dd->subsampling_ld[i];
      sub[i].flags = const_cast<int32_t*>(dd->subsampling_count[i]);
      up[i].idx = const_cast<int64_t*>(dd->upsampling[i]);
      up[i].rows = lv[i].n; up[i].width = dd->upsampling_width[i]; up[i].ld = dd->upsampling_ld[i];
      up[i].flags = const_cast<int32_t*>(dd->upsampling_count[i]);
    }
    // the caller collated (no grids of ours): build them for the spatial query order of the encoder's kernels -- the same
    // grids the per-op mirror builds (rdmnet_amd/model.py: run_encoder), so both give the same GroupNorm partials
    ENG_CHECK(build_level_grids());
    if (dd->collate_status && dd->n_collate_status > 0) {  // the collate's status words join the engine's own (checked below)
      ::rdm::launch<or_status_kernel_body, or_status_kernel, 64>(dim3(1), 0, r.st, dd->collate_status, static_cast<int>(dd->n_collate_status),
                         flags + 2 * call + 1);
      ENG_CHECK(launch_status("or_status_kernel"));
      call++;
    }
  } else {
  // ---------------------------------------------------------------- collate (data.py:13-77)
  // The forward consumes column 0 of upsampling[1..3] only (nearest_upsample, functional.py:6-22) and upsampling[0] not at all
  // (backbone.py:118-151 stops at the second level): a plain run skips that search (32 000 queries, a quarter of all) and keeps one
  // column of the others; the collate API and runs that keep their stage tensors build the reference's full tables.
  // (in a plain run up[0] stays the default-constructed Table -- null pointer, zero rows, never read -- and the status
  // rows `flags` are numbered in CALL order, i.e. without the skipped search: self0 sub0 self1 sub1 self2 sub2 up1 ... ; the
  // layout Engine.collate documents (rows 4 / 7 / 10 / 12 = the up-sampling tables) holds for runs that build all 13 tables.)
  const bool full_up = e->keep_taps || e->collate_only;
  // ... and keeps the neighbour / subsampling tables, which only its own KPConv layers and shortcut pools read, in 32 bits
  // (33 MB less written and 65 MB less read per pair; the int64 layout of the reference stays wherever a table leaves the
  // engine: stage tensors, rdm_engine_collate, rdm_engine_forward's data_dict)
  const bool i32 = !full_up;
  lv[0].n = n0; lv[0].n_ref = n_ref;
  lv[0].pts = e->alloc<float>(3 * n0);
  lv[0].lengths = e->alloc<int64_t>(2);
  ENG_ALLOC(lv[0].pts); ENG_ALLOC(lv[0].lengths);
  // (latency mode: each stream stacks the clouds for itself -- lv[0] is the side stream's copy, which everything but the
  // subsampling chain reads; that chain reads the caller's stream's own)
  float* pts0 = lv[0].pts;
  int64_t* len0 = lv[0].lengths;
  if (overlap_l0) {
    pts0 = e->alloc<float>(3 * n0);
    len0 = e->alloc<int64_t>(2);
    ENG_ALLOC(pts0); ENG_ALLOC(len0);
  }
  if (overlap_l0) side_guard.pending++;  // (from here to the join an early return waits for the side stream)
  ENG_CHECK(launch1d<concat_points_kernel_body, concat_points_kernel>("concat_points", 3 * n0, r.st, ref_points, n_ref, src_points, n_src, pts0, len0));
  int64_t* all_len = e->alloc<int64_t>(8);  // device lengths of levels 1..4, contiguous for one read-back
  ENG_ALLOC(all_len);
  float voxel = c.init_voxel_size;
  int64_t cap = n0;
  // levels whose subsampling runs as the multi-launch pipeline (RDM_GS_MULTI_LEVELS, developer knob; default: the first)
  static const int gs_multi_levels = [] { const char* v = ::rdm::dev_knob("RDM_GS_MULTI_LEVELS"); return v ? atoi(v) : 1; }();
  std::vector<char> side_queue(radius_redo_queue_bytes());
  for (int i = 1; i < 5; ++i) {
    voxel *= 2.f;  // data.py:23-28
    lv[i].pts = e->alloc<float>(3 * cap);
    lv[i].lengths = all_len + 2 * (i - 1);
    ENG_ALLOC(lv[i].pts);
    // level 0 -> 1 (16-20 k points per cloud): phases spread over the GPU when the clouds are large; the later levels run at
    // capacity `cap` with a few thousand real points: the single-workgroup kernel
    RDM_DUP_LOOP("gs")
    ENG_CHECK(grid_subsample_mode(i == 1 ? pts0 : lv[i - 1].pts, cap, i == 1 ? len0 : lv[i - 1].lengths, 2, voxel, lv[i].pts,
                                  lv[i].lengths, r.ws, r.ws_bytes, r.st, i <= gs_multi_levels ? 2 : 1));
    // capacity of the next level is unknown on the host until the read-back; run it at full capacity
    if (i == 1 && overlap_l0) {
      // beside that chain, on the side stream: everything that needs level 0 only.  Enqueued between the chain's first level
      // (0.18 ms of GPU time) and the rest, so that neither stream waits for the host to get to it.
      ENG_CHECK(launch1d<concat_points_kernel_body, concat_points_kernel>("concat_points", 3 * n0, rs.st, ref_points, n_ref, src_points, n_src, lv[0].pts,
                         lv[0].lengths));
      ENG_CHECK(build_level_grids(0, 1, rs.st));
      radius_redo_queue_reset(side_queue.data());
      ENG_CHECK(search(lv[0], grids[0], c.init_radius, c.neighbor_limits[0], nb[0], i32, side_queue.data(), rs.st));
      ENG_CHECK(radius_redo_flush(side_queue.data(), rs.st));
      ENG_CHECK(encoder_input(rs));
      ENG_CHECK(encoder_blocks(rs, 0, 2));
    }
  }
  int64_t host_len[8];
  // NOTE: each level was subsampled with n_points = cap (capacity): rows beyond the true count are
  // never read because the kernels walk `lengths`.
  ENG_CHECK(d2h(r, all_len, sizeof(host_len), host_len));
  if (overlap_l0) {  // host-side join (the caller's stream is idle here)
    ENG_CHECK(wait_on(e, e->side));
    side_guard.pending--;
  }
  for (int i = 1; i < 5; ++i) {
    lv[i].n_ref = host_len[2 * (i - 1)];
    lv[i].n = host_len[2 * (i - 1)] + host_len[2 * (i - 1) + 1];
    res->level_sizes[i] = lv[i].n;
    res->level_ref_sizes[i] = lv[i].n_ref;
  }
  res->level_sizes[0] = n0;
  res->level_ref_sizes[0] = n_ref;

  float radius = c.init_radius;
  ENG_CHECK(build_level_grids(overlap_l0 ? 1 : 0, 5));
  for (int i = 0; i < 5; ++i) {
    if (!(overlap_l0 && i == 0)) ENG_CHECK(search(lv[i], grids[i], radius, c.neighbor_limits[i], nb[i], i32));
    if (i < 4) ENG_CHECK(search(lv[i + 1], grids[i], radius, c.neighbor_limits[i], sub[i], i32));
    if (i > 0 && (full_up || i > 1)) ENG_CHECK(search(lv[i - 1], grids[i], radius, full_up ? c.neighbor_limits[i] : 1, up[i - 1]));
    radius *= 2.f;
  }
  ENG_CHECK(radius_redo_flush(redo_queue.data(), r.st));
  }  // collate
  for (int i = 0; i < 5; ++i) {
    tap(r, ("lengths" + std::to_string(i)).c_str(), lv[i].lengths, 1, 2, 2, 1);
    tap(r, ("points" + std::to_string(i)).c_str(), lv[i].pts, lv[i].n, 3, 3, 0);
    tap(r, ("neighbors" + std::to_string(i)).c_str(), nb[i].idx, nb[i].rows, nb[i].width, nb[i].stride(), 1);
    if (i < 4) {
      tap(r, ("subsampling" + std::to_string(i)).c_str(), sub[i].idx, sub[i].rows, sub[i].width, sub[i].stride(), 1);
      tap(r, ("upsampling" + std::to_string(i)).c_str(), up[i].idx, up[i].rows, up[i].width, up[i].stride(), 1);
    }
  }

  tap(r, "search_flags", flags, 32, 2, 2, 3);  // per search: [max neighbour count, status] (int32)
  if (e->collate_only) {
    res->arena_used = e->arena_off;
    return RDM_OK;
  }

  // kernel scratch of the network part, sized from the actual pyramid: Linear+GroupNorm of level l is at most
  // lv[l].n rows x (init_dim * 2^(l+1)) columns (backbone.py:27-70); the decoder and heads are narrower
  {
    size_t need = r.ws_bytes;
    for (int l = 0; l < c.num_stages; ++l)
      need = std::max(need, rdm_linear_group_norm_workspace_bytes(lv[l].n, int64_t(128) << l));
    if (need > r.ws_bytes) {
      r.ws_bytes = need;
      r.ws = e->alloc<char>(r.ws_bytes);
      ENG_ALLOC(r.ws);
    }
  }

  // ---------------------------------------------------------------- encoder (the callables above)
  if (!overlap_l0) {
    ENG_CHECK(encoder_input(r));
    ENG_CHECK(encoder_blocks(r, 0, 14));
  } else {
    ENG_CHECK(encoder_blocks(r, 2, 14));  // (input and blocks 0-1 ran on the side stream beside the subsampling)
  }
  const int64_t Nc = lv[4].n, nc_ref = lv[4].n_ref, Nf = lv[1].n, nf_ref = lv[1].n_ref;
  const int64_t D = c.out_dim;  // 256
  tap(r, "feats_c_enc", feats[4]);

  // ---------------------------------------------------------------- transformer #1 + heads
  Mat pts_c4{e->alloc<float>(4 * (Nc > 0 ? Nc : 1)), Nc, 4, 4};
  ENG_ALLOC(pts_c4.p);
  ENG_CHECK(launch1d<pad_points_kernel_body, pad_points_kernel>("pad_points", Nc, r.st, lv[4].pts, Nc, pts_c4.p));
  Mat buf_c = e->mat(Nc, D + 1);
  ENG_ALLOC(buf_c.p);
  Mat x_c = buf_c.cols_from(0, D);
  ENG_CHECK(thdroformer(r, "transformer", pts_c4, feats[4], nc_ref, c.num_layers, x_c));
  tap(r, "t1", x_c);
  Mat n2p_logit = buf_c.cols_from(D, 1);
  ENG_CHECK(linear(r, "proj_n2p_score", x_c, n2p_logit, 0, false));
  // The three score heads' sigmoids (and the n2n projection) feed output tensors only (model_infer.py:160-236: `n2p_scores`,
  // `p2p_scores`, `node_scores`; the n2p LOGIT is an input of the decoder and stays): a plain run, which hands out no stage
  // tensors, skips those four launches.
  const bool want_scores = e->keep_taps;
  float* n2p = nullptr;
  if (want_scores) {
    n2p = e->alloc<float>(Nc);
    ENG_ALLOC(n2p);
    ENG_CHECK(rdm_sigmoid_column(n2p_logit.p, n2p_logit.ld, Nc, n2p, r.st));
    tap(r, "n2p_scores", n2p, Nc, 1, 1, 0);
  }

  // ---------------------------------------------------------------- decoder (backbone.py:118-151)
  // Latency mode: the three wide GEMMs of the decoder run on the side stream beside the chain of small launches of the second
  // transformer, the grouping and the coarse matching (~80 launches of a few workgroups each).  Fork and join sit at read-backs
  // the host performs anyway (the NMS survivor counts, the coarse matching's correspondence count): once the host has seen the
  // caller's stream idle, the side stream needs no device-side dependency on it, and the join is a host wait on the side
  // stream next to the one on the caller's stream -- a device-side event pair between two busy streams costs ~0.15 ms per
  // pair here (tools/overlap_probe.py, tools/xsync_probe.hip), more than the overlap wins.
  Mat dec, feats_f;
  float* p2p = nullptr;
  auto run_decoder = [&](Run& rd) -> int {
    const std::string n4 = "decoder.decoder4.norm.norm", n3 = "decoder.decoder3.norm.norm";
    Mat l4, l3;
    ENG_CHECK(decoder_stage(rd, "decoder.decoder4.mlp", &n4, buf_c, up[3].idx, up[3].stride(), feats[3], lv[3].n, l4));
    ENG_CHECK(decoder_stage(rd, "decoder.decoder3.mlp", &n3, l4, up[2].idx, up[2].stride(), feats[2], lv[2].n, l3));
    ENG_CHECK(decoder_stage(rd, "decoder.decoder2.mlp", nullptr, l3, up[1].idx, up[1].stride(), feats[1], lv[1].n, dec));
    tap(r, "decoder", dec);
    feats_f = dec.cols_from(0, D);
    if (want_scores) {
      p2p = e->alloc<float>(Nf);
      ENG_ALLOC(p2p);
      ENG_CHECK(rdm_sigmoid_column(dec.p + D, dec.ld, Nf, p2p, rd.st));
      tap(r, "p2p_scores", p2p, Nf, 1, 1, 0);
    }
    return RDM_OK;
  };
  auto fork_decoder = [&]() -> int {  // after a host wait on the caller's stream: what the decoder reads was complete then, and nothing enqueued since writes it
    Run rd = r;
    rd.st = e->side;
    rd.ws = nullptr; rd.ws_bytes = 0;  // (decoder_stage then takes its scratch from the arena: r.ws belongs to the caller's stream)
    side_guard.pending++;
    return run_decoder(rd);
  };
  if (!overlap) ENG_CHECK(run_decoder(r));

  int64_t m_r = 0, m_s = 0, Mn = 0;
  float* nodes = nullptr;
  Mat buf2;
  if (c.use_vote) {
    // ---------------------------------------------------------------- vote (vote.py:83-117)
    Mat h = x_c;
    for (int i = 0; i < c.vote_mlp_layers; ++i) {
      Mat t1, t2;
      ENG_CHECK(linear(r, "vote.mlp_modules." + std::to_string(3 * i), h, t1));
      ENG_CHECK(layer_norm(r, "vote.mlp_modules." + std::to_string(3 * i + 1), t1, nullptr, 1, t2));
      h = t2;
    }
    Mat off;
    ENG_CHECK(linear(r, "vote.ctr_reg", h, off));
    float* shifted = e->alloc<float>(3 * (Nc > 0 ? Nc : 1));
    ENG_ALLOC(shifted);
    ENG_CHECK(rdm_vote_shift(lv[4].pts, off.p, off.ld, Nc, c.vote_limit[0], c.vote_limit[1], c.vote_limit[2], shifted, r.st));
    Mat off_f = off.cols_from(3, D), vfeats;
    ENG_CHECK(layer_norm(r, "vote.out_proj.0", x_c, &off_f, 0, vfeats));
    tap(r, "vote_xyz", shifted, Nc, 3, 3, 0);
    tap(r, "vote_feats", vfeats);
    float* n2n = nullptr;
    if (want_scores) {
      Mat n2n_logit;
      ENG_CHECK(linear(r, "proj_n2n_score", vfeats, n2n_logit));
      n2n = e->alloc<float>(Nc);
      ENG_ALLOC(n2n);
      ENG_CHECK(rdm_sigmoid_column(n2n_logit.p, n2n_logit.ld, Nc, n2n, r.st));
    }

    // ---------------------------------------------------------------- NMS (vote.py:13-40)
    Level nodes_all;
    nodes_all.pts = shifted; nodes_all.n = Nc; nodes_all.lengths = lv[4].lengths; nodes_all.n_ref = nc_ref;
    Table nms_t;
    Grid nms_grid;
    if (Nc <= 4096) {  // a few hundred superpoints: one cell per cloud (brute force) in one launch instead of the grid's seven
      nms_grid.n_s = Nc;
      nms_grid.bytes = rdm_radius_grid_workspace_bytes(Nc);
      nms_grid.ws = e->alloc<char>(nms_grid.bytes);
      ENG_ALLOC(nms_grid.ws);
      ENG_CHECK(radius_grid_build_trivial(shifted, Nc, lv[4].lengths, 2, nms_grid.ws, nms_grid.bytes, r.st));
    } else {
      ENG_CHECK(build_grid(nodes_all, c.nms_radius, nms_grid));
    }
    ENG_CHECK(search(nodes_all, nms_grid, c.nms_radius, c.neighbor_limits[4], nms_t));
    ENG_CHECK(radius_redo_flush(redo_queue.data(), r.st));  // (searches are recorded and run at the flush)
    uint8_t* keep = e->alloc<uint8_t>(Nc > 0 ? Nc : 1);
    ENG_ALLOC(keep);
    ENG_CHECK(rdm_nms(nms_t.idx, Nc, nms_t.width, nms_t.width, nms_t.flags, keep, r.st));
    tap(r, "nms_mask", keep, Nc, 1, 1, 2);
    int32_t* order = e->alloc<int32_t>(Nc > 0 ? Nc : 1);
    int32_t* kept = flags + 60;  // [2]
    ENG_ALLOC(order);
    // both clouds' compactions in one launch, which also stores the status words into the mapped read-back buffer
    ENG_CHECK(compact_indices_pair(keep, nc_ref, Nc, order, kept, flags, static_cast<int32_t*>(e->pinned_dev), 64, r.st));
    int32_t host_flags[64];
    ENG_CHECK(wait_stream(r));
    std::memcpy(host_flags, e->pinned, sizeof(host_flags));
    for (int i = 0; i < 2 * call; i += 2)
      if (host_flags[i + 1] != 0) {
        set_error("rdm_engine_run: a radius search failed (status %d: 2 = grid built for a smaller radius)", host_flags[i + 1]);
        return RDM_ERR_CAPACITY;
      }
    m_r = host_flags[60]; m_s = host_flags[61]; Mn = m_r + m_s;
    RDM_REQUIRE(m_r > 0 && m_s > 0, "rdm_engine_run: NMS left no superpoints");
    nodes = e->alloc<float>(3 * Mn);
    ENG_ALLOC(nodes);
    Mat sel_feats = e->mat(Mn, D);
    ENG_ALLOC(sel_feats.p);
    // (the (n2p, n2n) score pairs exist only in runs that hand out stage tensors: a plain run skips the score heads, and a
    // buffer nobody writes is not allocated -- ADVICE r4: a later consumer could otherwise read garbage)
    float* sel_scores = want_scores ? e->alloc<float>(2 * Mn) : nullptr;
    if (want_scores) ENG_ALLOC(sel_scores);
    Mat nodes4{e->alloc<float>(4 * Mn), Mn, 4, 4};
    ENG_ALLOC(nodes4.p);
    {  // the survivors' rows of every per-superpoint tensor with one launch
      SelectNodesArgs sa;
      sa.order = order; sa.nc_ref = static_cast<int>(nc_ref); sa.m_r = static_cast<int>(m_r); sa.m_n = static_cast<int>(Mn);
      sa.xyz = shifted; sa.feats = vfeats.p; sa.n2p = n2p; sa.n2n = n2n;
      sa.d = static_cast<int>(D); sa.ldf = static_cast<int>(vfeats.ld); sa.ldo = static_cast<int>(sel_feats.ld);
      sa.nodes = nodes; sa.nodes4 = nodes4.p; sa.out_feats = sel_feats.p; sa.scores = sel_scores;
      ::rdm::launch<select_nodes_kernel_body, select_nodes_kernel, 64>(dim3(static_cast<unsigned>(Mn)), 0, r.st, sa);
      ENG_CHECK(launch_status("select_nodes_kernel"));
    }
    tap(r, "nodes", nodes, Mn, 3, 3, 0);
    if (want_scores) tap(r, "node_scores", sel_scores, Mn, 2, 2, 0);

    // ---------------------------------------------------------------- transformer #2, normalise
    buf2 = e->mat(Mn, D);
    ENG_ALLOC(buf2.p);
    // (latency mode: the decoder is forked once the first two layers are enqueued -- everything it reads was complete at the
    // read-back above, and the host's 12 launches for it then do not hold up this chain)
    const std::function<int()> fork = fork_decoder;
    ENG_CHECK(thdroformer(r, "transformer2", nodes4, sel_feats, m_r, c.num_layers2, buf2, overlap ? &fork : nullptr));
    if (overlap && 2 * c.num_layers2 < 2) ENG_CHECK(fork_decoder());
    tap(r, "t2", buf2);
  } else {
    // infer.py:119-120 (Mulran) switches the vote layer off and model_infer.py:179-246 then leaves the
    // superpoints undefined; defined as the un-shifted coarse points with the first transformer's features
    int32_t host_flags[64];
    ENG_CHECK(d2h(r, flags, sizeof(host_flags), host_flags));
    for (int i = 0; i < 2 * call; i += 2)
      if (host_flags[i + 1] != 0) {
        set_error("rdm_engine_run: a radius search failed (status %d: 2 = grid built for a smaller radius)", host_flags[i + 1]);
        return RDM_ERR_CAPACITY;
      }
    m_r = nc_ref; m_s = Nc - nc_ref; Mn = Nc;
    RDM_REQUIRE(m_r > 0 && m_s > 0, "rdm_engine_run: a cloud has no superpoints");
    if (overlap) ENG_CHECK(fork_decoder());
    nodes = const_cast<float*>(lv[4].pts);
    buf2 = x_c;
    tap(r, "nodes", nodes, Mn, 3, 3, 0);
  }
  Mat fn = e->mat(Mn, D);
  ENG_ALLOC(fn.p);
  ENG_CHECK(rdm_l2_normalize(buf2.p, buf2.ld, Mn, D, fn.p, fn.ld, r.st));
  tap(r, "feats_c", fn);

  // ---------------------------------------------------------------- grouping + coarse matching
  const int K = c.points_in_patch;
  const float* pf_ref = lv[1].pts;
  const float* pf_src = lv[1].pts + 3 * nf_ref;
  const int64_t nf_src = Nf - nf_ref;
  int64_t* r_knn = e->alloc<int64_t>(m_r * K);
  int64_t* s_knn = e->alloc<int64_t>(m_s * K);
  uint8_t* r_km = e->alloc<uint8_t>(m_r * K);
  uint8_t* s_km = e->alloc<uint8_t>(m_s * K);
  uint8_t* r_nm = e->alloc<uint8_t>(m_r);
  uint8_t* s_nm = e->alloc<uint8_t>(m_s);
  ENG_ALLOC(r_knn); ENG_ALLOC(s_knn); ENG_ALLOC(r_km); ENG_ALLOC(s_km); ENG_ALLOC(r_nm); ENG_ALLOC(s_nm);
  int32_t* p2n_status = flags + 62;
  RDM_DUP_LOOP("p2n")
  ENG_CHECK(rdm_point_to_node_pair(pf_ref, nf_ref, nodes, m_r, pf_src, nf_src, nodes + 3 * m_r, m_s, K, r_knn, r_km, r_nm, s_knn,
                                   s_km, s_nm, p2n_status, r.ws, r.ws_bytes, r.st));  // both clouds, one set of launches
  const int kc = c.num_correspondences;
  int64_t* r_sel = e->alloc<int64_t>(kc);
  int64_t* s_sel = e->alloc<int64_t>(kc);
  float* node_sc = e->alloc<float>(kc);
  int32_t* n_sel = flags + 63;
  ENG_ALLOC(r_sel); ENG_ALLOC(s_sel); ENG_ALLOC(node_sc);
  {
    void* cm_ws = r.ws;
    size_t cm_bytes = rdm_coarse_matching_features_workspace_bytes(m_r, m_s);  // 12 B per superpoint pair
    if (cm_bytes > r.ws_bytes) {  // thousands of superpoints per cloud (very sparse input)
      cm_ws = e->alloc<char>(cm_bytes);
      ENG_ALLOC(cm_ws);
    } else {
      cm_bytes = r.ws_bytes;
    }
    RDM_DUP_LOOP("coarse")
  ENG_CHECK(rdm_coarse_matching_features(fn.p, fn.ld, m_r, fn.p + m_r * fn.ld, fn.ld, m_s, D, r_nm, s_nm,
                                           c.dual_normalization, kc, r_sel, s_sel, node_sc, n_sel, cm_ws, cm_bytes, r.st));
  }
  int32_t tail[2];
  ENG_CHECK(d2h(r, flags + 62, sizeof(tail), tail));
  if (overlap) {  // host-side join: the fine features are read from here on
    ENG_CHECK(wait_on(e, e->side));
    side_guard.pending--;
  }
  if (tail[0] != 0) {
    set_error("rdm_engine_run: a superpoint owns more than 4096 points");
    return RDM_ERR_CAPACITY;
  }
  const int64_t B = tail[1];
  RDM_REQUIRE(B > 0, "rdm_engine_run: no superpoint correspondences");
  tap(r, "ref_node_corr_indices", r_sel, B, 1, 1, 1);
  tap(r, "src_node_corr_indices", s_sel, B, 1, 1, 1);
  tap(r, "node_corr_scores", node_sc, B, 1, 1, 0);

  // ---------------------------------------------------------------- patches, Sinkhorn, LGR
  int64_t* r_idx = e->alloc<int64_t>(B * K);
  int64_t* s_idx = e->alloc<int64_t>(B * K);
  uint8_t* r_pm = e->alloc<uint8_t>(B * K);
  uint8_t* s_pm = e->alloc<uint8_t>(B * K);
  float* r_pts = e->alloc<float>(B * K * 3);
  float* s_pts = e->alloc<float>(B * K * 3);
  ENG_ALLOC(r_idx); ENG_ALLOC(s_idx); ENG_ALLOC(r_pm); ENG_ALLOC(s_pm); ENG_ALLOC(r_pts); ENG_ALLOC(s_pts);
  {  // the patch gathers as two launches: (knn indices, knn masks) x (ref, src), then the points x (ref, src); the patch
     // FEATURES are gathered inside the score GEMM's operand loads (rdm_patch_scores)
    const void* x1[4] = {r_knn, s_knn, r_km, s_km};
    const int64_t ns1[4] = {m_r, m_s, m_r, m_s}, w1[4] = {2 * K, 2 * K, K / 4, K / 4}, mm1[4] = {B, B, B, B};
    const int64_t* i1[4] = {r_sel, s_sel, r_sel, s_sel};
    void* y1[4] = {r_idx, s_idx, r_pm, s_pm};
    RDM_DUP_LOOP("rows")
  ENG_CHECK(gather_rows_multi(4, x1, ns1, w1, w1, i1, mm1, y1, w1, r.st));
    const void* x2[2] = {pf_ref, pf_src};
    const int64_t ns2[2] = {nf_ref, nf_src}, w2[2] = {3, 3}, lx2[2] = {3, 3};
    const int64_t mm2[2] = {B * K, B * K};
    const int64_t* i2[2] = {r_idx, s_idx};
    void* y2[2] = {r_pts, s_pts};
    RDM_DUP_LOOP("rows")
  ENG_CHECK(gather_rows_multi(2, x2, ns2, w2, lx2, i2, mm2, y2, w2, r.st));
  }
  float* sqrt_c = vecp(r, "__sqrt_out_dim");  // [K] x sqrt(D): the einsum's divisor (model_infer.py:311), uploaded at finalize
  float* scores = e->alloc<float>(B * K * K);
  float* ms = e->alloc<float>(B * (K + 1) * (K + 1));
  ENG_ALLOC(sqrt_c); ENG_ALLOC(scores); ENG_ALLOC(ms);
  static const bool materialise_patches = ::rdm::dev_knob("RDM_NO_PATCH_GATHER") != nullptr;  // developer knob (A/B): gather, then GEMM
  if (materialise_patches) {
    float* r_pf = e->alloc<float>(B * K * D);
    float* s_pf = e->alloc<float>(B * K * D);
    ENG_ALLOC(r_pf); ENG_ALLOC(s_pf);
    ENG_CHECK(rdm_gather_rows(feats_f.p, nf_ref, D, feats_f.ld, r_idx, B * K, r_pf, D, r.st));
    ENG_CHECK(rdm_gather_rows(feats_f.p + nf_ref * feats_f.ld, nf_src, D, feats_f.ld, s_idx, B * K, s_pf, D, r.st));
    ENG_CHECK(rdm_gemm(r_pf, D, K * D, s_pf, D, K * D, 1, scores, K, static_cast<int64_t>(K) * K, K, K, D, static_cast<int>(B),
                       nullptr, sqrt_c, 0, nullptr, 0, r.st));
  } else {
    ENG_CHECK(rdm_patch_scores(feats_f.p, feats_f.ld, nf_ref, r_idx, feats_f.p + nf_ref * feats_f.ld, feats_f.ld, nf_src, s_idx, B,
                               K, D, sqrt_c, scores, r.st));
  }
  ENG_CHECK(rdm_sinkhorn(scores, B, K, K, r_pm, s_pm, vecp(r, "optimal_transport.alpha"), c.sinkhorn_iterations, ms, r.st));
  tap(r, "patch_scores", scores, B * K, K, K, 0);
  tap(r, "matching_scores", ms, B * (K + 1), K + 1, K + 1, 0);
  tap(r, "ref_node_corr_knn_points", r_pts, B * K, 3, 3, 0);
  tap(r, "src_node_corr_knn_points", s_pts, B * K, 3, 3, 0);
  tap(r, "ref_node_corr_knn_masks", r_pm, B, K, K, 2);
  tap(r, "src_node_corr_knn_masks", s_pm, B, K, K, 2);

  const int64_t ccap = B * 2 * K;
  float* rc = e->alloc<float>(3 * ccap);
  float* sc = e->alloc<float>(3 * ccap);
  float* cs = e->alloc<float>(ccap);
  float* T = e->alloc<float>(16 + 4);  // pose and the three counters behind it: one read-back copy
  ENG_ALLOC(rc); ENG_ALLOC(sc); ENG_ALLOC(cs); ENG_ALLOC(T);
  int32_t* counts = reinterpret_cast<int32_t*>(T + 16);
  RDM_DUP_LOOP("lgr")
  ENG_CHECK(rdm_lgr(ms, r_pts, s_pts, r_pm, s_pm, B, K, c.acceptance_radius, c.correspondence_threshold,
                    c.num_refinement_steps, rc, sc, cs, T, counts, r.ws, r.ws_bytes, r.st));
  struct {
    float T[16];
    int32_t counts[4];
  } tailbuf;
  float* host_corr = reinterpret_cast<float*>(static_cast<char*>(e->pinned) + 4096);
  // (a plain launch in a lock-step group too: ONE dispatch per pair whatever the schedule -- what the profile summaries count pairs by)
  hipLaunchKernelGGL(export_result_kernel, dim3(16), dim3(256), 0, r.st, T, rc, sc, cs, static_cast<uint32_t*>(e->pinned_dev),
                     reinterpret_cast<float*>(static_cast<char*>(e->pinned_dev) + 4096), static_cast<int>(e->host_corr_cap));
  ENG_CHECK(launch_status("export_result_kernel"));
  ENG_CHECK(wait_stream(r));
  std::memcpy(tailbuf.T, r.e->pinned, 64);
  std::memcpy(tailbuf.counts, static_cast<char*>(r.e->pinned) + 64, 12);
  std::memcpy(res->transform, tailbuf.T, 64);
  res->n_correspondences = tailbuf.counts[0];
  res->n_hypotheses = tailbuf.counts[1];
  res->best_hypothesis = tailbuf.counts[2];
  res->n_ref_nodes = m_r;
  res->n_src_nodes = m_s;
  res->n_node_correspondences = B;
  res->ref_corr_points = rc;
  res->src_corr_points = sc;
  res->corr_scores = cs;
  res->transform_dev = T;
  {  // host views of the correspondences (valid until the next call on this engine)
    const int64_t nh = std::min<int64_t>(res->n_correspondences, e->host_corr_cap);
    res->n_host_correspondences = static_cast<int32_t>(nh);
    res->host_ref_corr_points = host_corr;
    res->host_src_corr_points = host_corr + 3 * nh;
    res->host_corr_scores = host_corr + 6 * nh;
  }
  res->arena_used = e->arena_off;
  for (int i = 0; i < e->prof_layers && !e->profile_shapes_only; ++i) {  // the stream is idle here (read-back above synchronised it)
    RDM_HIP_CHECK(hipEventElapsedTime(&e->prof[i].gather_ms, e->events[3 * i], e->events[3 * i + 1]));
    RDM_HIP_CHECK(hipEventElapsedTime(&e->prof[i].total_ms, e->events[3 * i], e->events[3 * i + 2]));
  }
  tap(r, "ref_corr_points", rc, res->n_correspondences, 3, 3, 0);
  tap(r, "src_corr_points", sc, res->n_correspondences, 3, 3, 0);
  tap(r, "corr_scores", cs, res->n_correspondences, 1, 1, 0);
  tap(r, "estimated_transform", T, 4, 4, 4, 0);
  return RDM_OK;
}

namespace {
// up to 24 device-to-device copies in one launch (the output_dict of a forward): workgroup -> (item, 4 KB chunk)
struct CopyBatch {
  const uint32_t* src[24];
  uint32_t* dst[24];
  int first_block[25];  // prefix of 1024-word chunks
  int words[24];
  int n;
};
__global__ __launch_bounds__(256) void copy_multi_kernel(CopyBatch b) {
  int it = 0;
  for (int k = 1; k < b.n; ++k) it += static_cast<int>(blockIdx.x) >= b.first_block[k] ? 1 : 0;
  const int base = (blockIdx.x - b.first_block[it]) * 1024;
  const uint32_t* s = b.src[it];
  uint32_t* d = b.dst[it];
  const int n = b.words[it];
  if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {  // workgroup-uniform: 16 bytes per thread
    const int i = base + 4 * threadIdx.x;
    if (i + 3 < n) {
      *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
    } else {
      for (int k = i; k < n && k < i + 4; ++k) d[k] = s[k];
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = base + u * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
  }
}
}  // namespace

extern "C" int rdm_engine_describe(rdm_engine* e, int n, const char* const* names, rdm_tensor_view* out) {
  RDM_REQUIRE(e && names && out && n >= 0, "rdm_engine_describe: bad arguments");
  for (int i = 0; i < n; ++i) {
    auto it = e->taps.find(names[i]);
    if (it == e->taps.end()) {
      set_error("rdm_engine_describe: no tensor named %s (call rdm_engine_keep_taps first)", names[i]);
      return RDM_ERR_ARG;
    }
    out[i] = it->second;
  }
  return RDM_OK;
}

// Copies `n` stage tensors of the last run (rdm_engine_keep_taps) into caller buffers with as few launches as possible.
// dst[i] must hold rows * ld * element size bytes of tensor names[i] (as rdm_engine_get_tensor reports them).
extern "C" int rdm_engine_export(rdm_engine* e, int n, const char* const* names, void* const* dst, void* stream) {
  RDM_REQUIRE(e && names && dst && n >= 0, "rdm_engine_export: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  CopyBatch b;
  b.n = 0;
  b.first_block[0] = 0;
  auto flush = [&]() -> int {
    if (b.n > 0 && b.first_block[b.n] > 0) {
      hipLaunchKernelGGL(copy_multi_kernel, dim3(static_cast<unsigned>(b.first_block[b.n])), dim3(256), 0, st, b);
      if (int rc = launch_status("copy_multi_kernel")) return rc;
    }
    b.n = 0;
    b.first_block[0] = 0;
    return RDM_OK;
  };
  for (int i = 0; i < n; ++i) {
    auto it = e->taps.find(names[i]);
    if (it == e->taps.end()) {
      set_error("rdm_engine_export: no tensor named %s (call rdm_engine_keep_taps first)", names[i]);
      return RDM_ERR_ARG;
    }
    const rdm_tensor_view& v = it->second;
    const size_t esz = v.dtype == 1 ? 8 : (v.dtype == 2 ? 1 : 4);  // 0 f32, 1 i64, 2 u8, 3 i32
    const size_t bytes = static_cast<size_t>(v.rows) * v.ld * esz;
    if (bytes == 0) continue;
    RDM_REQUIRE(dst[i], "rdm_engine_export: null destination for %s", names[i]);
    if (bytes % 4 != 0 || bytes / 4 > (size_t(1) << 30) || (reinterpret_cast<uintptr_t>(v.data) & 3) ||
        (reinterpret_cast<uintptr_t>(dst[i]) & 3)) {  // odd byte counts (mask tensors): the runtime's copy
      RDM_HIP_CHECK(hipMemcpyAsync(dst[i], v.data, bytes, hipMemcpyDeviceToDevice, st));
      continue;
    }
    b.src[b.n] = static_cast<const uint32_t*>(v.data);
    b.dst[b.n] = static_cast<uint32_t*>(dst[i]);
    b.words[b.n] = static_cast<int>(bytes / 4);
    b.first_block[b.n + 1] = b.first_block[b.n] + static_cast<int>((bytes / 4 + 1023) / 1024);
    if (++b.n == 24) ENG_CHECK(flush());
  }
  return flush();
}

extern "C" int rdm_copy_device(void* dst, const void* src, size_t bytes, void* stream) {
  RDM_REQUIRE(dst && src, "rdm_copy_device: null pointer");
  RDM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return rdm::RDM_OK;
}
