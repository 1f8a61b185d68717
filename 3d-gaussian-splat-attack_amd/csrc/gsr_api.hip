// gsr_api.hip -- C ABI of libgsraster.so (include/gsraster.h): context, workspace pool, stage launches.
//
// Host side of the boundary that replaces the reference's third-party `diff_gaussian_rasterization._C`
// (imported at reference gaussian_renderer/__init__.py:14).  No torch linkage: raw device pointers in,
// kernels enqueued on the caller's stream.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/gsraster.h"
#include "gsr_kernels.hip.h"
#include "gsr_sort.hip.h"
#include "gsr_knn.hip.h"
#include "gsr_pgd.hip.h"

using namespace gsr;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(stage, expr)                                                                          \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) return set_err(GSR_ERR_DEVICE, "%s: %s (%s)", stage, hipGetErrorString(_e), #expr); \
  } while (0)

#define LAUNCH_CHECK(stage)                                                                       \
  do {                                                                                            \
    hipError_t _e = hipGetLastError();                                                            \
    if (_e != hipSuccess) return set_err(GSR_ERR_DEVICE, "%s: launch failed: %s", stage, hipGetErrorString(_e)); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// workspace pool: grow-only caching of hipMalloc blocks per device; reuse is stream-ordered
// ---------------------------------------------------------------------------------------------
namespace {

struct Block {
  void* p;
  size_t bytes;
  hipStream_t stream;   // stream of the last user
  bool used;
};

struct Pool {
  std::mutex mu;
  std::vector<Block> blocks;
  size_t total = 0;
};

constexpr int MAX_DEV = 32;
constexpr size_t POOL_SOFT_CAP = size_t(24) << 30;   // of 288 GB: beyond this, free blocks of other streams are re-used
Pool g_pool[MAX_DEV];

int cur_dev() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) d = 0;
  return std::min(std::max(d, 0), MAX_DEV - 1);
}

void* pool_alloc(int dev, size_t bytes, hipStream_t st) {
  bytes = std::max<size_t>((bytes + 255) & ~size_t(255), 256);
  Pool& pl = g_pool[dev];
  std::lock_guard<std::mutex> lk(pl.mu);
  // Best fit among the free blocks last used on THIS stream (work enqueued later on the same stream is ordered
  // after the old user by the stream itself).  A block last used on another stream would need a host-side wait
  // for that stream, which would serialise views pipelined over several streams: it is taken only once the pool
  // already holds POOL_SOFT_CAP bytes, otherwise a new block is allocated for this stream.
  int best = -1, other = -1;
  for (size_t i = 0; i < pl.blocks.size(); ++i) {
    const Block& b = pl.blocks[i];
    if (b.used || b.bytes < bytes || b.bytes > bytes + bytes / 2 + (1u << 20)) continue;
    int& slot = (b.stream == st) ? best : other;
    if (slot < 0 || b.bytes < pl.blocks[slot].bytes) slot = (int)i;
  }
  if (best < 0 && other >= 0 && pl.total + bytes > POOL_SOFT_CAP) {
    (void)hipStreamSynchronize(pl.blocks[other].stream);   // cross-stream reuse: wait for the old user
    best = other;
  }
  if (best >= 0) {
    Block& b = pl.blocks[best];
    b.used = true;
    b.stream = st;
    return b.p;
  }
  void* p = nullptr;
  // leave head-room so that a slowly growing pair count re-uses the block instead of reallocating
  const size_t want = bytes + bytes / 8;
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();
    if (other >= 0) {                                    // out of memory: take the other stream's block
      Block& b = pl.blocks[other];
      (void)hipStreamSynchronize(b.stream);
      b.used = true;
      b.stream = st;
      return b.p;
    }
    // give the device back every idle block of this pool (none of them fits), then try once more
    std::vector<Block> keep;
    for (Block& b : pl.blocks) {
      if (b.used) { keep.push_back(b); continue; }
      (void)hipStreamSynchronize(b.stream);
      (void)hipFree(b.p);
      pl.total -= b.bytes;
    }
    pl.blocks.swap(keep);
    if (hipMalloc(&p, want) != hipSuccess && hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  pl.blocks.push_back(Block{p, want, st, true});
  pl.total += want;
  return p;
}

void pool_free(int dev, void* p) {
  if (!p) return;
  Pool& pl = g_pool[dev];
  std::lock_guard<std::mutex> lk(pl.mu);
  for (Block& b : pl.blocks)
    if (b.p == p) { b.used = false; return; }
}

// a slab = one pool block carved into 256-byte aligned pieces
struct Slab {
  char* base = nullptr;
  size_t cap = 0, cur = 0;
  template <typename T>
  T* take(size_t n) {
    cur = (cur + 255) & ~size_t(255);
    T* r = reinterpret_cast<T*>(base + cur);
    cur += n * sizeof(T);
    return r;
  }
};

struct SlabPlan {
  size_t bytes = 0;
  template <typename T>
  void add(size_t n) { bytes = ((bytes + 255) & ~size_t(255)) + n * sizeof(T); }
};

// pinned host word + event for the one D2H read per forward (per calling thread and device)
thread_local uint32_t* t_pinned[MAX_DEV] = {nullptr};
thread_local hipEvent_t t_count_event[MAX_DEV] = {nullptr};

uint32_t* pinned_word(int dev) {
  if (!t_pinned[dev]) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocDefault) != hipSuccess) return nullptr;
    t_pinned[dev] = static_cast<uint32_t*>(p);
  }
  return t_pinned[dev];
}

hipEvent_t count_event(int dev) {
  if (!t_count_event[dev]) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    t_count_event[dev] = e;
  }
  return t_count_event[dev];
}

// ---- per-stage profiling (process-wide: autograd runs backward on its own thread) ------------------
struct ProfSpan { int stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
uint32_t g_prof = 0;   // bit i: stage i is timed
std::vector<ProfSpan> g_spans;
std::vector<hipEvent_t> g_free_events;
float g_ms[GSR_STAGE_COUNT] = {0};
int64_t g_calls[GSR_STAGE_COUNT] = {0};

hipEvent_t get_event_locked() {
  if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

struct StageTimer {
  int stage; hipStream_t st; hipEvent_t a{}, b{}; bool on = false;
  // kernel_events: the stage is ONE kernel and the caller hands a and b to hipExtLaunchKernelGGL, which stamps them
  // with the dispatch's own start and end (what a profiler reports as the kernel's duration); events recorded on the
  // stream around a launch also count the time the dispatch waits behind other streams' kernels.
  bool kernel_events;
  StageTimer(int s, hipStream_t stream, bool kernel_ev = false) : stage(s), st(stream), kernel_events(kernel_ev) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    on = (g_prof >> stage) & 1u;
    if (on) { a = get_event_locked(); b = get_event_locked(); if (!kernel_events) (void)hipEventRecord(a, st); }
  }
  ~StageTimer() {
    if (on) {
      if (!kernel_events) (void)hipEventRecord(b, st);
      std::lock_guard<std::mutex> lk(g_prof_mu);
      g_spans.push_back(ProfSpan{stage, a, b});
    }
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct GsrCtx {
  int dev = 0;
  GsrSettings st{};
  int P = 0, K = 0;
  int gridx = 0, gridy = 0, ntiles = 0;
  uint32_t N = 0;
  // inputs (owned by the caller)
  bool raw = false;              // inputs are raw parameters (gsr_forward_raw)
  const float* sh_dc = nullptr;  // raw: _features_dc
  const float *means3D = nullptr, *shs = nullptr, *sh_objs = nullptr, *colors = nullptr, *opac = nullptr,
              *scales = nullptr, *rots = nullptr, *cov3d = nullptr;
  // kept workspace
  void* keep_blk = nullptr;
  void* rank_blk = nullptr;
  void* seg_blk = nullptr;        // boundary records of split tiles (null: no tile is split)
  float4* bnd = nullptr;
  uint32_t* segoff = nullptr;
  uint2* rec_item = nullptr;
  uint32_t* nrec = nullptr;
  uint32_t seg_shift = 0, rec_cap = 0;
  size_t keep_bytes = 0;
  float4 *R0 = nullptr, *R1 = nullptr, *R2 = nullptr;   // splat records in depth order
  float4 *G0 = nullptr, *G1 = nullptr, *G2 = nullptr;   // the same in storage order
  float* D = nullptr;             // [P,9] d rgb / d view direction (lane-group kernels, SH input, backward expected)
  bool lanegroup = false;         // K1 ran as k_pre_fwd: K8+K9 runs as k_pre_bwd
  uint32_t *order = nullptr, *off = nullptr, *offg = nullptr, *pair_rank = nullptr;
  uint2* ranges = nullptr;
  uint32_t* sched = nullptr;      // [ntiles] tiles longest-list-first + priority class
  float* final_T = nullptr;
  uint32_t* n_contrib = nullptr;
  unsigned long long* total64 = nullptr;   // exact (64-bit) number of pairs of the tile rects
};

static ViewArgs view_args(const GsrSettings& s) {
  ViewArgs va;
  va.vm = s.viewmatrix; va.pm = s.projmatrix; va.cam = s.campos;
  va.H = s.image_height; va.W = s.image_width;
  va.tanfovx = s.tanfovx; va.tanfovy = s.tanfovy; va.mod = s.scale_modifier; va.deg = s.sh_degree;
  return va;
}

constexpr unsigned long long MAX_PAIRS = 1ull << 31;   // 32-bit pair numbering with head-room

// per-call launch overrides carried in GsrSettings.flags (include/gsraster.h)
static int flag_fwd_npx(uint32_t f) { const int v = (f >> 4) & 7u; return v == 1 ? 1 : v == 2 ? 2 : v == 3 ? 4 : 0; }
static int flag_bwd_npx(uint32_t f) { const int v = (f >> 8) & 3u; return v == 1 ? 2 : v == 2 ? 4 : 0; }
static int flag_tile_map(uint32_t f) { const int v = (f >> 12) & 7u; return (v >= 1 && v <= 4) ? v - 1 : 3; }

static int ceil_log2(uint32_t v) {
  int b = 0;
  while ((1u << b) < v) ++b;
  return b;
}

extern "C" {

const char* gsr_last_error(void) { return g_err; }

void gsr_ctx_free(GsrCtx* c) {
  if (!c) return;
  pool_free(c->dev, c->keep_blk);
  pool_free(c->dev, c->rank_blk);
  pool_free(c->dev, c->seg_blk);
  delete c;
}

}  // extern "C"

// raw != 0: scales / rotations / opacities are the reference model's RAW parameters, shs is _features_rest and
// sh_dc is _features_dc (K must be 16); activations and their chain rule run inside K1 / K9.
// diagnostic: device buffer [ntiles][2] that the next backward composites stamp with their waves' start/end clocks
static std::atomic<unsigned long long*> g_wave_clock{nullptr};
static std::atomic<unsigned long long*> g_wave_clock_fwd{nullptr};

// second attribute segment of gsr_forward_raw2 (raw parameters of Pb more Gaussians, numbered after the first P - Pb)
struct SegB {
  int32_t Pb = 0;
  const float *xyz = nullptr, *features_dc = nullptr, *features_rest = nullptr, *objects_dc = nullptr,
              *opacity = nullptr, *scaling = nullptr, *rotation = nullptr;
};

static int forward_impl(const GsrSettings* s, int32_t P, int32_t K, const float* means3D, const float* shs,
                        const float* sh_dc, const float* sh_objs, const float* colors_precomp, const float* opacities,
                        const float* scales, const float* rotations, const float* cov3D_precomp, float* out_color,
                        float* out_objects, int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered, void* stream,
                        bool raw, const SegB* segb = nullptr) {
  if (ctx_out) *ctx_out = nullptr;
  if (!s || !out_color) return set_err(GSR_ERR_INVALID, "gsr_forward: null settings / out_color");
  if (P < 0 || s->image_height <= 0 || s->image_width <= 0)
    return set_err(GSR_ERR_INVALID, "gsr_forward: bad sizes P=%d H=%d W=%d", P, s->image_height, s->image_width);
  if (P > 0) {   // an empty scene carries no data pointers: it renders the background
    if (!radii || !means3D || !opacities)
      return set_err(GSR_ERR_INVALID, "gsr_forward: null means3D / opacities / radii");
    if ((shs == nullptr) == (colors_precomp == nullptr))
      return set_err(GSR_ERR_INVALID, "gsr_forward: provide exactly one of shs / colors_precomp");
    const bool has_sr = scales != nullptr && rotations != nullptr;
    if (((scales != nullptr) != (rotations != nullptr)) || (has_sr == (cov3D_precomp != nullptr)))
      return set_err(GSR_ERR_INVALID, "gsr_forward: provide exactly one of (scales, rotations) / cov3D_precomp");
  }
  if (P > 0 && shs && (s->sh_degree < 0 || s->sh_degree > 3 || K < (s->sh_degree + 1) * (s->sh_degree + 1)))
    return set_err(GSR_ERR_INVALID, "gsr_forward: sh_degree %d needs K >= %d, got K=%d (degree must be 0..3)",
                   s->sh_degree, (s->sh_degree + 1) * (s->sh_degree + 1), K);
  if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos)
    return set_err(GSR_ERR_INVALID, "gsr_forward: settings tensors (bg, viewmatrix, projmatrix, campos) must be device pointers");
  if (out_objects == nullptr && sh_objs != nullptr) sh_objs = nullptr;   // objects not wanted

  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = cur_dev();
  const int H = s->image_height, W = s->image_width;
  const int gridx = (W + TILE - 1) / TILE, gridy = (H + TILE - 1) / TILE;
  const int ntiles = gridx * gridy;
  if (gridx > 4095 || gridy > 4095) return set_err(GSR_ERR_INVALID, "gsr_forward: image larger than 65520 px per side");
  if ((uint32_t)P > RANK_MASK) return set_err(GSR_ERR_INVALID, "gsr_forward: more than 2^28 Gaussians");
  const size_t HW = (size_t)H * W;

  GsrCtx* c = new (std::nothrow) GsrCtx();
  if (!c) return set_err(GSR_ERR_NOMEM, "gsr_forward: host allocation failed");
  c->dev = dev; c->st = *s; c->P = P; c->K = K; c->gridx = gridx; c->gridy = gridy; c->ntiles = ntiles;
  c->means3D = means3D; c->shs = shs; c->sh_objs = sh_objs; c->colors = colors_precomp; c->opac = opacities;
  c->scales = scales; c->rots = rotations; c->cov3d = cov3D_precomp;
  c->raw = raw; c->sh_dc = sh_dc;

  const size_t Pp = (size_t)std::max(P, 1);
  // ---- kept slab ---------------------------------------------------------------------------
  SlabPlan kp;
  kp.add<float4>(3 * Pp);   // G records (storage order)
  kp.add<uint32_t>(Pp); kp.add<uint32_t>(Pp + 1); kp.add<uint32_t>(Pp + 1);   // order, off, offg
  kp.add<uint2>(ntiles); kp.add<float>(HW); kp.add<uint32_t>(HW); kp.add<unsigned long long>(2); kp.add<uint32_t>(ntiles);
  // the SH layouts the reference uses (and precomputed colours) take the lane-group kernels
  c->lanegroup = raw || (shs && K == 16) || colors_precomp != nullptr;
  const bool want_D = c->lanegroup && shs != nullptr && ctx_out != nullptr;
  if (want_D) kp.add<float>(9 * Pp);
  c->keep_bytes = kp.bytes + 256;
  c->keep_blk = pool_alloc(dev, c->keep_bytes, st);
  // ---- scratch slab (released at the end of forward) ----------------------------------------
  const uint32_t tblP = radix_table_words((uint32_t)Pp);
  SlabPlan sp;
  sp.add<uint32_t>(Pp); sp.add<uint32_t>(Pp); sp.add<uint32_t>(Pp); sp.add<uint32_t>(Pp);   // dkey a/b, order b, tcnt
  sp.add<uint32_t>(tblP); sp.add<uint32_t>(RS_BINS);
  sp.add<uint32_t>(Pp / SCAN_CHUNK + 2); sp.add<unsigned long long>(Pp / SCAN_CHUNK + 2);
  void* scratch_blk = pool_alloc(dev, sp.bytes + 256, st);
  if (!c->keep_blk || !scratch_blk) {
    pool_free(dev, scratch_blk);
    gsr_ctx_free(c);
    return set_err(GSR_ERR_NOMEM, "gsr_forward: workspace allocation failed (P=%d, %dx%d)", P, W, H);
  }
  Slab ks{static_cast<char*>(c->keep_blk), c->keep_bytes, 0};
  c->G0 = ks.take<float4>(3 * Pp); c->G1 = c->G0 + 1; c->G2 = c->G0 + 2;   // interleaved 48-byte records
  c->R0 = c->G0; c->R1 = c->G1; c->R2 = c->G2;                              // the compositors gather them by Gaussian index
  c->order = ks.take<uint32_t>(Pp); c->off = ks.take<uint32_t>(Pp + 1); c->offg = ks.take<uint32_t>(Pp + 1);
  c->ranges = ks.take<uint2>(ntiles); c->final_T = ks.take<float>(HW); c->n_contrib = ks.take<uint32_t>(HW);
  c->total64 = ks.take<unsigned long long>(2);
  c->sched = ks.take<uint32_t>(ntiles);
  if (want_D) c->D = ks.take<float>(9 * Pp);
  Slab ss{static_cast<char*>(scratch_blk), sp.bytes + 256, 0};
  float4* G0 = c->G0; float4* G1 = c->G1; float4* G2 = c->G2;
  uint32_t* dkeyA = ss.take<uint32_t>(Pp); uint32_t* dkeyB = ss.take<uint32_t>(Pp); uint32_t* orderB = ss.take<uint32_t>(Pp);
  uint32_t* tcnt = ss.take<uint32_t>(Pp);
  uint32_t* table = ss.take<uint32_t>(tblP); uint32_t* tsums = ss.take<uint32_t>(RS_BINS);
  uint32_t* psums = ss.take<uint32_t>(Pp / SCAN_CHUNK + 2);
  unsigned long long* psums64 = ss.take<unsigned long long>(Pp / SCAN_CHUNK + 2);

  void* pairs_blk[4] = {nullptr, nullptr, nullptr, nullptr};
  void* tbl_blk = nullptr;
  auto fail = [&](int code) {
    pool_free(dev, scratch_blk);
    pool_free(dev, tbl_blk);
    for (void* b : pairs_blk) if (b && b != c->rank_blk) pool_free(dev, b);
    gsr_ctx_free(c);
    return code;
  };
#define F_TRY(stage, expr)                                                                                   \
  do {                                                                                                       \
    hipError_t _e = (expr);                                                                                  \
    if (_e != hipSuccess) return fail(set_err(GSR_ERR_DEVICE, "%s: %s (%s)", stage, hipGetErrorString(_e), #expr)); \
  } while (0)
#define F_LAUNCH(stage)                                                                                            \
  do {                                                                                                             \
    hipError_t _e = hipGetLastError();                                                                             \
    if (_e != hipSuccess) return fail(set_err(GSR_ERR_DEVICE, "%s: launch failed: %s", stage, hipGetErrorString(_e))); \
  } while (0)

  const ViewArgs va = view_args(*s);
  const dim3 blk(256);
  const dim3 gridP((unsigned)((Pp + 255) / 256));
  const dim3 blkPre(PRE_BLOCK), gridPre((unsigned)((Pp + PRE_BLOCK - 1) / PRE_BLOCK));
  uint32_t N = 0;
  if (P > 0) {
    {
      StageTimer t(GSR_STAGE_PREPROCESS, st);
      if (c->lanegroup) {
        PreArgs pa;
        pa.P = P; pa.va = va; pa.means = means3D; pa.scales = scales; pa.rots = rotations; pa.cov3d = cov3D_precomp;
        pa.opac = opacities; pa.sh = shs; pa.sh_dc = sh_dc; pa.colors = colors_precomp; pa.radii = radii;
        pa.G0 = G0; pa.G1 = G1; pa.G2 = G2; pa.D = c->D; pa.dkey = dkeyA; pa.tcnt = tcnt;
        pa.Pa = segb ? P - segb->Pb : P;
        pa.means_b = segb ? segb->xyz : nullptr; pa.scales_b = segb ? segb->scaling : nullptr;
        pa.rots_b = segb ? segb->rotation : nullptr; pa.opac_b = segb ? segb->opacity : nullptr;
        pa.sh_b = segb ? segb->features_rest : nullptr; pa.sh_dc_b = segb ? segb->features_dc : nullptr;
        if (raw) hipLaunchKernelGGL((k_pre_fwd<true>), gridPre, blkPre, 0, st, pa);
        else hipLaunchKernelGGL((k_pre_fwd<false>), gridPre, blkPre, 0, st, pa);
      } else
        hipLaunchKernelGGL((k_preprocess<false, false>), gridPre, blkPre, 0, st, P, K, va, means3D, scales, rotations,
                           cov3D_precomp, opacities, shs, sh_dc, colors_precomp, radii, G0, G1, G2, dkeyA, tcnt);
      // storage-order numbering of the (tile, Gaussian) pairs: where the backward puts its partial rows
      scan_exclusive_u32(tcnt, c->offg, (uint32_t)P, psums, c->offg + P, st, psums64, c->total64);
      F_LAUNCH("preprocess");
    }
    // The total of that scan IS the pair count N the host needs to size the sort buffers.  Start its read-back
    // now and wait for it only after the depth sort / gather / scan below are enqueued: the GPU keeps working
    // while the host learns N, instead of idling through a stream synchronise later.
    uint32_t* pinned = pinned_word(dev);
    hipEvent_t n_ready = count_event(dev);
    if (!pinned || !n_ready) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: pinned host word / event allocation failed"));
    // one 8-byte copy: the exact 64-bit total (N is its low word once it is known to be below 2^31)
    F_TRY("read pair count", hipMemcpyAsync(pinned + 2, c->total64, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    F_TRY("read pair count", hipEventRecord(n_ready, st));
    uint32_t* skey;
    {
      StageTimer t(GSR_STAGE_DEPTH_SORT, st);
      // stable argsort of the depth keys: order[r] = Gaussian index of depth rank r
      const int res = radix_sort_pairs(dkeyA, c->order, dkeyB, orderB, (uint32_t)P, 0, 32, true, table, tsums, st);
      F_LAUNCH("depth sort");
      // 4 passes => result back in (dkeyA, c->order); keep the code honest if the pass count changes
      skey = res ? dkeyB : dkeyA;
      if (res) F_TRY("depth sort", hipMemcpyAsync(c->order, orderB, sizeof(uint32_t) * P, hipMemcpyDeviceToDevice, st));
    }
    {
      StageTimer t(GSR_STAGE_BIN, st);
      uint32_t* cnt = orderB;   // free again after the sort
      (void)skey;
      hipLaunchKernelGGL(k_rank_counts, gridP, blk, 0, st, P, c->order, tcnt, cnt);
      scan_exclusive_u32(cnt, c->off, (uint32_t)P, psums, c->off + P, st);
      F_LAUNCH("pack/scan");
      F_TRY("read pair count", hipEventSynchronize(n_ready));
      unsigned long long exact = 0;
      memcpy(&exact, pinned + 2, sizeof(exact));
      N = (uint32_t)exact;
      if (exact >= MAX_PAIRS)   // NSUB * N must stay below 2^32
        return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: %llu (tile, Gaussian) pairs exceed the supported %llu "
                            "(splats cover too many tiles: check scales / scale_modifier)", exact, MAX_PAIRS));
    }
  } else {
    F_TRY("init", hipMemsetAsync(c->off, 0, sizeof(uint32_t), st));
    F_TRY("init", hipMemsetAsync(c->offg, 0, sizeof(uint32_t), st));
  }
  c->N = N;
  F_TRY("ranges", hipMemsetAsync(c->ranges, 0, sizeof(uint2) * ntiles, st));
  if (N > 0) {
    const uint32_t tblN = radix_table_words(N);
    for (int i = 0; i < 4; ++i) {
      pairs_blk[i] = pool_alloc(dev, sizeof(uint32_t) * (size_t)N, st);
      if (!pairs_blk[i]) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: pair buffers (N=%u) allocation failed", N));
    }
    tbl_blk = pool_alloc(dev, sizeof(uint32_t) * ((size_t)tblN + RS_BINS), st);
    if (!tbl_blk) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: sort table allocation failed"));
    uint32_t* tileA = static_cast<uint32_t*>(pairs_blk[0]); uint32_t* rankA = static_cast<uint32_t*>(pairs_blk[1]);
    uint32_t* tileB = static_cast<uint32_t*>(pairs_blk[2]); uint32_t* rankB = static_cast<uint32_t*>(pairs_blk[3]);
    uint32_t* tableN = static_cast<uint32_t*>(tbl_blk);
    uint32_t* tsumsN = tableN + tblN;
    {
      StageTimer t(GSR_STAGE_BIN, st);
      const int cull = (s->flags & GSR_FLAG_NO_CULL) ? 0 : 1;
      hipLaunchKernelGGL(k_emit, dim3((N + EMIT_SLOTS - 1) / EMIT_SLOTS), blk, 0, st, c->off, c->order, (uint32_t)P, N, c->G0,
                         c->G1, c->G2, gridx, W, H, (uint32_t)ntiles, cull, tileA, rankA);
      F_LAUNCH("emit");
    }
    int res;
    {
      StageTimer t(GSR_STAGE_TILE_SORT, st);
      // keys are 0..ntiles (ntiles = culled pair): one more value than there are tiles
      res = radix_sort_pairs(tileA, rankA, tileB, rankB, N, 0, ceil_log2((uint32_t)ntiles + 1), false, tableN, tsumsN, st);
      hipLaunchKernelGGL(k_ranges, dim3((N + 255) / 256), blk, 0, st, N, (uint32_t)ntiles, res ? tileB : tileA, c->ranges);
      F_LAUNCH("tile sort");
    }
    c->pair_rank = res ? rankB : rankA;
    c->rank_blk = res ? pairs_blk[3] : pairs_blk[1];
    pool_free(dev, tbl_blk);
    tbl_blk = nullptr;
    for (void* b : pairs_blk) if (b != c->rank_blk) pool_free(dev, b);
    for (void*& b : pairs_blk) b = nullptr;
  }
  {
    StageTimer t(GSR_STAGE_RENDER_FWD, st);
    RenderArgs ra;
    ra.ranges = c->ranges; ra.pair_rank = c->pair_rank; ra.R0 = c->R0; ra.R1 = c->R1; ra.R2 = c->R2;
    ra.sh_objs = sh_objs; ra.bg = s->bg; ra.W = W; ra.H = H; ra.gridx = gridx; ra.ntiles = ntiles;
    ra.sh_objs_b = segb ? segb->objects_dc : nullptr; ra.Pa = segb ? P - segb->Pb : P;
    const int map_mode_f = flag_tile_map(s->flags);
    ra.map_mode = map_mode_f;
    ra.sched = c->sched;
    ra.wave_clock = g_wave_clock_fwd.load();
    // Long tile lists are split into segments for the backward (gsr_kernels.hip.h, "Segments"): the forward stores the
    // per-pixel (T, C) at the segment boundaries.  Not with object channels (their 16 running sums are not stored), not
    // for a forward-only call, not under GSR_FLAG_NO_SEGMENTS.
    static const int seg_shift_env = [] { const char* e = getenv("GSR_SEG_SHIFT"); int v = e ? atoi(e) : 8; return (v >= 6 && v <= 16) ? v : 8; }();
    ra.bnd = nullptr; ra.segoff = nullptr; ra.seg_shift = 0;
    if (N > 0 && ctx_out && !(out_objects && sh_objs) && !(s->flags & GSR_FLAG_NO_SEGMENTS)) {
      const uint32_t per = N >> seg_shift_env;
      c->seg_shift = (uint32_t)seg_shift_env;
      c->rec_cap = per + std::min<uint32_t>((uint32_t)ntiles, per) + 1u;   // sum over split tiles of ceil(len / seg)
      SlabPlan gp;
      gp.add<float4>((size_t)c->rec_cap * PXL * 64); gp.add<uint2>(c->rec_cap); gp.add<uint32_t>(ntiles); gp.add<uint32_t>(4);
      c->seg_blk = pool_alloc(dev, gp.bytes + 256, st);
      if (!c->seg_blk) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: segment boundary buffer (N=%u) allocation failed", N));
      Slab gs{static_cast<char*>(c->seg_blk), gp.bytes + 256, 0};
      c->bnd = gs.take<float4>((size_t)c->rec_cap * PXL * 64); c->rec_item = gs.take<uint2>(c->rec_cap);
      c->segoff = gs.take<uint32_t>(ntiles); c->nrec = gs.take<uint32_t>(4);
      F_TRY("segments", hipMemsetAsync(c->rec_item, 0, sizeof(uint2) * c->rec_cap, st));
      ra.bnd = c->bnd; ra.segoff = c->segoff; ra.seg_shift = c->seg_shift;
    }
    if (map_mode_f == 3 || c->bnd)
      hipLaunchKernelGGL(k_tile_schedule, dim3(1), dim3(1024), 0, st, ntiles, c->ranges, c->sched, c->seg_shift, c->segoff,
                         c->rec_item, c->rec_cap, c->nrec);
    ra.out_color = out_color; ra.out_objects = out_objects; ra.final_T = c->final_T; ra.n_contrib = c->n_contrib;
    const dim3 blkT(64);
    // pixels per lane of K6: fewer = more, shorter waves per tile (see k_render_fwd); images with fewer tiles than
    // half the chip's wave slots are split down to one 16x4 strip per wave.  GSR_FLAG_FWD_SPLIT(n) overrides.
    const int fwd_npx = flag_fwd_npx(s->flags) ? flag_fwd_npx(s->flags) : (ntiles < 4096 ? 1 : 2);
    const dim3 gridT(render_grid(ntiles * (PXL / fwd_npx)));
    // the tile's waves as one workgroup that stages every batch once (k_render_fwd's WPB): S-nyc-1M gathers 596 -> 367 MB
    // but runs 0.197 -> 0.224 ms (two workgroup barriers per batch, and the waves of a tile wait for each other), so
    // it is opt-in: GSR_FLAG_FWD_SHARED, or GSR_K6_SHARED=1 in the environment
    static const int k6_env = [] { const char* e = getenv("GSR_K6_SHARED"); return e ? atoi(e) : 0; }();
    const bool k6_shared = k6_env != 0 || (s->flags & GSR_FLAG_FWD_SHARED) != 0;
    const dim3 gridS(render_grid(ntiles)), blkS2(128), blkS4(256);
    if (out_objects && sh_objs) {
      if (fwd_npx == 4) hipLaunchKernelGGL((k_render_fwd<true, 4>), gridT, blkT, 0, st, ra);
      else if (fwd_npx == 2 && k6_shared) hipLaunchKernelGGL((k_render_fwd<true, 2, 2>), gridS, blkS2, 0, st, ra);
      else if (fwd_npx == 2) hipLaunchKernelGGL((k_render_fwd<true, 2>), gridT, blkT, 0, st, ra);
      else if (k6_shared) hipLaunchKernelGGL((k_render_fwd<true, 1, 4>), gridS, blkS4, 0, st, ra);
      else hipLaunchKernelGGL((k_render_fwd<true, 1>), gridT, blkT, 0, st, ra);
    } else {
      if (out_objects) F_TRY("objects", hipMemsetAsync(out_objects, 0, sizeof(float) * NUM_OBJ * HW, st));
      if (fwd_npx == 4) hipLaunchKernelGGL((k_render_fwd<false, 4>), gridT, blkT, 0, st, ra);
      else if (fwd_npx == 2 && k6_shared) hipLaunchKernelGGL((k_render_fwd<false, 2, 2>), gridS, blkS2, 0, st, ra);
      else if (fwd_npx == 2) hipLaunchKernelGGL((k_render_fwd<false, 2>), gridT, blkT, 0, st, ra);
      else if (k6_shared) hipLaunchKernelGGL((k_render_fwd<false, 1, 4>), gridS, blkS4, 0, st, ra);
      else hipLaunchKernelGGL((k_render_fwd<false, 1>), gridT, blkT, 0, st, ra);
    }
    F_LAUNCH("render forward");
  }
  pool_free(dev, scratch_blk);
  if (num_rendered) *num_rendered = (int64_t)N;
  if (ctx_out) *ctx_out = c; else gsr_ctx_free(c);
  return GSR_OK;
#undef F_TRY
#undef F_LAUNCH
}

extern "C" {

int gsr_forward(const GsrSettings* s, int32_t P, int32_t K, const float* means3D, const float* shs,
                const float* sh_objs, const float* colors_precomp, const float* opacities, const float* scales,
                const float* rotations, const float* cov3D_precomp, float* out_color, float* out_objects,
                int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered, void* stream) {
  return forward_impl(s, P, K, means3D, shs, nullptr, sh_objs, colors_precomp, opacities, scales, rotations,
                      cov3D_precomp, out_color, out_objects, radii, ctx_out, num_rendered, stream, false);
}

int gsr_forward_raw2(const GsrSettings* s, int32_t Pa, const float* xyz_a, const float* features_dc_a,
                     const float* features_rest_a, const float* objects_dc_a, const float* opacity_logit_a,
                     const float* log_scaling_a, const float* rotation_raw_a, int32_t Pb, const float* xyz_b,
                     const float* features_dc_b, const float* features_rest_b, const float* objects_dc_b,
                     const float* opacity_logit_b, const float* log_scaling_b, const float* rotation_raw_b,
                     float* out_color, float* out_objects, int32_t* radii, int64_t* num_rendered, void* stream) {
  if (Pa < 0 || Pb < 0 || (long long)Pa + Pb > 0x7FFFFFFFll) return set_err(GSR_ERR_INVALID, "gsr_forward_raw2: bad sizes Pa=%d Pb=%d", Pa, Pb);
  if (Pb == 0)
    return forward_impl(s, Pa, 16, xyz_a, features_rest_a, features_dc_a, objects_dc_a, nullptr, opacity_logit_a,
                        log_scaling_a, rotation_raw_a, nullptr, out_color, out_objects, radii, nullptr, num_rendered, stream, true);
  if (Pa == 0)
    return forward_impl(s, Pb, 16, xyz_b, features_rest_b, features_dc_b, objects_dc_b, nullptr, opacity_logit_b,
                        log_scaling_b, rotation_raw_b, nullptr, out_color, out_objects, radii, nullptr, num_rendered, stream, true);
  if (!xyz_a || !features_dc_a || !features_rest_a || !opacity_logit_a || !log_scaling_a || !rotation_raw_a || !xyz_b ||
      !features_dc_b || !features_rest_b || !opacity_logit_b || !log_scaling_b || !rotation_raw_b)
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw2: null parameter tensor");
  if (out_objects && ((objects_dc_a == nullptr) != (objects_dc_b == nullptr)))
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw2: object features must be given for both segments or for neither");
  SegB b;
  b.Pb = Pb; b.xyz = xyz_b; b.features_dc = features_dc_b; b.features_rest = features_rest_b; b.objects_dc = objects_dc_b;
  b.opacity = opacity_logit_b; b.scaling = log_scaling_b; b.rotation = rotation_raw_b;
  return forward_impl(s, Pa + Pb, 16, xyz_a, features_rest_a, features_dc_a, objects_dc_a, nullptr, opacity_logit_a,
                      log_scaling_a, rotation_raw_a, nullptr, out_color, out_objects, radii, nullptr, num_rendered, stream,
                      true, &b);
}

int gsr_forward_raw(const GsrSettings* s, int32_t P, const float* xyz, const float* features_dc,
                    const float* features_rest, const float* objects_dc, const float* opacity_logit,
                    const float* log_scaling, const float* rotation_raw, float* out_color, float* out_objects,
                    int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered, void* stream) {
  if (P > 0 && (!features_dc || !features_rest || !log_scaling || !rotation_raw))
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw: null features_dc / features_rest / log_scaling / rotation_raw");
  return forward_impl(s, P, 16, xyz, features_rest, features_dc, objects_dc, nullptr, opacity_logit, log_scaling,
                      rotation_raw, nullptr, out_color, out_objects, radii, ctx_out, num_rendered, stream, true);
}

}  // extern "C"

static int backward_impl(GsrCtx* c, const float* grad_color, const float* grad_objects, float* dmeans3D, float* dmeans2D,
                         float* dshs, float* dsh_dc, float* dsh_objs, float* dcolors_precomp, float* dopacities,
                         float* dscales, float* drotations, float* dcov3D, void* stream) {
  if (!c) return set_err(GSR_ERR_STATE, "gsr_backward: null context");
  if (!grad_color) return set_err(GSR_ERR_INVALID, "gsr_backward: grad_color is null");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = c->dev;
  const int P = c->P;
  if (P == 0) return GSR_OK;
  const bool obj = grad_objects != nullptr && c->sh_objs != nullptr;
  // Only colour-side gradients wanted (SH / precomputed colours / object features): K7 and K8+K9 drop the geometry
  // sums and the projection chain rule (the colour attack; BASELINE configs 2 and 3).
  const bool geom = dmeans3D || dmeans2D || dopacities || dscales || drotations || dcov3D;
  const uint32_t N = c->N;
  void* part_blk = nullptr;
  void* pobj_blk = nullptr;
  float4* part = nullptr;
  float4* part_obj = nullptr;
  // K7 runs one wave per 16x(4*npx) part of a tile; each writes its own partial row per list entry (4 pixels per lane:
  // one wave per tile; 2: two, and K8/K9 then sums twice the rows).  With long lists walked as segments every work item
  // is short whatever the tile count, so one wave per tile it is (S-hydrant-full, 2500 tiles: K7 0.169 -> 0.194 ms but
  // K8+K9 0.141 -> 0.084 ms, 2563 -> 2811 views/s); only without segments (object channels, GSR_FLAG_NO_SEGMENTS) an
  // image with fewer tiles than the chip has wave slots is split in two.  GSR_FLAG_BWD_SPLIT(n) overrides.
  const bool segs_on = c->bnd != nullptr && !obj;
  const int bwd_npx = flag_bwd_npx(c->st.flags) ? flag_bwd_npx(c->st.flags) : ((c->ntiles < 4096 && !segs_on) ? 2 : 4);
  const uint32_t nsub = (uint32_t)(PXL / bwd_npx);
  if (N > 0) {
    part_blk = pool_alloc(dev, sizeof(float4) * PART_F4 * (size_t)N * nsub, st);
    if (obj) pobj_blk = pool_alloc(dev, sizeof(float4) * 4 * (size_t)N * nsub, st);
    if (!part_blk || (obj && !pobj_blk)) {
      pool_free(dev, part_blk); pool_free(dev, pobj_blk);
      return set_err(GSR_ERR_NOMEM, "gsr_backward: partial-gradient buffer (N=%u) allocation failed", N);
    }
    part = static_cast<float4*>(part_blk);
    part_obj = static_cast<float4*>(pobj_blk);
  }
  auto done = [&](int code) { pool_free(dev, part_blk); pool_free(dev, pobj_blk); return code; };
  // 64-bit tag of this call: K7 stamps it into every partial row it writes, K8/K9 ignores rows without it
  static std::atomic<uint64_t> tag_counter{0x243F6A8885A308D3ull};
  uint64_t z = tag_counter.fetch_add(0x9E3779B97F4A7C15ull) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  const uint32_t tag_lo = (uint32_t)z, tag_hi = (uint32_t)(z >> 32);
  if (N > 0) {
    StageTimer t(GSR_STAGE_RENDER_BWD, st, true);
    RenderBwdArgs ra;
    ra.tag_lo = tag_lo; ra.tag_hi = tag_hi;
    ra.ranges = c->ranges; ra.pair_rank = c->pair_rank; ra.offg = c->offg; ra.R0 = c->R0; ra.R1 = c->R1; ra.R2 = c->R2;
    ra.sh_objs = c->sh_objs; ra.bg = c->st.bg; ra.W = c->st.image_width; ra.H = c->st.image_height;
    ra.map_mode = flag_tile_map(c->st.flags);
    ra.sched = c->sched;
    ra.wave_clock = g_wave_clock.load();
    ra.gridx = c->gridx; ra.ntiles = c->ntiles; ra.final_T = c->final_T; ra.n_contrib = c->n_contrib;
    ra.grad_color = grad_color; ra.grad_objects = obj ? grad_objects : nullptr; ra.part = part; ra.part_obj = part_obj;
    // split tiles: one extra work item per boundary record, in front of the per-tile items
    const bool segs = c->bnd != nullptr && !obj;
    ra.bnd = segs ? c->bnd : nullptr; ra.segoff = c->segoff; ra.rec_item = c->rec_item; ra.nrec = c->nrec;
    ra.seg_shift = c->seg_shift; ra.extra_blocks = segs ? c->rec_cap * nsub : 0u;
    const dim3 gridT(ra.extra_blocks + (unsigned)render_grid(c->ntiles * (int)nsub)), blk(64);
#define LAUNCH_K7(kern)                                                                      \
  do {                                                                                       \
    if (t.on) hipExtLaunchKernelGGL(kern, gridT, blk, 0, st, t.a, t.b, 0, ra);               \
    else hipLaunchKernelGGL(kern, gridT, blk, 0, st, ra);                                    \
  } while (0)
    if (obj) {
      if (geom) {
        if (bwd_npx == 4) LAUNCH_K7((k_render_bwd<true, 4, true>));
        else LAUNCH_K7((k_render_bwd<true, 2, true>));
      } else {
        if (bwd_npx == 4) LAUNCH_K7((k_render_bwd<true, 4, false>));
        else LAUNCH_K7((k_render_bwd<true, 2, false>));
      }
    } else if (geom) {
      if (bwd_npx == 4) LAUNCH_K7((k_render_bwd<false, 4, true>));
      else LAUNCH_K7((k_render_bwd<false, 2, true>));
    } else {
      if (bwd_npx == 4) LAUNCH_K7((k_render_bwd<false, 4, false>));
      else LAUNCH_K7((k_render_bwd<false, 2, false>));
    }
#undef LAUNCH_K7
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return done(set_err(GSR_ERR_DEVICE, "render backward: launch failed: %s", hipGetErrorString(e)));
  }
  {
    StageTimer t(GSR_STAGE_PREPROCESS_BWD, st);
    PreBwdArgs pa;
    pa.P = P; pa.K = c->K; pa.va = view_args(c->st);
    pa.offg = c->offg; pa.G0 = c->G0; pa.G1 = c->G1; pa.G2 = c->G2;
    pa.part = part; pa.part_obj = obj ? part_obj : nullptr;
    pa.tag_lo = tag_lo; pa.tag_hi = tag_hi; pa.nsub = nsub;
    pa.means = c->means3D; pa.scales = c->scales; pa.rots = c->rots; pa.cov3d = c->cov3d; pa.sh = c->shs;
    pa.sh_dc = c->sh_dc; pa.dsh_dc = dsh_dc; pa.D = c->D;
    pa.dmeans3D = dmeans3D; pa.dmeans2D = dmeans2D; pa.dsh = c->shs ? dshs : nullptr; pa.dsh_objs = dsh_objs;
    pa.dcolors = c->colors ? dcolors_precomp : nullptr; pa.dopac = dopacities;
    pa.dscales = c->cov3d ? nullptr : dscales; pa.drots = c->cov3d ? nullptr : drotations;
    pa.dcov3d = c->cov3d ? dcov3D : nullptr;
    const dim3 gridK9((unsigned)((P + PRE_BLOCK - 1) / PRE_BLOCK));
    // SH rows of 16 coefficients x 3 channels (the only layout the reference uses) go through LDS
    const bool sh_lds = c->shs != nullptr && pa.dsh != nullptr && c->K == 16;
    (void)sh_lds;
    if (c->raw && ((pa.dsh == nullptr) != (pa.dsh_dc == nullptr)))
      return done(set_err(GSR_ERR_INVALID, "gsr_backward_raw: dfeatures_dc and dfeatures_rest must both be given"));
    if (c->lanegroup && c->shs && geom && !c->D)
      return done(set_err(GSR_ERR_STATE, "gsr_backward: the forward of this context was run without its backward state"));
    if (c->lanegroup) {
      if (c->raw) {
        if (geom) hipLaunchKernelGGL((k_pre_bwd<true, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
        else hipLaunchKernelGGL((k_pre_bwd<true, false>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
      } else {
        if (geom) hipLaunchKernelGGL((k_pre_bwd<false, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
        else hipLaunchKernelGGL((k_pre_bwd<false, false>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
      }
    } else {
      if (geom) hipLaunchKernelGGL((k_preprocess_bwd<false, false, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
      else hipLaunchKernelGGL((k_preprocess_bwd<false, false, false>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return done(set_err(GSR_ERR_DEVICE, "preprocess backward: launch failed: %s", hipGetErrorString(e)));
  }
  return done(GSR_OK);
}

extern "C" {

int gsr_backward(GsrCtx* c, const float* grad_color, const float* grad_objects, float* dmeans3D, float* dmeans2D,
                 float* dshs, float* dsh_objs, float* dcolors_precomp, float* dopacities, float* dscales,
                 float* drotations, float* dcov3D, void* stream) {
  if (c && c->raw) return set_err(GSR_ERR_STATE, "gsr_backward: context came from gsr_forward_raw; use gsr_backward_raw");
  return backward_impl(c, grad_color, grad_objects, dmeans3D, dmeans2D, dshs, nullptr, dsh_objs, dcolors_precomp,
                       dopacities, dscales, drotations, dcov3D, stream);
}

int gsr_backward_raw(GsrCtx* c, const float* grad_color, const float* grad_objects, float* dxyz, float* dmeans2D,
                     float* dfeatures_dc, float* dfeatures_rest, float* dobjects_dc, float* dopacity_logit,
                     float* dlog_scaling, float* drotation_raw, void* stream) {
  if (c && !c->raw) return set_err(GSR_ERR_STATE, "gsr_backward_raw: context came from gsr_forward; use gsr_backward");
  return backward_impl(c, grad_color, grad_objects, dxyz, dmeans2D, dfeatures_rest, dfeatures_dc, dobjects_dc, nullptr,
                       dopacity_logit, dlog_scaling, drotation_raw, nullptr, stream);
}

int gsr_mark_visible(const GsrSettings* s, int32_t P, const float* means3D, uint8_t* present, void* stream) {
  if (!s || !s->viewmatrix || !means3D || !present) return set_err(GSR_ERR_INVALID, "gsr_mark_visible: null argument");
  if (P <= 0) return GSR_OK;
  hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), P,
                     s->viewmatrix, means3D, present);
  LAUNCH_CHECK("mark_visible");
  return GSR_OK;
}

int gsr_query(int32_t what, int64_t* out) {
  if (!out) return set_err(GSR_ERR_INVALID, "gsr_query: null out");
  switch (what) {
    case 0: *out = GSR_VERSION; return GSR_OK;
    case 1: {
      Pool& pl = g_pool[cur_dev()];
      std::lock_guard<std::mutex> lk(pl.mu);
      *out = (int64_t)pl.total;
      return GSR_OK;
    }
    default: return set_err(GSR_ERR_INVALID, "gsr_query: unknown item %d", what);
  }
}

int gsr_ctx_info(const GsrCtx* c, int32_t what, int64_t* out) {
  if (!c || !out) return set_err(GSR_ERR_INVALID, "gsr_ctx_info: null argument");
  switch (what) {
    case 0: *out = (int64_t)c->N; return GSR_OK;
    case 1: *out = -1; return GSR_OK;
    case 2: *out = (int64_t)(c->keep_bytes + sizeof(uint32_t) * (size_t)c->N); return GSR_OK;
    default: return set_err(GSR_ERR_INVALID, "gsr_ctx_info: unknown item %d", what);
  }
}

int gsr_ctx_export(const GsrCtx* c, int32_t what, void* dst, int64_t dst_bytes, void* stream) {
  if (!c || !dst) return set_err(GSR_ERR_INVALID, "gsr_ctx_export: null argument");
  const size_t HW = (size_t)c->st.image_height * c->st.image_width;
  const void* src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case 0: src = c->ranges; bytes = sizeof(uint2) * (size_t)c->ntiles; break;
    case 1: src = c->pair_rank; bytes = sizeof(uint32_t) * (size_t)c->N; break;
    case 2: src = c->n_contrib; bytes = sizeof(uint32_t) * HW; break;
    case 3: src = c->final_T; bytes = sizeof(float) * HW; break;
    case 4: src = c->order; bytes = sizeof(uint32_t) * (size_t)c->P; break;
    case 5: src = c->off; bytes = sizeof(uint32_t) * ((size_t)c->P + 1); break;
    case 6: src = c->R0; bytes = sizeof(float4) * 3 * (size_t)c->P; break;   // depth-ordered records [P][3] float4
    case 7: src = c->G0; bytes = sizeof(float4) * 3 * (size_t)c->P; break;   // storage-ordered records
    default: return set_err(GSR_ERR_INVALID, "gsr_ctx_export: unknown item %d", what);
  }
  if ((int64_t)bytes > dst_bytes)
    return set_err(GSR_ERR_INVALID, "gsr_ctx_export: item %d needs %zu bytes, buffer has %lld", what, bytes, (long long)dst_bytes);
  if (bytes == 0) return GSR_OK;
  HIP_TRY("ctx export", hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return GSR_OK;
}

void gsr_trim_pool(void) {
  const int dev = cur_dev();
  (void)hipDeviceSynchronize();
  Pool& pl = g_pool[dev];
  std::lock_guard<std::mutex> lk(pl.mu);
  std::vector<Block> keep;
  for (Block& b : pl.blocks) {
    if (b.used) keep.push_back(b);
    else { (void)hipFree(b.p); pl.total -= b.bytes; }
  }
  pl.blocks.swap(keep);
}

int gsr_pgd_step(float* x, const float* grad, const float* x0, int64_t rows, int32_t cols, float alpha, float epsilon,
                 int32_t l2, void* stream) {
  if (rows < 0 || cols < 1 || cols > PGD_MAX_COLS)
    return set_err(GSR_ERR_INVALID, "gsr_pgd_step: rows=%lld cols=%d (1..%d columns)", (long long)rows, cols, PGD_MAX_COLS);
  if (rows == 0) return GSR_OK;
  if (!x || !grad || !x0) return set_err(GSR_ERR_INVALID, "gsr_pgd_step: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = cur_dev();
  const size_t n = (size_t)rows * (size_t)cols;
  const unsigned blocks = (unsigned)((rows + 63) / 64);
  if (!l2) {
    hipLaunchKernelGGL((k_pgd_step<false>), dim3(blocks), dim3(64), 0, st, x, grad, x0, (size_t)rows, cols, alpha, epsilon,
                       (const double*)nullptr, 0);
  } else {
    const int nb = (int)std::min<size_t>((n + 4095) / 4096, 1024);
    void* blk = pool_alloc(dev, sizeof(double) * (size_t)nb, st);
    if (!blk) return set_err(GSR_ERR_NOMEM, "gsr_pgd_step: allocation failed");
    double* partial = static_cast<double*>(blk);
    hipLaunchKernelGGL(k_pgd_sumsq, dim3(nb), dim3(256), 0, st, grad, n, partial);
    hipLaunchKernelGGL((k_pgd_step<true>), dim3(blocks), dim3(64), 0, st, x, grad, x0, (size_t)rows, cols, alpha, epsilon,
                       partial, nb);
    pool_free(dev, blk);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "gsr_pgd_step: launch failed: %s", hipGetErrorString(e));
  return GSR_OK;
}

int gsr_knn_dist2(const float* points, int32_t P, float* mean_dist2, void* stream) {
  if (P < 0 || (P > 0 && (!points || !mean_dist2))) return set_err(GSR_ERR_INVALID, "gsr_knn_dist2: null argument");
  if (P == 0) return GSR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = cur_dev();
  const uint32_t n = (uint32_t)P;
  const uint32_t tbl = radix_table_words(n);
  // ---- bounding box (one small read-back: this routine is scene set-up, not the per-view path) ----------------
  float host_bb[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  void* bb_blk = pool_alloc(dev, 256, st);
  if (!bb_blk) return set_err(GSR_ERR_NOMEM, "gsr_knn_dist2: allocation failed");
  float* bbox = static_cast<float*>(bb_blk);
  hipError_t e = hipMemcpyAsync(bbox, host_bb, sizeof(host_bb), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_knn_bbox, dim3(std::min<uint32_t>((n + 255) / 256, 1024u)), dim3(256), 0, st, points, P, bbox);
    e = hipMemcpyAsync(host_bb, bbox, sizeof(host_bb), hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  pool_free(dev, bb_blk);
  if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "gsr_knn_dist2: bounding box: %s", hipGetErrorString(e));
  // ---- grid: about three points per cell, at most 256 cells per axis ---------------------------------------------
  double ext[3], vol = 1.0, emax = 0.0;
  for (int a = 0; a < 3; ++a) { ext[a] = (double)host_bb[3 + a] - (double)host_bb[a]; emax = std::max(emax, ext[a]); }
  if (!(emax > 0.0)) emax = 1.0;                       // all points coincide
  for (int a = 0; a < 3; ++a) { ext[a] = std::max(ext[a], 1e-6 * emax); vol *= ext[a]; }
  const double cell = std::cbrt(vol * 3.0 / (double)P);
  KnnGrid g;
  int dims[3];
  float inv[3], size[3];
  for (int a = 0; a < 3; ++a) {
    dims[a] = (int)std::min(256.0, std::max(1.0, std::ceil(ext[a] / cell)));
    const double c = ext[a] / dims[a];
    inv[a] = (float)(1.0 / c);
    size[a] = (float)c;
  }
  g.minx = host_bb[0]; g.miny = host_bb[1]; g.minz = host_bb[2];
  g.inv_cx = inv[0]; g.inv_cy = inv[1]; g.inv_cz = inv[2];
  g.sx = size[0]; g.sy = size[1]; g.sz = size[2]; g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
  const uint32_t ncells = (uint32_t)dims[0] * dims[1] * dims[2];
  // ---- workspace -----------------------------------------------------------------------------------------------
  const size_t words = (size_t)4 * n + tbl + RS_BINS + 2 * (size_t)ncells + 64;
  void* blk = pool_alloc(dev, sizeof(uint32_t) * words, st);
  if (!blk) return set_err(GSR_ERR_NOMEM, "gsr_knn_dist2: workspace allocation failed");
  uint32_t* k0 = static_cast<uint32_t*>(blk);
  uint32_t* v0 = k0 + n; uint32_t* k1 = v0 + n; uint32_t* v1 = k1 + n;
  uint32_t* table = v1 + n; uint32_t* sums = table + tbl;
  uint2* cell_range = reinterpret_cast<uint2*>(sums + RS_BINS + (((uintptr_t)(sums + RS_BINS) & 4) ? 1 : 0));
  hipLaunchKernelGGL(k_knn_cells, dim3((n + 255) / 256), dim3(256), 0, st, points, P, g, k0);
  const int res = radix_sort_pairs(k0, v0, k1, v1, n, 0, ceil_log2(ncells), true, table, sums, st);
  const uint32_t* skeys = res ? k1 : k0;
  const uint32_t* sidx = res ? v1 : v0;
  if (ceil_log2(ncells) == 0) {   // a single cell: no pass ran, build the identity permutation by hand
    std::vector<uint32_t> iota(n);
    for (uint32_t i = 0; i < n; ++i) iota[i] = i;
    (void)hipMemcpyAsync(v0, iota.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, st);
    (void)hipStreamSynchronize(st);
  }
  (void)hipMemsetAsync(cell_range, 0, sizeof(uint2) * ncells, st);
  hipLaunchKernelGGL(k_knn_ranges, dim3((n + 255) / 256), dim3(256), 0, st, n, skeys, cell_range);
  hipLaunchKernelGGL(k_knn_search, dim3((n + 255) / 256), dim3(256), 0, st, points, P, g, sidx, cell_range, mean_dist2);
  pool_free(dev, blk);
  LAUNCH_CHECK("knn");
  return GSR_OK;
}

int gsr_debug_wave_clock(unsigned long long* buf) {
  g_wave_clock.store(buf);
  return GSR_OK;
}

int gsr_debug_wave_clock_fwd(unsigned long long* buf) {
  g_wave_clock_fwd.store(buf);
  return GSR_OK;
}

int gsr_test_scan(const uint32_t* in, uint32_t* out, uint32_t n, void* stream) {
  if (!in || !out) return set_err(GSR_ERR_INVALID, "gsr_test_scan: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = cur_dev();
  void* blk = pool_alloc(dev, sizeof(uint32_t) * ((size_t)n / SCAN_CHUNK + 2), st);
  if (!blk) return set_err(GSR_ERR_NOMEM, "gsr_test_scan: allocation failed");
  scan_exclusive_u32(in, out, n, static_cast<uint32_t*>(blk), out + n, st);
  pool_free(dev, blk);
  LAUNCH_CHECK("test scan");
  return GSR_OK;
}

int gsr_test_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t n, int32_t begin_bit, int32_t end_bit, int32_t iota,
                        void* stream) {
  if (!keys || !vals) return set_err(GSR_ERR_INVALID, "gsr_test_sort_pairs: null argument");
  if (n == 0) return GSR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = cur_dev();
  const uint32_t tbl = radix_table_words(n);
  void* blk = pool_alloc(dev, sizeof(uint32_t) * ((size_t)2 * n + tbl + RS_BINS), st);
  if (!blk) return set_err(GSR_ERR_NOMEM, "gsr_test_sort_pairs: allocation failed");
  uint32_t* k1 = static_cast<uint32_t*>(blk);
  uint32_t* v1 = k1 + n;
  uint32_t* table = v1 + n;
  uint32_t* sums = table + tbl;
  const int res = radix_sort_pairs(keys, vals, k1, v1, n, begin_bit, end_bit, iota != 0, table, sums, st);
  if (res) {
    (void)hipMemcpyAsync(keys, k1, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(vals, v1, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st);
  }
  pool_free(dev, blk);
  LAUNCH_CHECK("test sort");
  return GSR_OK;
}

void gsr_profile(int32_t enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof = (uint32_t)enable;
  for (ProfSpan& s : g_spans) { g_free_events.push_back(s.a); g_free_events.push_back(s.b); }
  g_spans.clear();
  for (int i = 0; i < GSR_STAGE_COUNT; ++i) { g_ms[i] = 0.f; g_calls[i] = 0; }
}

int gsr_profile_timeline(float* out, int max_spans) {
  // diagnostic: (stage, start ms, end ms) of every recorded span, relative to the first span's start; consumes them
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  hipEvent_t base = g_spans.empty() ? nullptr : g_spans[0].a;
  for (ProfSpan& s : g_spans) {
    if (hipEventSynchronize(s.b) != hipSuccess) break;
    float t0 = 0.f, t1 = 0.f;
    if (n < max_spans && out && hipEventElapsedTime(&t0, base, s.a) == hipSuccess &&
        hipEventElapsedTime(&t1, base, s.b) == hipSuccess) {
      out[3 * n] = (float)s.stage; out[3 * n + 1] = t0; out[3 * n + 2] = t1;
      ++n;
    }
  }
  for (ProfSpan& s : g_spans) { g_free_events.push_back(s.a); g_free_events.push_back(s.b); }
  g_spans.clear();
  (void)hipGetLastError();
  return n;
}

int gsr_profile_read(float* ms, int64_t* calls) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ProfSpan& s : g_spans) {
    hipError_t e = hipEventSynchronize(s.b);
    if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "gsr_profile_read: %s", hipGetErrorString(e));
    float m = 0.f;
    if (hipEventElapsedTime(&m, s.a, s.b) == hipSuccess) { g_ms[s.stage] += m; g_calls[s.stage] += 1; }
    g_free_events.push_back(s.a); g_free_events.push_back(s.b);
  }
  g_spans.clear();
  for (int i = 0; i < GSR_STAGE_COUNT; ++i) {
    if (ms) ms[i] = g_ms[i];
    if (calls) calls[i] = g_calls[i];
  }
  return GSR_OK;
}

}  // extern "C"
