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
#include <chrono>
#include <mutex>
#include <thread>
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
Pool g_pool[MAX_DEV];

// Soft cap of a device's pool: beyond it, free blocks last used on OTHER streams are re-used (behind a host-side wait
// for that stream) instead of allocating more.  An eighth of the device's memory (36 GB of an MI355X's 288 GB), at
// least 2 GB; GSR_POOL_CAP_MB overrides.
size_t pool_soft_cap() {
  static const size_t cap = [] {
    if (const char* e = getenv("GSR_POOL_CAP_MB")) { const long long v = atoll(e); if (v > 0) return (size_t)v << 20; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); return size_t(24) << 30; }
    return std::max<size_t>(total_b / 8, size_t(2) << 30);
  }();
  return cap;
}

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
  // already holds pool_soft_cap() bytes, otherwise a new block is allocated for this stream.
  int best = -1, other = -1;
  for (size_t i = 0; i < pl.blocks.size(); ++i) {
    const Block& b = pl.blocks[i];
    if (b.used || b.bytes < bytes || b.bytes > bytes + bytes / 2 + (1u << 20)) continue;
    int& slot = (b.stream == st) ? best : other;
    if (slot < 0 || b.bytes < pl.blocks[slot].bytes) slot = (int)i;
  }
  if (best < 0 && other >= 0 && pl.total + bytes > pool_soft_cap()) {
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
  size_t got = want;                                     // what the device allocation really holds
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
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      got = bytes;                                       // no head-room left: the block is recorded at its true size
      if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
    }
  }
  pl.blocks.push_back(Block{p, got, st, true});
  pl.total += got;
  return p;
}

void pool_free(int dev, void* p) {
  if (!p) return;
  Pool& pl = g_pool[dev];
  std::lock_guard<std::mutex> lk(pl.mu);
  for (Block& b : pl.blocks)
    if (b.p == p) { b.used = false; return; }
}

// A live block is being used on another stream than the one it was taken for (a kept context re-rendered or
// differentiated on stream `st`, ordered behind its earlier users by the caller): when it is freed, stream-ordered reuse
// has to follow THAT stream.
void pool_retag(int dev, void* p, hipStream_t st) {
  if (!p) return;
  Pool& pl = g_pool[dev];
  std::lock_guard<std::mutex> lk(pl.mu);
  for (Block& b : pl.blocks)
    if (b.p == p) { b.stream = st; return; }
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

// Host-visible slot of ONE forward's pair count.  The kernel that computes the count (k_storage_scan_hist) stores it
// straight into this pinned, device-mapped host memory and then stores the forward's token behind a system-scope fence;
// the host polls the token.  No copy and no event sits in the stream for it (a 32-byte device-to-host copy is a blit
// kernel plus an event: ~15 us of the forward's critical path, measured).  Slots are pooled and belong to a forward (its
// context, or the call itself when no context is kept), not to the calling thread.
enum { HS_N64 = 0, HS_OVF = 2, HS_TOKEN = 3 };
struct CountSlot {
  volatile uint32_t* host = nullptr;     // 16 words, pinned + mapped
  uint32_t* dev = nullptr;               // the same memory as the device sees it
  uint32_t token = 0;                    // what the forward that owns the slot will write last
};
std::mutex g_slot_mu;
std::vector<CountSlot> g_free_slots;
std::atomic<uint32_t> g_token{1};

bool slot_get(CountSlot& s) {
  bool have = false;
  {
    std::lock_guard<std::mutex> lk(g_slot_mu);
    if (!g_free_slots.empty()) { s = g_free_slots.back(); g_free_slots.pop_back(); have = true; }
  }
  if (!have) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return false; }
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); return false; }
    s.host = static_cast<volatile uint32_t*>(p);
    s.dev = static_cast<uint32_t*>(d);
    s.host[HS_TOKEN] = 0u;
  }
  uint32_t t = g_token.fetch_add(1);
  if (t == 0u) t = g_token.fetch_add(1);
  s.token = t;
  return true;
}

void slot_put(CountSlot& s) {
  if (!s.host) return;
  std::lock_guard<std::mutex> lk(g_slot_mu);
  g_free_slots.push_back(s);
  s = CountSlot{};
}

inline bool slot_landed(const CountSlot& s) { return s.host[HS_TOKEN] == s.token; }

// Waits until the forward that owns the slot has published its count: spins briefly (the write lands a few tens of
// microseconds after the forward's first kernels), then yields; falls back to a stream synchronise after two seconds.
bool slot_wait(const CountSlot& s, hipStream_t st) {
  for (int i = 0; i < 4000; ++i) {
    if (slot_landed(s)) return true;
    __builtin_ia32_pause();
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (!slot_landed(s)) {
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      if (hipStreamSynchronize(st) != hipSuccess) return false;
      return slot_landed(s);
    }
  }
  return true;
}

// Pair counts seen by earlier forwards, per (device, P, H, W): the capacity guess of an asynchronous-count forward
// (GSR_FLAG_ASYNC_COUNT).  An entry is dropped when a forward overflowed its guess, so the next one counts synchronously.
struct CapKey { int dev, P, H, W; };
struct CapEntry { CapKey k; unsigned long long n; };
std::mutex g_cap_mu;
std::vector<CapEntry> g_caps;

bool cap_lookup(const CapKey& k, unsigned long long& n) {
  std::lock_guard<std::mutex> lk(g_cap_mu);
  for (const CapEntry& e : g_caps)
    if (e.k.dev == k.dev && e.k.P == k.P && e.k.H == k.H && e.k.W == k.W) { n = e.n; return true; }
  return false;
}

void cap_store(const CapKey& k, unsigned long long n, bool erase) {
  std::lock_guard<std::mutex> lk(g_cap_mu);
  for (size_t i = 0; i < g_caps.size(); ++i) {
    const CapKey& q = g_caps[i].k;
    if (q.dev == k.dev && q.P == k.P && q.H == k.H && q.W == k.W) {
      if (erase) { g_caps[i] = g_caps.back(); g_caps.pop_back(); }
      else g_caps[i].n = n;
      return;
    }
  }
  if (!erase) {
    if (g_caps.size() >= 256) g_caps.clear();
    g_caps.push_back(CapEntry{k, n});
  }
}

// Side stream of a caller stream: K1's colour half (SH -> RGB, the bulk of K1's bytes) runs there, beside the binning
// chain of the same view, and is joined before the forward compositor.  One side stream + two events per
// (device, caller stream), created on first use and kept.
struct SideStream { int dev; hipStream_t main, side; hipEvent_t fork, join; };
std::mutex g_side_mu;
std::vector<SideStream> g_sides;

bool side_stream_for(int dev, hipStream_t main, SideStream& out) {
  std::lock_guard<std::mutex> lk(g_side_mu);
  for (const SideStream& s : g_sides)
    if (s.dev == dev && s.main == main) { out = s; return true; }
  SideStream s{dev, main, nullptr, nullptr, nullptr};
  // lowest priority: its one kernel (SH -> RGB) has the whole binning chain's duration to finish in, and must not take
  // wave slots from the chain's short kernels, which sit on the view's critical path
  int least = 0, greatest = 0;
  static const int prio_env = [] { const char* e = getenv("GSR_SIDE_PRIORITY"); return e ? atoi(e) : 1; }();
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
  if (hipStreamCreateWithPriority(&s.side, hipStreamNonBlocking, prio_env ? least : 0) != hipSuccess ||
      hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  g_sides.push_back(s);
  out = s;
  return true;
}

// EXPERIMENT (GSR_COMP_CUMASK=w0,w1,... hex words): the two VALU-bound compositors K6 / K7 of a caller
// stream run on a companion stream restricted to the given compute units (hipExtStreamCreateWithCUMask), so that the
// HBM- and latency-bound kernels of OTHER views' streams always find the remaining CUs free of compositor waves.
// Two event hops per compositor launch.  Off unless the variable is set.
struct CompStream { int dev; hipStream_t main, comp; hipEvent_t in, out; };
std::mutex g_comp_mu;
std::vector<CompStream> g_comps;
static const std::vector<uint32_t>& comp_mask_env() {
  static const std::vector<uint32_t> m = [] {
    std::vector<uint32_t> v;
    const char* e = getenv("GSR_COMP_CUMASK");
    if (e) {
      const char* p = e;
      while (*p) {
        char* end = nullptr;
        const unsigned long w = strtoul(p, &end, 16);
        if (end == p) break;
        v.push_back((uint32_t)w);
        p = (*end == ',') ? end + 1 : end;
      }
    }
    return v;
  }();
  return m;
}
bool comp_stream_for(int dev, hipStream_t main, CompStream& out) {
  const std::vector<uint32_t>& mask = comp_mask_env();
  if (mask.empty()) return false;
  std::lock_guard<std::mutex> lk(g_comp_mu);
  for (const CompStream& s : g_comps)
    if (s.dev == dev && s.main == main) { out = s; return true; }
  CompStream s{dev, main, nullptr, nullptr, nullptr};
  if (hipExtStreamCreateWithCUMask(&s.comp, (uint32_t)mask.size(), mask.data()) != hipSuccess ||
      hipEventCreateWithFlags(&s.in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s.out, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  g_comps.push_back(s);
  out = s;
  return true;
}
// hop onto the compositor stream (returns the stream to launch on) and back
static hipStream_t comp_enter(int dev, hipStream_t st, CompStream& cs, bool& used) {
  used = comp_stream_for(dev, st, cs);
  if (!used) return st;
  if (hipEventRecord(cs.in, st) != hipSuccess || hipStreamWaitEvent(cs.comp, cs.in, 0) != hipSuccess) {
    (void)hipGetLastError();
    used = false;
    return st;
  }
  return cs.comp;
}
// (false: the join failed -- the caller's stream is NOT ordered behind the compositor stream, which is an error to report)
static bool comp_leave(hipStream_t st, const CompStream& cs, bool used) {
  if (!used) return true;
  const bool ok = hipEventRecord(cs.out, cs.comp) == hipSuccess && hipStreamWaitEvent(st, cs.out, 0) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  return ok;
}

// Count slots of forwards whose context was released before the copy landed (forward-only calls with an asynchronous
// count): harvested, without blocking, at the start of later forwards.
struct PendingSlot { CountSlot slot; CapKey key; };
std::mutex g_pending_mu;
std::vector<PendingSlot> g_pending;

void pending_harvest() {
  std::lock_guard<std::mutex> lk(g_pending_mu);
  for (size_t i = 0; i < g_pending.size();) {
    PendingSlot& ps = g_pending[i];
    if (slot_landed(ps.slot)) {
      const unsigned long long n = (unsigned long long)ps.slot.host[HS_N64] | ((unsigned long long)ps.slot.host[HS_N64 + 1] << 32);
      cap_store(ps.key, n, ps.slot.host[HS_OVF] != 0u);
      slot_put(ps.slot);
      g_pending[i] = g_pending.back();
      g_pending.pop_back();
    } else {
      ++i;
    }
  }
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
  // A batch of views rendered as one virtual scene (gsr_forward_raw_batch; gsr_kernels.hip.h, ViewDev): B views, view v owns
  // the virtual Gaussians [v * Ppad, v * Ppad + P) and the tiles [v * tpv, (v + 1) * tpv); Pv = B * Ppad, ntiles = B * tpv.
  // One view: B = 1, Ppad = Pv = P, tpv = ntiles, vpack = null.
  int B = 1, Ppad = 0, Pv = 0, tpv = 0;
  std::vector<GsrSettings> views;   // B > 1: the views' settings (device pointers owned by the caller); st = views[0]
  ViewDev* vpack = nullptr;
  // Pairs: `nbound` is what the host sized every pair-proportional buffer and grid for -- the exact count when the
  // forward waited for it, the capacity guess of an asynchronous-count forward otherwise; the device-side count lives
  // in dv[DV_N] and reaches the host through `slot` (n_known: already read).
  uint32_t nbound = 0;
  bool n_known = false, overflow = false;
  unsigned long long n64 = 0;
  CountSlot slot;
  hipStream_t fwd_stream = nullptr;   // the forward's stream (fallback of the count wait)
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
  uint32_t seg_shift = 0, rec_cap = 0;
  size_t keep_bytes = 0;
  float4 *G0 = nullptr, *G1 = nullptr, *G2 = nullptr;   // splat records, storage order (the compositors gather them)
  float* D = nullptr;             // [P,9] d rgb / d view direction (lane-group kernels, SH input, backward expected)
  double* abc = nullptr;          // [P] needle marks under GSR_FLAG_NEEDLE_DOUBLE (K1 -> K9; backward expected)
  bool lanegroup = false;         // K1 ran as k_pre_geom + k_pre_color: K8+K9 runs as k_pre_bwd
  uint32_t *order = nullptr, *off = nullptr, *offg = nullptr, *pair_rank = nullptr;
  uint2* ranges = nullptr;
  uint32_t* sched = nullptr;      // [ntiles] tiles longest-list-first + priority class
  float* final_T = nullptr;
  uint32_t* n_contrib = nullptr;
  uint32_t* dv = nullptr;         // device-side scalars of this forward (gsr_sort.hip.h: DV_*)
  // what gsr_ctx_rerender needs beyond the above
  bool async_count = false;       // the forward sized its pair buffers from a guess: K6 looks at dv[DV_OVF]
  bool fwd_only = false;          // kept by gsr_forward_raw2_keep: re-renderable, not differentiable
  bool has_b = false;             // two attribute segments (gsr_forward_raw2_keep): Gaussians >= P - segb.Pb read segb
  struct {
    int32_t Pb = 0;
    const float *xyz = nullptr, *features_dc = nullptr, *features_rest = nullptr, *objects_dc = nullptr,
                *opacity = nullptr, *scaling = nullptr, *rotation = nullptr;
  } b;
  bool objects_out = false;       // the forward composited the 16 object channels
  bool D_stale = false;           // the last re-render skipped d colour / d direction: no geometry backward until the next
  double* sumsq_out = nullptr;    // gsr_ctx_request_sumsq: where the next overwrite-mode raw backward leaves its six sums of squares
};

// Host copy of the forward's device-side scalars: waits for the (early) copy if it has not landed yet.
static int ctx_resolve_count(GsrCtx* c) {
  if (c->n_known || !c->slot.host) return GSR_OK;
  if (!slot_wait(c->slot, c->fwd_stream)) return GSR_ERR_DEVICE;
  c->n64 = (unsigned long long)c->slot.host[HS_N64] | ((unsigned long long)c->slot.host[HS_N64 + 1] << 32);
  c->overflow = c->slot.host[HS_OVF] != 0u;
  c->n_known = true;
  const CapKey key{c->dev, c->Pv, c->st.image_height, c->st.image_width};
  cap_store(key, c->n64, c->overflow);
  slot_put(c->slot);
  return GSR_OK;
}

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
  if (c->slot.host) {                  // the count was never looked at: take it if it has landed, else leave the slot
    if (slot_landed(c->slot)) {                            // to be harvested by a later forward (no host wait here)
      (void)ctx_resolve_count(c);
    } else {
      std::lock_guard<std::mutex> lk(g_pending_mu);
      g_pending.push_back(PendingSlot{c->slot, CapKey{c->dev, c->Pv, c->st.image_height, c->st.image_width}});
      c->slot = CountSlot{};
    }
  }
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

// K6 of a context whose binning (pair list, tile ranges, schedule, boundary-record plan) and splat records are in place:
// the last launch of a forward, and all that a re-render of a kept context needs behind the colour kernel.
static int launch_render_fwd(GsrCtx* c, float* out_color, float* out_objects, hipStream_t st_main) {
  const GsrSettings* s = &c->st;
  CompStream comp_s{};
  bool comp_used = false;
  hipStream_t st = comp_enter(c->dev, st_main, comp_s, comp_used);
  const int W = s->image_width, H = s->image_height, ntiles = c->ntiles;
  const size_t HW = (size_t)H * W;
  const float* sh_objs = out_objects ? c->sh_objs : nullptr;
  RenderArgs ra;
  ra.ranges = c->ranges; ra.pair_rank = c->pair_rank; ra.R0 = c->G0; ra.R1 = c->G1; ra.R2 = c->G2;
  ra.sh_objs = sh_objs; ra.bg = s->bg; ra.W = W; ra.H = H; ra.gridx = c->gridx; ra.ntiles = ntiles;
  ra.sh_objs_b = c->has_b ? c->b.objects_dc : nullptr; ra.Pa = c->has_b ? c->P - c->b.Pb : c->P;
  ra.map_mode = flag_tile_map(s->flags);
  ra.sched = c->sched;
  ra.dv = c->async_count ? c->dv : nullptr;
  ra.wave_clock = g_wave_clock_fwd.load();
  ra.bnd = c->bnd; ra.segoff = c->bnd ? c->segoff : nullptr; ra.seg_shift = c->bnd ? c->seg_shift : 0u;
  ra.out_color = out_color; ra.out_objects = out_objects; ra.final_T = c->final_T; ra.n_contrib = c->n_contrib;
  ra.tpv = c->tpv; ra.vpack = c->vpack;
  const dim3 blkT(64);
  // pixels per lane of K6: fewer = more, shorter waves per tile (see k_render_fwd); images with fewer tiles than
  // half the chip's wave slots are split down to one 16x4 strip per wave.  GSR_FLAG_FWD_SPLIT(n) overrides.
  // From 32 000 tiles on (a 4K image, a batch of four or more 1080p views) one wave per tile: four rounds of waves fill the
  // chip either way, and a tile's list is then staged once instead of once per half tile (S-airport-4K K6 0.215 -> 0.207 ms,
  // an 8-view batch of S-nyc-1M 0.140 -> 0.129 ms per view).  Below that the longer work items cost more in the kernel's
  // tail than they save: 8160 tiles 0.196 -> 0.268 ms, an 8-view batch of S-hydrant-full (20 000 tiles) 0.042 -> 0.061.
  const int fwd_npx = flag_fwd_npx(s->flags) ? flag_fwd_npx(s->flags) : (ntiles < 4096 ? 1 : (ntiles < 32000 ? 2 : 4));
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
    if (out_objects) {
      hipError_t e = hipMemsetAsync(out_objects, 0, sizeof(float) * NUM_OBJ * HW, st);
      if (e != hipSuccess) {
        (void)comp_leave(st_main, comp_s, comp_used);
        return set_err(GSR_ERR_DEVICE, "objects: %s", hipGetErrorString(e));
      }
    }
    if (fwd_npx == 4) hipLaunchKernelGGL((k_render_fwd<false, 4>), gridT, blkT, 0, st, ra);
    else if (fwd_npx == 2 && k6_shared) hipLaunchKernelGGL((k_render_fwd<false, 2, 2>), gridS, blkS2, 0, st, ra);
    else if (fwd_npx == 2) hipLaunchKernelGGL((k_render_fwd<false, 2>), gridT, blkT, 0, st, ra);
    else if (k6_shared) hipLaunchKernelGGL((k_render_fwd<false, 1, 4>), gridS, blkS4, 0, st, ra);
    else hipLaunchKernelGGL((k_render_fwd<false, 1>), gridT, blkT, 0, st, ra);
  }
  hipError_t e = hipGetLastError();
  const bool joined = comp_leave(st_main, comp_s, comp_used);
  if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "render forward: launch failed: %s", hipGetErrorString(e));
  if (!joined) return set_err(GSR_ERR_DEVICE, "render forward: joining the compositor stream failed");
  return GSR_OK;
}

static int forward_impl(const GsrSettings* s, int32_t P, int32_t K, const float* means3D, const float* shs,
                        const float* sh_dc, const float* sh_objs, const float* colors_precomp, const float* opacities,
                        const float* scales, const float* rotations, const float* cov3D_precomp, float* out_color,
                        float* out_objects, int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered, void* stream,
                        bool raw, const SegB* segb = nullptr, bool fwd_only = false, int nviews = 1) {
  // nviews > 1 (gsr_forward_raw_batch): `s` points at nviews settings; the batch is one virtual scene (GsrCtx::B)
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
  // objects not wanted -- unless the features are to be kept for the backward (GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY)
  if (out_objects == nullptr && sh_objs != nullptr && !(s->flags & GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY)) sh_objs = nullptr;

  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = cur_dev();
  const int H = s->image_height, W = s->image_width;
  const int gridx = (W + TILE - 1) / TILE, gridy = (H + TILE - 1) / TILE;
  const int tpv = gridx * gridy;                                       // tiles per view
  if (gridx > 4095 || gridy > 4095) return set_err(GSR_ERR_INVALID, "gsr_forward: image larger than 65520 px per side");
  // a batch: B views as one virtual scene of B * Ppad Gaussians and B * tpv tiles (a batch of one is an ordinary forward)
  const int B = (nviews > 1 && P > 0) ? nviews : 1;
  const int Ppad = B > 1 ? (int)(((size_t)P + BATCH_PAD - 1) / BATCH_PAD * BATCH_PAD) : P;
  if ((size_t)B * (size_t)Ppad > (size_t)RANK_MASK)
    return set_err(GSR_ERR_INVALID, "gsr_forward: more than 2^28 Gaussians (times views of a batch)");
  if ((size_t)B * (size_t)tpv >= (size_t)SCHED_TILE_MASK) return set_err(GSR_ERR_INVALID, "gsr_forward: more than 2^28 tiles in the batch");
  const int Pv = B > 1 ? B * Ppad : P;                                 // virtual Gaussians
  const int ntiles = B * tpv;                                          // virtual tiles
  const size_t HW = (size_t)H * W * (size_t)B;                         // pixels of all views

  GsrCtx* c = new (std::nothrow) GsrCtx();
  if (!c) return set_err(GSR_ERR_NOMEM, "gsr_forward: host allocation failed");
  c->dev = dev; c->st = *s; c->P = P; c->K = K; c->gridx = gridx; c->gridy = gridy; c->ntiles = ntiles;
  c->B = B; c->Ppad = Ppad; c->Pv = Pv; c->tpv = tpv;
  if (B > 1) c->views.assign(s, s + B);
  c->means3D = means3D; c->shs = shs; c->sh_objs = sh_objs; c->colors = colors_precomp; c->opac = opacities;
  c->scales = scales; c->rots = rotations; c->cov3d = cov3D_precomp;
  c->raw = raw; c->sh_dc = sh_dc;
  c->fwd_only = fwd_only; c->objects_out = out_objects != nullptr && sh_objs != nullptr;
  if (segb) {
    c->has_b = true;
    c->b.Pb = segb->Pb; c->b.xyz = segb->xyz; c->b.features_dc = segb->features_dc; c->b.features_rest = segb->features_rest;
    c->b.objects_dc = segb->objects_dc; c->b.opacity = segb->opacity; c->b.scaling = segb->scaling; c->b.rotation = segb->rotation;
  }

  const size_t Pp = (size_t)std::max(Pv, 1);
  pending_harvest();
  // Asynchronous pair count (GSR_FLAG_ASYNC_COUNT, or GSR_ASYNC_COUNT=1 in the environment): the host does not wait
  // for the pair count; buffers and grids are sized from the count an earlier forward of the same (P, H, W) saw, with
  // head-room.  Without such an entry this forward counts synchronously (and leaves the entry behind).
  static const int async_env = [] { const char* e = getenv("GSR_ASYNC_COUNT"); return e ? atoi(e) : 0; }();
  unsigned long long cap_pairs = MAX_PAIRS - 1;
  bool async_count = false;
  if (P > 0 && (async_env != 0 || (s->flags & GSR_FLAG_ASYNC_COUNT))) {
    unsigned long long seen = 0;
    if (cap_lookup(CapKey{dev, Pv, H, W}, seen)) {
      cap_pairs = std::min<unsigned long long>(seen + seen / 4 + 65536ull, MAX_PAIRS - 1);
      async_count = true;
    }
  }
  // ---- kept slab ---------------------------------------------------------------------------
  SlabPlan kp;
  kp.add<float4>(REC * Pp);   // G records (storage order)
  kp.add<uint32_t>(Pp); kp.add<uint32_t>(Pp + 1); kp.add<uint32_t>(Pp + 1);   // order, off, offg
  kp.add<uint2>(ntiles); kp.add<float>(HW); kp.add<uint32_t>(HW); kp.add<uint32_t>(DV_WORDS); kp.add<uint32_t>(ntiles);
  // the SH layouts the reference uses (and precomputed colours) take the lane-group kernels
  c->lanegroup = raw || (shs && K == 16) || colors_precomp != nullptr;
  const bool want_D = c->lanegroup && shs != nullptr && ctx_out != nullptr && !fwd_only;
  if (want_D) kp.add<float>(9 * Pp);
  const bool needle_double = (s->flags & GSR_FLAG_NEEDLE_DOUBLE) != 0u;
  const bool want_abc = needle_double && c->lanegroup && ctx_out != nullptr && !fwd_only;
  if (want_abc) kp.add<double>(Pp);
  if (B > 1) kp.add<ViewDev>((size_t)B);
  c->keep_bytes = kp.bytes + 256;
  c->keep_blk = pool_alloc(dev, c->keep_bytes, st);
  // ---- scratch slab (released at the end of forward) ----------------------------------------
  const uint32_t nbP = (uint32_t)((Pp + DCHUNK - 1) / DCHUNK);             // chunks of the storage scan / depth sort
  const uint32_t nk1 = (uint32_t)((Pp + PREG_BLOCK - 1) / PREG_BLOCK);     // workgroups of K1's geometry half (all views)
  const uint32_t nk1v = B > 1 ? (uint32_t)(Ppad / PREG_BLOCK) : nk1;       // ... per view
  const uint32_t nkc = (uint32_t)(((size_t)std::max(P, 1) + PREF_BLOCK - 1) / PREF_BLOCK);   // ... of its colour half, per view
  // the depth sort: three passes of <= 11 bits (a latency-bound million keys) or four of <= 8 (several million: throughput)
  static const int depth_passes_env = [] { const char* e = getenv("GSR_DEPTH_PASSES"); int v = e ? atoi(e) : 0; return (v == 3 || v == 4) ? v : 0; }();
  const int depth_passes = depth_passes_env ? depth_passes_env : (Pp >= (size_t(3) << 20) ? 4 : 3);
  const bool group_sums = nk1 > K1_GROUP_MIN;                              // K2 reads K1's sums through group sums (k_bout_group_sum)
  const uint32_t ngroups = (nk1 + K1_GROUP - 1) / K1_GROUP;
  SlabPlan sp;
  sp.add<uint32_t>(Pp); sp.add<uint32_t>(Pp); sp.add<uint32_t>(Pp); sp.add<uint32_t>(Pp); sp.add<uint32_t>(Pp);   // dkey, k1, vtmp, v2, tcnt
  sp.add<uint32_t>((size_t)RS_BINS_DEV * nbP); sp.add<uint32_t>(RS_BINS_DEV);
  sp.add<uint4>(nk1); sp.add<uint32_t>(nbP + 2);
  if (group_sums) sp.add<uint4>(ngroups);
  sp.add<uint8_t>(Pp);
  void* scratch_blk = pool_alloc(dev, sp.bytes + 256, st);
  if (!c->keep_blk || !scratch_blk) {
    pool_free(dev, scratch_blk);
    gsr_ctx_free(c);
    return set_err(GSR_ERR_NOMEM, "gsr_forward: workspace allocation failed (P=%d, %dx%d)", P, W, H);
  }
  Slab ks{static_cast<char*>(c->keep_blk), c->keep_bytes, 0};
  c->G0 = ks.take<float4>(REC * Pp); c->G1 = c->G0 + 1; c->G2 = c->G0 + 2;   // interleaved records, REC float4 apart
  c->order = ks.take<uint32_t>(Pp); c->off = ks.take<uint32_t>(Pp + 1); c->offg = ks.take<uint32_t>(Pp + 1);
  c->ranges = ks.take<uint2>(ntiles); c->final_T = ks.take<float>(HW); c->n_contrib = ks.take<uint32_t>(HW);
  c->dv = ks.take<uint32_t>(DV_WORDS);
  c->sched = ks.take<uint32_t>(ntiles);
  if (want_D) c->D = ks.take<float>(9 * Pp);
  if (want_abc) c->abc = ks.take<double>(Pp);
  if (B > 1) c->vpack = ks.take<ViewDev>((size_t)B);
  Slab ss{static_cast<char*>(scratch_blk), sp.bytes + 256, 0};
  float4* G0 = c->G0; float4* G1 = c->G1; float4* G2 = c->G2;
  uint32_t* dkey = ss.take<uint32_t>(Pp); uint32_t* k1 = ss.take<uint32_t>(Pp); uint32_t* vtmp = ss.take<uint32_t>(Pp);
  uint32_t* v2 = ss.take<uint32_t>(Pp); uint32_t* tcnt = ss.take<uint32_t>(Pp);
  uint32_t* table = ss.take<uint32_t>((size_t)RS_BINS_DEV * nbP); uint32_t* tsums = ss.take<uint32_t>(RS_BINS_DEV);
  uint4* bout = ss.take<uint4>(nk1);
  uint32_t* psums = ss.take<uint32_t>(nbP + 2);
  uint4* gsum = group_sums ? ss.take<uint4>(ngroups) : nullptr;
  uint8_t* tcnt8 = ss.take<uint8_t>(Pp);

  void* pairs_blk[4] = {nullptr, nullptr, nullptr, nullptr};
  void* tbl_blk = nullptr;
  SideStream side{};
  bool side_used = false;
  auto fail = [&](int code) {
    if (side_used) (void)hipStreamWaitEvent(st, side.join, 0);   // nothing of this forward may outlive its workspace
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
  const dim3 blkPre(PREG_BLOCK), gridPre(nk1v), blkCol(PREF_BLOCK), gridCol(nkc);
  const int cull = (s->flags & GSR_FLAG_NO_CULL) ? 0 : 1;
  uint32_t nbound = 0;
  // K1's colour half: on the side stream unless the caller turned that off (GSR_FLAG_NO_SIDE_STREAM / GSR_SIDE_STREAM=0).
  // GSR_FORK_LATE=1 (experiment) starts it behind the depth sort instead of behind the geometry half.
  PreArgs color_pa[MAX_BATCH] = {};
  bool want_color = false;
  static const int fork_late = [] { const char* e = getenv("GSR_FORK_LATE"); return e ? atoi(e) : 0; }();
  auto launch_color = [&]() -> int {
    static const int side_env = [] { const char* e = getenv("GSR_SIDE_STREAM"); return e ? atoi(e) : 1; }();
    hipStream_t cs = st;
    if (side_env != 0 && !(s->flags & GSR_FLAG_NO_SIDE_STREAM) && side_stream_for(dev, st, side)) {
      F_TRY("side stream", hipEventRecord(side.fork, st));
      F_TRY("side stream", hipStreamWaitEvent(side.side, side.fork, 0));
      cs = side.side;
      side_used = true;
    }
    // beside the chain: a thin grid (GSR_COLOR_BLOCKS workgroups looping over the chunks); alone: one per chunk
    static const int col_blocks = [] { const char* e = getenv("GSR_COLOR_BLOCKS"); int v = e ? atoi(e) : 512; return v > 0 ? v : 512; }();
    const dim3 gridC(side_used ? std::min<unsigned>(gridCol.x, (unsigned)col_blocks) : gridCol.x);
    // a batch of views: ONE launch reads every SH row once for all the views that see the Gaussian (GSR_BATCH_COLOR=0:
    // the single-view kernel once per view -- the A/B form)
    static const int batch_color_env = [] { const char* e = getenv("GSR_BATCH_COLOR"); return e ? atoi(e) : 1; }();
    if (B > 1 && raw && batch_color_env != 0) {
      PreColorBatchArgs ca;
      ca.P = P; ca.B = B; ca.Ppad = Ppad; ca.deg = s->sh_degree; ca.vpack = c->vpack; ca.means = means3D; ca.sh = shs; ca.sh_dc = sh_dc;
      ca.Pa = segb ? P - segb->Pb : P; ca.means_b = segb ? segb->xyz : nullptr;
      ca.sh_b = segb ? segb->features_rest : nullptr; ca.sh_dc_b = segb ? segb->features_dc : nullptr;
      ca.tcnt = tcnt; ca.offg = nullptr; ca.G1 = G1; ca.G2 = G2; ca.D = c->D;
      hipLaunchKernelGGL(k_pre_color_batch, gridC, blkCol, 0, cs, ca);
    } else {
      for (int v = 0; v < B; ++v) {
        if (raw) hipLaunchKernelGGL((k_pre_color<true>), gridC, blkCol, 0, cs, color_pa[v]);
        else hipLaunchKernelGGL((k_pre_color<false>), gridC, blkCol, 0, cs, color_pa[v]);
      }
    }
    if (side_used) F_TRY("side stream", hipEventRecord(side.join, side.side));
    return GSR_OK;
  };
  if (P > 0) {
    {
      StageTimer t(GSR_STAGE_PREPROCESS, st);
      PreBlockOut bo;
      bo.bout = bout;
      bo.ranges = P >= tpv ? c->ranges : nullptr;         // the tile ranges are cleared by K1's first threads
      bo.ntiles = tpv;
      if (!bo.ranges) {                                   // fewer Gaussians than tiles: spans (0xFFFFFFFF, 0) by two fills
        F_TRY("ranges", hipMemset2DAsync(c->ranges, sizeof(uint2), 0xFF, sizeof(uint32_t), ntiles, st));
        F_TRY("ranges", hipMemset2DAsync(reinterpret_cast<char*>(c->ranges) + sizeof(uint32_t), sizeof(uint2), 0, sizeof(uint32_t), ntiles, st));
      }
      if (B > 1) {
        // the views' constants in one device array (the compositors find a tile's background there)
        ViewPtrs vp{};
        for (int v = 0; v < B; ++v) {
          vp.vm[v] = s[v].viewmatrix; vp.pm[v] = s[v].projmatrix; vp.cam[v] = s[v].campos; vp.bg[v] = s[v].bg;
          vp.tanfovx[v] = s[v].tanfovx; vp.tanfovy[v] = s[v].tanfovy;
        }
        hipLaunchKernelGGL(k_pack_views, dim3(B), dim3(64), 0, st, vp, B, c->vpack);
      }
      if (c->lanegroup) {
        // (a batch: one launch per view over that view's padded range of the virtual scene -- the same kernels, their
        // per-Gaussian arrays offset by v * Ppad; everything behind K1 then runs once over the B * Ppad virtual Gaussians)
        // (a batch: ONE launch of the geometry kernel over the B views' padded ranges -- view = workgroup / workgroups per
        // view; everything behind K1 then runs once over the B * Ppad virtual Gaussians.  The colour kernel still runs once
        // per view, its per-Gaussian arrays offset by v * Ppad.)
        PreArgs pa;
        pa.P = P; pa.Pfill = B > 1 ? Ppad : P; pa.va = va;
        pa.vpack = B > 1 ? c->vpack : nullptr; pa.bpv = (int)nk1v;
        pa.means = means3D; pa.scales = scales; pa.rots = rotations; pa.cov3d = cov3D_precomp;
        pa.opac = opacities; pa.sh = shs; pa.sh_dc = sh_dc; pa.colors = colors_precomp; pa.radii = radii;
        pa.G0 = G0; pa.G1 = G1; pa.G2 = G2; pa.D = c->D; pa.dkey = dkey; pa.tcnt = tcnt; pa.offg = nullptr;
        pa.tcnt8 = tcnt8;
        pa.abc = c->abc;
        pa.Pa = segb ? P - segb->Pb : P;
        pa.means_b = segb ? segb->xyz : nullptr; pa.scales_b = segb ? segb->scaling : nullptr;
        pa.rots_b = segb ? segb->rotation : nullptr; pa.opac_b = segb ? segb->opacity : nullptr;
        pa.sh_b = segb ? segb->features_rest : nullptr; pa.sh_dc_b = segb ? segb->features_dc : nullptr;
        pa.cull = cull; pa.bo = bo;
        const dim3 gridG(nk1);
        if (needle_double) {
          if (raw) hipLaunchKernelGGL((k_pre_geom<true, true>), gridG, blkPre, 0, st, pa);
          else hipLaunchKernelGGL((k_pre_geom<false, true>), gridG, blkPre, 0, st, pa);
        } else {
          if (raw) hipLaunchKernelGGL((k_pre_geom<true>), gridG, blkPre, 0, st, pa);
          else hipLaunchKernelGGL((k_pre_geom<false>), gridG, blkPre, 0, st, pa);
        }
        for (int v = 0; v < B; ++v) {
          const size_t o = (size_t)v * (size_t)Ppad;
          color_pa[v] = pa;
          color_pa[v].va = view_args(s[v]); color_pa[v].vpack = nullptr;
          color_pa[v].G0 = G0 + REC * o; color_pa[v].G1 = G1 + REC * o; color_pa[v].G2 = G2 + REC * o;
          color_pa[v].D = c->D ? c->D + 9 * o : nullptr; color_pa[v].tcnt = tcnt + o;
        }
        want_color = !colors_precomp;
        if (want_color && !fork_late) { const int rcol = launch_color(); if (rcol != GSR_OK) return rcol; }
      } else
        hipLaunchKernelGGL(k_preprocess, gridPre, blkPre, 0, st, P, K, va, (cull ? 1 : 0) | (needle_double ? 2 : 0), means3D, scales, rotations, cov3D_precomp,
                           opacities, shs, colors_precomp, radii, G0, G1, G2, dkey, tcnt, bo);
      // storage-order numbering of the pairs (where the backward puts its partial rows), the depth sort's digit width
      // and first histogram, the device-side pair count -- published to the host slot by the kernel itself: one launch
      // stream capture (hipGraph): the forward must not wait for anything, so the count has to be asynchronous, and it is
      // not published to the host at all (a replayed graph would keep writing into a slot that has long been recycled)
      hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
      const bool capturing = hipStreamIsCapturing(st, &cap_st) == hipSuccess && cap_st == hipStreamCaptureStatusActive;
      if (capturing && !async_count)
        return fail(set_err(GSR_ERR_STATE, "gsr_forward: a stream capture needs GSR_FLAG_ASYNC_COUNT and an earlier forward of "
                            "the same (P, H, W) on this device (the pair count cannot be waited for while capturing)"));
      if (!capturing && !slot_get(c->slot)) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: pinned host slot allocation failed"));
      c->fwd_stream = st;
      if (group_sums) hipLaunchKernelGGL(k_bout_group_sum, dim3(ngroups), dim3(K1_GROUP), 0, st, (const uint4*)bout, nk1, gsum);
      hipLaunchKernelGGL(k_storage_scan_hist, dim3(nbP), blk, 0, st, (uint32_t)Pv, (const uint32_t*)tcnt, (const uint32_t*)dkey,
                         (const uint4*)bout, (const uint4*)gsum, c->offg, table, nbP, c->dv, cap_pairs, c->slot.dev, c->slot.token,
                         (uint32_t)depth_passes);
      F_LAUNCH("preprocess");
    }
    {
      StageTimer t(GSR_STAGE_DEPTH_SORT, st);
      // Stable argsort of the live depth keys in three passes whose digit width the device chose from the keys' range:
      // order[r] = Gaussian of depth rank r, r < V = dv[DV_V].  Pass 0 drops the Gaussians that emit nothing (its
      // histogram came from k_storage_scan_hist), the last pass also gathers cnt[r] = tiles touched by rank r.
      const uint32_t* nV = c->dv + DV_V;
      DigitSpec d0{c->dv, 0, 0, 0u}, d1{c->dv, 1, 0, 0u}, d2{c->dv, 2, 0, 0u};
      if (depth_passes == 4) {
        // millions of keys (a batch of views, a large scene): four passes of <= 8 bits through the 256-bin kernels; passes
        // 1-3 in the chunk size the tile sort would pick for that many keys
        DigitSpec d3{c->dv, 3, 0, 0u};
        const int rn = radix_rounds_for((uint32_t)Pv);
        radix_pass<RS_BINS>(dkey, nullptr, k1, vtmp, (uint32_t)Pv, nullptr, d0, DROUNDS, table, tsums, false, 1, 1,
                            c->dv + DV_V, nullptr, nullptr, st);
        radix_pass<RS_BINS>(k1, vtmp, dkey, v2, (uint32_t)Pv, nV, d1, rn, table, tsums, true, 0, 0, nullptr, nullptr, nullptr, st);
        radix_pass<RS_BINS>(dkey, v2, k1, vtmp, (uint32_t)Pv, nV, d2, rn, table, tsums, true, 0, 0, nullptr, nullptr, nullptr, st);
        radix_pass<RS_BINS>(k1, vtmp, nullptr, c->order, (uint32_t)Pv, nV, d3, rn, table, tsums, true, 0, 0, nullptr, tcnt, v2, st,
                            nullptr, 0, c->lanegroup ? tcnt8 : nullptr);
      } else {
      radix_pass<RS_BINS_DEV>(dkey, nullptr, k1, vtmp, (uint32_t)Pv, nullptr, d0, DROUNDS, table, tsums, false, 1, 1,
                              c->dv + DV_V, nullptr, nullptr, st);
      radix_pass<RS_BINS_DEV>(k1, vtmp, dkey, v2, (uint32_t)Pv, nV, d1, DROUNDS, table, tsums, true, 0, 0, nullptr, nullptr,
                              nullptr, st);
      radix_pass<RS_BINS_DEV>(dkey, v2, nullptr, c->order, (uint32_t)Pv, nV, d2, DROUNDS, table, tsums, true, 0, 0, nullptr,
                              tcnt, vtmp, st, nullptr, 0, c->lanegroup ? tcnt8 : nullptr);
      }
      F_LAUNCH("depth sort");
    }
    if (want_color && fork_late) { const int rcol = launch_color(); if (rcol != GSR_OK) return rcol; }
    if (!async_count) {
      // (the rank-order scan is enqueued after this wait: it records the owners of the emission chunks' first slots, an
      // array sized by N; the depth sort keeps the GPU busy well past the host's wake-up)
      if (ctx_resolve_count(c) != GSR_OK) return fail(set_err(GSR_ERR_DEVICE, "gsr_forward: reading the pair count failed"));
      if (c->n64 >= MAX_PAIRS)   // NSUB * N must stay below 2^32
        return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: %llu (tile, Gaussian) pairs exceed the supported %llu "
                            "(splats cover too many tiles: check scales / scale_modifier)", c->n64, MAX_PAIRS));
      nbound = (uint32_t)c->n64;
    } else {
      nbound = (uint32_t)cap_pairs;
    }
  } else {
    F_TRY("init", hipMemsetAsync(c->off, 0, sizeof(uint32_t), st));
    F_TRY("init", hipMemsetAsync(c->offg, 0, sizeof(uint32_t), st));
    F_TRY("init", hipMemsetAsync(c->dv, 0, sizeof(uint32_t) * DV_WORDS, st));
    F_TRY("ranges", hipMemsetAsync(c->ranges, 0, sizeof(uint2) * ntiles, st));
    c->n_known = true;
  }
  c->nbound = nbound;
  c->async_count = async_count;
  if (nbound == 0 && P > 0) F_TRY("init", hipMemsetAsync(c->off, 0, sizeof(uint32_t), st));
  if (nbound > 0) {
    const int rounds = radix_rounds_for(nbound);
    const uint32_t chunkN = (uint32_t)rs_chunk(rounds);
    const uint32_t nbN = (nbound + chunkN - 1) / chunkN;
    const uint32_t tblN = RS_BINS * nbN;
    for (int i = 0; i < 4; ++i) {
      pairs_blk[i] = pool_alloc(dev, sizeof(uint32_t) * (size_t)nbound, st);
      if (!pairs_blk[i]) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: pair buffers (N=%u) allocation failed", nbound));
    }
    const uint32_t ngrain = nbound / EMIT_GRAIN + 4;            // chunk_first entries
    tbl_blk = pool_alloc(dev, sizeof(uint32_t) * ((size_t)tblN + RS_BINS + ngrain), st);
    if (!tbl_blk) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: sort table allocation failed"));
    uint32_t* tileA = static_cast<uint32_t*>(pairs_blk[0]); uint32_t* rankA = static_cast<uint32_t*>(pairs_blk[1]);
    uint32_t* tileB = static_cast<uint32_t*>(pairs_blk[2]); uint32_t* rankB = static_cast<uint32_t*>(pairs_blk[3]);
    uint32_t* tableN = static_cast<uint32_t*>(tbl_blk);
    uint32_t* tsumsN = tableN + tblN;
    uint32_t* chunk_first = tsumsN + RS_BINS;
    const int tile_bits = ceil_log2((uint32_t)ntiles + 1);   // keys are 0..ntiles (ntiles = culled pair)
    {
      StageTimer t(GSR_STAGE_BIN, st);
      // off[r] = pairs emitted by the ranks in front of r, off[V] = their total; chunk_first[c] = rank that owns slot
      // c * EMIT_GRAIN
      // (two launches: block sums, then carry + local scan.  Round 3's one-launch variant, in which a block waited for its
      // predecessors' published sums, saved 2.4 us and could, in principle, give up waiting with nothing but a poisoned
      // image to show for it: removed)
      scan_exclusive_u32(depth_passes == 4 ? v2 : vtmp, c->off, (uint32_t)Pv, c->dv + DV_V, psums, st, chunk_first, (uint32_t)EMIT_GRAIN, ngrain);
      F_LAUNCH("rank scan");
      int sh0; uint32_t mask0;
      radix_first_digit(tile_bits, sh0, mask0);
      // a batch: the view of a virtual Gaussian is g / Ppad (one multiply by ceil(2^32 / Ppad) and one correction)
      const uint32_t e_ppad = B > 1 ? (uint32_t)Ppad : 0u;
      const uint32_t e_magic = B > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)Ppad - 1ull) / (uint64_t)Ppad) : 0u;
      if (rounds == RS_ROUNDS_MIN)
        hipLaunchKernelGGL((k_emit<RS_ROUNDS_MIN>), dim3(nbN), dim3(rs_chunk(RS_ROUNDS_MIN) / EMIT_PER_THREAD), 0, st, (const uint32_t*)c->off, (const uint32_t*)c->order,
                           (const uint32_t*)chunk_first, (const uint32_t*)c->dv, (const float4*)c->G0, (const float4*)c->G1, (const float4*)c->G2, gridx, W, H,
                           (uint32_t)ntiles, cull, tileA, rankA, tableN, nbN, mask0, e_ppad, e_magic, (uint32_t)tpv);
      else
        hipLaunchKernelGGL((k_emit<RS_ROUNDS_MAX>), dim3(nbN), dim3(rs_chunk(RS_ROUNDS_MAX) / EMIT_PER_THREAD), 0, st, (const uint32_t*)c->off, (const uint32_t*)c->order,
                           (const uint32_t*)chunk_first, (const uint32_t*)c->dv, (const float4*)c->G0, (const float4*)c->G1, (const float4*)c->G2, gridx, W, H,
                           (uint32_t)ntiles, cull, tileA, rankA, tableN, nbN, mask0, e_ppad, e_magic, (uint32_t)tpv);
      F_LAUNCH("emit");
    }
    int res;
    {
      StageTimer t(GSR_STAGE_TILE_SORT, st);
      res = radix_sort_pairs(tileA, rankA, tileB, rankB, nbound, c->dv + DV_N, 0, tile_bits, false, tableN, tsumsN, st, true,
                             reinterpret_cast<uint32_t*>(c->ranges), (uint32_t)ntiles);
      F_LAUNCH("tile sort");
    }
    c->pair_rank = res ? rankB : rankA;
    c->rank_blk = res ? pairs_blk[3] : pairs_blk[1];
    pool_free(dev, tbl_blk);
    tbl_blk = nullptr;
    for (void* b : pairs_blk) if (b != c->rank_blk) pool_free(dev, b);
    for (void*& b : pairs_blk) b = nullptr;
  }
  if (side_used) F_TRY("side stream", hipStreamWaitEvent(st, side.join, 0));   // the colours are in place from here on
  {
    StageTimer t(GSR_STAGE_RENDER_FWD, st);
    // Long tile lists are split into segments for the backward (gsr_kernels.hip.h, "Segments"): the forward stores the
    // per-pixel (T, C) at the segment boundaries -- with object channels too (round 5): a backward without dL/dobjects then
    // still walks segments; one WITH dL/dobjects ignores the records (the 16 running object sums are not stored).  Not
    // for a forward-only call, not under GSR_FLAG_NO_SEGMENTS.
    static const int seg_shift_env = [] { const char* e = getenv("GSR_SEG_SHIFT"); int v = e ? atoi(e) : 8; return (v >= 6 && v <= 16) ? v : 8; }();
    if (nbound > 0 && ctx_out && !fwd_only && !(s->flags & GSR_FLAG_NO_SEGMENTS)) {
      const uint32_t per = nbound >> seg_shift_env;
      c->seg_shift = (uint32_t)seg_shift_env;
      // sum over split tiles of ceil(len / seg) <= N / seg + min(T, N / seg): every split tile's records always fit
      c->rec_cap = per + std::min<uint32_t>((uint32_t)ntiles, per) + 1u;
      SlabPlan gp;
      gp.add<float4>((size_t)c->rec_cap * PXL * 64); gp.add<uint2>(c->rec_cap); gp.add<uint32_t>(ntiles);
      c->seg_blk = pool_alloc(dev, gp.bytes + 256, st);
      if (!c->seg_blk) return fail(set_err(GSR_ERR_NOMEM, "gsr_forward: segment boundary buffer (N=%u) allocation failed", nbound));
      Slab gs{static_cast<char*>(c->seg_blk), gp.bytes + 256, 0};
      c->bnd = gs.take<float4>((size_t)c->rec_cap * PXL * 64); c->rec_item = gs.take<uint2>(c->rec_cap);
      c->segoff = gs.take<uint32_t>(ntiles);
    }
    // always: it also turns the empty spans the tile sort left untouched into (0, 0)
    // (more than 64 KB of dynamic LDS at 4K: a single workgroup may use all of the CU's 160 KB on gfx950)
    // If the limit cannot be raised the lengths of the tiles beyond the default 64 KB's worth are read from HBM instead.
    static const int sched_lds_cap = [] {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_schedule),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SCHED_LDS_TILES);
      (void)hipGetLastError();
      return e == hipSuccess ? SCHED_LDS_TILES : SCHED_LDS_TILES_DEFAULT;
    }();
    static const int batch_sched_env = [] { const char* e = getenv("GSR_BATCH_SCHED"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL(k_tile_schedule, dim3((c->segoff ? 2 : 1) * B), dim3(1024), sched_lds_bytes(tpv, sched_lds_cap), st, tpv,
                       sched_lds_cap, c->ranges, c->sched, c->seg_shift, c->segoff, c->rec_item, c->rec_cap, c->dv + DV_NREC, B,
                       batch_sched_env);
    F_LAUNCH("tile schedule");
    const int rk6 = launch_render_fwd(c, out_color, out_objects, st);
    if (rk6 != GSR_OK) return fail(rk6);
  }
  pool_free(dev, scratch_blk);
  if (num_rendered) *num_rendered = c->n_known ? (int64_t)c->n64 : (int64_t)-1;   // -1: not known yet (asynchronous count)
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

int gsr_forward_raw2_keep(const GsrSettings* s, int32_t Pa, const float* xyz_a, const float* features_dc_a,
                          const float* features_rest_a, const float* objects_dc_a, const float* opacity_logit_a,
                          const float* log_scaling_a, const float* rotation_raw_a, int32_t Pb, const float* xyz_b,
                          const float* features_dc_b, const float* features_rest_b, const float* objects_dc_b,
                          const float* opacity_logit_b, const float* log_scaling_b, const float* rotation_raw_b,
                          float* out_color, float* out_objects, int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered,
                          void* stream) {
  if (Pa < 0 || Pb < 0 || (long long)Pa + Pb > 0x7FFFFFFFll) return set_err(GSR_ERR_INVALID, "gsr_forward_raw2: bad sizes Pa=%d Pb=%d", Pa, Pb);
  if (Pb == 0)
    return forward_impl(s, Pa, 16, xyz_a, features_rest_a, features_dc_a, objects_dc_a, nullptr, opacity_logit_a,
                        log_scaling_a, rotation_raw_a, nullptr, out_color, out_objects, radii, ctx_out, num_rendered, stream, true,
                        nullptr, true);
  if (Pa == 0)
    return forward_impl(s, Pb, 16, xyz_b, features_rest_b, features_dc_b, objects_dc_b, nullptr, opacity_logit_b,
                        log_scaling_b, rotation_raw_b, nullptr, out_color, out_objects, radii, ctx_out, num_rendered, stream, true,
                        nullptr, true);
  if (!xyz_a || !features_dc_a || !features_rest_a || !opacity_logit_a || !log_scaling_a || !rotation_raw_a || !xyz_b ||
      !features_dc_b || !features_rest_b || !opacity_logit_b || !log_scaling_b || !rotation_raw_b)
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw2: null parameter tensor");
  if (out_objects && ((objects_dc_a == nullptr) != (objects_dc_b == nullptr)))
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw2: object features must be given for both segments or for neither");
  SegB b;
  b.Pb = Pb; b.xyz = xyz_b; b.features_dc = features_dc_b; b.features_rest = features_rest_b; b.objects_dc = objects_dc_b;
  b.opacity = opacity_logit_b; b.scaling = log_scaling_b; b.rotation = rotation_raw_b;
  return forward_impl(s, Pa + Pb, 16, xyz_a, features_rest_a, features_dc_a, objects_dc_a, nullptr, opacity_logit_a,
                      log_scaling_a, rotation_raw_a, nullptr, out_color, out_objects, radii, ctx_out, num_rendered, stream,
                      true, &b, true);
}

int gsr_forward_raw2(const GsrSettings* s, int32_t Pa, const float* xyz_a, const float* features_dc_a,
                     const float* features_rest_a, const float* objects_dc_a, const float* opacity_logit_a,
                     const float* log_scaling_a, const float* rotation_raw_a, int32_t Pb, const float* xyz_b,
                     const float* features_dc_b, const float* features_rest_b, const float* objects_dc_b,
                     const float* opacity_logit_b, const float* log_scaling_b, const float* rotation_raw_b,
                     float* out_color, float* out_objects, int32_t* radii, int64_t* num_rendered, void* stream) {
  return gsr_forward_raw2_keep(s, Pa, xyz_a, features_dc_a, features_rest_a, objects_dc_a, opacity_logit_a, log_scaling_a,
                               rotation_raw_a, Pb, xyz_b, features_dc_b, features_rest_b, objects_dc_b, opacity_logit_b,
                               log_scaling_b, rotation_raw_b, out_color, out_objects, radii, nullptr, num_rendered, stream);
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

// A batch of views of ONE set of raw parameters through one launch chain (include/gsraster.h): the views must agree in
// image size, scale modifier, SH degree, flags; cameras, tan(fov / 2) and backgrounds are per view.
int gsr_forward_raw_batch(const GsrSettings* s, int32_t B, int32_t P, const float* xyz, const float* features_dc,
                          const float* features_rest, const float* opacity_logit, const float* log_scaling,
                          const float* rotation_raw, float* out_color, int32_t* radii, GsrCtx** ctx_out, int64_t* num_rendered,
                          void* stream) {
  if (ctx_out) *ctx_out = nullptr;
  if (!s || B < 1 || B > MAX_BATCH) return set_err(GSR_ERR_INVALID, "gsr_forward_raw_batch: 1..%d views, got %d", MAX_BATCH, B);
  if (P > 0 && (!features_dc || !features_rest || !log_scaling || !rotation_raw))
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw_batch: null features_dc / features_rest / log_scaling / rotation_raw");
  for (int v = 0; v < B; ++v) {
    if (!s[v].bg || !s[v].viewmatrix || !s[v].projmatrix || !s[v].campos)
      return set_err(GSR_ERR_INVALID, "gsr_forward_raw_batch: view %d: settings tensors (bg, viewmatrix, projmatrix, campos) must be device pointers", v);
    if (s[v].image_height != s[0].image_height || s[v].image_width != s[0].image_width || s[v].scale_modifier != s[0].scale_modifier ||
        s[v].sh_degree != s[0].sh_degree || s[v].flags != s[0].flags)
      return set_err(GSR_ERR_INVALID, "gsr_forward_raw_batch: view %d differs from view 0 in image size, scale modifier, SH degree or "
                     "flags (a batch shares them)", v);
  }
  if (s[0].flags & GSR_FLAG_NEEDLE_DOUBLE) return set_err(GSR_ERR_INVALID, "gsr_forward_raw_batch: GSR_FLAG_NEEDLE_DOUBLE is a single-view flag");
  if (!out_color) return set_err(GSR_ERR_INVALID, "gsr_forward_raw_batch: out_color is null");
  if (P == 0 && B > 1) {                 // an empty scene: B backgrounds; the context (of view 0) has nothing to differentiate
    const size_t img = (size_t)3 * (size_t)s[0].image_height * (size_t)s[0].image_width;
    for (int v = 0; v < B; ++v) {
      const int rc = forward_impl(s + v, 0, 16, xyz, features_rest, features_dc, nullptr, nullptr, opacity_logit, log_scaling,
                                  rotation_raw, nullptr, out_color + (size_t)v * img, nullptr, radii, v == 0 ? ctx_out : nullptr,
                                  num_rendered, stream, true);
      if (rc != GSR_OK) return rc;
    }
    return GSR_OK;
  }
  return forward_impl(s, P, 16, xyz, features_rest, features_dc, nullptr, nullptr, opacity_logit, log_scaling, rotation_raw,
                      nullptr, out_color, nullptr, radii, ctx_out, num_rendered, stream, true, nullptr, false, B);
}

// gsr_forward_raw_batch for TWO parameter sets as one scene per view (the success renders of a batch, reference
// attack.py:513-530 inside the loop over the batch's cameras): forward only; a kept context is re-renderable, not differentiable.
int gsr_forward_raw2_batch(const GsrSettings* s, int32_t B, int32_t Pa, const float* xyz_a, const float* features_dc_a,
                           const float* features_rest_a, const float* opacity_logit_a, const float* log_scaling_a,
                           const float* rotation_raw_a, int32_t Pb, const float* xyz_b, const float* features_dc_b,
                           const float* features_rest_b, const float* opacity_logit_b, const float* log_scaling_b,
                           const float* rotation_raw_b, float* out_color, int32_t* radii, GsrCtx** ctx_out,
                           int64_t* num_rendered, void* stream) {
  if (ctx_out) *ctx_out = nullptr;
  if (Pa <= 0 || Pb <= 0 || (long long)Pa + Pb > 0x7FFFFFFFll)
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw2_batch: both segments must hold Gaussians (Pa=%d Pb=%d)", Pa, Pb);
  if (!s || B < 1 || B > MAX_BATCH) return set_err(GSR_ERR_INVALID, "gsr_forward_raw2_batch: 1..%d views, got %d", MAX_BATCH, B);
  if (!xyz_a || !features_dc_a || !features_rest_a || !opacity_logit_a || !log_scaling_a || !rotation_raw_a || !xyz_b ||
      !features_dc_b || !features_rest_b || !opacity_logit_b || !log_scaling_b || !rotation_raw_b)
    return set_err(GSR_ERR_INVALID, "gsr_forward_raw2_batch: null parameter tensor");
  for (int v = 0; v < B; ++v) {
    if (!s[v].bg || !s[v].viewmatrix || !s[v].projmatrix || !s[v].campos)
      return set_err(GSR_ERR_INVALID, "gsr_forward_raw2_batch: view %d: settings tensors (bg, viewmatrix, projmatrix, campos) must be device pointers", v);
    if (s[v].image_height != s[0].image_height || s[v].image_width != s[0].image_width || s[v].scale_modifier != s[0].scale_modifier ||
        s[v].sh_degree != s[0].sh_degree || s[v].flags != s[0].flags)
      return set_err(GSR_ERR_INVALID, "gsr_forward_raw2_batch: view %d differs from view 0 in image size, scale modifier, SH degree or "
                     "flags (a batch shares them)", v);
  }
  if (s[0].flags & GSR_FLAG_NEEDLE_DOUBLE) return set_err(GSR_ERR_INVALID, "gsr_forward_raw2_batch: GSR_FLAG_NEEDLE_DOUBLE is a single-view flag");
  if (!out_color) return set_err(GSR_ERR_INVALID, "gsr_forward_raw2_batch: out_color is null");
  SegB b;
  b.Pb = Pb; b.xyz = xyz_b; b.features_dc = features_dc_b; b.features_rest = features_rest_b; b.objects_dc = nullptr;
  b.opacity = opacity_logit_b; b.scaling = log_scaling_b; b.rotation = rotation_raw_b;
  return forward_impl(s, Pa + Pb, 16, xyz_a, features_rest_a, features_dc_a, nullptr, nullptr, opacity_logit_a, log_scaling_a,
                      rotation_raw_a, nullptr, out_color, nullptr, radii, ctx_out, num_rendered, stream, true, &b, true, B);
}

// Re-render of a kept context whose colour inputs (SH coefficients) may have changed and nothing else has: the colour
// half of K1 over the Gaussians that emit pairs, then K6 over the kept lists.  include/gsraster.h has the contract.
int gsr_ctx_rerender(GsrCtx* c, const float* features_dc, const float* features_rest, const float* features_dc_b,
                     const float* features_rest_b, const float* bg, float* out_color, float* out_objects, uint32_t flags,
                     void* stream) {
  if (!c) return set_err(GSR_ERR_STATE, "gsr_ctx_rerender: null context");
  if (!out_color) return set_err(GSR_ERR_INVALID, "gsr_ctx_rerender: out_color is null");
  // a batch context (gsr_forward_raw_batch / gsr_forward_raw2_batch): out_color [B,3,H,W], bg [B,3] or null; no object channels
  if (c->B > 1 && out_objects)
    return set_err(GSR_ERR_INVALID, "gsr_ctx_rerender: a batch context (%d views) has no object channels", c->B);
  if (c->P > 0 && (!c->lanegroup || !c->shs))
    return set_err(GSR_ERR_INVALID, "gsr_ctx_rerender: the context was not rendered from SH coefficients (raw parameters, "
                   "or shs with K = 16): there is no colour stage to run again");
  if (out_objects && !c->objects_out)
    return set_err(GSR_ERR_INVALID, "gsr_ctx_rerender: the context's forward did not composite object features");
  if (!c->raw && features_dc)
    return set_err(GSR_ERR_INVALID, "gsr_ctx_rerender: a gsr_forward context takes its [P,16,3] coefficients in `features_rest`");
  if (!c->has_b && (features_dc_b || features_rest_b))
    return set_err(GSR_ERR_INVALID, "gsr_ctx_rerender: the context has one attribute segment");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ctx_resolve_count(c) != GSR_OK) return set_err(GSR_ERR_DEVICE, "gsr_ctx_rerender: reading the forward's pair count failed");
  if (c->overflow)
    return set_err(GSR_ERR_OVERFLOW, "gsr_ctx_rerender: the context's forward overflowed its pair capacity (%llu pairs, "
                   "capacity %u): render again with gsr_forward", c->n64, c->nbound);
  if ((flags & GSR_RERENDER_FIRST_SEGMENT_ONLY) && (features_dc_b || features_rest_b))
    return set_err(GSR_ERR_INVALID, "gsr_ctx_rerender: GSR_RERENDER_FIRST_SEGMENT_ONLY with new second-segment coefficients");
  // every argument check has passed: only now does the kept context take the call's pointers and follow its stream (a
  // rejected call leaves the context exactly as it was)
  pool_retag(c->dev, c->keep_blk, st); pool_retag(c->dev, c->rank_blk, st); pool_retag(c->dev, c->seg_blk, st);
  c->sumsq_out = nullptr;                // a request armed for a backward of the previous render does not carry over
  if (features_rest) c->shs = features_rest;
  if (features_dc) c->sh_dc = features_dc;
  if (features_rest_b) c->b.features_rest = features_rest_b;
  if (features_dc_b) c->b.features_dc = features_dc_b;
  if (bg && c->B > 1) {
    for (int v = 0; v < c->B; ++v) c->views[v].bg = bg + 3 * (size_t)v;
    c->st.bg = bg;
  } else if (bg) {
    c->st.bg = bg;
  }
  if (c->P > 0 && c->B > 1) {
    // the batch's colour kernel over the Gaussians that emit pairs in ANY view (every SH row read once), the views' constants
    // packed again first: the CONTENTS of a background tensor may have changed, and the compositors read it from there
    StageTimer t(GSR_STAGE_PREPROCESS, st);
    ViewPtrs vp{};
    for (int v = 0; v < c->B; ++v) {
      const GsrSettings& sv = c->views[v];
      vp.vm[v] = sv.viewmatrix; vp.pm[v] = sv.projmatrix; vp.cam[v] = sv.campos; vp.bg[v] = sv.bg;
      vp.tanfovx[v] = sv.tanfovx; vp.tanfovy[v] = sv.tanfovy;
    }
    hipLaunchKernelGGL(k_pack_views, dim3(c->B), dim3(64), 0, st, vp, c->B, c->vpack);
    PreColorBatchArgs ca;
    ca.P = c->P; ca.B = c->B; ca.Ppad = c->Ppad; ca.deg = c->st.sh_degree; ca.vpack = c->vpack; ca.means = c->means3D;
    ca.sh = c->shs; ca.sh_dc = c->sh_dc; ca.tcnt = nullptr; ca.offg = c->offg; ca.G1 = c->G1; ca.G2 = c->G2;
    ca.Pa = c->has_b ? c->P - c->b.Pb : c->P; ca.means_b = c->has_b ? c->b.xyz : nullptr;
    ca.sh_b = c->has_b ? c->b.features_rest : nullptr; ca.sh_dc_b = c->has_b ? c->b.features_dc : nullptr;
    // two segments, the second one's coefficients untouched since the last render: its colour words are still right
    if (c->has_b && (flags & GSR_RERENDER_FIRST_SEGMENT_ONLY)) ca.P = ca.Pa;
    const bool skip_D = (flags & GSR_RERENDER_COLOR_GRADS_ONLY) != 0u;
    ca.D = skip_D ? nullptr : c->D;
    c->D_stale = c->D != nullptr && skip_D;
    const dim3 gridC((unsigned)((ca.P + PREF_BLOCK - 1) / PREF_BLOCK)), blkC(PREF_BLOCK);
    hipLaunchKernelGGL(k_pre_color_batch, gridC, blkC, 0, st, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "rerender colours (batch): launch failed: %s", hipGetErrorString(e));
  } else if (c->P > 0) {
    StageTimer t(GSR_STAGE_PREPROCESS, st);
    PreArgs pa{};
    pa.P = c->P; pa.va = view_args(c->st); pa.means = c->means3D; pa.sh = c->shs; pa.sh_dc = c->sh_dc;
    pa.Pa = c->has_b ? c->P - c->b.Pb : c->P;
    pa.means_b = c->has_b ? c->b.xyz : nullptr;
    pa.sh_b = c->has_b ? c->b.features_rest : nullptr; pa.sh_dc_b = c->has_b ? c->b.features_dc : nullptr;
    pa.G0 = c->G0; pa.G1 = c->G1; pa.G2 = c->G2;
    const bool skip_D = (flags & GSR_RERENDER_COLOR_GRADS_ONLY) != 0u;
    pa.D = skip_D ? nullptr : c->D;
    c->D_stale = c->D != nullptr && skip_D;
    pa.tcnt = nullptr; pa.offg = c->offg;
    // two segments, the second one's coefficients untouched since the last render: its colour words are still right
    if (c->has_b && (flags & GSR_RERENDER_FIRST_SEGMENT_ONLY)) pa.P = pa.Pa;
    const dim3 gridC((unsigned)((pa.P + PREF_BLOCK - 1) / PREF_BLOCK)), blkC(PREF_BLOCK);
    if (c->raw) hipLaunchKernelGGL((k_pre_color<true>), gridC, blkC, 0, st, pa);
    else hipLaunchKernelGGL((k_pre_color<false>), gridC, blkC, 0, st, pa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "rerender colours: launch failed: %s", hipGetErrorString(e));
  }
  StageTimer t(GSR_STAGE_RENDER_FWD, st);
  return launch_render_fwd(c, out_color, out_objects, st);
}

}  // extern "C"

static bool batch_k9_fused() {
  static const int env = [] { const char* e = getenv("GSR_BATCH_K9"); return e ? atoi(e) : 1; }();
  return env != 0;
}

static int backward_impl(GsrCtx* c, const float* grad_color, const float* grad_objects, float* dmeans3D, float* dmeans2D,
                         float* dshs, float* dsh_dc, float* dsh_objs, float* dcolors_precomp, float* dopacities,
                         float* dscales, float* drotations, float* dcov3D, void* stream, bool accumulate = false,
                         int nchunks = 1, gsr_chunk_fn chunk_done = nullptr, void* chunk_user = nullptr,
                         int64_t view_stride = 0) {
  // view_stride != 0 (gsr_backward_raw_batch_views): a batch context's PER-VIEW gradients -- the attribute-gradient pointers
  // are view 0's buffers, view v's lie v * view_stride floats further; every view's buffers are overwritten
  if (!c) return set_err(GSR_ERR_STATE, "gsr_backward: null context");
  // gsr_ctx_request_sumsq is one-shot: the request is taken (and the context disarmed) here, whatever this call's fate --
  // a backward that fails early must not leave the next one writing six doubles to a buffer that may be gone by then
  double* ss_out = c->sumsq_out;
  c->sumsq_out = nullptr;
  if (!grad_color) return set_err(GSR_ERR_INVALID, "gsr_backward: grad_color is null");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = c->dev;
  const int P = c->P;
  if (c->fwd_only)
    return set_err(GSR_ERR_STATE, "gsr_backward: the context was kept by gsr_forward_raw2_keep (re-render only, no backward state)");
  if (P == 0) return GSR_OK;
  const bool obj = grad_objects != nullptr && c->sh_objs != nullptr;
  // Only colour-side gradients wanted (SH / precomputed colours / object features): K7 and K8+K9 drop the geometry
  // sums and the projection chain rule (the colour attack; BASELINE configs 2 and 3).
  const bool geom = dmeans3D || dmeans2D || dopacities || dscales || drotations || dcov3D;
  // an asynchronous-count forward: its pair count must have fitted the capacity guess (else the image it produced was
  // poisoned with NaN and nothing downstream of it is meaningful).  (A forward recorded into a hipGraph publishes
  // nothing to the host: c->slot is empty and this is a no-op.)
  if (ctx_resolve_count(c) != GSR_OK) return set_err(GSR_ERR_DEVICE, "gsr_backward: reading the forward's pair count failed");
  if (c->overflow)
    return set_err(GSR_ERR_OVERFLOW, "gsr_backward: the forward emitted %llu (tile, Gaussian) pairs, more than the capacity "
                   "%u guessed from earlier views (GSR_FLAG_ASYNC_COUNT); its image was filled with NaN -- render again",
                   c->n64, c->nbound);
  const uint32_t N = c->nbound;
  // the context's blocks are read here, on this stream: whoever takes them over after gsr_ctx_free orders behind it
  pool_retag(dev, c->keep_blk, st); pool_retag(dev, c->rank_blk, st); pool_retag(dev, c->seg_blk, st);
  void* part_blk = nullptr;
  void* pobj_blk = nullptr;
  float4* part = nullptr;
  float4* part_obj = nullptr;
  // K7 runs one wave per 16x(4*npx) part of a tile; each writes its own partial row per list entry (4 pixels per lane:
  // one wave per tile; 2: two, and K8/K9 then sums twice the rows).  With long lists walked as segments every work item
  // is short whatever the tile count, so one wave per tile it is (S-hydrant-full, 2500 tiles: K7 0.169 -> 0.194 ms but
  // K8+K9 0.141 -> 0.084 ms, 2563 -> 2811 views/s); only without segments (object channels, GSR_FLAG_NO_SEGMENTS) an
  // image with fewer tiles than the chip has wave slots is split in two.  GSR_FLAG_BWD_SPLIT(n) overrides.
  // (The rule looks at the tiles of ONE view, also for a batch: the split decides how many partial rows K9 adds up per
  // pair -- the float32 association of the sums -- and a batch's per-view gradients are the single-view call's bit for bit.
  // Found by tests/diag_fuzz_batch.py with GSR_FLAG_NO_SEGMENTS on batches of >= 4096 tiles of views with fewer.)
  const bool segs_on = c->bnd != nullptr && !obj;
  const int bwd_npx = flag_bwd_npx(c->st.flags) ? flag_bwd_npx(c->st.flags) : ((c->tpv < 4096 && !segs_on) ? 2 : 4);
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
    ra.ranges = c->ranges; ra.pair_rank = c->pair_rank; ra.offg = c->offg; ra.R0 = c->G0; ra.R1 = c->G1; ra.R2 = c->G2;
    ra.sh_objs = c->sh_objs; ra.bg = c->st.bg; ra.W = c->st.image_width; ra.H = c->st.image_height;
    ra.map_mode = flag_tile_map(c->st.flags);
    ra.sched = c->sched;
    ra.wave_clock = g_wave_clock.load();
    ra.gridx = c->gridx; ra.ntiles = c->ntiles; ra.final_T = c->final_T; ra.n_contrib = c->n_contrib;
    ra.grad_color = grad_color; ra.grad_objects = obj ? grad_objects : nullptr; ra.part = part; ra.part_obj = part_obj;
    // split tiles: one extra work item per boundary record, in front of the per-tile items
    const bool segs = c->bnd != nullptr && !obj;
    ra.bnd = segs ? c->bnd : nullptr; ra.segoff = c->segoff; ra.rec_item = c->rec_item; ra.nrec = c->dv + DV_NREC;
    ra.seg_shift = c->seg_shift; ra.extra_blocks = segs ? c->rec_cap * nsub : 0u;
    ra.tpv = c->tpv; ra.vpack = c->vpack;
    const dim3 gridT(ra.extra_blocks + (unsigned)render_grid(c->ntiles * (int)nsub)), blk(64);
    CompStream comp_s{};
    bool comp_used = false;
    hipStream_t st7 = comp_enter(dev, st, comp_s, comp_used);
#define LAUNCH_K7(kern)                                                                      \
  do {                                                                                       \
    if (t.on) hipExtLaunchKernelGGL(kern, gridT, blk, 0, st7, t.a, t.b, 0, ra);              \
    else hipLaunchKernelGGL(kern, gridT, blk, 0, st7, ra);                                   \
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
    const bool joined = comp_leave(st, comp_s, comp_used);
    if (e != hipSuccess) return done(set_err(GSR_ERR_DEVICE, "render backward: launch failed: %s", hipGetErrorString(e)));
    if (!joined) return done(set_err(GSR_ERR_DEVICE, "render backward: joining the compositor stream failed"));
  }
  {
    StageTimer t(GSR_STAGE_PREPROCESS_BWD, st);
    PreBwdArgs pa;
    pa.P = P; pa.g0 = 0; pa.K = c->K; pa.va = view_args(c->st);
    pa.offg = c->offg; pa.G0 = c->G0; pa.G1 = c->G1; pa.G2 = c->G2;
    pa.part = part; pa.part_obj = obj ? part_obj : nullptr;
    pa.tag_lo = tag_lo; pa.tag_hi = tag_hi; pa.nsub = nsub;
    pa.means = c->means3D; pa.scales = c->scales; pa.rots = c->rots; pa.cov3d = c->cov3d; pa.sh = c->shs;
    pa.sh_dc = c->sh_dc; pa.dsh_dc = dsh_dc; pa.D = c->D; pa.abc = c->abc;
    pa.needle_double = (c->st.flags & GSR_FLAG_NEEDLE_DOUBLE) != 0u ? 1 : 0;
    pa.dmeans3D = dmeans3D; pa.dmeans2D = dmeans2D; pa.dsh = c->shs ? dshs : nullptr; pa.dsh_objs = dsh_objs;
    pa.dcolors = c->colors ? dcolors_precomp : nullptr; pa.dopac = dopacities;
    pa.dscales = c->cov3d ? nullptr : dscales; pa.drots = c->cov3d ? nullptr : drotations;
    pa.dcov3d = c->cov3d ? dcov3D : nullptr;
    pa.accumulate = accumulate ? 1 : 0;
    pa.vpack = nullptr; pa.Ppad = 0; pa.Pscene = P; pa.vstride = 0;
    if (c->raw && ((pa.dsh == nullptr) != (pa.dsh_dc == nullptr)))
      return done(set_err(GSR_ERR_INVALID, "gsr_backward_raw: dfeatures_dc and dfeatures_rest must both be given"));
    if (c->lanegroup && c->shs && geom && !c->D)
      return done(set_err(GSR_ERR_STATE, "gsr_backward: the forward of this context was run without its backward state"));
    if (c->lanegroup && c->shs && geom && c->D_stale)
      return done(set_err(GSR_ERR_STATE, "gsr_backward: geometry gradients asked of a context whose last gsr_ctx_rerender was "
                          "told GSR_RERENDER_COLOR_GRADS_ONLY"));
    // The per-Gaussian stage covers the Gaussians in `nchunks` ranges (multiples of 64), one launch each; after a
    // range's launch is enqueued the caller is told (chunk_done): its gradients are complete in stream order, so a
    // collective over that range can be issued while the next range is still being computed.
    nchunks = std::max(1, std::min(nchunks, (P + 63) / 64));
    const int per = (((P + nchunks - 1) / nchunks) + 63) / 64 * 64;
    // gsr_ctx_request_sumsq (one-shot): the overwriting raw-parameter kernel also leaves per-workgroup sums of squares of
    // what it writes; one small launch behind it adds them up per tensor (fixed order) into the caller's six doubles
    // a batch of views: ONE launch of k_pre_bwd_batch per range walks the B views and writes the gradients once
    // (GSR_BATCH_K9=0: one k_pre_bwd launch per view instead, the others in accumulate mode -- the A/B and the bit-exact form)
    const bool batch_fused = c->B > 1 && c->raw && c->lanegroup && batch_k9_fused() && view_stride == 0;
    void* ss_blk = nullptr;
    const int ss_blocks = batch_fused ? ((P + 63) / 64) * BATCH_K9_WAVES : (P + PRE_BLOCK - 1) / PRE_BLOCK;
    pa.sumsq = nullptr;
    if (ss_out) {
      if (!(c->lanegroup && c->raw) || accumulate || nchunks != 1 || (c->B > 1 && !batch_fused))
        return done(set_err(GSR_ERR_INVALID, "gsr_ctx_request_sumsq: served by an overwriting gsr_backward_raw* over one range only"));
      ss_blk = pool_alloc(dev, sizeof(float) * SUMSQ_W * (size_t)ss_blocks * PRE_WAVES, st);
      if (!ss_blk) return done(set_err(GSR_ERR_NOMEM, "gsr_backward_raw: sum-of-squares partials allocation failed"));
      pa.sumsq = static_cast<float*>(ss_blk);
    }
    for (int ck = 0; ck < nchunks; ++ck) {
      const int gb = std::min(ck * per, P), ge = (ck == nchunks - 1) ? P : std::min((ck + 1) * per, P);
      // (a batch of views: one launch per view over the same range of Gaussians, the first overwriting -- unless the caller
      // asked for accumulation -- and the others adding; the per-view arrays of the virtual scene are offset by v * Ppad)
      if (batch_fused && ge > gb) {
        PreBwdBatchArgs ba;
        ba.P = ge; ba.g0 = gb; ba.B = c->B; ba.Ppad = c->Ppad; ba.vpack = c->vpack;
        ba.H = c->st.image_height; ba.W = c->st.image_width; ba.deg = c->st.sh_degree; ba.mod = c->st.scale_modifier;
        ba.offg = c->offg; ba.G0 = c->G0; ba.G1 = c->G1; ba.G2 = c->G2; ba.part = part;
        ba.tag_lo = tag_lo; ba.tag_hi = tag_hi; ba.nsub = nsub;
        ba.means = c->means3D; ba.scales = c->scales; ba.rots = c->rots; ba.D = c->D;
        ba.dmeans3D = dmeans3D; ba.dmeans2D = dmeans2D; ba.dsh = dshs; ba.dsh_dc = dsh_dc; ba.dopac = dopacities;
        ba.dscales = dscales; ba.drots = drotations; ba.sumsq = pa.sumsq;
        const dim3 gridB((unsigned)((ge - gb + 63) / 64)), blkB(64 * BATCH_K9_WAVES);
        const size_t lds = sizeof(float) * 3 * 64 * (size_t)c->B;
        if (geom) {
          if (accumulate) hipLaunchKernelGGL((k_pre_bwd_batch<true, true>), gridB, blkB, lds, st, ba);
          else hipLaunchKernelGGL((k_pre_bwd_batch<true, false>), gridB, blkB, lds, st, ba);
        } else {
          if (accumulate) hipLaunchKernelGGL((k_pre_bwd_batch<false, true>), gridB, blkB, lds, st, ba);
          else hipLaunchKernelGGL((k_pre_bwd_batch<false, false>), gridB, blkB, lds, st, ba);
        }
      }
      // per-view gradients of a batch (gsr_backward_raw_batch_views): ONE launch whose grid.y is the view -- the B per-view
      // launches' ramp-ups and tails happen once (GSR_BATCH_K9_VIEWS=0: one launch per view, the A/B form; same bits)
      static const int views_one_launch = [] { const char* e = getenv("GSR_BATCH_K9_VIEWS"); return e ? atoi(e) : 1; }();
      const bool views_fused = !batch_fused && view_stride != 0 && c->B > 1 && c->raw && c->lanegroup && views_one_launch != 0;
      if (views_fused && ge > gb) {
        pa.vpack = c->vpack; pa.Ppad = c->Ppad; pa.Pscene = P; pa.vstride = (long long)view_stride;
        pa.accumulate = 0; pa.abc = nullptr;
        pa.g0 = gb; pa.P = ge;
        const dim3 gridV((unsigned)((ge - gb + PRE_BLOCK - 1) / PRE_BLOCK), (unsigned)c->B);
        if (geom) hipLaunchKernelGGL((k_pre_bwd<true, true>), gridV, dim3(PRE_BLOCK), 0, st, pa);
        else hipLaunchKernelGGL((k_pre_bwd<true, false>), gridV, dim3(PRE_BLOCK), 0, st, pa);
        pa.vpack = nullptr;
      }
      for (int v = 0; !batch_fused && !views_fused && ge > gb && v < c->B; ++v) {
        const size_t o = (size_t)v * (size_t)c->Ppad;
        const bool acc_v = view_stride == 0 && (accumulate || v > 0);
        if (c->B > 1) {
          if (view_stride != 0) {               // this view's own gradient buffers
            const size_t vs = (size_t)v * (size_t)view_stride;
            pa.dmeans3D = dmeans3D ? dmeans3D + vs : nullptr; pa.dsh = dshs ? dshs + vs : nullptr;
            pa.dsh_dc = dsh_dc ? dsh_dc + vs : nullptr; pa.dopac = dopacities ? dopacities + vs : nullptr;
            pa.dscales = dscales ? dscales + vs : nullptr; pa.drots = drotations ? drotations + vs : nullptr;
          }
          pa.va = view_args(c->views[v]);
          pa.offg = c->offg + o; pa.G0 = c->G0 + REC * o; pa.G1 = c->G1 + REC * o; pa.G2 = c->G2 + REC * o;
          pa.D = c->D ? c->D + 9 * o : nullptr; pa.abc = c->abc ? c->abc + o : nullptr;
          pa.dmeans2D = dmeans2D ? dmeans2D + 3 * (size_t)v * (size_t)P : nullptr;
          pa.accumulate = acc_v ? 1 : 0;
        }
        pa.g0 = gb; pa.P = ge;
        const dim3 gridK9((unsigned)((ge - gb + PRE_BLOCK - 1) / PRE_BLOCK));
        if (c->lanegroup) {
          const bool ndl = pa.needle_double != 0 && pa.abc != nullptr;
          if (c->raw && acc_v) {
            if (geom && ndl) hipLaunchKernelGGL((k_pre_bwd<true, true, true, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
            else if (geom) hipLaunchKernelGGL((k_pre_bwd<true, true, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
            else hipLaunchKernelGGL((k_pre_bwd<true, false, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
          } else if (c->raw) {
            if (geom && ndl) hipLaunchKernelGGL((k_pre_bwd<true, true, false, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
            else if (geom) hipLaunchKernelGGL((k_pre_bwd<true, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
            else hipLaunchKernelGGL((k_pre_bwd<true, false>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
          } else {
            if (geom && ndl) hipLaunchKernelGGL((k_pre_bwd<false, true, false, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
            else if (geom) hipLaunchKernelGGL((k_pre_bwd<false, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
            else hipLaunchKernelGGL((k_pre_bwd<false, false>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
          }
        } else {
          if (geom && pa.needle_double) hipLaunchKernelGGL((k_preprocess_bwd<true, true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
          else if (geom) hipLaunchKernelGGL((k_preprocess_bwd<true>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
          else hipLaunchKernelGGL((k_preprocess_bwd<false>), gridK9, dim3(PRE_BLOCK), 0, st, pa);
        }
      }
      if (chunk_done) chunk_done(chunk_user, ck, (int64_t)gb, (int64_t)ge);
    }
    if (ss_out) {
      hipLaunchKernelGGL(k_sumsq_reduce, dim3(6), dim3(256), 0, st, pa.sumsq, ss_blocks * PRE_WAVES, ss_out);
      pool_free(dev, ss_blk);
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

int gsr_backward_raw_into(GsrCtx* c, const float* grad_color, const float* grad_objects, float* dxyz, float* dmeans2D,
                          float* dfeatures_dc, float* dfeatures_rest, float* dobjects_dc, float* dopacity_logit,
                          float* dlog_scaling, float* drotation_raw, int32_t accumulate, void* stream) {
  if (c && !c->raw) return set_err(GSR_ERR_STATE, "gsr_backward_raw_into: context came from gsr_forward; use gsr_backward");
  return backward_impl(c, grad_color, grad_objects, dxyz, dmeans2D, dfeatures_rest, dfeatures_dc, dobjects_dc, nullptr,
                       dopacity_logit, dlog_scaling, drotation_raw, nullptr, stream, accumulate != 0);
}

int gsr_backward_raw_chunked(GsrCtx* c, const float* grad_color, const float* grad_objects, float* dxyz, float* dmeans2D,
                             float* dfeatures_dc, float* dfeatures_rest, float* dobjects_dc, float* dopacity_logit,
                             float* dlog_scaling, float* drotation_raw, int32_t accumulate, int32_t nchunks,
                             gsr_chunk_fn chunk_done, void* user, void* stream) {
  if (c && !c->raw) return set_err(GSR_ERR_STATE, "gsr_backward_raw_chunked: context came from gsr_forward; use gsr_backward");
  return backward_impl(c, grad_color, grad_objects, dxyz, dmeans2D, dfeatures_rest, dfeatures_dc, dobjects_dc, nullptr,
                       dopacity_logit, dlog_scaling, drotation_raw, nullptr, stream, accumulate != 0, nchunks, chunk_done, user);
}

int gsr_backward_raw_batch_into(GsrCtx* c, const float* grad_color, float* dxyz, float* dmeans2D, float* dfeatures_dc,
                                float* dfeatures_rest, float* dopacity_logit, float* dlog_scaling, float* drotation_raw,
                                int32_t accumulate, void* stream) {
  if (c && !c->raw) return set_err(GSR_ERR_STATE, "gsr_backward_raw_batch_into: context came from gsr_forward; use gsr_backward");
  return backward_impl(c, grad_color, nullptr, dxyz, dmeans2D, dfeatures_rest, dfeatures_dc, nullptr, nullptr, dopacity_logit,
                       dlog_scaling, drotation_raw, nullptr, stream, accumulate != 0);
}

int gsr_backward_raw_batch_views(GsrCtx* c, const float* grad_color, float* dxyz, float* dmeans2D, float* dfeatures_dc,
                                 float* dfeatures_rest, float* dopacity_logit, float* dlog_scaling, float* drotation_raw,
                                 int64_t view_stride, void* stream) {
  if (c && !c->raw) return set_err(GSR_ERR_STATE, "gsr_backward_raw_batch_views: context came from gsr_forward; use gsr_backward");
  if (view_stride <= 0) return set_err(GSR_ERR_INVALID, "gsr_backward_raw_batch_views: view_stride must be positive (floats between two views' buffers)");
  return backward_impl(c, grad_color, nullptr, dxyz, dmeans2D, dfeatures_rest, dfeatures_dc, nullptr, nullptr, dopacity_logit,
                       dlog_scaling, drotation_raw, nullptr, stream, false, 1, nullptr, nullptr, view_stride);
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
    case 0:
      if (ctx_resolve_count(const_cast<GsrCtx*>(c)) != GSR_OK) return set_err(GSR_ERR_DEVICE, "gsr_ctx_info: reading the pair count failed");
      *out = (int64_t)c->n64;
      return GSR_OK;
    case 1: *out = -1; return GSR_OK;
    case 2: *out = (int64_t)(c->keep_bytes + sizeof(uint32_t) * (size_t)c->nbound); return GSR_OK;
    case 3: *out = (int64_t)c->nbound; return GSR_OK;
    case 4: *out = (int64_t)c->B; return GSR_OK;
    case 5: *out = (int64_t)c->Ppad; return GSR_OK;
    default: return set_err(GSR_ERR_INVALID, "gsr_ctx_info: unknown item %d", what);
  }
}

int gsr_ctx_export(const GsrCtx* c, int32_t what, void* dst, int64_t dst_bytes, void* stream) {
  if (!c || !dst) return set_err(GSR_ERR_INVALID, "gsr_ctx_export: null argument");
  const size_t HW = (size_t)c->st.image_height * c->st.image_width * (size_t)c->B;   // (a batch: the B views one after another)
  const size_t PV = (size_t)c->Pv;                                                    // (... B * Ppad virtual Gaussians)
  const void* src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case 0: src = c->ranges; bytes = sizeof(uint2) * (size_t)c->ntiles; break;
    case 1:
      if (ctx_resolve_count(const_cast<GsrCtx*>(c)) != GSR_OK) return set_err(GSR_ERR_DEVICE, "gsr_ctx_export: reading the pair count failed");
      src = c->pair_rank; bytes = sizeof(uint32_t) * (size_t)(c->overflow ? 0 : c->n64);
      break;
    case 2: src = c->n_contrib; bytes = sizeof(uint32_t) * HW; break;
    case 3: src = c->final_T; bytes = sizeof(float) * HW; break;
    case 4: src = c->order; bytes = sizeof(uint32_t) * PV; break;
    case 5: src = c->off; bytes = sizeof(uint32_t) * (PV + 1); break;
    case 6:                                                                   // (kept for old callers: same as 7)
    case 7: {                                                                 // splat records [P][3] float4, storage order
      const size_t need = sizeof(float4) * 3 * PV;
      if ((int64_t)need > dst_bytes)
        return set_err(GSR_ERR_INVALID, "gsr_ctx_export: item %d needs %zu bytes, buffer has %lld", what, need, (long long)dst_bytes);
      if (PV == 0) return GSR_OK;
      // the records sit REC float4 apart in the workspace: copied out packed, 48 bytes each
      HIP_TRY("ctx export", hipMemcpy2DAsync(dst, sizeof(float4) * 3, c->G0, sizeof(float4) * REC, sizeof(float4) * 3,
                                             PV, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
      return GSR_OK;
    }
    case 8: src = c->dv; bytes = sizeof(uint32_t) * DV_WORDS; break;          // device-side scalars (DV_*)
    case 9: src = c->offg; bytes = sizeof(uint32_t) * (PV + 1); break;
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

int gsr_ctx_request_sumsq(GsrCtx* c, double* out6) {
  if (!c) return set_err(GSR_ERR_STATE, "gsr_ctx_request_sumsq: null context");
  if (!c->raw || !c->lanegroup)
    return set_err(GSR_ERR_INVALID, "gsr_ctx_request_sumsq: only contexts of gsr_forward_raw (raw parameters) produce the sums");
  if (c->B > 1 && !batch_k9_fused())
    return set_err(GSR_ERR_INVALID, "gsr_ctx_request_sumsq: a batch context under GSR_BATCH_K9=0 runs one accumulating launch per view");
  c->sumsq_out = out6;
  return GSR_OK;
}

int gsr_pgd_step_normed(float* x, const float* grad, const float* x0, int64_t rows, int32_t cols, float alpha, float epsilon,
                        const double* sumsq, void* stream) {
  if (rows < 0 || cols < 1 || cols > PGD_MAX_COLS)
    return set_err(GSR_ERR_INVALID, "gsr_pgd_step_normed: rows=%lld cols=%d (1..%d columns)", (long long)rows, cols, PGD_MAX_COLS);
  if (rows == 0) return GSR_OK;
  if (!x || !grad || !x0 || !sumsq) return set_err(GSR_ERR_INVALID, "gsr_pgd_step_normed: null argument");
  const int64_t rpw = pgd_rows_per_wave(cols);
  hipLaunchKernelGGL((k_pgd_step<true>), dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(64), 0, static_cast<hipStream_t>(stream), x, grad,
                     x0, (size_t)rows, cols, alpha, epsilon, sumsq, 1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "gsr_pgd_step_normed: launch failed: %s", hipGetErrorString(e));
  return GSR_OK;
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
  const int64_t rpw = pgd_rows_per_wave(cols);
  const unsigned blocks = (unsigned)((rows + rpw - 1) / rpw);
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

int gsr_pgd_step_multi(int32_t n, float* const* x, const float* const* grad, const float* const* x0, const int64_t* rows,
                       const int32_t* cols, const float* alpha, const float* epsilon, int32_t l2, const double* const* sumsq,
                       void* stream) {
  if (n < 0 || n > PGD_MAX_TENSORS) return set_err(GSR_ERR_INVALID, "gsr_pgd_step_multi: n=%d (0..%d tensors)", n, PGD_MAX_TENSORS);
  if (n == 0) return GSR_OK;
  if (!x || !grad || !x0 || !rows || !cols || !alpha || !epsilon) return set_err(GSR_ERR_INVALID, "gsr_pgd_step_multi: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int dev = cur_dev();
  PgdMulti m;
  memset(&m, 0, sizeof(m));
  unsigned long long blocks = 0;
  size_t need = 0;                                       // doubles of partial sums this call has to make itself
  for (int t = 0; t < n; ++t) {
    if (rows[t] < 0 || cols[t] < 1 || cols[t] > PGD_MAX_COLS)
      return set_err(GSR_ERR_INVALID, "gsr_pgd_step_multi: tensor %d: rows=%lld cols=%d (1..%d columns)", t, (long long)rows[t], cols[t],
                     PGD_MAX_COLS);
    if (rows[t] > 0 && (!x[t] || !grad[t] || !x0[t])) return set_err(GSR_ERR_INVALID, "gsr_pgd_step_multi: tensor %d: null pointer", t);
    m.x[t] = x[t]; m.g[t] = grad[t]; m.x0[t] = x0[t];
    m.rows[t] = (unsigned long long)rows[t]; m.cols[t] = cols[t]; m.alpha[t] = alpha[t]; m.eps[t] = epsilon[t];
    m.first[t] = (unsigned)blocks;
    blocks += (unsigned long long)((rows[t] + pgd_rows_per_wave(cols[t]) - 1) / pgd_rows_per_wave(cols[t]));
    if (blocks > 0x7fffffffull) return set_err(GSR_ERR_INVALID, "gsr_pgd_step_multi: too many rows for one launch");
    if (l2 && rows[t] > 0) {
      if (sumsq && sumsq[t]) { m.partial[t] = sumsq[t]; m.nb[t] = 1; }
      else {
        m.nb[t] = (int)std::min<size_t>(((size_t)rows[t] * (size_t)cols[t] + 4095) / 4096, 1024);
        need += (size_t)m.nb[t];
      }
    }
  }
  for (int t = n; t <= PGD_MAX_TENSORS; ++t) m.first[t] = (unsigned)blocks;
  m.n = n;
  if (blocks == 0) return GSR_OK;
  void* blk = nullptr;
  if (need) {
    blk = pool_alloc(dev, sizeof(double) * need, st);
    if (!blk) return set_err(GSR_ERR_NOMEM, "gsr_pgd_step_multi: allocation failed");
    double* partial = static_cast<double*>(blk);
    for (int t = 0; t < n; ++t) {
      if (!l2 || rows[t] == 0 || m.partial[t]) continue;
      hipLaunchKernelGGL(k_pgd_sumsq, dim3(m.nb[t]), dim3(256), 0, st, grad[t], (size_t)rows[t] * (size_t)cols[t], partial);
      m.partial[t] = partial;
      partial += m.nb[t];
    }
  }
  if (l2) hipLaunchKernelGGL((k_pgd_step_multi<true>), dim3((unsigned)blocks), dim3(64), 0, st, m);
  else hipLaunchKernelGGL((k_pgd_step_multi<false>), dim3((unsigned)blocks), dim3(64), 0, st, m);
  if (blk) pool_free(dev, blk);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(GSR_ERR_DEVICE, "gsr_pgd_step_multi: launch failed: %s", hipGetErrorString(e));
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
  const int res = radix_sort_pairs(k0, v0, k1, v1, n, nullptr, 0, ceil_log2(ncells), true, table, sums, st);
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
  void* blk = pool_alloc(dev, sizeof(uint32_t) * ((size_t)n / 2048 + 2), st);
  if (!blk) return set_err(GSR_ERR_NOMEM, "gsr_test_scan: allocation failed");
  scan_exclusive_u32(in, out, n, nullptr, static_cast<uint32_t*>(blk), st);
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
  const int res = radix_sort_pairs(keys, vals, k1, v1, n, nullptr, begin_bit, end_bit, iota != 0, table, sums, st);
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
