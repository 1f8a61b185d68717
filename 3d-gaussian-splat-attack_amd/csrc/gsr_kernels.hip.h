// gsr_kernels.hip.h -- the raster path's kernels, hand-written for gfx950 (wave64, LDS-staged tile
// lists, DPP wave reductions).  Stage names follow SURVEY.md section 2.2 (K1..K10).
//
// Data layout in HBM (all float32 / uint32):
//   G0,G1,G2[REC * g]   per-Gaussian screen geometry written by K1, 3 x float4:
//                 G0=(px,py,A,B)  G1=(C,opacity,r,g)  G2=(b, depth, rectx_bits, recty_bits); (px, py) = the pixel centre
//                 RELATIVE to the first pixel of the reference rect's first tile (16 (minx - offx), ...), from double-precision projection
//                 rectx_bits = minx | maxx<<12 | offx<<24,  recty_bits likewise (tile units; off: RECT_OFF_MAX); a colour whose SH sum was
//                 clamped at 0 is stored as -0.0f (the clamp flag of the backward)
//   dkey[g]       float bits of view depth (positive => order-preserving), 0xFFFFFFFF for a Gaussian that emits no pair
//   tcnt[g]       tiles of its (tightened) rect; offg[g] = exclusive scan of tcnt in storage order (numbers the
//                 backward's partial rows), offg[P] = N
//   order[r]      Gaussian index of depth rank r (stable radix argsort of the live dkey), r < V
//   off[r]        exclusive scan of tiles touched in depth order, off[V] = N (numbers the emitted pairs)
//   dv[16]        device-side scalars of the forward (gsr_sort.hip.h, DV_*): N, V, kmin, digit width, overflow, ...
//   pair_tile/pair_rank[N]  (tile id, Gaussian | strip mask << 28) pairs, emitted in depth-rank order, then stably
//                 sorted by tile id (=> depth order inside a tile)
//   ranges[t]     [start,end) of tile t in the sorted pair list
//   final_T, n_contrib [H*W]  per-pixel transmittance / last contributing list position (1-based)
//   part[N][12]   backward: per-(tile,Gaussian) partial sums written at the pair's storage-order slot, so
//                 that the rows of one Gaussian are contiguous and K8/K9 reduces them without atomics
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_math.h"
#include "gsr_sort.hip.h"

namespace gsr {

struct ViewArgs {
  const float* vm;
  const float* pm;
  const float* cam;
  int H, W;
  float tanfovx, tanfovy, mod;
  int deg;
};

__device__ __forceinline__ void load_view(View& v, const ViewArgs& a) {
  make_view(v, a.vm, a.pm, a.cam, a.H, a.W, a.tanfovx, a.tanfovy, a.mod, a.deg);
}

// ------------------------------------------------------------------------------------------------
// A BATCH of views (gsr_forward_raw_batch; reference attack.py:476-485 renders the B cameras of a batch one after another
// from one set of attributes).  The batch is laid out as ONE virtual scene: view v owns the virtual Gaussians
// [v * Ppad, v * Ppad + P) (Ppad = P rounded up to a multiple of BATCH_PAD, the tail of a view's range emits nothing) and the
// virtual tiles [v * T, (v + 1) * T); scans, both sorts, the emission, the tile schedule and the two compositors then run
// ONCE over B * Ppad Gaussians and B * T tiles -- the stable tile sort keeps every (view, tile) list in depth order because
// the global depth order restricted to one view is that view's depth order.  ViewDev: what a compositor needs of view v
// (its background) and what the per-Gaussian kernels need (matrices, camera position, tan(fov / 2)), gathered once per
// forward from the B settings' device tensors.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_BATCH = 16;
constexpr int BATCH_PAD = 16384;              // a multiple of every chunk size the per-Gaussian kernels and scans use
struct ViewDev {
  float vm[16];
  float pm[16];
  float cam[4];
  float bg[4];
  float tanfovx, tanfovy, pad0, pad1;
};
struct ViewPtrs {
  const float* vm[MAX_BATCH];
  const float* pm[MAX_BATCH];
  const float* cam[MAX_BATCH];
  const float* bg[MAX_BATCH];
  float tanfovx[MAX_BATCH], tanfovy[MAX_BATCH];
};
__global__ void __launch_bounds__(64) k_pack_views(ViewPtrs p, int B, ViewDev* __restrict__ out) {
  const int v = blockIdx.x, t = threadIdx.x;
  if (v >= B) return;
  ViewDev& o = out[v];
  if (t < 16) { o.vm[t] = p.vm[v][t]; o.pm[t] = p.pm[v][t]; }
  if (t < 3) { o.cam[t] = p.cam[v][t]; o.bg[t] = p.bg[v][t]; }
  if (t == 3) { o.cam[3] = 0.f; o.bg[3] = 0.f; o.tanfovx = p.tanfovx[v]; o.tanfovy = p.tanfovy[v]; o.pad0 = 0.f; o.pad1 = 0.f; }
}

constexpr uint32_t RECT_MASK = 0xFFFu;
// Bits 24..31 of a rect word: how many tiles the TIGHTENED rect's first tile lies behind the first tile of the reference's
// 3-sigma rect.  The stored centre is relative to the latter -- the same origin whether or not the footprint cull ran, so
// that its float32 rounding, and with it every pixel, is bit-identical under GSR_FLAG_NO_CULL -- and the compositors get
// it back as (tightened min - this offset).  A tightening of more than 255 tiles is cut short (a few more pairs emitted).
constexpr int RECT_OFF_MAX = 255;
// The three float4 of a splat record are interleaved (48 contiguous bytes per Gaussian): X0/X1/X2 below are the same
// array offset by 0/1/2 float4 and are indexed [REC * i].  REC = 3: 48 contiguous bytes per Gaussian, half of the records
// straddle two 64-byte sectors.  -DGSR_REC=4 puts them at a 64-byte pitch (one sector per gather): measured in round 3,
// the compositors' counted traffic falls (K6 597 -> 500 MB, K7 749 -> 686 MB) and their run time does not move, while
// every small kernel that touches the records gets a little slower (one-stream rate 1123 -> 1111 views/s): not the default.
#ifndef GSR_REC
#define GSR_REC 3
#endif
constexpr int REC = GSR_REC;
constexpr int RANK_BITS = 28;                       // pair value = depth rank | strip mask << 28
constexpr uint32_t RANK_MASK = (1u << RANK_BITS) - 1u;

// ------------------------------------------------------------------------------------------------
// K1 and K8+K9 launch shape.  Their waves never talk to each other about Gaussians (each wave owns 64 of them, wave
// barriers only); a workgroup is a dispatch granule plus, in K1, the unit that hands the storage-order scan its partial
// sums: per workgroup the tiles touched by its Gaussians and the smallest / largest live depth key (the depth sort
// chooses its digit width from that range), so that no separate reduction pass over tcnt / dkey is needed.
// ------------------------------------------------------------------------------------------------
constexpr int PRE_WAVES = 1;                 // K8+K9: one wave + its LDS finds room beside other streams' compositing waves
constexpr int PRE_BLOCK = 64 * PRE_WAVES;
constexpr int PREF_WAVES = 4;                // K1, colour half
constexpr int PREF_BLOCK = 64 * PREF_WAVES;
constexpr int PREG_WAVES = 4;                // K1, geometry half (small workgroups find room beside other streams' compositing waves)
constexpr int PREG_BLOCK = 64 * PREG_WAVES;

struct PreBlockOut {
  uint4* bout;       // [ceil(P / PREG_BLOCK)] per workgroup: (tiles touched, smallest live depth key or 0xFFFFFFFF, largest or 0, -)
  uint2* ranges;     // [ntiles] set to the empty span (0xFFFFFFFF, 0) here (one tile per thread) when P >= ntiles
  int ntiles;
};

// cnt: tiles this thread's Gaussian touches; key: its depth key, 0xFFFFFFFF when it emits nothing
__device__ __forceinline__ void pre_block_epilogue(const PreBlockOut& o, int g, uint32_t cnt, uint32_t key) {
  __shared__ uint32_t red[3][PREG_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (o.ranges && g < o.ntiles) o.ranges[g] = make_uint2(0xFFFFFFFFu, 0u);     // (start, end) for atomicMin / atomicMax
  uint32_t s = cnt, mn = key, mx = key == 0xFFFFFFFFu ? 0u : key;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s += (uint32_t)__shfl_xor((int)s, d, 64);
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, 64));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  }
  if (lane == 0) { red[0][wave] = s; red[1][wave] = mn; red[2][wave] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t ts = 0, tmn = 0xFFFFFFFFu, tmx = 0u;
#pragma unroll
    for (int w = 0; w < PREG_WAVES; ++w) { ts += red[0][w]; tmn = min(tmn, red[1][w]); tmx = max(tmx, red[2][w]); }
    o.bout[blockIdx.x] = make_uint4(ts, tmn, tmx, 0u);
  }
}

// ------------------------------------------------------------------------------------------------
// K1, generic form: one thread per Gaussian in storage order, any SH coefficient count K (the layouts the reference
// uses -- K = 16, the raw dc | rest pair, precomputed colours -- take k_pre_geom / k_pre_color below).
// cull bit 0: the tile rect is shrunk to the alpha >= 1/255 footprint's bounding box (tighten_rect); bit 1: needles get their
// conic from the double chain (GSR_FLAG_NEEDLE_DOUBLE).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PREG_BLOCK) k_preprocess(int P, int K, ViewArgs va, int cull, const float* __restrict__ means,
                                                    const float* __restrict__ scales, const float* __restrict__ rots,
                                                    const float* __restrict__ cov3d, const float* __restrict__ opac,
                                                    const float* __restrict__ sh, const float* __restrict__ colors,
                                                    int32_t* __restrict__ radii, float4* __restrict__ G0,
                                                    float4* __restrict__ G1, float4* __restrict__ G2,
                                                    uint32_t* __restrict__ dkey, uint32_t* __restrict__ tcnt, PreBlockOut bo) {
  const int g = blockIdx.x * PREG_BLOCK + threadIdx.x;
  uint32_t cnt = 0, key = 0xFFFFFFFFu;
  if (g < P) {
    View v;
    load_view(v, va);
    const float p[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
    float c6[6];
    if (cov3d) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c6[i] = cov3d[6 * g + i];
    } else {
      const float sc[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
      const float4 q4 = reinterpret_cast<const float4*>(rots)[g];
      const float q[4] = {q4.x, q4.y, q4.z, q4.w};
      cov3d_from_scale_rot(sc, va.mod, q, c6);
    }
    Splat s;
    const float o = opac[g];
    const bool ok = project_splat(v, p, c6, s) && opacity_ok(o);
    radii[g] = ok ? s.radius : 0;
    if (ok) {
      if ((cull & 2) && is_needle(s.ca, s.cb, s.cc)) {       // cull bit 1: GSR_FLAG_NEEDLE_DOUBLE
        // a needle's conic (compositors and footprint tests): the same chain in double (gsr_math.h cov2d_accurate)
        float scd[3] = {0.f, 0.f, 0.f}, qd[4] = {0.f, 0.f, 0.f, 0.f};
        if (!cov3d) {
          scd[0] = scales[3 * g]; scd[1] = scales[3 * g + 1]; scd[2] = scales[3 * g + 2];
          const float4 q4d = reinterpret_cast<const float4*>(rots)[g];
          qd[0] = q4d.x; qd[1] = q4d.y; qd[2] = q4d.z; qd[3] = q4d.w;
        }
        double ca, cb, cc;
        cov2d_accurate(v, p, scd, va.mod, qd, cov3d ? c6 : nullptr, ca, cb, cc);
        needle_conic_to_float(ca, cb, cc, s.A, s.B, s.C);
      }
      const int fminx = s.rminx, fminy = s.rminy;      // first tile of the reference's rect: the origin of the stored centre
      if (cull & 1) tighten_rect(s.px, s.py, s.A, s.B, s.C, o, v.gridx, v.gridy, s.rminx, s.rminy, s.rmaxx, s.rmaxy);
      s.rminx = min(s.rminx, fminx + RECT_OFF_MAX); s.rminy = min(s.rminy, fminy + RECT_OFF_MAX);
      cnt = (uint32_t)((s.rmaxx - s.rminx) * (s.rmaxy - s.rminy));
      if (cnt != 0u) {
        float rgb[3];
        uint32_t cl = 0;
        if (colors) { rgb[0] = colors[3 * g]; rgb[1] = colors[3 * g + 1]; rgb[2] = colors[3 * g + 2]; }
        else cl = sh_to_rgb(va.deg, sh + (size_t)g * K * 3, p, v.cam, rgb);
        key = __float_as_uint(s.depth);
        const uint32_t rx = (uint32_t)s.rminx | ((uint32_t)s.rmaxx << 12) | ((uint32_t)(s.rminx - fminx) << 24);
        const uint32_t ry = (uint32_t)s.rminy | ((uint32_t)s.rmaxy << 12) | ((uint32_t)(s.rminy - fminy) << 24);
        // a clamped colour is stored as -0.0f: the backward reads the clamp flags off the sign bits
        if (cl & 1u) rgb[0] = -0.0f;
        if (cl & 2u) rgb[1] = -0.0f;
        if (cl & 4u) rgb[2] = -0.0f;
        G0[REC * g] = make_float4((float)(s.pxd - (double)(fminx * TILE)), (float)(s.pyd - (double)(fminy * TILE)), s.A, s.B);
        G1[REC * g] = make_float4(s.C, o, rgb[0], rgb[1]);
        G2[REC * g] = make_float4(rgb[2], s.depth, __uint_as_float(rx), __uint_as_float(ry));
      }
    }
    dkey[g] = key;
    tcnt[g] = cnt;
  }
  pre_block_epilogue(bo, g, cnt, key);
}

// K10
__global__ void __launch_bounds__(256) k_mark_visible(int P, const float* __restrict__ vm,
                                                      const float* __restrict__ means, uint8_t* __restrict__ present) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P) return;
  const float z = means[3 * g] * vm[2] + means[3 * g + 1] * vm[6] + means[3 * g + 2] * vm[10] + vm[14];
  present[g] = z > NEAR_Z ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// K2: storage-order numbering of the (tile, Gaussian) pairs + everything the depth sort needs before its first scatter,
// in ONE launch behind K1.  Block b owns DCHUNK = 2048 Gaussians (= one chunk of the depth sort's first pass):
//   * its carry is the sum of K1's workgroup sums in front of it (a few thousand L2-resident words, summed redundantly
//     by every block instead of by a dependent single-block kernel), likewise the global min / max of the live depth keys;
//   * offg[g] = exclusive scan of tiles touched in storage order (numbers the backward's partial rows; offg[P] = total);
//   * the digit width w of the depth sort from the key range, and this chunk's histogram of the first digit of the live
//     keys (culled Gaussians are not counted: the first scatter pass drops them), written to the radix table;
//   * the last block publishes the device-side scalars: pair count (exact 64-bit, and the 32-bit count the rest of
//     the forward uses -- 0 with the overflow flag set when it exceeds `cap` or 2^31), kmin, w.
// ------------------------------------------------------------------------------------------------
constexpr int DROUNDS = RS_ROUNDS_MIN;               // the depth sort always works in 2048-key chunks (LDS: 2048 bins)
constexpr int DCHUNK = rs_chunk(DROUNDS);
constexpr int DITEMS = DCHUNK / 256;
static_assert(DCHUNK % PREG_BLOCK == 0, "a scan chunk covers whole K1 workgroups");

// Every block of K2 adds up K1's per-workgroup sums in front of its chunk: 3907 words at 1 M Gaussians, but quadratic in
// the Gaussian count (17 us at 1 M, 45 us at 2 M, and a batch of views is one scene of B * P).  Above K1_GROUP_MIN
// workgroups a tiny launch in front of K2 folds every K1_GROUP consecutive workgroup sums into one (sum as 64 bits in
// .x | .w << 32, min, max), and a K2 block reads the group sums plus the at most K1_GROUP - 1 workgroups of its own group.
constexpr int K1_GROUP = 64;
constexpr uint32_t K1_GROUP_MIN = 4096;
static_assert((K1_GROUP * PREG_BLOCK) % rs_chunk(RS_ROUNDS_MIN) == 0, "a group covers whole scan chunks");
__global__ void __launch_bounds__(K1_GROUP) k_bout_group_sum(const uint4* __restrict__ bout, uint32_t nk1, uint4* __restrict__ gsum) {
  const uint32_t i = blockIdx.x * K1_GROUP + threadIdx.x;
  unsigned long long s = 0;
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  if (i < nk1) { const uint4 bo = bout[i]; s = bo.x; mn = bo.y; mx = bo.z; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s += __shfl_xor(s, d, 64);
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, 64));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  }
  if (threadIdx.x == 0) gsum[blockIdx.x] = make_uint4((uint32_t)s, mn, mx, (uint32_t)(s >> 32));
}

__global__ void __launch_bounds__(256) k_storage_scan_hist(uint32_t P, const uint32_t* __restrict__ tcnt,
                                                           const uint32_t* __restrict__ dkey,
                                                           const uint4* __restrict__ bout, const uint4* __restrict__ gsum,
                                                           uint32_t* __restrict__ offg,
                                                           uint32_t* __restrict__ table, uint32_t nb,
                                                           uint32_t* __restrict__ dv, unsigned long long cap,
                                                           uint32_t* host_slot, uint32_t host_token, uint32_t npass) {
  __shared__ uint32_t tmp[4];
  __shared__ unsigned long long tmp64[4];
  __shared__ uint32_t h[RS_BINS_DEV];
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  const uint32_t base = b * DCHUNK + tid * DITEMS;
  // this chunk's tiles-touched counts and depth keys are requested first: they do not depend on the reduction below
  uint32_t v[DITEMS], kk[DITEMS];
#pragma unroll
  for (int i = 0; i < DITEMS; ++i) {
    const bool in = base + i < P;
    v[i] = in ? tcnt[base + i] : 0u;
    kk[i] = in ? dkey[base + i] : RS_DROP_KEY;
  }
  const uint32_t nk1 = (P + PREG_BLOCK - 1) / PREG_BLOCK;
  const uint32_t front = b * (DCHUNK / PREG_BLOCK);     // K1 workgroups in front of this chunk
  uint32_t part = 0, mn = 0xFFFFFFFFu, mx = 0u;
  unsigned long long all = 0;
  if (gsum != nullptr) {
    const uint32_t ng = (nk1 + K1_GROUP - 1) / K1_GROUP, gf = front / K1_GROUP;
    for (uint32_t i = tid; i < ng; i += 256) {
      const uint4 gs = gsum[i];
      part += i < gf ? gs.x : 0u;
      all += (unsigned long long)gs.x | ((unsigned long long)gs.w << 32);
      mn = min(mn, gs.y); mx = max(mx, gs.z);
    }
    const uint32_t i = gf * K1_GROUP + tid;                // the workgroups of the chunk's own group in front of it
    if (tid < K1_GROUP && i < front) part += bout[i].x;
  } else {
    for (uint32_t i = tid; i < nk1; i += 256) {
      const uint4 bo = bout[i];
      part += i < front ? bo.x : 0u;
      all += bo.x;
      mn = min(mn, bo.y); mx = max(mx, bo.z);
    }
  }
  uint32_t carry;
  block_excl_scan_256(part, tmp, carry);
  const uint32_t kmin = block_min_u32(mn, tmp), kmax = block_max_u32(mx, tmp);
  const uint32_t w = depth_digit_width(kmin, kmax, npass), mask = (1u << w) - 1u, nbins = 1u << w;
  for (uint32_t d = tid; d < nbins; d += 256) h[d] = 0;
  __syncthreads();
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < DITEMS; ++i) {
    sum += v[i];
    if (kk[i] != RS_DROP_KEY) atomicAdd(&h[(kk[i] - kmin) & mask], 1u);
  }
  uint32_t total;
  uint32_t run = block_excl_scan_256(sum, tmp, total) + carry;
#pragma unroll
  for (int i = 0; i < DITEMS; ++i) {
    if (base + i < P) offg[base + i] = run;
    run += v[i];
  }
  __syncthreads();
  for (uint32_t d = tid; d < nbins; d += 256) table[d * nb + b] = h[d];
  if (b == nb - 1) {
    const unsigned long long n64 = block_sum_u64(all, tmp64);
    if (tid == 0) {
      offg[P] = carry + total;
      const bool ovf = n64 > cap;
      dv[DV_N] = ovf ? 0u : (uint32_t)n64;
      dv[DV_KMIN] = kmin; dv[DV_W] = w; dv[DV_OVF] = ovf ? 1u : 0u;
      dv[DV_NREC] = 0u;                                      // (a batch's tile schedule counts the boundary records up from here)
      dv[DV_N64] = (uint32_t)n64; dv[DV_N64 + 1] = (uint32_t)(n64 >> 32);
      if (host_slot) {
        // pinned, device-mapped host memory: count and flag first, then -- behind a system-scope fence -- the token the
        // host is polling for (words: 0,1 = count, 2 = overflow flag, 3 = token)
        host_slot[0] = (uint32_t)n64; host_slot[1] = (uint32_t)(n64 >> 32); host_slot[2] = ovf ? 1u : 0u;
        __threadfence_system();
        __hip_atomic_store(&host_slot[3], host_token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K3: emit (tile, Gaussian) pairs, one thread per OUTPUT slot (perfectly balanced, coalesced writes), in depth-rank
// order.  A block emits one chunk of the tile sort's first pass and leaves that pass's digit histogram of its chunk in
// the radix table: the sort's first histogram launch (a full read of the keys) disappears.
// ------------------------------------------------------------------------------------------------
// A (tile, Gaussian) pair of the Gaussian's tile rect is kept only if some pixel of the tile can pass
// the reference's own alpha test (alpha >= 1/255); pairs that cannot are given the key `ntiles`, sort to the
// tail and are never composited.  The same test per 16x4 strip gives a 4-bit mask that rides in the top bits of
// the pair's value, so K6/K7 skip strips with scalar bit tests instead of evaluating the splat there.  The rendered
// image, radii and gradients are unchanged by construction (the test is conservative); only the work shrinks.
// `cull` = 0 keeps every pair.  dv: device-side counts (DV_N pairs, DV_V ranks).
// A block = CHUNK / 8 threads emits CHUNK slots, eight per thread with their dependent chains (owner -> order[] ->
// record -> mask) interleaved.  The owner of a slot comes from a running maximum over marks in LDS, not from a search.
constexpr int EMIT_PER_THREAD = 8;        // slots per thread, in EMIT_BATCHES rounds of EMIT_ILV interleaved chains
constexpr int EMIT_ILV = 4;
constexpr int EMIT_GRAIN = rs_chunk(RS_ROUNDS_MIN);     // slots per chunk_first entry (the smaller of the two chunk sizes)
template <int ROUNDS>
__global__ void __launch_bounds__(rs_chunk(ROUNDS) / EMIT_PER_THREAD)
k_emit(const uint32_t* __restrict__ off, const uint32_t* __restrict__ order, const uint32_t* __restrict__ chunk_first,
       const uint32_t* __restrict__ dv, const float4* __restrict__ R0, const float4* __restrict__ R1, const float4* __restrict__ R2, int gridx, int W, int H,
       uint32_t ntiles, int cull, uint32_t* __restrict__ pair_tile, uint32_t* __restrict__ pair_rank,
       uint32_t* __restrict__ table, uint32_t nb, uint32_t digit_mask, uint32_t ppad, uint32_t ppad_magic, uint32_t tpv) {
  // ppad != 0: a batch of views as one virtual scene (see ViewDev) -- Gaussian g belongs to view g / ppad (ppad_magic =
  // ceil(2^32 / ppad): the quotient from one multiply and one correction), whose tiles are keyed from view * tpv on;
  // `ntiles` is then the batch's tile count B * tpv (the key of a culled pair)
  constexpr int CHUNK = rs_chunk(ROUNDS);
  constexpr int THREADS = CHUNK / EMIT_PER_THREAD;
  constexpr int WAVES = THREADS / 64;
  __shared__ uint32_t s_off[CHUNK + 1];
  __shared__ __attribute__((aligned(16))) uint32_t s_own[CHUNK];   // per slot: its owner's index in s_off
  __shared__ uint32_t s_r[2];
  __shared__ uint32_t s_w[WAVES];
  __shared__ uint32_t h[RS_BINS];
  const uint32_t N = dv[DV_N], V = dv[DV_V];
  const uint32_t e0 = blockIdx.x * CHUNK;
  if (e0 >= N) return;
  const uint32_t e1 = min(e0 + (uint32_t)CHUNK, N);
  // first and last rank with slots in this chunk, from the owners of the chunk starts that the rank scan recorded
  // (chunk_first, one entry per EMIT_GRAIN slots): no search of off[] in global memory
  if (threadIdx.x == 0) s_r[0] = chunk_first[blockIdx.x * (CHUNK / EMIT_GRAIN)];
  if (threadIdx.x == 64) {
    uint32_t rh = V - 1u;
    if (e1 < N) {                                           // another chunk follows: e1 is its first slot
      const uint32_t rn = chunk_first[(blockIdx.x + 1u) * (CHUNK / EMIT_GRAIN)];
      rh = off[rn] == e1 ? rn - 1u : rn;                    // ranks all own at least one slot: off[] is strictly increasing
    }
    s_r[1] = rh;
  }
  if (threadIdx.x < RS_BINS) h[threadIdx.x] = 0;
  uint4* s_own4 = reinterpret_cast<uint4*>(s_own);
  s_own4[2 * threadIdx.x] = make_uint4(0, 0, 0, 0);
  s_own4[2 * threadIdx.x + 1] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const uint32_t r_lo = s_r[0], r_hi = s_r[1];
  const uint32_t span = r_hi - r_lo + 1;                    // <= CHUNK: every rank of the chunk owns a slot of it
  // Slot -> owner without a search: rank i of the chunk marks the slot its run starts on (rank 0's run starts at or
  // before e0: it owns the slots in front of the first mark), and a running maximum over the slots spreads the marks.
  for (uint32_t i = threadIdx.x; i < span; i += THREADS) {
    const uint32_t v = off[r_lo + i];
    s_off[i] = v;
    if (i) s_own[v - e0] = i;
  }
  __syncthreads();
  {
    uint4 a = s_own4[2 * threadIdx.x], b = s_own4[2 * threadIdx.x + 1];
    a.y = max(a.y, a.x); a.z = max(a.z, a.y); a.w = max(a.w, a.z);
    b.x = max(b.x, a.w); b.y = max(b.y, b.x); b.z = max(b.z, b.y); b.w = max(b.w, b.z);
    uint32_t t = b.w;                                       // inclusive maximum over the lanes below, then the waves below
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = __shfl_up(t, d, 64);
      if ((int)(threadIdx.x & 63) >= d) t = max(t, u);
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = t;
    uint32_t c = __shfl_up(t, 1, 64);
    if ((threadIdx.x & 63) == 0) c = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES - 1; ++w) if ((int)(threadIdx.x >> 6) > w) c = max(c, s_w[w]);
    a.x = max(a.x, c); a.y = max(a.y, c); a.z = max(a.z, c); a.w = max(a.w, c);
    b.x = max(b.x, c); b.y = max(b.y, c); b.z = max(b.z, c); b.w = max(b.w, c);
    s_own4[2 * threadIdx.x] = a; s_own4[2 * threadIdx.x + 1] = b;
  }
  __syncthreads();
#pragma unroll 1
  for (int bt = 0; bt < EMIT_PER_THREAD / EMIT_ILV; ++bt) {
  uint32_t e[EMIT_ILV], o[EMIT_ILV], g[EMIT_ILV];
  bool on[EMIT_ILV];
#pragma unroll
  for (int k = 0; k < EMIT_ILV; ++k) {
    e[k] = e0 + threadIdx.x + (bt * EMIT_ILV + k) * THREADS;
    on[k] = e[k] < e1;
    uint32_t r = r_lo;
    o[k] = 0;
    if (on[k]) {
      const uint32_t i = s_own[e[k] - e0];
      r = r_lo + i; o[k] = s_off[i];
    }
    g[k] = on[k] ? order[r] : 0u;                          // the pair's value: the Gaussian (storage index)
  }
  float4 ra[EMIT_ILV], rb[EMIT_ILV], rc[EMIT_ILV];
#pragma unroll
  for (int k = 0; k < EMIT_ILV; ++k) {
    rc[k] = R2[REC * g[k]];
    if (cull) { ra[k] = R0[REC * g[k]]; rb[k] = R1[REC * g[k]]; }
  }
#pragma unroll
  for (int k = 0; k < EMIT_ILV; ++k) {
    if (!on[k]) continue;
    const uint32_t rx = __float_as_uint(rc[k].z), ry = __float_as_uint(rc[k].w);
    const uint32_t minx = rx & RECT_MASK, wx = ((rx >> 12) & RECT_MASK) - minx, miny = ry & RECT_MASK;
    const uint32_t local = e[k] - o[k];
    // local / wx through a float reciprocal (local < 2^24: a rect has at most 4095^2 tiles), corrected by one step
    uint32_t dy = (uint32_t)((float)local * __builtin_amdgcn_rcpf((float)wx));
    dy -= (dy * wx > local) ? 1u : 0u;
    dy += ((dy + 1u) * wx <= local) ? 1u : 0u;
    const uint32_t dx = local - dy * wx;
    const uint32_t tx = minx + dx, ty = miny + dy;
    uint32_t key = ty * (uint32_t)gridx + tx;
    if (ppad != 0u) {
      uint32_t view = __umulhi(g[k], ppad_magic);          // floor(g / ppad) or one more
      view -= (view * ppad > g[k]) ? 1u : 0u;
      key += view * tpv;
    }
    uint32_t mask = 0xFu;   // one bit per 16x4 strip of the tile that the Gaussian can reach
    if (cull) {
      const float x0 = (float)(tx * TILE);
      const float x1 = fminf(x0 + (float)(TILE - 1), (float)(W - 1));
      // (the record's centre is relative to the rect's first tile: back to image coordinates for the footprint test)
      mask = strip_masks4(ra[k].x + (float)((int)(minx - (rx >> 24)) * TILE), ra[k].y + (float)((int)(miny - (ry >> 24)) * TILE),
                          ra[k].z, ra[k].w, rb[k].x, rb[k].y, x0,
                          x1, (float)(ty * TILE), (float)(H - 1));
      if (mask == 0) key = ntiles;
    }
    pair_tile[e[k]] = key;
    pair_rank[e[k]] = g[k] | (mask << RANK_BITS);
    atomicAdd(&h[key & digit_mask], 1u);
  }
  }
  __syncthreads();
  if (threadIdx.x <= digit_mask) table[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// K5 (tile ranges) has no kernel of its own: the last pass of the tile sort leaves every tile's span of the sorted list
// in ranges[] (gsr_sort.hip.h, key_ranges), and k_tile_schedule turns the spans of tiles without pairs into (0, 0).

// ------------------------------------------------------------------------------------------------
// K6 / K7 common: one WAVE (= one 64-thread workgroup) per 16x16 tile, 4 pixels per lane
// (rows y0 + {0,4,8,12} + lane/16).  A list entry is staged once in LDS and broadcast to 256 pixels;
// there is no block barrier anywhere; each 16x4 strip is entered only if a conservative wave-ballot test
// says some pixel of it can pass the reference's alpha test, and the strip body itself is branch-free
// (selects), so the exec mask is never juggled.  Blocks b and b+8 land on the same XCD (observed
// round-robin dispatch): each XCD gets one contiguous band of tiles so neighbouring tiles share their
// splat records in that XCD's L2.
// ------------------------------------------------------------------------------------------------
struct RenderArgs {
  const uint2* ranges;
  const uint32_t* pair_rank;
  const float4* R0;
  const float4* R1;
  const float4* R2;
  const float* sh_objs;   // [P,16] or null
  const float* sh_objs_b; // second segment's object features (Gaussians g >= Pa), see PreArgs
  int Pa;
  const float* bg;
  int W, H, gridx, ntiles, map_mode;
  const uint32_t* sched;      // map mode 3: tiles longest-list-first + priority class (k_tile_schedule)
  float* out_color;       // [3,H,W]
  float* out_objects;     // [16,H,W] or null
  float* final_T;         // [H*W]
  uint32_t* n_contrib;    // [H*W]
  // segment boundaries of long tile lists (see "Segments" below); bnd == nullptr: none are kept
  float4* bnd;            // [records][256] (T, C_r, C_g, C_b) per pixel, pixel = strip * 64 + lane
  const uint32_t* segoff; // [ntiles] first boundary record of the tile, SEG_NONE for tiles that are not split
  uint32_t seg_shift;     // log2 of the segment length
  const uint32_t* dv;     // device-side scalars of an asynchronous-count forward (dv[DV_OVF] != 0: the image is poisoned), or null
  unsigned long long* wave_clock;   // diagnostic (gsr_debug_wave_clock_fwd): [ntiles * NSUB][2] start/end, 100 MHz
  // a batch of views (see ViewDev): ntiles = B * tpv tile ids, view = tile / tpv; the per-pixel arrays and out_color hold
  // the B images one after another; vpack[view].bg is that view's background.  One view: tpv = ntiles, vpack = null.
  int tpv;
  const ViewDev* vpack;
};

// ------------------------------------------------------------------------------------------------
// Segments.  Front-to-back compositing is associative in (T, C): the state after list position b, (T_b, C_b), is all
// that the entries behind b need from the entries in front of it.  The forward walk of a tile whose list is longer
// than one segment (2^seg_shift entries) stores that state per pixel at every segment boundary, plus its final (T, C);
// the backward then walks every SEGMENT as an independent work item on its own wave: a pixel whose last contributor
// lies behind the segment starts from T_b and from the colour it shows behind b,
//     Acc_b = ((C_final - C_b) . g + T_final (bg . g)) / T_b,
// instead of from (T_final, bg . g) -- the same two recursions the unsplit walk carries (T /= 1 - alpha,
// Acc = alpha c.g + (1 - alpha) Acc), cut at b.  The serial walk of the ~1000-entry lists, which used to set the run
// time of the backward composite long after the average SIMD had run dry, becomes 2^seg_shift entries at most.
// A tile's records: one per interior boundary j = 1 .. nseg-1 (record segoff + j - 1), then the final state
// (record segoff + nseg - 1).  rec_item[r] = {tile, j} names the extra backward work item of record r (j = 0: final
// record, no item).  Tiles that are not split (segoff = SEG_NONE) run exactly the code they ran before.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SEG_NONE = 0xFFFFFFFFu;

constexpr int PXL = 4;   // pixels per lane
constexpr float LOG2E = 1.4426950408889634f;

// Work-item index of this block.  mode 1: blocks b and b+8 run on one XCD (observed round-robin dispatch), so XCD x
// is handed the contiguous band [x*ceil(n/8), ...): neighbouring tiles share one L2.  mode 0: identity (items are
// dealt round-robin over the XCDs: no L2 sharing, but image regions of different cost spread over all XCDs).
// mode 2: bands of 32 consecutive items are dealt round-robin: local sharing inside a band, global spread.
__device__ __forceinline__ int item_of_block(int b, int nitems, int mode) {
  if (mode == 0) return b;
  if (mode == 1) {
    const int tpx = (nitems + 7) >> 3;
    return (b >> 3) < tpx ? (b & 7) * tpx + (b >> 3) : nitems;   // the grid is padded: surplus blocks exit
  }
  // mode 2: block b -> XCD x = b&7, slot s = b>>3 on that XCD; XCD x owns bands x, x+8, x+16, ... of 32 items
  const int x = b & 7, s = b >> 3;
  return ((s >> 5) * 8 + x) * 32 + (s & 31);
}

// ------------------------------------------------------------------------------------------------
// Tile schedule (map mode 3, the default).  A tile's list is walked serially by its wave(s), so the longest list
// is the critical path of K6 and K7 and, sharing its SIMD with three to seven other waves, it sets the kernels'
// run time long after the average SIMD has run dry.  sched[] lists the tiles longest-first (counting sort on
// len/4, one block) so long lists start at t = 0 and short ones fill in behind them, and carries a 2-bit
// priority class (length relative to the longest list) that the render kernels hand to s_setprio: the long
// lists issue ahead of their SIMD's other waves.  Results do not depend on the schedule.
// ------------------------------------------------------------------------------------------------
constexpr int SCHED_BINS = 4096;                      // one bin per list length (clamped): few same-bin LDS atomics
constexpr int SCHED_LDS_TILES = 65536;               // list lengths kept in LDS as 16-bit words (a 4K image has 32 400 tiles)
constexpr uint32_t SCHED_TILE_MASK = (1u << 28) - 1u;
__device__ __forceinline__ uint32_t wave_max_u32_fwd(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}
// exclusive scan of one value per thread over the 1024-thread block (16 waves); `ws` is 16 words of LDS
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* ws, uint32_t& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();                                   // ws may still be read from a previous call
  if (lane == 63) ws[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t x = ws[w];
    if (w < wave) base += x;
    tot += x;
  }
  total = tot;
  return base + inc - v;
}
// seg_shift = 0: no tile is split.  rec_cap: capacity of the boundary-record buffer (a tile whose records would not
// fit stays unsplit); nrec_out receives the number of records in use.  Tiles whose span is still the empty
// (0xFFFFFFFF, 0) the tile sort starts from are given (0, 0).
// Dynamic LDS: 2 * min(ntiles, SCHED_LDS_TILES) bytes (sched_lds_bytes) -- the tiles' list lengths are read from HBM
// once, eight loads in flight per thread, and kept as 16-bit words for the two passes behind the histogram (a length
// of 65535 or more is stored as 0xFFFF and read again from `ranges` where its exact value matters).  Until round 4 the
// lengths of images with more than 12288 tiles were re-read from HBM one dependent load at a time: 92 us at 4K.
// lds_cap: how many list lengths the launch's dynamic LDS holds -- SCHED_LDS_TILES when the 64 KB default limit could be
// raised (gfx950: 160 KB per workgroup), SCHED_LDS_TILES_DEFAULT otherwise; tiles beyond it are re-read from `ranges`.
constexpr int SCHED_LDS_TILES_DEFAULT = 23552;       // 46 KB of 16-bit lengths + the 16 KB histogram + scalars < 64 KB
inline size_t sched_lds_bytes(int ntiles, int lds_cap) { return 2u * (size_t)((std::min(ntiles, lds_cap) + 7) & ~7); }
// A batch of B views (ViewDev): the launch has B workgroups per role -- workgroup = (role, view) -- each working on its view's
// `ntiles` tiles [view * ntiles, (view + 1) * ntiles) exactly as the single-view launch does on its image; the B
// longest-first orders are interleaved (rank r of view v goes to position r * B + v: the views of one scene have similar
// length distributions, and only the order of the last few rounds of waves matters), priorities are relative to the view's
// own longest list, and the boundary records of a view's split tiles are numbered from a base the view takes from the shared
// counter *nrec_out by one atomic add (the caller -- K2 -- zeroed it; records need unique ranges, not tile order).
__global__ void __launch_bounds__(1024) k_tile_schedule(int ntiles, int lds_cap, uint2* __restrict__ ranges,
                                                        uint32_t* __restrict__ sched, uint32_t seg_shift,
                                                        uint32_t* __restrict__ segoff, uint2* __restrict__ rec_item,
                                                        uint32_t rec_cap, uint32_t* __restrict__ nrec_out, int B,
                                                        int view_major) {
  __shared__ uint32_t hist[SCHED_BINS];
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t smax;
  __shared__ uint32_t sbase;
  extern __shared__ unsigned short slen[];
  const int t = threadIdx.x, lane = t & 63;
  const int view = (int)(blockIdx.x % (uint32_t)B), role = (int)(blockIdx.x / (uint32_t)B);
  const int tile0 = view * ntiles;                       // this workgroup's tiles: tile0 + [0, ntiles)
  ranges += tile0;
  if (segoff != nullptr) segoff += tile0;
  const int lds_tiles = min(ntiles, lds_cap);
  // list length of tile i, exact (segment plan) / clamped to 65535 (bins and priorities: both saturate far below)
  auto tile_len = [&](int i) -> uint32_t {
    if (i < lds_tiles) {
      const uint32_t v = slen[i];
      if (v != 0xFFFFu) return v;
    }
    const uint2 r = ranges[i];
    return r.y - r.x;
  };
  auto tile_len_sat = [&](int i) -> uint32_t {
    if (i < lds_tiles) return slen[i];
    const uint2 r = ranges[i];
    return min(r.y - r.x, 0xFFFFu);
  };
  // Two roles when tiles can be split (segoff != null): role 0 makes the schedule, role 1 the segment plan --
  // both from the list lengths, which each reads for itself (two CUs instead of one workgroup's phases back to back).
  const bool do_sched = role == 0;
  const bool do_plan = segoff != nullptr && role == (int)(gridDim.x / (uint32_t)B) - 1;
#pragma unroll
  for (int k = 0; k < SCHED_BINS / 1024; ++k) hist[t + k * 1024] = 0;
  if (t == 0) smax = 0;
  __syncthreads();
  uint32_t mymax = 0;
  for (int i0 = t; i0 < ntiles; i0 += 8 * 1024) {
    // eight independent loads in flight per thread
    uint2 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 1024;
      r[k] = i < ntiles ? ranges[i] : make_uint2(0u, 0u);
      if (r[k].x > r[k].y) {                 // a tile without pairs still holds the empty span (0xFFFFFFFF, 0)
        r[k] = make_uint2(0u, 0u);
        if (do_sched) ranges[i] = r[k];      // (the other block sees either form as an empty list)
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 1024;
      if (i < ntiles) {
        const uint32_t len = r[k].y - r[k].x;
        if (i < lds_tiles) slen[i] = (unsigned short)min(len, 0xFFFFu);
        if (do_sched) atomicAdd(&hist[min(len, (uint32_t)SCHED_BINS - 1u)], 1u);
        mymax = max(mymax, len);
      }
    }
  }
  if (do_sched) {
  mymax = wave_max_u32_fwd(mymax);
  if (lane == 0) atomicMax(&smax, mymax);
  __syncthreads();
  // exclusive scan over the bins in DESCENDING length order: thread t owns bins 4095 - 4t .. 4095 - 4t - 3
  {
    constexpr int BPT = SCHED_BINS / 1024;
    uint32_t c[BPT], s = 0;
#pragma unroll
    for (int k = 0; k < BPT; ++k) { c[k] = hist[SCHED_BINS - 1 - (t * BPT + k)]; s += c[k]; }
    uint32_t total;
    uint32_t run = block_excl_scan_1024(s, wsum, total);
#pragma unroll
    for (int k = 0; k < BPT; ++k) { hist[SCHED_BINS - 1 - (t * BPT + k)] = run; run += c[k]; }   // becomes the bin's cursor
  }
  __syncthreads();
  const float prio_scale = 4.0f / (float)(smax + 1u);
  for (int i0 = t; i0 < ntiles; i0 += 4 * 1024) {        // four independent cursor updates in flight per thread
    uint32_t len[4], p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * 1024;
      len[k] = i < ntiles ? tile_len_sat(i) : 0u;
      p[k] = i < ntiles ? atomicAdd(&hist[min(len[k], (uint32_t)SCHED_BINS - 1u)], 1u) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * 1024;
      if (i < ntiles) {
        const uint32_t prio = min(3u, (uint32_t)((float)len[k] * prio_scale));    // 0..3: length relative to the longest list
        // (view_major: view after view, each longest-first -- the waves in flight then gather the records of one or two views)
        const size_t pos = view_major ? (size_t)tile0 + p[k] : (size_t)p[k] * (size_t)B + view;
        sched[pos] = (uint32_t)(tile0 + i) | (prio << 28);
      }
    }
  }
  }
  if (!do_plan) return;
  __syncthreads();                                         // slen[] of this block is complete
  // ---- boundary records of the split tiles: exclusive scan of nseg over the tiles, in tile order ------------------
  // thread t owns the consecutive tiles [t * tpt, (t + 1) * tpt): ONE block scan instead of one per 1024 tiles
  const uint32_t seg_len = 1u << seg_shift;
  const int tpt = (ntiles + 1023) / 1024;
  const int i_lo = t * tpt, i_hi = min(i_lo + tpt, ntiles);
  uint32_t mine = 0;
  if (seg_shift != 0u)
    for (int i = i_lo; i < i_hi; ++i) {
      const uint32_t len = tile_len(i);
      if (len > seg_len) mine += (len + seg_len - 1u) >> seg_shift;
    }
  uint32_t total;
  uint32_t run = block_excl_scan_1024(mine, wsum, total);
  if (B > 1) {                                             // a batch: this view's records start at a base of its own
    if (t == 0) sbase = atomicAdd(nrec_out, total);
    __syncthreads();
    run += sbase;
  }
  for (int i = i_lo; i < i_hi; ++i) {
    const uint32_t len = tile_len(i);
    uint32_t nseg = 0;
    if (seg_shift != 0u && len > seg_len) nseg = (len + seg_len - 1u) >> seg_shift;
    uint32_t off = SEG_NONE;
    if (nseg != 0u && run + nseg <= rec_cap) {
      off = run;
      for (uint32_t j = 1; j < nseg; ++j) rec_item[run + j - 1u] = make_uint2((uint32_t)(tile0 + i), j);
      rec_item[run + nseg - 1u] = make_uint2((uint32_t)(tile0 + i), 0u);
    }
    segoff[i] = off;
    run += nseg;
  }
  if (B == 1 && t == 0) *nrec_out = min(total, rec_cap);
}
__device__ __forceinline__ void set_wave_priority(uint32_t prio) {
  if (prio == 3u) __builtin_amdgcn_s_setprio(3);
  else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
  else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
}
inline int render_grid(int nitems) { return 256 * ((nitems + 255) / 256); }   // covers every mapping mode

// Staged form of a splat: the conic is pre-scaled so that p2 = log2(e) * power comes out of two FMAs.  c.y is a
// slightly LOWERED bound on the p2 at which alpha reaches 1/255 (a per-pixel prefilter: the reference's exact alpha
// test is applied afterwards) whose four lowest mantissa bits are replaced by the pair's strip mask (bit k: strip k
// of the tile can be reached) -- a 2e-6 relative nudge, far inside the bound's own 1e-4 margin.
// The staged centre is relative to the first pixel of the TILE being composited (tx, ty): the record's centre is
// relative to the rect's first tile (rx, ry bits), the difference is a multiple of 16 -- two small numbers, so that
// dx = centre - column keeps float32's full resolution at any image size.
struct StagedSplat { float4 a; float4 b; float2 c; };
__device__ __forceinline__ StagedSplat stage_splat(const float4 r0, const float4 r1, const float4 r2, uint32_t mask, int tx, int ty) {
  StagedSplat s;
  const float bch = r2.x;
  const uint32_t rxw = __float_as_uint(r2.z), ryw = __float_as_uint(r2.w);
  const int minx = (int)(rxw & RECT_MASK) - (int)(rxw >> 24), miny = (int)(ryw & RECT_MASK) - (int)(ryw >> 24);
  s.a = make_float4(r0.x + (float)((minx - tx) * TILE), r0.y + (float)((miny - ty) * TILE), -0.5f * LOG2E * r0.z, -LOG2E * r0.w);
  s.b = make_float4(-0.5f * LOG2E * r1.x, r1.y, r1.z, r1.w);
  const float t = -__log2f(255.0f * r1.y);             // +inf for opacity 0: never a candidate
  const float thr = t - 1e-4f * (fabsf(t) + 1.0f);
  s.c = make_float2(bch, __uint_as_float((__float_as_uint(thr) & ~0xFu) | (mask & 0xFu)));
  return s;
}

constexpr float PX_OFF = 1.0e30f;   // y coordinate of a finished / out-of-image pixel: p2 = -inf, alpha = 0

// NPX = pixels per lane: 4 -> one wave per tile, 2 -> two waves (16x8 halves), 1 -> four waves (16x4 strips).
// Fewer pixels per wave = shorter dependent chain per list entry and more, smaller work items for the
// dispatcher to balance (a tile's list length sets its wave's run time); more = staging amortised further.
// WPB = waves per workgroup: 1 -> every wave is its own workgroup and stages the tile's list for itself; NSUB -> the
// tile's waves form ONE workgroup and stage each batch (64 entries per wave) once for all of them: the records are
// gathered once per tile instead of once per wave, at the price of two workgroup barriers per batch.
#ifndef GSR_K6_OBJ_WAVES
#define GSR_K6_OBJ_WAVES 5          // waves per SIMD the two-strip object variant is compiled for
#endif
#ifndef GSR_K6_OBJ_STAGE_ALL
#define GSR_K6_OBJ_STAGE_ALL 0      // 1: stage the object features of every list entry (rounds 1-4); 0: only of the entries
#endif                              // that reach this wave's strips
#ifndef GSR_K6_OBJ_SCALAR
#define GSR_K6_OBJ_SCALAR 0         // 0: object features staged through LDS; 1: fetched by scalar loads into SGPRs (measured
                                    // slower on S-nyc-1M: K6 0.353 ms against 0.304 -- EXPERIMENTS.md, round 5)
#endif
template <bool OBJ, int NPX, int WPB = 1>
__global__ void __launch_bounds__(64 * WPB, (OBJ && NPX == 2 && WPB == 1) ? GSR_K6_OBJ_WAVES : 1) k_render_fwd(RenderArgs a) {
  constexpr int NSUB = PXL / NPX;
  static_assert(WPB == 1 || WPB == NSUB, "a shared workgroup holds all the waves of a tile");
  constexpr int BATCH = 64 * WPB;
  __shared__ float4 s0[BATCH];
  __shared__ float4 s1[BATCH];
  __shared__ float4 s2[BATCH];                             // (blue, threshold | mask, Gaussian index, -): 16-byte pitch like s0 /
                                                           // s1, so that the three reads of an entry share ONE address register
  __shared__ __attribute__((aligned(16))) float so[(OBJ && !GSR_K6_OBJ_SCALAR) ? BATCH : 1][NUM_OBJ];
  __shared__ uint32_t salive[WPB];
  const int lane = threadIdx.x & 63;
  const int wv = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // wave-uniform, and known to be
  int item;
  if (a.map_mode == 3) {
    if ((int)blockIdx.x * WPB >= a.ntiles * NSUB) return;
    const uint32_t sc = a.sched[blockIdx.x * WPB / NSUB];
    item = (int)(sc & SCHED_TILE_MASK) * NSUB + (WPB > 1 ? wv : (int)(blockIdx.x % NSUB));
    set_wave_priority(sc >> 28);
  } else {
    item = item_of_block((int)blockIdx.x, a.ntiles * NSUB / WPB, a.map_mode) * WPB + wv;
    if (item >= a.ntiles * NSUB) return;
  }
  const int tile = item / NSUB, sub = item - tile * NSUB;
  const int view = a.tpv < a.ntiles ? __builtin_amdgcn_readfirstlane(tile / a.tpv) : 0;     // a batch of views: ViewDev
  const int ltile = tile - view * a.tpv;
  const int tx = ltile % a.gridx, ty = ltile / a.gridx;
  const uint2 rg = a.ranges[tile];
  // a split tile (see "Segments"): this wave stores (T, C) of its pixels at every segment boundary it walks past
  // (also with object channels composited: the records hold T and the colour sums, which is all a backward WITHOUT
  // dL/dobjects needs -- the attack never differentiates the object map -- to walk the list in segments; a backward that
  // is handed dL/dobjects walks whole lists, the 16 running object sums are not stored)
  const uint32_t rec0 = (a.bnd != nullptr) ? a.segoff[tile] : SEG_NONE;
  const uint32_t seg_mask = (1u << a.seg_shift) - 1u;
  if (a.wave_clock && lane == 0) a.wave_clock[2 * item] = wall_clock64();
  const int x = tx * TILE + (lane & 15);
  const float pxf = (float)(lane & 15);                    // pixel coordinates relative to the tile's first pixel (stage_splat)
  int y[NPX];
  float pyf[NPX], T[NPX], C[NPX][3];
  float O[OBJ ? NPX : 1][NUM_OBJ];
  uint32_t last[NPX];
  uint32_t alive = 0;                       // bit k: strip k still has an unfinished pixel (wave-uniform)
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    y[k] = ty * TILE + sub * (4 * NPX) + (lane >> 4) + 4 * k;
    const bool inside = x < a.W && y[k] < a.H;
    pyf[k] = inside ? (float)(y[k] - ty * TILE) : PX_OFF;
    T[k] = 1.f;
    C[k][0] = C[k][1] = C[k][2] = 0.f;
    last[k] = 0;
    if (__ballot(inside) != 0ull) alive |= 1u << k;
    if (OBJ) {
#pragma unroll
      for (int c = 0; c < NUM_OBJ; ++c) O[k][c] = 0.f;
    }
  }
  for (uint32_t base = rg.x; base < rg.y && (WPB > 1 || alive); base += BATCH) {
    const int slot = (int)threadIdx.x;                    // the batch entry this thread stages
    const uint32_t i = base + (uint32_t)slot;
    uint32_t mine = 0;
    uint32_t rown = 0;                                    // the staged entry's Gaussian (0 past the end of the list: a valid row)
    if (i < rg.y) {
      const uint32_t pv = a.pair_rank[i];
      const uint32_t r = pv & RANK_MASK;
      const float4 c = a.R2[REC * r];
      // staged mask bits: this wave's strips (own workgroup) or all four strips of the tile (shared workgroup)
      mine = WPB > 1 ? (pv >> RANK_BITS) & 0xFu : ((pv >> RANK_BITS) >> (sub * NPX)) & ((1u << NPX) - 1u);
      const StagedSplat sp = stage_splat(a.R0[REC * r], a.R1[REC * r], c, mine, tx, ty);
      rown = r;                                           // the pair's value IS the Gaussian's storage index
      s0[slot] = sp.a; s1[slot] = sp.b; s2[slot] = make_float4(sp.c.x, sp.c.y, __uint_as_float(r), 0.f);
      if (OBJ && !GSR_K6_OBJ_SCALAR && (GSR_K6_OBJ_STAGE_ALL || mine != 0u)) {   // (an entry no strip of this wave reaches is never read)
        const float4* src = reinterpret_cast<const float4*>(r >= (uint32_t)a.Pa ? a.sh_objs_b + (size_t)(r - (uint32_t)a.Pa) * NUM_OBJ
                                                                                 : a.sh_objs + (size_t)r * NUM_OBJ);
        float4* dst = reinterpret_cast<float4*>(&so[slot][0]);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
      }
    } else if (WPB > 1) {
      s2[slot] = make_float4(0.f, 0.f, 0.f, 0.f);         // past the end of the list: reaches no strip
    }
    if (WPB > 1) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int jb = 0; jb < BATCH; jb += 64) {              // the batch, 64 entries at a time
    if (WPB > 1) mine = (__float_as_uint(s2[jb + lane].y) >> (sub * NPX)) & ((1u << NPX) - 1u);
    // entries of the batch that reach this wave's strips: the walk visits only these (an entry of the tile that
    // touches only the other waves' strips costs nothing here)
    {
    // sb[k]: the batch entries that reach strip k while it is alive (one scalar bit test per entry and strip, no mask to
    // fetch from the staged record); todo: their union.  A strip that finishes takes its entries out of both.
    uint64_t sb[NPX], todo = 0ull;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      sb[k] = (alive >> k) & 1u ? __ballot(((mine >> k) & 1u) != 0u) : 0ull;
      todo |= sb[k];
    }
    if (todo != 0ull) {
    // One entry of the batch composited onto this wave's strips.  The walk below is unrolled by two with the entries in
    // TWO register sets (a, b), each loaded while the other is composited: with one set rotated every iteration
    // (`e = n; n = load`) the compiler ends every entry on seven register-to-register moves of the prefetched record.
    auto strip_done = [&](const int k) {
      alive &= ~(1u << k);
      sb[k] = 0ull;
      uint64_t rest = 0ull;
#pragma unroll
      for (int q = 0; q < NPX; ++q) rest |= sb[q];
      todo &= rest;
    };
    // The 16 object features of an entry are the same for all 64 lanes: they are fetched with SCALAR loads (one
    // s_load_dwordx16 from the feature table, issued one entry ahead like the staged record) into scalar registers and
    // enter the sixteen accumulations as scalar operands.  Rounds 1-4 staged them through LDS and read them back as four
    // 16-byte vector words per strip: 16 vector registers per register set, which is what kept the object variant on the
    // round-3 walk (the trimmed walk's second register set cost it a wave per SIMD: K6 0.30 -> 0.36 ms).
    struct Feat { float4 q[4]; };
    const uint32_t rvec = OBJ ? (WPB > 1 ? __float_as_uint(s2[jb + lane].z) : rown) : 0u;   // lane l: Gaussian of entry jb + l
    auto load_feat = [&](const int j) -> Feat {
      Feat f;
      if (OBJ && GSR_K6_OBJ_SCALAR) {
        const uint32_t og = __builtin_amdgcn_readlane(rvec, j);
        const float* row = og >= (uint32_t)a.Pa ? a.sh_objs_b + (size_t)(og - (uint32_t)a.Pa) * NUM_OBJ : a.sh_objs + (size_t)og * NUM_OBJ;
        // (the feature table is read-only for the whole launch: read through the constant address space, a uniform address
        // there is a scalar load; left in the global address space the compiler issues vector loads into 16 VGPRs)
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef const f4v __attribute__((address_space(4))) * const_f4p;
        const const_f4p src = (const_f4p)(uintptr_t)row;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f4v v = src[q]; f.q[q] = make_float4(v.x, v.y, v.z, v.w); }
      } else {
        f.q[0] = f.q[1] = f.q[2] = f.q[3] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return f;
    };
    auto composite = [&](const float4& e0, const float4& e1, const float2& e2, const Feat& fe, const int jc) {
      const uint32_t pos = base - rg.x + (uint32_t)(jb + jc) + 1;
      {
      const float dx = e0.x - pxf;
      const float qa = e0.z * dx * dx, bdx = e0.w * dx;
#pragma unroll
      for (int k = 0; k < NPX; ++k) {
        if ((sb[k] >> jc) & 1ull) {
          const float dy = e0.y - pyf[k];
          const float p2 = fmaf(dy, fmaf(e1.x, dy, bdx), qa);
          const float G = __builtin_amdgcn_exp2f(p2);
          const float alpha = fminf(ALPHA_CAP, e1.y * G);
          const float Tn = T[k] * (1.f - alpha);
          // valid = the reference's alpha test (finished pixels: p2 = -inf, alpha = 0), stop = the pixel ends in front
          // of this entry, contrib = the entry is blended.  With two or four strips per wave the lane masks are kept as
          // scalars (s_and / s_andn2 of ballots, selects through inverse_ballot, instead of a second pair of vector
          // compares): 72 -> 64 VGPRs at four strips, 0.203 -> 0.200 ms on S-nyc-1M; the one-strip kernel is 2 % faster
          // with the plain form (S-hydrant-full 0.2095 vs 0.2140 ms).
          bool stop, contrib;
          uint64_t sm;
          if (NPX >= 2) {
            const uint64_t vm = __builtin_amdgcn_ballot_w64(p2 <= 0.f) & __builtin_amdgcn_ballot_w64(alpha >= ALPHA_MIN);
            const uint64_t lt = __builtin_amdgcn_ballot_w64(Tn < T_STOP);
            sm = vm & lt;
            stop = __builtin_amdgcn_inverse_ballot_w64(sm);
            contrib = __builtin_amdgcn_inverse_ballot_w64(vm & ~lt);
          } else {
            const bool valid = (p2 <= 0.f) && (alpha >= ALPHA_MIN);
            stop = valid && (Tn < T_STOP);
            contrib = valid && !stop;
            sm = 0;
          }
          const float w = contrib ? alpha * T[k] : 0.f;
          C[k][0] = fmaf(e1.z, w, C[k][0]); C[k][1] = fmaf(e1.w, w, C[k][1]); C[k][2] = fmaf(e2.x, w, C[k][2]);
          if (OBJ && GSR_K6_OBJ_SCALAR) {
            const float f[NUM_OBJ] = {fe.q[0].x, fe.q[0].y, fe.q[0].z, fe.q[0].w, fe.q[1].x, fe.q[1].y, fe.q[1].z, fe.q[1].w,
                                      fe.q[2].x, fe.q[2].y, fe.q[2].z, fe.q[2].w, fe.q[3].x, fe.q[3].y, fe.q[3].z, fe.q[3].w};
#pragma unroll
            for (int c = 0; c < NUM_OBJ; ++c) O[k][c] = fmaf(f[c], w, O[k][c]);
          } else if (OBJ) {
            // the entry's 16 object features as four 16-byte LDS reads, right in front of their use (left to itself the
            // compiler pairs them for packed FMAs at odd offsets: one 12-byte read, six 2 x 4-byte reads and two singles)
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v* sp4 = reinterpret_cast<const f4v*>(&so[jb + jc][0]);
            f4v q0 = sp4[0], q1 = sp4[1], q2 = sp4[2], q3 = sp4[3];
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));     // the four words stay four 128-bit registers
            const float f[NUM_OBJ] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
            for (int c = 0; c < NUM_OBJ; ++c) O[k][c] = fmaf(f[c], w, O[k][c]);
          }
          T[k] = contrib ? Tn : T[k];
          last[k] = contrib ? pos : last[k];
          // A pixel ends in front of an entry only now and then: its y moves off the image inside that rare branch.
          // (A strip that has just finished takes its entries out of `todo`; with the wave's last strip `todo` is empty:
          // the walk's only exit test is "no entry left".)
          if (NPX >= 2 ? sm != 0ull : __ballot(stop) != 0ull) {
            pyf[k] = stop ? PX_OFF : pyf[k];
            if (__builtin_amdgcn_ballot_w64(pyf[k] < PX_OFF) == 0ull) strip_done(k);
          }
        }
      }
      }
    };
    // Next entry of the batch that reaches this wave (the scalar unit is shared by the CU's four SIMDs and this
    // loop leans on it: three scalar instructions instead of the eight the compiler makes of the C expressions).
    // After the last entry s_ff1 returns -1: the prefetch then reads slot 63, which is never used.
    int j = __builtin_ctzll(todo);
    float4 a0 = s0[jb + j], a1 = s1[jb + j];
    float2 a2 = make_float2(s2[jb + j].x, s2[jb + j].y);
    Feat fa = load_feat(j);
    // (the entry behind the one on which the wave's last pixel finished has been fetched already: it is skipped on its
    // empty strip mask, and the cleared `todo` ends the walk behind it)
    while (true) {
      int jc = j, jraw;
      asm volatile("s_bitset0_b64 %0, %2\n\ts_ff1_i32_b64 %1, %0" : "+s"(todo), "=s"(jraw) : "s"(jc));
      j = jraw & 63;                          // prefetch the next entry while this one is composited
      const float4 b0 = s0[jb + j], b1 = s1[jb + j];
      const float2 b2 = make_float2(s2[jb + j].x, s2[jb + j].y);
      const Feat fb = load_feat(j);
      composite(a0, a1, a2, fa, jc);
      if (jraw < 0) break;
      jc = j;
      asm volatile("s_bitset0_b64 %0, %2\n\ts_ff1_i32_b64 %1, %0" : "+s"(todo), "=s"(jraw) : "s"(jc));
      j = jraw & 63;
      a0 = s0[jb + j]; a1 = s1[jb + j]; a2 = make_float2(s2[jb + j].x, s2[jb + j].y);
      fa = load_feat(j);
      composite(b0, b1, b2, fb, jc);
      if (jraw < 0) break;
    }
    }
    }
    if (WPB == 1) __builtin_amdgcn_wave_barrier();
    if (rec0 != SEG_NONE) {
      const uint32_t pos_end = base - rg.x + (uint32_t)jb + 64u;   // list positions 1 .. pos_end are behind us
      if ((pos_end & seg_mask) == 0u && pos_end < rg.y - rg.x) {
        float4* rec = a.bnd + ((size_t)(rec0 + (pos_end >> a.seg_shift) - 1u) * PXL + sub * NPX) * 64 + lane;
#pragma unroll
        for (int k = 0; k < NPX; ++k) rec[k * 64] = make_float4(T[k], C[k][0], C[k][1], C[k][2]);
      }
    }
    }
    if (WPB > 1) {                                        // the workgroup goes on while any of its waves has pixels left
      if (lane == 0) salive[wv] = alive;
      __syncthreads();
      uint32_t any = 0;
#pragma unroll
      for (int w = 0; w < WPB; ++w) any |= salive[w];
      if (any == 0u) break;
    }
  }
  if (rec0 != SEG_NONE) {                               // final state of a split tile: record rec0 + nseg - 1
    const uint32_t nseg = (rg.y - rg.x + seg_mask) >> a.seg_shift;
    float4* rec = a.bnd + ((size_t)(rec0 + nseg - 1u) * PXL + sub * NPX) * 64 + lane;
#pragma unroll
    for (int k = 0; k < NPX; ++k) rec[k * 64] = make_float4(T[k], C[k][0], C[k][1], C[k][2]);
  }
  if (a.wave_clock && lane == 0) a.wave_clock[2 * item + 1] = wall_clock64();
  const float* bgp = a.vpack != nullptr ? a.vpack[view].bg : a.bg;
  float bg0 = bgp[0], bg1 = bgp[1], bg2 = bgp[2];
  if (a.dv != nullptr && a.dv[DV_OVF] != 0u) {
    // the pair count overflowed the capacity this forward was sized for: nothing was composited -- make that unmissable
    bg0 = bg1 = bg2 = __uint_as_float(0x7FC00000u);
#pragma unroll
    for (int k = 0; k < NPX; ++k) T[k] = 1.f;
  }
  const size_t HW = (size_t)a.H * a.W;
  float* const oc = a.out_color + 3 * HW * (size_t)view;
  float* const fT = a.final_T + HW * (size_t)view;
  uint32_t* const nc = a.n_contrib + HW * (size_t)view;
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    if (x < a.W && y[k] < a.H) {
      const size_t pix = (size_t)y[k] * a.W + x;
      oc[pix] = C[k][0] + T[k] * bg0;
      oc[HW + pix] = C[k][1] + T[k] * bg1;
      oc[2 * HW + pix] = C[k][2] + T[k] * bg2;
      fT[pix] = T[k];
      nc[pix] = last[k];
      if (OBJ) {
#pragma unroll
        for (int c = 0; c < NUM_OBJ; ++c) a.out_objects[c * HW + pix] = O[k][c];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wave64 sum via DPP: after the call lanes 48..63 hold the total
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov(float v) {
  // (quad permutations read no lane outside the row: bound_ctrl lets the compiler fold the move into the add that
  // follows instead of clearing a destination register first)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, CTRL < 0x100));
}
__device__ __forceinline__ float wave_sum_to_hi(float v) {
  v += dpp_mov<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_mov<0x140, 0xF>(v);   // row_mirror  -> every lane holds its row's sum
  v += dpp_mov<0x142, 0xA>(v);   // row_bcast15 into rows 1,3
  v += dpp_mov<0x143, 0xC>(v);   // row_bcast31 into rows 2,3
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  return v;
}

// ------------------------------------------------------------------------------------------------
// K7: backward composite, same tiling, list walked back to front.  Per (tile, entry) the wave reduces
// nine sums over its 256 pixels and stores ONE 48-byte row at the pair's emission slot: no global
// atomics, bitwise reproducible.  Row = (Sq dx, Sq dy, Sq dx^2, Sq dx dy, Sq dy^2, S G dL/dalpha,
// S w g_r, S w g_g, S w g_b, tag_lo, tag_hi, -) with q = o G dL/dalpha.
// The 64-lane sums are not butterflies (a DPP add issues at half the rate of a plain one and nine values
// need 9 x 6 of them): the per-lane values are parked in LDS as they are (ds_write does not occupy the VALU;
// only dg and db are folded into one register so that an entry is 64 chunks of 8 floats), and after every
// contributing entry the wave sums them TRANSPOSED: lane L adds the 8 floats of chunk L with plain v_add (2
// ds_read_b128), three DPP adds join the eight chunks of a value, and the lanes store their row words directly.
// ------------------------------------------------------------------------------------------------
// Layout of one parked register (round 5): 64 floats = 256 bytes, so that the eight registers of an entry sit at
// ds_write2st64 offsets 0..7 of ONE address register (the 96-float pitch of rounds 2-4 needed three more, formed per
// entry).  Chunk c (8 floats) lies at position pos(c) = 4 (c & 1) + (c >> 1): the even chunks -- written by lanes 0..31,
// element e of chunk 2q by lane 8q + e -- fill floats 0..31 and the odd ones -- lanes 32..63 -- floats 32..63, each half-wave
// pass of a ds_write_b32 touching every bank exactly once; the odd chunks store their two 16-byte halves swapped (element
// e at e ^ 4), and the transposed sum reads them in that order: the eight lanes of a ds_read_b128 pass then cover the 32
// banks exactly once as well (positions p and p + 4 start 128 bytes apart: without the swap they would share banks).
constexpr int RED_REG = 64;
struct RenderBwdArgs {
  const uint2* ranges;
  const uint32_t* pair_rank;
  const uint32_t* offg;   // [P+1] exclusive scan of tiles touched in STORAGE order: numbers the partial rows
  const float4* R0;
  const float4* R1;
  const float4* R2;
  const float* sh_objs;
  const float* bg;
  int W, H, gridx, ntiles, map_mode;
  const uint32_t* sched;      // map mode 3: tiles longest-list-first + priority class (k_tile_schedule)
  const float* final_T;
  const uint32_t* n_contrib;
  const float* grad_color;    // [3,H,W]
  const float* grad_objects;  // [16,H,W] or null
  float4* part;               // [N][3]
  float4* part_obj;           // [N][4] or null
  // split tiles (see "Segments"): the first extra_blocks workgroups are the extra work items, one per boundary record
  const float4* bnd;          // null: no tile is split
  const uint32_t* segoff;
  const uint2* rec_item;
  const uint32_t* nrec;
  uint32_t seg_shift, extra_blocks;
  uint32_t tag_lo, tag_hi;    // stamped into every row written by this call
  unsigned long long* wave_clock;   // diagnostic (gsr_debug_wave_clock): [ntiles][2] start/end of each tile's wave, 100 MHz
  int tpv;                    // a batch of views, as in RenderArgs: view = tile / tpv; grad_color holds the B images' gradients
  const ViewDev* vpack;
};

constexpr int PART_F4 = 3;

// NPX = pixels per lane as in K6: 4 -> one wave per tile, 2 -> two waves per tile (16x8 halves), each writing its
// OWN partial row (row = NSUB * slot + half; K8/K9 sums the NSUB rows of a pair, rows of halves the entry does not
// reach stay stale and are skipped by their tag).  The serial walk of the longest list is the kernel's critical
// path: halving the per-entry work of a wave shortens it, and the smaller register state lets six waves share a SIMD.
// GEOM = false: only dL/dcolour (and dL/dobject features) is wanted -- the colour attack, BASELINE configs 2 and 3.
// The walk then keeps T and the blend weight only (no running colour term, no dL/dalpha, no conic / mean / opacity
// sums): three sums per entry instead of nine, two parked registers instead of five, eight entries per transposed sum.
// Six waves per SIMD for the one-wave-per-tile kernels without object channels: registers capped at 80 (8 bytes of
// scratch per lane) and the parked-sum buffer kept to 2.5 KB, so that neither registers nor LDS (5.4 KB per wave)
// stop the sixth wave.  Measured on S-nyc-1M: 0.332 -> 0.323 ms, pipelined 1233 -> 1267 views/s; seven and eight
// waves per SIMD spill 36 / 64 bytes per lane and run 0.365 / 0.51 ms.
template <bool OBJ, int NPX, bool GEOM>
__global__ void __launch_bounds__(64, (!OBJ && NPX == 4) ? 6 : 1) k_render_bwd(RenderBwdArgs a) {
  constexpr int NSUB = PXL / NPX;
  // Registers parked per contributing entry: the per-lane sums as they are, except that the LAST one holds two values
  // (dg in lanes < 32, db in lanes >= 32, folded by one v_permlane32_swap + add): 8 registers for the nine sums,
  // 2 for the three colour sums.  (v_permlane32_swap issues at 8.5 cycles per wave on gfx950 against 2.8 for a plain
  // add -- tests/ubench/valu_rate.hip -- so only the one fold that makes the chunk count a power of two is kept;
  // folding all nine values into five registers cost five of them per entry.)
  constexpr int NREG = GEOM ? 8 : 2;
  constexpr int RB = 64 / (8 * NREG);             // entries between two transposed sums: 1, or 4 (8 * NREG chunks each)
  constexpr int RENTRY = NREG * RED_REG;
  __shared__ float4 s0[64];
  __shared__ float4 s1[64];
  __shared__ float2 s2[64];
  __shared__ uint32_t sslot[64];
  __shared__ __attribute__((aligned(16))) float so[OBJ ? 64 : 1][NUM_OBJ];
  __shared__ __attribute__((aligned(16))) float sred[RB * RENTRY];
  const int lane = threadIdx.x;
  int item;
  uint32_t seg = 0;                                   // segment of the tile's list this wave walks
  if (blockIdx.x < a.extra_blocks) {
    // extra work item of a split tile: segment j >= 1.  They come first in the grid: each is a full segment long.
    const uint32_t r = blockIdx.x / NSUB;
    if (r >= *a.nrec) return;
    const uint2 it = a.rec_item[r];
    if (it.y == 0u) return;                           // the tile's final-state record: no work item
    item = (int)it.x * NSUB + (int)(blockIdx.x % NSUB);
    seg = it.y;
    set_wave_priority(3u);
  } else {
    const int b = (int)(blockIdx.x - a.extra_blocks);
    if (a.map_mode == 3) {
      if (b >= a.ntiles * NSUB) return;
      const uint32_t sc = a.sched[b / NSUB];
      item = (int)(sc & SCHED_TILE_MASK) * NSUB + (b % NSUB);
      set_wave_priority(sc >> 28);
    } else {
      item = item_of_block(b, a.ntiles * NSUB, a.map_mode);
      if (item >= a.ntiles * NSUB) return;
    }
  }
  const int tile = item / NSUB, sub = item - tile * NSUB;
  const int view = a.tpv < a.ntiles ? __builtin_amdgcn_readfirstlane(tile / a.tpv) : 0;     // a batch of views: ViewDev
  const int ltile = tile - view * a.tpv;
  const int tx = ltile % a.gridx, ty = ltile / a.gridx;
  const uint2 rg = a.ranges[tile];
  // list positions (a_pos, b_pos] are this wave's; an unsplit tile: the whole list
  const uint32_t rec0 = (!OBJ && a.bnd != nullptr) ? a.segoff[tile] : SEG_NONE;
  uint32_t a_pos = 0, b_pos = rg.y - rg.x, nseg = 1;
  if (!OBJ && rec0 != SEG_NONE) {
    nseg = (rg.y - rg.x + (1u << a.seg_shift) - 1u) >> a.seg_shift;
    a_pos = seg << a.seg_shift;
    b_pos = min(a_pos + (1u << a.seg_shift), rg.y - rg.x);
  }
  const bool clocked = a.wave_clock != nullptr && seg == 0u;
  if (clocked && lane == 0) a.wave_clock[2 * item] = wall_clock64();
  // Where this lane parks its partials inside a 64-float register (layout: RED_REG above): lanes 0..31 fill the even chunks
  // (floats 0..31), lanes 32..63 the odd ones (floats 32..63, their 16-byte halves swapped) -- each half-wave pass of a
  // ds_write_b32 touches every bank once.
  const int red_wofs = ((lane >> 5) * 4 + ((lane & 31) >> 3)) * 8 + ((lane & 7) ^ ((lane >> 5) << 2));
  // transposed-sum roles: lane = (entry e, register rr, chunk ch of 8 floats).  The joins leave in chunks >= 4 first
  // the total of the chunks of the same parity, then the total of all eight.  Row word this lane stores: chunk 4 of
  // register rr stores value rr (the colour sums keep their words 6..8 without the geometry sums); the folded last
  // register holds dg in its even chunks (lanes < 32 of the fold) and db in its odd ones: chunk 4 stores dg, chunk 5 db;
  // chunks 6 and 7 of register 0 stamp the tag words 9 and 10
  const int red_e = lane / (8 * NREG), red_rr = (lane % (8 * NREG)) >> 3, red_ch = lane & 7;
  const bool red_last = red_rr == NREG - 1;
  const int red_word = red_last ? (red_ch == 4 ? 7 : (red_ch == 5 ? 8 : -1))
                                : (red_ch == 4 ? red_rr + (GEOM ? 0 : 6)
                                               : (red_rr == 0 && red_ch == 6 ? 9 : (red_rr == 0 && red_ch == 7 ? 10 : -1)));
  const float red_tag = __uint_as_float(red_word == 10 ? a.tag_hi : a.tag_lo);
  int red_j = 0;       // lane b: batch index j of the b-th parked entry
  int red_n = 0;       // parked entries (wave uniform)
  const int x = tx * TILE + (lane & 15);
  const float pxf = (float)(lane & 15);                    // relative to the tile's first pixel, like the staged centres
  const float* bgp = a.vpack != nullptr ? a.vpack[view].bg : a.bg;
  const float bg0 = bgp[0], bg1 = bgp[1], bg2 = bgp[2];
  const size_t HW = (size_t)a.H * a.W;
  const float* const fT = a.final_T + HW * (size_t)view;
  const uint32_t* const nc = a.n_contrib + HW * (size_t)view;
  const float* const gcol = a.grad_color + 3 * HW * (size_t)view;
  float pyf[NPX], T[NPX], Acc[NPX], g0[NPX], g1[NPX], g2[NPX];
  float gO[OBJ ? NPX : 1][NUM_OBJ];
  uint32_t ncon[NPX], smax[NPX];
  uint32_t maxc = 0;
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int y = ty * TILE + sub * (4 * NPX) + (lane >> 4) + 4 * k;
    pyf[k] = (float)(y - ty * TILE);
    Acc[k] = 0.f;
    if (x < a.W && y < a.H) {
      const size_t pix = (size_t)y * a.W + x;
      T[k] = fT[pix];
      ncon[k] = nc[pix];
      g0[k] = gcol[pix]; g1[k] = gcol[HW + pix]; g2[k] = gcol[2 * HW + pix];
      // Acc_i = sum over the entries j behind i of alpha_j (c_j.g) prod_{i<k<j} (1 - alpha_k): what the pixel shows
      // behind entry i, dotted with dL/dC.  The background is the list's last "entry" (alpha 1, colour bg): seeding
      // Acc with bg.g makes T_i (c_i.g - Acc_i) carry the -T_final/(1-alpha_i) (bg.g) term of dL/dalpha_i by itself.
      Acc[k] = bg0 * g0[k] + bg1 * g1[k] + bg2 * g2[k];
      if (OBJ) {
#pragma unroll
        for (int c = 0; c < NUM_OBJ; ++c) gO[k][c] = a.grad_objects[c * HW + pix];
      }
      if (!OBJ && rec0 != SEG_NONE) {
        if (ncon[k] > b_pos) {
          // the pixel's last contributor lies behind this segment: start from the forward's state at b_pos
          const size_t px = (size_t)(sub * NPX + k) * 64 + lane;
          const float4 B = a.bnd[(size_t)(rec0 + seg) * (PXL * 64) + px];
          if (GEOM) {
            const float4 F = a.bnd[(size_t)(rec0 + nseg - 1u) * (PXL * 64) + px];
            const float behind = fmaf(F.y - B.y, g0[k], fmaf(F.z - B.z, g1[k], (F.w - B.w) * g2[k]));
            Acc[k] = fmaf(T[k], Acc[k], behind) / B.x;          // T[k] = T_final, Acc[k] = bg . g here
          }
          T[k] = B.x;
          ncon[k] = b_pos;
        } else if (ncon[k] <= a_pos) {
          ncon[k] = 0;                                           // finished in front of this segment
        }
      }
    } else {
      T[k] = 1.f; ncon[k] = 0; g0[k] = g1[k] = g2[k] = 0.f;
      if (OBJ) {
#pragma unroll
        for (int c = 0; c < NUM_OBJ; ++c) gO[k][c] = 0.f;
      }
    }
    smax[k] = __builtin_amdgcn_readfirstlane(wave_max_u32(ncon[k]));   // last list position this strip uses
    maxc = max(maxc, smax[k]);
  }
  for (int hi = (int)maxc; hi > (int)a_pos; hi -= 64) {
    const int lo = max(hi - 64, (int)a_pos);
    const int cnt = hi - lo;
    if (lane < cnt) {
      const uint32_t pv = a.pair_rank[rg.x + lo + lane];
      const uint32_t r = pv & RANK_MASK;
      const float4 c = a.R2[REC * r];
      // strips whose last contributor lies in front of this list position are done with it: clear their bits
      // here, once per staged entry, instead of testing pos <= smax[k] per strip in the walk
      const uint32_t pos = (uint32_t)(lo + lane + 1);
      uint32_t live = 0;
#pragma unroll
      for (int k = 0; k < NPX; ++k) live |= (pos <= smax[k] ? 1u : 0u) << k;
      const StagedSplat sp = stage_splat(a.R0[REC * r], a.R1[REC * r], c, ((pv >> RANK_BITS) >> (sub * NPX)) & live, tx, ty);
      s0[lane] = sp.a; s1[lane] = sp.b; s2[lane] = sp.c;
      const uint32_t rx = __float_as_uint(c.z), ry = __float_as_uint(c.w);
      const uint32_t minx = rx & RECT_MASK, wx = ((rx >> 12) & RECT_MASK) - minx, miny = ry & RECT_MASK;
      sslot[lane] = (a.offg[r] + ((uint32_t)ty - miny) * wx + ((uint32_t)tx - minx)) * NSUB + sub;
      if (OBJ) {
        const float4* src = reinterpret_cast<const float4*>(a.sh_objs + (size_t)r * NUM_OBJ);
        float4* dst = reinterpret_cast<float4*>(&so[lane][0]);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // One list entry: its strips, the parking of its per-lane sums, and the transposed sum when the buffer is full.  The
    // walk below is unrolled by two over TWO register sets (a, b), each loaded while the other is processed: with one set
    // rotated every iteration the compiler ends every entry on five 64-bit register moves of the prefetched record.
    auto entry = [&](const float4& e0, const float4& e1, const float2& e2, const int j) {
      const uint32_t pos = (uint32_t)(lo + j + 1);
      const uint32_t m = __builtin_amdgcn_readfirstlane(__float_as_uint(e2.y)) & 0xFu;
      if (m != 0u) {
      const float dx = e0.x - pxf;
      const float qa = e0.z * dx * dx, bdx = e0.w * dx;
      float sq = 0.f, sqy = 0.f, sqyy = 0.f, dop = 0.f, dr = 0.f, dg = 0.f, db = 0.f;
      float dobj[OBJ ? NUM_OBJ : 1];
      if (OBJ) {
#pragma unroll
        for (int c = 0; c < NUM_OBJ; ++c) dobj[c] = 0.f;
      }
      bool hit = false;
      // One strip's share of this entry.  `cand`: the pixel may contribute (prefilter on p2 and on its last contributor).
      auto strip = [&](const int k, const float dy, const float p2, const bool cand) {
        const float G = __builtin_amdgcn_exp2f(p2);
        const float oG = e1.y * G;
        const float alpha = fminf(ALPHA_CAP, oG);
        const bool valid = cand && (p2 <= 0.f) && (alpha >= ALPHA_MIN);
        // An entry this pixel skips is carried through the recursions as alpha = 0, which leaves T, the
        // running colour term and every sum bit-for-bit unchanged (x*1, x+0, 0*x are exact): one select
        // here instead of one per state variable.
        const float ae = valid ? alpha : 0.f;
        const float inv1m = __builtin_amdgcn_rcpf(1.f - ae);
        T[k] *= inv1m;
        const float w = ae * T[k];
        dr = fmaf(w, g0[k], dr); dg = fmaf(w, g1[k], dg); db = fmaf(w, g2[k], db);
        if (OBJ) {
#pragma unroll
          for (int c = 0; c < NUM_OBJ; ++c) dobj[c] = fmaf(w, gO[k][c], dobj[c]);
        }
        if (GEOM) {
          float cg = fmaf(e1.z, g0[k], fmaf(e1.w, g1[k], e2.x * g2[k]));
          if (OBJ) {
#pragma unroll
            for (int c = 0; c < NUM_OBJ; ++c) cg = fmaf(so[j][c], gO[k][c], cg);
          }
          const float dcg = cg - Acc[k];
          const float dLda = valid ? T[k] * dcg : 0.f;
          Acc[k] = fmaf(ae, dcg, Acc[k]);                  // ae*cg + (1-ae)*Acc: now includes this entry
          dop = fmaf(G, dLda, dop);
          const float q = oG * dLda;
          const float qy = q * dy;
          sq += q; sqy += qy; sqyy = fmaf(qy, dy, sqyy);
        }
      };
      // A strip is entered only if the entry's mask has it, and evaluated only if some pixel of it is a candidate.
      // (Measured alternatives, both slower on S-nyc-1M: evaluating every strip of the mask without the ballot branch
      // 0.330 -> 0.347 ms, evaluating strips in straight-line pairs for two interleaved dependency chains -> 0.398 ms.
      // The kernel is bound by the NUMBER of vector instructions, not by their latency.)
#pragma unroll
      for (int k = 0; k < NPX; ++k) {
        if (m & (1u << k)) {
          const float dy = e0.y - pyf[k];
          const float p2 = fmaf(dy, fmaf(e1.x, dy, bdx), qa);
          const bool cand = (p2 >= e2.y) && (pos <= ncon[k]);
          if (__ballot(cand) != 0ull) {
            hit = true;
            strip(k, dy, p2, cand);
          }
        }
      }
      if (hit) {
        // the per-lane sums are parked in LDS (ds_write does not occupy the VALU); dg and db share one register:
        // v_permlane32_swap exchanges the upper half of one register with the lower half of another, adding the two
        // afterwards leaves dg (summed over lanes l, l+32) in lanes < 32 and db in lanes >= 32.  (The clang builtin
        // returns a broken second result in ROCm 7.2, hence inline asm; the s_nop covers the VALU-write ->
        // permlane-read wait states, which hipcc does not insert around asm.)
        // (dg and db themselves: the entry is done with them, and swapping copies costs a register move)
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(dg), "+v"(db));
        const float fa = dg, fb = db;
        float* w = &sred[red_n * RENTRY + red_wofs];
        if (GEOM) {
          const float a0 = sq * dx;
          w[0] = a0;                                     // value 0: S q dx
          w[RED_REG] = sqy;                              // value 1: S q dy
          w[2 * RED_REG] = a0 * dx;                      // value 2: S q dx^2
          w[3 * RED_REG] = sqy * dx;                     // value 3: S q dx dy
          w[4 * RED_REG] = sqyy;                         // value 4: S q dy^2
          w[5 * RED_REG] = dop;                          // value 5: S G dL/dalpha
          w[6 * RED_REG] = dr;                           // value 6
          w[7 * RED_REG] = fa + fb;                      // values 7 | 8
        } else {
          w[0] = dr;
          w[RED_REG] = fa + fb;
        }
        if (RB > 1) red_j = lane == red_n ? j : red_j;      // (one entry per sum: its index is the scalar j itself)
        ++red_n;
        if (OBJ) {
#pragma unroll
          for (int c = 0; c < NUM_OBJ; ++c) dobj[c] = wave_sum_to_hi(dobj[c]);
          if (lane == 63) {
            float4* ro = a.part_obj + (size_t)sslot[j] * 4;
            ro[0] = make_float4(dobj[0], dobj[1], dobj[2], dobj[3]);
            ro[1] = make_float4(dobj[4], dobj[5], dobj[6], dobj[7]);
            ro[2] = make_float4(dobj[8], dobj[9], dobj[10], dobj[11]);
            ro[3] = make_float4(dobj[12], dobj[13], dobj[14], dobj[15]);
          }
        }
      }
      }
      if (red_n == RB || (j == 0 && red_n > 0)) {     // sslot[] is re-staged after j == 0: drain before that
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int jm = RB > 1 ? __builtin_amdgcn_ds_bpermute(red_e << 2, red_j) : j;
        if (red_e < red_n) {
          // chunk red_ch of register red_rr of parked entry red_e; the odd chunks' halves are stored swapped
          const float* chp = &sred[red_e * RENTRY + red_rr * RED_REG + ((red_ch & 1) * 4 + (red_ch >> 1)) * 8];
          const float4 q0 = *reinterpret_cast<const float4*>(chp + ((red_ch & 1) << 2));
          const float4 q1 = *reinterpret_cast<const float4*>(chp + (((red_ch & 1) ^ 1) << 2));
          // three packed adds + one (v_pk_add_f32) instead of seven scalar ones
          typedef float f2v __attribute__((ext_vector_type(2)));
          const f2v h0 = f2v{q0.x, q0.y} + f2v{q0.z, q0.w}, h1 = f2v{q1.x, q1.y} + f2v{q1.z, q1.w};
          const f2v hs = h0 + h1;
          float t = hs.x + hs.y;
          t += dpp_mov<0x4E, 0xF>(t);                    // quad_perm [2,3,0,1]: chunks c, c ^ 2
          t += dpp_mov<0x114, 0xF>(t);                   // row_shr:4: chunks >= 4 now hold their parity's total
          const float t4 = t + dpp_mov<0xB1, 0xF>(t);    // quad_perm [1,0,3,2]: both parities, an unfolded register's value
          if (red_word >= 0) {
            float* row = reinterpret_cast<float*>(a.part + (size_t)sslot[jm] * PART_F4);
            // words 9 and 10 carry this backward call's 64-bit tag: rows that no wave writes keep whatever the
            // workspace held and are recognised as stale by K8/K9, so the partial-row buffer is never cleared
            row[red_word] = red_word >= 9 ? red_tag : (red_last ? t : t4);
          }
        }
        red_n = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    };
    float4 a0 = s0[cnt - 1], a1 = s1[cnt - 1];
    float2 a2 = s2[cnt - 1];
    for (int j = cnt - 1; j >= 0; j -= 2) {
      const int jb = max(j - 1, 0);           // prefetch the next entry while this one is processed
      const float4 b0 = s0[jb], b1 = s1[jb];
      const float2 b2 = s2[jb];
      entry(a0, a1, a2, j);
      if (j == 0) break;
      const int ja = max(j - 2, 0);
      a0 = s0[ja]; a1 = s1[ja]; a2 = s2[ja];
      entry(b0, b1, b2, j - 1);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (clocked && lane == 0) a.wave_clock[2 * item + 1] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// K8+K9 fused with the per-Gaussian reduction of the partial rows: one thread per Gaussian, in STORAGE
// order, so every attribute read and every gradient write of a wave is one contiguous span; the rows of a
// Gaussian's (tile, Gaussian) pairs are contiguous too (slots are numbered in storage order).
// ------------------------------------------------------------------------------------------------
struct PreBwdArgs {
  int P, K;               // P: END of the range of Gaussians this launch covers (= the scene's count for a whole launch)
  int g0;                 // first Gaussian of the range (a multiple of 64); block b owns Gaussians g0 + 64 b ...
  ViewArgs va;
  const uint32_t* offg;   // [P+1] exclusive scan of tiles touched, storage order
  const float4* G0;
  const float4* G1;
  const float4* G2;
  const float4* part;
  const float4* part_obj;
  uint32_t tag_lo, tag_hi;   // rows whose words 9,10 differ are stale (not written by this backward)
  uint32_t nsub;             // partial rows per (tile, Gaussian) pair: one per K7 wave of the tile
  const float* means;
  const float* scales;
  const float* rots;
  const float* cov3d;
  const float* sh;        // RAW: _features_rest
  const float* sh_dc;     // RAW: _features_dc
  const float* D;         // [P,9] d rgb / d view direction left by k_pre_color (lane-group kernels only)
  const double* abc;      // [P] k_pre_geom's needle marks under GSR_FLAG_NEEDLE_DOUBLE (a number = needle, NaN = not)
  float* dmeans3D;
  float* dmeans2D;
  float* dsh;             // RAW: gradient of _features_rest
  float* dsh_dc;          // RAW: gradient of _features_dc
  float* dsh_objs;
  float* dcolors;
  float* dopac;
  float* dscales;
  float* drots;
  float* dcov3d;
  int accumulate;         // != 0 (k_pre_bwd only): the 59 attribute gradients are ADDED to (Gaussians without pairs are
                          // left alone); dmeans2D and dsh_objs, which belong to one view, are overwritten regardless
  int needle_double;      // GSR_FLAG_NEEDLE_DOUBLE was set in the forward (generic k_preprocess_bwd recomputes the double chain)
  float* sumsq;           // k_pre_bwd<RAW = true, ., ACC = false> only, or null: [workgroups][SUMSQ_W] per-workgroup sums of
                          // squares of the gradients this launch writes, per attribute tensor (SUMSQ_* below)
  // k_pre_bwd over ALL views of a batch context in one launch (per-view gradients, gsr_backward_raw_batch_views): grid.y =
  // view; the per-view arrays of the virtual scene (offg, records, D) lie view * Ppad further, dmeans2D view * Pscene * 3,
  // the attribute gradients view * vstride floats; the view's camera comes from vpack[view].  vpack == null: one view, `va`.
  const ViewDev* vpack;
  int Ppad, Pscene;
  long long vstride;
};
// Slots of PreBwdArgs::sumsq / gsr_ctx_request_sumsq: the six tensors the reference's L2 steps normalise over (attack.py:
// 53-119, 138-173: a global norm per tensor, _features_dc and _features_rest separately).
enum { SUMSQ_XYZ = 0, SUMSQ_DC = 1, SUMSQ_REST = 2, SUMSQ_OPACITY = 3, SUMSQ_SCALING = 4, SUMSQ_ROTATION = 5, SUMSQ_W = 8 };

// The gradient's sum of squares per tensor from the per-workgroup partials k_pre_bwd leaves (fixed summation order:
// reproducible): block k sums slot k over all workgroups in double.
__global__ void __launch_bounds__(256) k_sumsq_reduce(const float* __restrict__ part, int nblocks, double* __restrict__ out) {
  __shared__ double wsum[4];
  const int k = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += (double)part[(size_t)i * SUMSQ_W + k];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[k] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// Generic form (any K): one thread per Gaussian walks its own partial rows and its own SH row.
// GEOM = false: the rows carry the three colour sums only (K7 without the geometry sums) and only the SH / colour /
// object-feature gradients are produced.  NDL: GSR_FLAG_NEEDLE_DOUBLE was set in the forward (needles run needle_bwd_d).
template <bool GEOM, bool NDL = false>
__global__ void __launch_bounds__(PRE_BLOCK) k_preprocess_bwd(PreBwdArgs a) {
  const int g = a.g0 + blockIdx.x * PRE_BLOCK + threadIdx.x;
  const int K = a.K;
  if (g >= a.P) return;
  const uint32_t o0 = a.offg[g] * a.nsub, o1 = a.offg[g + 1] * a.nsub;
  if (o1 == o0) {   // no pairs: zero gradients
    if (a.dmeans3D) { a.dmeans3D[3 * g] = 0.f; a.dmeans3D[3 * g + 1] = 0.f; a.dmeans3D[3 * g + 2] = 0.f; }
    if (a.dmeans2D) { a.dmeans2D[3 * g] = 0.f; a.dmeans2D[3 * g + 1] = 0.f; a.dmeans2D[3 * g + 2] = 0.f; }
    if (a.dsh) for (int i = 0; i < 3 * K; ++i) a.dsh[(size_t)g * K * 3 + i] = 0.f;
    if (a.dsh_objs) for (int i = 0; i < NUM_OBJ; ++i) a.dsh_objs[(size_t)g * NUM_OBJ + i] = 0.f;
    if (a.dcolors) { a.dcolors[3 * g] = 0.f; a.dcolors[3 * g + 1] = 0.f; a.dcolors[3 * g + 2] = 0.f; }
    if (a.dopac) a.dopac[g] = 0.f;
    if (a.dscales) { a.dscales[3 * g] = 0.f; a.dscales[3 * g + 1] = 0.f; a.dscales[3 * g + 2] = 0.f; }
    if (a.drots) { a.drots[4 * g] = 0.f; a.drots[4 * g + 1] = 0.f; a.drots[4 * g + 2] = 0.f; a.drots[4 * g + 3] = 0.f; }
    if (a.dcov3d) for (int i = 0; i < 6; ++i) a.dcov3d[6 * g + i] = 0.f;
    return;
  }
  // (this kernel always overwrites its outputs: accumulation into a caller's bucket exists for the raw-parameter path
  // only, which runs k_pre_bwd -- gsr_backward_raw_into)
  auto put = [](float* p, float v) { *p = v; };
  float mx = 0.f, my = 0.f, mxx = 0.f, mxy = 0.f, myy = 0.f, dop = 0.f, dr = 0.f, dg = 0.f, db = 0.f;
  for (uint32_t e = o0; e < o1; ++e) {
    const float4 p0 = a.part[(size_t)e * PART_F4], p1 = a.part[(size_t)e * PART_F4 + 1], p2 = a.part[(size_t)e * PART_F4 + 2];
    if (__float_as_uint(p2.y) != a.tag_lo || __float_as_uint(p2.z) != a.tag_hi) continue;
    if (GEOM) { mx += p0.x; my += p0.y; mxx += p0.z; mxy += p0.w; myy += p1.x; dop += p1.y; }
    dr += p1.z; dg += p1.w; db += p2.x;
  }
  if (a.dsh_objs) {
    float acc_o[NUM_OBJ];
#pragma unroll
    for (int c = 0; c < NUM_OBJ; ++c) acc_o[c] = 0.f;
    if (a.part_obj) {
      for (uint32_t e = o0; e < o1; ++e) {
        const float4 tg = a.part[(size_t)e * PART_F4 + 2];
        if (__float_as_uint(tg.y) != a.tag_lo || __float_as_uint(tg.z) != a.tag_hi) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = a.part_obj[(size_t)e * 4 + q];
          acc_o[4 * q] += v.x; acc_o[4 * q + 1] += v.y; acc_o[4 * q + 2] += v.z; acc_o[4 * q + 3] += v.w;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NUM_OBJ; ++c) put(&a.dsh_objs[(size_t)g * NUM_OBJ + c], acc_o[c]);
  }
  View v;
  load_view(v, a.va);
  const float4 e0 = a.G0[REC * g], e1 = a.G1[REC * g], e2 = a.G2[REC * g];
  const float A = e0.z, B = e0.w, C = e1.x;
  // dL/d(pixel centre) = -(A mx + B my, B mx + C my); screen-space means are reported in NDC units
  const float dndcx = -(A * mx + B * my) * 0.5f * (float)v.W;
  const float dndcy = -(B * mx + C * my) * 0.5f * (float)v.H;
  const float dA = -0.5f * mxx, dB = -mxy, dC = -0.5f * myy;
  if (GEOM && a.dmeans2D) { put(&a.dmeans2D[3 * g], dndcx); put(&a.dmeans2D[3 * g + 1], dndcy); put(&a.dmeans2D[3 * g + 2], 0.f); }
  if (GEOM && a.dopac) put(&a.dopac[g], dop);
  const float p[3] = {a.means[3 * g], a.means[3 * g + 1], a.means[3 * g + 2]};
  float dp[3] = {0.f, 0.f, 0.f};
  if (a.dcolors) { put(&a.dcolors[3 * g], dr); put(&a.dcolors[3 * g + 1], dg); put(&a.dcolors[3 * g + 2], db); }
  if (a.sh) {
    const uint32_t cl = (__float_as_uint(e1.z) >> 31) | ((__float_as_uint(e1.w) >> 31) << 1) | ((__float_as_uint(e2.x) >> 31) << 2);
    const float drgb[3] = {(cl & 1u) ? 0.f : dr, (cl & 2u) ? 0.f : dg, (cl & 4u) ? 0.f : db};
    if (a.dsh) {
      sh_to_rgb_bwd(v.sh_degree, K, a.sh + (size_t)g * K * 3, p, v.cam, drgb, a.dsh + (size_t)g * K * 3, dp);
    } else {      // dL/dSH not wanted: only the view-direction term of dL/dmean (the basis has at most 16 functions)
      float scratch[48];
      sh_to_rgb_bwd(v.sh_degree, 16, a.sh + (size_t)g * K * 3, p, v.cam, drgb, scratch, dp);
    }
  }
  if (GEOM) {
    float c6[6];
    float sc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.cov3d) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c6[i] = a.cov3d[6 * g + i];
    } else {
      sc[0] = a.scales[3 * g]; sc[1] = a.scales[3 * g + 1]; sc[2] = a.scales[3 * g + 2];
      const float4 q4 = reinterpret_cast<const float4*>(a.rots)[g];
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
      cov3d_from_scale_rot(sc, a.va.mod, q, c6);
    }
    float dc6[6];
    float ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    // GSR_FLAG_NEEDLE_DOUBLE: a needle (the forward's test, on the float32 chain's 2D covariance) runs its chain rule in double
    bool needle = false;
    if (NDL) {
      float tt[3];
      view_transform(v, p, tt);
      ProjLin pl;
      proj_linear(v, tt, pl);
      float fa, fb, fc;
      cov2d_from_M(pl.M, c6, fa, fb, fc);
      needle = is_needle(fa, fb, fc);
    }
    if (NDL && needle) {
      double dpd[3] = {(double)dp[0], (double)dp[1], (double)dp[2]}, dsd[3], dqd[4], dS6[6];
      needle_bwd_d(v, p, sc, a.va.mod, q, a.cov3d ? c6 : nullptr, (double)dA, (double)dB, (double)dC, (double)dndcx, (double)dndcy,
                   dpd, dsd, dqd, dS6);
      for (int i = 0; i < 3; ++i) { dp[i] = (float)dpd[i]; ds[i] = (float)dsd[i]; }
      for (int i = 0; i < 4; ++i) dq[i] = (float)dqd[i];
      for (int i = 0; i < 6; ++i) dc6[i] = (float)dS6[i];
    } else {
      project_splat_bwd(v, p, c6, dA, dB, dC, dndcx, dndcy, dp, dc6);
      if (!a.cov3d && (a.dscales || a.drots)) cov3d_bwd(sc, a.va.mod, q, dc6, ds, dq);
    }
    if (a.dmeans3D) { put(&a.dmeans3D[3 * g], dp[0]); put(&a.dmeans3D[3 * g + 1], dp[1]); put(&a.dmeans3D[3 * g + 2], dp[2]); }
    if (a.cov3d) {
      if (a.dcov3d) for (int i = 0; i < 6; ++i) put(&a.dcov3d[6 * g + i], dc6[i]);
    } else if (a.dscales || a.drots) {
      if (a.dscales) { put(&a.dscales[3 * g], ds[0]); put(&a.dscales[3 * g + 1], ds[1]); put(&a.dscales[3 * g + 2], ds[2]); }
      if (a.drots) { put(&a.drots[4 * g], dq[0]); put(&a.drots[4 * g + 1], dq[1]); put(&a.drots[4 * g + 2], dq[2]); put(&a.drots[4 * g + 3], dq[3]); }
    }
  }
}


// ================================================================================================
// K1 and K8+K9 in LANE-GROUP form (the SH layouts the reference uses: K = 16 coefficients, or the raw dc | rest pair).
//
// The per-Gaussian geometry is one thread per Gaussian, but everything that touches a 192-byte SH row is done by a
// GROUP OF FOUR LANES per Gaussian: lane q of the group owns coefficients 4q .. 4q+3 (12 consecutive floats), so a wave
// moves SH data with 16-byte loads / stores of consecutive addresses (16 Gaussians x 4 lanes per instruction) and the
// rows never pass through LDS.  What an owner lane hands to its group is a 13-word slot, so a wave needs 3.3 KB of LDS
// instead of 13 KB and the kernels are no longer limited to three waves per SIMD; K1 reads the SH rows of the
// SURVIVORS only (compacted by a ballot), and leaves d(rgb)/d(view direction) (9 floats) behind so that the backward
// gets the view-direction term of dL/dmean without reading the coefficients again.
// ================================================================================================
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };   // 16-byte access at 4-byte alignment
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };

constexpr int SLOT_W = 13;     // words per hand-over slot (odd: 16 groups reading 16 slots hit 16 different banks)

__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  return v;
}

// the 12 floats of lane q: coefficients 4q .. 4q+3, three channels each
template <bool RAW>
__device__ __forceinline__ void load_sh12(const float* __restrict__ sh, const float* __restrict__ sh_dc, uint32_t g, int q,
                                          float v[12]) {
  if (RAW) {
    const float* rest = sh + (size_t)g * 45 + 12 * q;                       // coefficients 4q+1 .. : 9 floats
    const float* head = q == 0 ? sh_dc + (size_t)g * 3 : rest - 3;          // coefficient 4q
    const F3u h = *reinterpret_cast<const F3u*>(head);
    const F4u r0 = *reinterpret_cast<const F4u*>(rest);
    const F4u r1 = *reinterpret_cast<const F4u*>(rest + 4);
    const float r2 = rest[8];
    v[0] = h.x; v[1] = h.y; v[2] = h.z;
    v[3] = r0.x; v[4] = r0.y; v[5] = r0.z; v[6] = r0.w; v[7] = r1.x; v[8] = r1.y; v[9] = r1.z; v[10] = r1.w; v[11] = r2;
  } else {
    const float4* src = reinterpret_cast<const float4*>(sh + (size_t)g * 48 + 12 * q);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2];
    v[0] = r0.x; v[1] = r0.y; v[2] = r0.z; v[3] = r0.w; v[4] = r1.x; v[5] = r1.y; v[6] = r1.z; v[7] = r1.w;
    v[8] = r2.x; v[9] = r2.y; v[10] = r2.z; v[11] = r2.w;
  }
}

template <bool RAW>
__device__ __forceinline__ void store_sh12(float* __restrict__ dsh, float* __restrict__ dsh_dc, uint32_t g, int q,
                                           const float v[12]) {
  if (RAW) {
    float* rest = dsh + (size_t)g * 45 + 12 * q;
    float* head = q == 0 ? dsh_dc + (size_t)g * 3 : rest - 3;
    *reinterpret_cast<F3u*>(head) = F3u{v[0], v[1], v[2]};
    *reinterpret_cast<F4u*>(rest) = F4u{v[3], v[4], v[5], v[6]};
    *reinterpret_cast<F4u*>(rest + 4) = F4u{v[7], v[8], v[9], v[10]};
    rest[8] = v[11];
  } else {
    float4* dst = reinterpret_cast<float4*>(dsh + (size_t)g * 48 + 12 * q);
    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
    dst[2] = make_float4(v[8], v[9], v[10], v[11]);
  }
}

__device__ __forceinline__ float pick4(int q, float a, float b, float c, float d) {
  return q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d));
}

struct PreArgs {
  int P;
  ViewArgs va;
  const float* means;
  const float* scales;
  const float* rots;
  const float* cov3d;
  const float* opac;
  const float* sh;        // [P,16,3], or RAW: _features_rest [P,15,3]
  const float* sh_dc;     // RAW: _features_dc [P,1,3]
  const float* colors;    // precomputed colours [P,3] (then no SH)
  // second attribute segment (gsr_forward_raw2: attacked target + frozen background rendered as ONE scene without
  // concatenating their tensors): Gaussians g >= Pa read element g - Pa of these
  int Pa;                 // == P when there is one segment
  const float* means_b;
  const float* scales_b;
  const float* rots_b;
  const float* opac_b;
  const float* sh_b;
  const float* sh_dc_b;
  int Pfill;              // >= P (a view of a batch: the view's padded range, see ViewDev): Gaussians [P, Pfill) emit nothing
  const ViewDev* vpack;   // k_pre_geom over a batch of views in one launch (else null): see there
  int bpv;                // ... its workgroups per view (Pfill / PREG_BLOCK)
  int32_t* radii;
  float4* G0;
  float4* G1;
  float4* G2;
  float* D;               // [P,9] d rgb_c / d dir_axis (before the clamp), or null: not wanted (no backward follows)
  double* abc;            // [P] k_pre_geom<., NDL>: needle marks for K9 (the needle's 2D covariance entry a, or NaN: not a needle), or null
  uint32_t* dkey;         // float bits of the view depth; 0xFFFFFFFF for a Gaussian that emits no pair
  uint32_t* tcnt;         // tiles of the (tightened) rect
  uint8_t* tcnt8;         // k_pre_geom: the same saturated at 255 (the depth sort's last pass gathers tiles-per-rank from it), or null
  const uint32_t* offg;   // colour kernel with tcnt == null (re-render of a kept context): a Gaussian emits pairs iff
                          // offg[g + 1] != offg[g] (the storage-order scan of tcnt, which the context keeps)
  int cull;               // != 0: shrink the rect to the alpha >= 1/255 footprint (tighten_rect)
  PreBlockOut bo;
};

// A colour c >= 0 whose pre-clamp value was negative is stored as -0.0f: the compositors see 0 either way (x + -0 = x,
// w * -0 adds nothing), and the backward reads the clamp flag off the sign bit instead of from a separate word -- so
// the colour words of a record belong to the colour kernel alone and the geometry kernel never touches them.
__device__ __forceinline__ float clamp_flagged(float c) { return c > 0.f ? c : (c < 0.f ? -0.0f : 0.0f); }
__device__ __forceinline__ uint32_t clamp_bits_of(float r, float g, float b) {
  return (__float_as_uint(r) >> 31) | ((__float_as_uint(g) >> 31) << 1) | ((__float_as_uint(b) >> 31) << 2);
}

// K1, geometry half: one thread per Gaussian -- projection, culls, tile rect, depth key, tiles touched, the geometric
// words of the 48-byte record ((px,py,A,B) (C,opacity,.,.) (.,depth,rect x,rect y)) -- and the workgroup sums the storage
// scan starts from.  Reads 44 bytes per Gaussian.  With precomputed colours it writes those too (nothing else to do).
template <bool RAW, bool NDL = false>
__global__ void __launch_bounds__(PREG_BLOCK) k_pre_geom(PreArgs a) {
  // A batch of views in ONE launch (a.vpack != null; ViewDev): view = block / bpv, the block's Gaussians are those of that
  // view's padded range; the per-view arrays (records, keys, counts, needle marks) are indexed at vo + g, radii at view * P + g.
  const int view = a.vpack != nullptr ? (int)(blockIdx.x / (uint32_t)a.bpv) : 0;
  const int g = (int)(blockIdx.x - (uint32_t)(view * a.bpv)) * PREG_BLOCK + threadIdx.x;
  const size_t vo = (size_t)view * (size_t)a.bpv * PREG_BLOCK;
  uint32_t cnt = 0, key = 0xFFFFFFFFu;
  if (g < a.P) {
    View v;
    if (a.vpack != nullptr) {
      const ViewDev& vd = a.vpack[view];
      make_view(v, vd.vm, vd.pm, vd.cam, a.va.H, a.va.W, vd.tanfovx, vd.tanfovy, a.va.mod, a.va.deg);
    } else {
      load_view(v, a.va);
    }
    const bool second = g >= a.Pa;                       // which attribute segment this Gaussian lives in
    const int gl = second ? g - a.Pa : g;
    const float* means = second ? a.means_b : a.means;
    const float p[3] = {means[3 * gl], means[3 * gl + 1], means[3 * gl + 2]};
    float c6[6];
    float sc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.cov3d) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c6[i] = a.cov3d[6 * g + i];
    } else {
      const float* scales = second ? a.scales_b : a.scales;
      sc[0] = scales[3 * gl]; sc[1] = scales[3 * gl + 1]; sc[2] = scales[3 * gl + 2];
      const float4 q4 = reinterpret_cast<const float4*>(second ? a.rots_b : a.rots)[gl];
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
      if (RAW) {
        sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]);
        float inv_n;
        act_normalize4(q, q, inv_n);
      }
      cov3d_from_scale_rot(sc, a.va.mod, q, c6);
    }
    Splat s;
    const float oraw = second ? a.opac_b[gl] : a.opac[gl];
    const float op = RAW ? act_sigmoid(oraw) : oraw;
    const bool vis = project_splat(v, p, c6, s) && opacity_ok(op);
    a.radii[(size_t)view * (size_t)a.P + g] = vis ? s.radius : 0;                     // the reference's radius, whatever the footprint test says
    if (vis) {
      // NDL (GSR_FLAG_NEEDLE_DOUBLE): a needle's conic -- what the compositors and the footprint tests use -- comes from the
      // same chain in double, on the same float32 (activated) inputs (gsr_math.h is_needle / cov2d_accurate); every other
      // splat, and every splat without the flag, keeps the published float32 conic.  Radius, rect and culls above are the
      // float32 chain's either way.  (A template parameter: the double chain costs the kernel 20 registers.)
      double ca = (double)s.ca, cb = (double)s.cb, cc = (double)s.cc;
      const bool ndl = NDL && is_needle(s.ca, s.cb, s.cc);
      if (ndl) {
        cov2d_accurate(v, p, sc, a.va.mod, q, a.cov3d ? c6 : nullptr, ca, cb, cc);
        needle_conic_to_float(ca, cb, cc, s.A, s.B, s.C);
      }
      if (NDL && a.abc) a.abc[vo + (size_t)g] = ndl ? ca : __longlong_as_double(0x7FF8000000000000ll);   // K9's needle mark
      const int fminx = s.rminx, fminy = s.rminy;      // first tile of the reference's rect: the origin of the stored centre
      if (a.cull) tighten_rect(s.px, s.py, s.A, s.B, s.C, op, v.gridx, v.gridy, s.rminx, s.rminy, s.rmaxx, s.rmaxy);
      s.rminx = min(s.rminx, fminx + RECT_OFF_MAX); s.rminy = min(s.rminy, fminy + RECT_OFF_MAX);
      cnt = (uint32_t)((s.rmaxx - s.rminx) * (s.rmaxy - s.rminy));
      if (cnt != 0u) {
        key = __float_as_uint(s.depth);
        const uint32_t rx = (uint32_t)s.rminx | ((uint32_t)s.rmaxx << 12) | ((uint32_t)(s.rminx - fminx) << 24);
        const uint32_t ry = (uint32_t)s.rminy | ((uint32_t)s.rmaxy << 12) | ((uint32_t)(s.rminy - fminy) << 24);
        a.G0[REC * (vo + g)] = make_float4((float)(s.pxd - (double)(fminx * TILE)), (float)(s.pyd - (double)(fminy * TILE)), s.A, s.B);
        float* g1 = reinterpret_cast<float*>(&a.G1[REC * (vo + g)]);
        float* g2 = reinterpret_cast<float*>(&a.G2[REC * (vo + g)]);
        if (a.colors) {
          a.G1[REC * (vo + g)] = make_float4(s.C, op, a.colors[3 * g], a.colors[3 * g + 1]);
          a.G2[REC * (vo + g)] = make_float4(a.colors[3 * g + 2], s.depth, __uint_as_float(rx), __uint_as_float(ry));
        } else {
          *reinterpret_cast<float2*>(g1) = make_float2(s.C, op);
          g2[1] = s.depth;
          *reinterpret_cast<float2*>(g2 + 2) = make_float2(__uint_as_float(rx), __uint_as_float(ry));
        }
      }
    }
    a.dkey[vo + g] = key;
    a.tcnt[vo + g] = cnt;
    if (a.tcnt8) a.tcnt8[vo + g] = (uint8_t)min(cnt, 255u);
  } else if (g < a.Pfill) {
    a.dkey[vo + g] = 0xFFFFFFFFu;
    a.tcnt[vo + g] = 0u;
    if (a.tcnt8) a.tcnt8[vo + g] = 0;
  }
  PreBlockOut bo = a.bo;
  if (bo.ranges != nullptr) bo.ranges += (size_t)view * (size_t)bo.ntiles;     // this view's tiles
  pre_block_epilogue(bo, g, cnt, key);
}

// K1, colour half: SH -> RGB for the Gaussians that emit pairs, by FOUR LANES per Gaussian (see above), + d colour /
// d view direction for the backward.  Reads the 192-byte SH rows of the survivors only, writes the three colour words of
// their records.  It depends on the geometry half only through tcnt[] and is needed by nobody before the forward
// compositor: the library runs it on a side stream, beside the binning chain (which leaves most of the chip idle).
template <bool RAW>
__global__ void __launch_bounds__(PREF_BLOCK) k_pre_color(PreArgs a) {
  __shared__ float slots[PREF_WAVES * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ws = &slots[wave * 64 * 4];
  // grid-stride over the 256-Gaussian chunks: on the side stream the kernel is launched with FEWER workgroups than
  // chunks, so that it holds only a couple of waves per SIMD while the binning chain's short kernels come and go
  for (int chunk = blockIdx.x; chunk * PREF_BLOCK < a.P; chunk += gridDim.x) {
  const int g = chunk * PREF_BLOCK + wave * 64 + lane;
  const bool ok = g < a.P && (a.tcnt ? a.tcnt[g] != 0u : a.offg[g + 1] != a.offg[g]);
  const uint64_t live = __ballot(ok);
  const int nlive = __popcll(live);
  if (nlive == 0) continue;
  if (ok) {
    const bool second = g >= a.Pa;
    const int gl = second ? g - a.Pa : g;
    const float* means = second ? a.means_b : a.means;
    const float* cam = a.va.cam;
    const float dx = means[3 * gl] - cam[0], dy = means[3 * gl + 1] - cam[1], dz = means[3 * gl + 2] - cam[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float* sl = ws + 4 * __popcll(live & ((1ull << lane) - 1ull));
    sl[0] = dx * inv; sl[1] = dy * inv; sl[2] = dz * inv; sl[3] = __uint_as_float((uint32_t)g);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int q = lane & 3, grp = lane >> 2;
  const int deg = a.va.deg;
  for (int r0 = 0; r0 < nlive; r0 += 16) {
    const int si = r0 + grp;
    if (si < nlive) {
      const float4 slv = *reinterpret_cast<const float4*>(ws + 4 * si);
      const float x = slv.x, y = slv.y, z = slv.z;
      const uint32_t gg = __float_as_uint(slv.w);
      float sv[12];
      if (gg >= (uint32_t)a.Pa) load_sh12<RAW>(a.sh_b, a.sh_dc_b, gg - (uint32_t)a.Pa, q, sv);
      else load_sh12<RAW>(a.sh, a.sh_dc, gg, q, sv);
      float b[16], gx[16], gy[16], gz[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) b[k] = 0.f;
      sh_basis(deg, x, y, z, b);
      float bs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bs[j] = pick4(q, b[j], b[4 + j], b[8 + j], b[12 + j]);
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { c0 = fmaf(bs[j], sv[3 * j], c0); c1 = fmaf(bs[j], sv[3 * j + 1], c1); c2 = fmaf(bs[j], sv[3 * j + 2], c2); }
      c0 = quad_sum(c0) + 0.5f; c1 = quad_sum(c1) + 0.5f; c2 = quad_sum(c2) + 0.5f;
      if (a.D) {
        sh_basis_grad(deg, x, y, z, gx, gy, gz);
        float d[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // d[3c + axis]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ex = pick4(q, gx[j], gx[4 + j], gx[8 + j], gx[12 + j]);
          const float ey = pick4(q, gy[j], gy[4 + j], gy[8 + j], gy[12 + j]);
          const float ez = pick4(q, gz[j], gz[4 + j], gz[8 + j], gz[12 + j]);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            d[3 * c] = fmaf(ex, sv[3 * j + c], d[3 * c]);
            d[3 * c + 1] = fmaf(ey, sv[3 * j + c], d[3 * c + 1]);
            d[3 * c + 2] = fmaf(ez, sv[3 * j + c], d[3 * c + 2]);
          }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) d[i] = quad_sum(d[i]);
        if (q < 3) {
          const float dx_ = pick4(q, d[0], d[3], d[6], 0.f), dy_ = pick4(q, d[1], d[4], d[7], 0.f),
                      dz_ = pick4(q, d[2], d[5], d[8], 0.f);
          *reinterpret_cast<F3u*>(a.D + (size_t)gg * 9 + 3 * q) = F3u{dx_, dy_, dz_};
        }
      }
      // the colour words of the record: (., ., r, g) of its second float4 and (b, ., ., .) of its third
      if (q == 0) *reinterpret_cast<float2*>(reinterpret_cast<float*>(&a.G1[REC * gg]) + 2) = make_float2(clamp_flagged(c0), clamp_flagged(c1));
      else if (q == 1) *reinterpret_cast<float*>(&a.G2[REC * gg]) = clamp_flagged(c2);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the slots are rewritten by the next chunk
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}


// K1, colour half, for a BATCH of views (ViewDev; raw parameters): the 192-byte SH row of a Gaussian that emits pairs in ANY
// view is read ONCE, by its four lanes, and evaluated for every view that sees it -- the view's camera position is a scalar
// load, the colour words go to the record of (view, Gaussian), the nine d colour / d direction values next to them.  B
// single-view launches read the rows of the 66 % visible Gaussians B times: 1.0 GB per 8-view batch of S-nyc-1M, 0.17 GB here.
struct PreColorBatchArgs {
  int P, B, Ppad, deg;
  const ViewDev* vpack;
  const float* means;
  const float* sh;        // _features_rest [P,15,3]
  const float* sh_dc;     // _features_dc [P,1,3]
  int Pa;                 // Gaussians g >= Pa live in the second attribute segment (gsr_forward_raw2_batch); == P: one segment
  const float* means_b;
  const float* sh_b;
  const float* sh_dc_b;
  const uint32_t* tcnt;   // [B * Ppad] tiles touched by (view, Gaussian)
  const uint32_t* offg;   // tcnt == null (re-render of a kept batch context): (view, Gaussian) emits pairs iff
                          // offg[i + 1] != offg[i], i = view * Ppad + g (the storage-order scan the context keeps)
  float4* G1;             // records of the virtual scene (G0 + 1, G0 + 2)
  float4* G2;
  float* D;               // [B * Ppad, 9] or null
};
__global__ void __launch_bounds__(PREF_BLOCK) k_pre_color_batch(PreColorBatchArgs a) {
  __shared__ float slots[PREF_WAVES * 64 * 4];
  __shared__ uint32_t smask[PREF_WAVES * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ws = &slots[wave * 64 * 4];
  uint32_t* wm = &smask[wave * 64];
  for (int chunk = blockIdx.x; chunk * PREF_BLOCK < a.P; chunk += gridDim.x) {
    const int g = chunk * PREF_BLOCK + wave * 64 + lane;
    uint32_t seen = 0;                                     // bit v: view v has pairs of this Gaussian
    if (g < a.P) {
      if (a.tcnt != nullptr) {
#pragma unroll 4
        for (int v = 0; v < a.B; ++v) seen |= (a.tcnt[(size_t)v * (size_t)a.Ppad + g] != 0u ? 1u : 0u) << v;
      } else {
#pragma unroll 4
        for (int v = 0; v < a.B; ++v) {
          const size_t i = (size_t)v * (size_t)a.Ppad + g;
          seen |= (a.offg[i + 1] != a.offg[i] ? 1u : 0u) << v;
        }
      }
    }
    const bool ok = seen != 0u;
    const uint64_t live = __ballot(ok);
    const int nlive = __popcll(live);
    if (nlive == 0) continue;
    if (ok) {
      const int si = __popcll(live & ((1ull << lane) - 1ull));
      float* sl = ws + 4 * si;
      const bool second = g >= a.Pa;
      const int gl = second ? g - a.Pa : g;
      const float* means = second ? a.means_b : a.means;
      sl[0] = means[3 * gl]; sl[1] = means[3 * gl + 1]; sl[2] = means[3 * gl + 2]; sl[3] = __uint_as_float((uint32_t)g);
      wm[si] = seen;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int q = lane & 3, grp = lane >> 2;
    for (int r0 = 0; r0 < nlive; r0 += 16) {
      const int si = r0 + grp;
      if (si < nlive) {
        const float4 slv = *reinterpret_cast<const float4*>(ws + 4 * si);
        const uint32_t gg = __float_as_uint(slv.w);
        const uint32_t mv = wm[si];
        float sv[12];
        if (gg >= (uint32_t)a.Pa) load_sh12<true>(a.sh_b, a.sh_dc_b, gg - (uint32_t)a.Pa, q, sv);
        else load_sh12<true>(a.sh, a.sh_dc, gg, q, sv);
#pragma unroll 1
        for (int v = 0; v < a.B; ++v) {
          if (!((mv >> v) & 1u)) continue;
          const float* cam = a.vpack[v].cam;
          const float dx = slv.x - cam[0], dy = slv.y - cam[1], dz = slv.z - cam[2];
          const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
          const float x = dx * inv, y = dy * inv, z = dz * inv;
          float b[16], gx[16], gy[16], gz[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) b[k] = 0.f;
          sh_basis(a.deg, x, y, z, b);
          float bs[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bs[j] = pick4(q, b[j], b[4 + j], b[8 + j], b[12 + j]);
          float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) { c0 = fmaf(bs[j], sv[3 * j], c0); c1 = fmaf(bs[j], sv[3 * j + 1], c1); c2 = fmaf(bs[j], sv[3 * j + 2], c2); }
          c0 = quad_sum(c0) + 0.5f; c1 = quad_sum(c1) + 0.5f; c2 = quad_sum(c2) + 0.5f;
          const size_t rec = (size_t)v * (size_t)a.Ppad + gg;
          if (a.D) {
            sh_basis_grad(a.deg, x, y, z, gx, gy, gz);
            float d[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // d[3c + axis]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float ex = pick4(q, gx[j], gx[4 + j], gx[8 + j], gx[12 + j]);
              const float ey = pick4(q, gy[j], gy[4 + j], gy[8 + j], gy[12 + j]);
              const float ez = pick4(q, gz[j], gz[4 + j], gz[8 + j], gz[12 + j]);
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                d[3 * c] = fmaf(ex, sv[3 * j + c], d[3 * c]);
                d[3 * c + 1] = fmaf(ey, sv[3 * j + c], d[3 * c + 1]);
                d[3 * c + 2] = fmaf(ez, sv[3 * j + c], d[3 * c + 2]);
              }
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) d[i] = quad_sum(d[i]);
            if (q < 3) {
              const float dx_ = pick4(q, d[0], d[3], d[6], 0.f), dy_ = pick4(q, d[1], d[4], d[7], 0.f),
                          dz_ = pick4(q, d[2], d[5], d[8], 0.f);
              *reinterpret_cast<F3u*>(a.D + rec * 9 + 3 * q) = F3u{dx_, dy_, dz_};
            }
          }
          if (q == 0) *reinterpret_cast<float2*>(reinterpret_cast<float*>(&a.G1[REC * rec]) + 2) = make_float2(clamp_flagged(c0), clamp_flagged(c1));
          else if (q == 1) *reinterpret_cast<float*>(&a.G2[REC * rec]) = clamp_flagged(c2);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the slots are rewritten by the next chunk
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}


// K8+K9, lane-group form: per-Gaussian reduction of the partial rows and the projection chain rule by one thread per
// Gaussian (rows staged through LDS in coalesced chunks, as before); dL/dSH written by four lanes per Gaussian straight
// from (basis(dir), dL/drgb) -- 16-byte stores of consecutive addresses, zeros for culled Gaussians included -- and the
// view-direction term of dL/dmean from the nine d rgb / d dir values K1 left behind: the SH coefficients are not read.
constexpr int ROW_CHUNK = 128;     // partial rows staged per round (6 KB per wave)
constexpr int HAND_W = 7;          // hand-over: unit direction (3) + clamped dL/drgb (3), odd stride

// ACC: the outputs are added to instead of overwritten (PreBwdArgs::accumulate) -- a template parameter so that the
// overwriting kernel contains no loads of its outputs at all.
// NDL (GSR_FLAG_NEEDLE_DOUBLE): the Gaussians k_pre_geom marked as needles (a number, not NaN, in their first abc word) run
// their whole chain rule in double (gsr_math.h needle_bwd_d); a separate instantiation, because the double chain's
// registers cost the kernel a wave per SIMD.
template <bool RAW, bool GEOM, bool ACC = false, bool NDL = false>
__global__ void __launch_bounds__(PRE_BLOCK) k_pre_bwd(PreBwdArgs a) {
  __shared__ float4 srow[PRE_WAVES * ROW_CHUNK * PART_F4];
  __shared__ float shand[PRE_WAVES * 64 * HAND_W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw0 = a.g0 + blockIdx.x * PRE_BLOCK + wave * 64;    // first Gaussian of this wave
  const int g = gw0 + lane;
  float4* wrow = &srow[wave * ROW_CHUNK * PART_F4];
  float* hand = &shand[wave * 64 * HAND_W];
  const int nw = min(64, a.P - gw0);                     // Gaussians this wave owns (may be <= 0)
  if (nw <= 0) return;
  const int vw = a.vpack != nullptr ? (int)blockIdx.y : 0;
  if (a.vpack != nullptr) {
    // this workgroup's view of a batch: its slice of the virtual scene's arrays, its own output buffers (uniform offsets)
    const size_t o = (size_t)vw * (size_t)a.Ppad;
    const size_t vs = (size_t)vw * (size_t)a.vstride;
    a.offg += o; a.G0 += REC * o; a.G1 += REC * o; a.G2 += REC * o;
    if (a.D) a.D += 9 * o;
    if (a.dmeans2D) a.dmeans2D += 3 * (size_t)vw * (size_t)a.Pscene;
    if (a.dmeans3D) a.dmeans3D += vs;
    if (a.dsh) a.dsh += vs;
    if (a.dsh_dc) a.dsh_dc += vs;
    if (a.dopac) a.dopac += vs;
    if (a.dscales) a.dscales += vs;
    if (a.drots) a.drots += vs;
  }
  uint32_t o0 = 0, o1 = 0;
  if (g < a.P) { o0 = a.offg[g] * a.nsub; o1 = a.offg[g + 1] * a.nsub; }
  constexpr bool acc = ACC;
  auto put = [](float* p, float v) { if (ACC) *p += v; else *p = v; };
  // the record and the position of a Gaussian with rows are requested now, ahead of the row summation that does not
  // depend on them (the kernel is a chain of dependent memory phases per wave: every phase started early is time won)
  float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0, e2 = e0;
  float p[3] = {0.f, 0.f, 0.f};
  if (o1 != o0) {
    e0 = a.G0[REC * g]; e1 = a.G1[REC * g]; e2 = a.G2[REC * g];
    p[0] = a.means[3 * g]; p[1] = a.means[3 * g + 1]; p[2] = a.means[3 * g + 2];
  }
  // ---- sum this Gaussian's partial rows (the wave's rows are one contiguous span) ------------------------------------
  // (the second moments are summed in double: what they feed -- dL/dconic -> dL/dcov2D -- cancels to first order for an
  // elongated splat, see project_splat_bwd, and amplifies the rounding of a float32 running sum over hundreds of rows)
  float dop = 0.f, dr = 0.f, dg = 0.f, db = 0.f;
  double mx = 0.0, my = 0.0, mxx = 0.0, mxy = 0.0, myy = 0.0;      // (conic . (mx, my) cancels the same way)
  {
    const uint32_t S = a.offg[gw0] * a.nsub, E = a.offg[gw0 + nw] * a.nsub;
    const bool big = (o1 - o0) > (uint32_t)ROW_CHUNK;
    for (uint32_t c0 = S; c0 < E; c0 += (uint32_t)ROW_CHUNK) {
      const uint32_t rows = min((uint32_t)ROW_CHUNK, E - c0);
      const float4* src = a.part + (size_t)c0 * PART_F4;
      for (uint32_t i = lane; i < rows * PART_F4; i += 64) wrow[i] = src[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (!big) {
        const uint32_t lo = max(o0, c0), hi = min(o1, c0 + rows);
        for (uint32_t e = lo; e < hi; ++e) {
          const float4* r = &wrow[(e - c0) * PART_F4];
          const float4 p0 = r[0], p1 = r[1], p2 = r[2];
          if (__float_as_uint(p2.y) != a.tag_lo || __float_as_uint(p2.z) != a.tag_hi) continue;
          if (GEOM) { mx += (double)p0.x; my += (double)p0.y; mxx += (double)p0.z; mxy += (double)p0.w; myy += (double)p1.x; dop += p1.y; }
          dr += p1.z; dg += p1.w; db += p2.x;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint64_t bm = __ballot(big);
    while (bm) {                                         // a Gaussian with more rows than a chunk: the whole wave sums it
      const int L = __ffsll((unsigned long long)bm) - 1;
      bm &= bm - 1;
      const uint32_t b0 = __shfl(o0, L, 64), b1 = __shfl(o1, L, 64);
      float t[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      double t2[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      for (uint32_t e = b0 + lane; e < b1; e += 64) {
        const float4 p0 = a.part[(size_t)e * PART_F4], p1 = a.part[(size_t)e * PART_F4 + 1], p2 = a.part[(size_t)e * PART_F4 + 2];
        if (__float_as_uint(p2.y) != a.tag_lo || __float_as_uint(p2.z) != a.tag_hi) continue;
        if (GEOM) { t2[3] += (double)p0.x; t2[4] += (double)p0.y; t2[0] += (double)p0.z; t2[1] += (double)p0.w; t2[2] += (double)p1.x; t[5] += p1.y; }
        t[6] += p1.z; t[7] += p1.w; t[8] += p2.x;
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) t[i] = __shfl(wave_sum_to_hi(t[i]), 63, 64);
      if (GEOM) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
          for (int sft = 32; sft > 0; sft >>= 1) t2[i] += __shfl_xor(t2[i], sft, 64);
        }
      }
      if (lane == L) { mx = t2[3]; my = t2[4]; mxx = t2[0]; mxy = t2[1]; myy = t2[2]; dop = t[5]; dr = t[6]; dg = t[7]; db = t[8]; }
    }
  }
  // ---- phase A: chain rule per Gaussian ------------------------------------------------------------------------------
  float hdir[3] = {0.f, 0.f, 0.f}, hrgb[3] = {0.f, 0.f, 0.f};
  // squares of what this lane writes, per attribute tensor (PreBwdArgs::sumsq): the L2 step's global norms come out of
  // the kernel that produces the gradients instead of a second pass over them (a Gaussian without pairs writes zeros)
  float ssq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_ss = RAW && !ACC && a.sumsq != nullptr;
  // (a wave without Gaussians returns above without writing its slot of the partial sums: with one wave per workgroup
  // there is no such wave inside the grid -- k_sumsq_reduce reads ss_blocks * PRE_WAVES slots)
  static_assert(PRE_WAVES == 1, "more waves per workgroup: empty waves must zero their sum-of-squares slot");
  auto flush_sumsq = [&]() {
    if (!want_ss) return;
#pragma unroll
    for (int k = 0; k < 6; ++k) ssq[k] = wave_sum_to_hi(ssq[k]);
    if (lane == 63) {
      float* dst = a.sumsq + (size_t)(blockIdx.x * PRE_WAVES + wave) * SUMSQ_W;
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = ssq[k];
    }
  };
  if (g < a.P) {
    if (o1 == o0 && acc) {
      // nothing to add to the caller's bucket for a Gaussian without pairs; the two per-VIEW outputs (screen-space
      // gradient, object features) are not part of the bucket and are overwritten in either mode
      if (a.dmeans2D) { a.dmeans2D[3 * g] = 0.f; a.dmeans2D[3 * g + 1] = 0.f; a.dmeans2D[3 * g + 2] = 0.f; }
      if (a.dsh_objs) for (int i = 0; i < NUM_OBJ; ++i) a.dsh_objs[(size_t)g * NUM_OBJ + i] = 0.f;
    } else if (o1 == o0) {   // culled: zero gradients (dL/dSH: phase B writes the zeros)
      if (a.dmeans3D) { a.dmeans3D[3 * g] = 0.f; a.dmeans3D[3 * g + 1] = 0.f; a.dmeans3D[3 * g + 2] = 0.f; }
      if (a.dmeans2D) { a.dmeans2D[3 * g] = 0.f; a.dmeans2D[3 * g + 1] = 0.f; a.dmeans2D[3 * g + 2] = 0.f; }
      if (a.dsh_objs) for (int i = 0; i < NUM_OBJ; ++i) a.dsh_objs[(size_t)g * NUM_OBJ + i] = 0.f;
      if (a.dcolors) { a.dcolors[3 * g] = 0.f; a.dcolors[3 * g + 1] = 0.f; a.dcolors[3 * g + 2] = 0.f; }
      if (a.dopac) a.dopac[g] = 0.f;
      if (a.dscales) { a.dscales[3 * g] = 0.f; a.dscales[3 * g + 1] = 0.f; a.dscales[3 * g + 2] = 0.f; }
      if (a.drots) { a.drots[4 * g] = 0.f; a.drots[4 * g + 1] = 0.f; a.drots[4 * g + 2] = 0.f; a.drots[4 * g + 3] = 0.f; }
      if (a.dcov3d) for (int i = 0; i < 6; ++i) a.dcov3d[6 * g + i] = 0.f;
    } else {
      if (a.dsh_objs) {
        float acc_o[NUM_OBJ];
#pragma unroll
        for (int c = 0; c < NUM_OBJ; ++c) acc_o[c] = 0.f;
        if (a.part_obj) {
          for (uint32_t e = o0; e < o1; ++e) {
            const float4 tg = a.part[(size_t)e * PART_F4 + 2];
            if (__float_as_uint(tg.y) != a.tag_lo || __float_as_uint(tg.z) != a.tag_hi) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 v4 = a.part_obj[(size_t)e * 4 + q];
              acc_o[4 * q] += v4.x; acc_o[4 * q + 1] += v4.y; acc_o[4 * q + 2] += v4.z; acc_o[4 * q + 3] += v4.w;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NUM_OBJ; ++c) a.dsh_objs[(size_t)g * NUM_OBJ + c] = acc_o[c];      // per view: never added to
      }
      View v;
      if (a.vpack != nullptr) {
        const ViewDev& vd = a.vpack[vw];
        make_view(v, vd.vm, vd.pm, vd.cam, a.va.H, a.va.W, vd.tanfovx, vd.tanfovy, a.va.mod, a.va.deg);
      } else {
        load_view(v, a.va);
      }
      const float A = e0.z, B = e0.w, C = e1.x;
      // dL/d(pixel centre) = -(A mx + B my, B mx + C my); screen-space means are reported in NDC units
      const float dndcx = (float)(-((double)A * mx + (double)B * my) * (0.5 * (double)v.W));
      const float dndcy = (float)(-((double)B * mx + (double)C * my) * (0.5 * (double)v.H));
      const double dA = -0.5 * mxx, dB = -mxy, dC = -0.5 * myy;
      if (GEOM && a.dmeans2D) { a.dmeans2D[3 * g] = dndcx; a.dmeans2D[3 * g + 1] = dndcy; a.dmeans2D[3 * g + 2] = 0.f; }   // per view
      if (GEOM && a.dopac) {
        const float dov = RAW ? dop * e1.y * (1.f - e1.y) : dop;                       // e1.y = sigmoid(raw opacity)
        put(&a.dopac[g], dov);
        ssq[SUMSQ_OPACITY] = dov * dov;
      }
      float dp[3] = {0.f, 0.f, 0.f};
      if (a.dcolors) { put(&a.dcolors[3 * g], dr); put(&a.dcolors[3 * g + 1], dg); put(&a.dcolors[3 * g + 2], db); }
      if (a.sh) {
        const uint32_t cl = clamp_bits_of(e1.z, e1.w, e2.x);
        hrgb[0] = (cl & 1u) ? 0.f : dr; hrgb[1] = (cl & 2u) ? 0.f : dg; hrgb[2] = (cl & 4u) ? 0.f : db;
        const float vx = p[0] - v.cam[0], vy = p[1] - v.cam[1], vz = p[2] - v.cam[2];
        const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
        hdir[0] = vx * inv; hdir[1] = vy * inv; hdir[2] = vz * inv;
        if (GEOM) {
          // view-direction path of dL/dmean: dL/dd = D^T dL/drgb, then d = v/|v|: dL/dv = (dL/dd - d (d . dL/dd)) / |v|
          const float* Dg = a.D + (size_t)g * 9;
          const float ddx = Dg[0] * hrgb[0] + Dg[3] * hrgb[1] + Dg[6] * hrgb[2];
          const float ddy = Dg[1] * hrgb[0] + Dg[4] * hrgb[1] + Dg[7] * hrgb[2];
          const float ddz = Dg[2] * hrgb[0] + Dg[5] * hrgb[1] + Dg[8] * hrgb[2];
          const float dot = hdir[0] * ddx + hdir[1] * ddy + hdir[2] * ddz;
          dp[0] += (ddx - hdir[0] * dot) * inv;
          dp[1] += (ddy - hdir[1] * dot) * inv;
          dp[2] += (ddz - hdir[2] * dot) * inv;
        }
      }
      if (GEOM) {
        float c6[6];
        float sc[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
        float inv_qn = 1.f;
        if (a.cov3d) {
#pragma unroll
          for (int i = 0; i < 6; ++i) c6[i] = a.cov3d[6 * g + i];
        } else {
          sc[0] = a.scales[3 * g]; sc[1] = a.scales[3 * g + 1]; sc[2] = a.scales[3 * g + 2];
          const float4 q4 = reinterpret_cast<const float4*>(a.rots)[g];
          q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
          if (RAW) {
            sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]);
            act_normalize4(q, q, inv_qn);
          }
          cov3d_from_scale_rot(sc, a.va.mod, q, c6);
        }
        float dc6[6];
        float ds[3], dq[4];
        bool needle = false;
        if (NDL && a.abc) { const double m0 = a.abc[(size_t)g]; needle = m0 == m0; }      // NaN: an ordinary splat
        if (NDL && needle) {
          // a needle: the whole chain rule in double on the float32 inputs (like its forward), activations included
          double dpd[3] = {(double)dp[0], (double)dp[1], (double)dp[2]}, dsd[3], dqd[4], dS6[6];
          const double dnx = -((double)A * mx + (double)B * my) * (0.5 * (double)v.W);
          const double dny = -((double)B * mx + (double)C * my) * (0.5 * (double)v.H);
          needle_bwd_d(v, p, sc, a.va.mod, q, a.cov3d ? c6 : nullptr, dA, dB, dC, dnx, dny, dpd, dsd, dqd, dS6);
          dp[0] = (float)dpd[0]; dp[1] = (float)dpd[1]; dp[2] = (float)dpd[2];
#pragma unroll
          for (int i = 0; i < 6; ++i) dc6[i] = (float)dS6[i];
          if (RAW) {
            dsd[0] *= (double)sc[0]; dsd[1] *= (double)sc[1]; dsd[2] *= (double)sc[2];     // d exp(x) = exp(x)
            const double dot = (double)q[0] * dqd[0] + (double)q[1] * dqd[1] + (double)q[2] * dqd[2] + (double)q[3] * dqd[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) dqd[i] = (dqd[i] - (double)q[i] * dot) * (double)inv_qn;
          }
          ds[0] = (float)dsd[0]; ds[1] = (float)dsd[1]; ds[2] = (float)dsd[2];
          dq[0] = (float)dqd[0]; dq[1] = (float)dqd[1]; dq[2] = (float)dqd[2]; dq[3] = (float)dqd[3];
        } else {
          project_splat_bwd(v, p, c6, dA, dB, dC, dndcx, dndcy, dp, dc6);
        }
        if (a.dmeans3D) {
          put(&a.dmeans3D[3 * g], dp[0]); put(&a.dmeans3D[3 * g + 1], dp[1]); put(&a.dmeans3D[3 * g + 2], dp[2]);
          ssq[SUMSQ_XYZ] = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
        }
        if (a.cov3d) {
          if (a.dcov3d) for (int i = 0; i < 6; ++i) put(&a.dcov3d[6 * g + i], dc6[i]);
        } else if (a.dscales || a.drots) {
          if (!(NDL && needle)) {
            cov3d_bwd(sc, a.va.mod, q, dc6, ds, dq);
            if (RAW) {
              ds[0] *= sc[0]; ds[1] *= sc[1]; ds[2] *= sc[2];     // d exp(x) = exp(x)
              act_normalize4_bwd(q, inv_qn, dq, dq);
            }
          }
          if (a.dscales) {
            put(&a.dscales[3 * g], ds[0]); put(&a.dscales[3 * g + 1], ds[1]); put(&a.dscales[3 * g + 2], ds[2]);
            ssq[SUMSQ_SCALING] = ds[0] * ds[0] + ds[1] * ds[1] + ds[2] * ds[2];
          }
          if (a.drots) {
            put(&a.drots[4 * g], dq[0]); put(&a.drots[4 * g + 1], dq[1]); put(&a.drots[4 * g + 2], dq[2]); put(&a.drots[4 * g + 3], dq[3]);
            ssq[SUMSQ_ROTATION] = dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3];
          }
        }
      }
    }
  }
  if (a.dsh == nullptr) { flush_sumsq(); return; }
  // ---- phase B: four lanes per Gaussian write dL/dSH = basis(dir) x dL/drgb, 48 floats per Gaussian, coalesced -------
  {
    float* h = hand + HAND_W * lane;
    h[0] = hdir[0]; h[1] = hdir[1]; h[2] = hdir[2]; h[3] = hrgb[0]; h[4] = hrgb[1]; h[5] = hrgb[2];
    h[6] = (o1 != o0) ? 1.f : 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int q = lane & 3, grp = lane >> 2;
  const int deg = a.va.deg;
#pragma unroll 1
  for (int r0 = 0; r0 < nw; r0 += 16) {
    const int si = r0 + grp;
    if (si < nw) {
      const float* h = hand + HAND_W * si;
      const float x = h[0], y = h[1], z = h[2], g0 = h[3], g1 = h[4], g2 = h[5];
      float b[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) b[k] = 0.f;
      sh_basis(deg, x, y, z, b);
      float out[12];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bj = pick4(q, b[j], b[4 + j], b[8 + j], b[12 + j]);
        out[3 * j] = bj * g0; out[3 * j + 1] = bj * g1; out[3 * j + 2] = bj * g2;
      }
      if (acc) {
        if (h[6] != 0.f) {
          float cur[12];
          load_sh12<RAW>(a.dsh, a.dsh_dc, (uint32_t)(gw0 + si), q, cur);
#pragma unroll
          for (int j = 0; j < 12; ++j) out[j] += cur[j];
          store_sh12<RAW>(a.dsh, a.dsh_dc, (uint32_t)(gw0 + si), q, out);
        }
      } else {
        store_sh12<RAW>(a.dsh, a.dsh_dc, (uint32_t)(gw0 + si), q, out);
        if (want_ss) {
          // lane q = 0 holds coefficient 0 (= _features_dc) in out[0..2]; everything else is _features_rest
          const float head = out[0] * out[0] + out[1] * out[1] + out[2] * out[2];
          float rest = 0.f;
#pragma unroll
          for (int j = 3; j < 12; ++j) rest = fmaf(out[j], out[j], rest);
          ssq[SUMSQ_DC] += q == 0 ? head : 0.f;
          ssq[SUMSQ_REST] += q == 0 ? rest : rest + head;
        }
      }
    }
  }
  flush_sumsq();
}


// ------------------------------------------------------------------------------------------------
// K8+K9 of a BATCH of views (gsr_backward_raw_batch_into; raw parameters): one wave per 64 consecutive Gaussians walks
// the B views of the virtual scene (ViewDev) -- per view the partial rows of its (tile, Gaussian) pairs (contiguous per
// wave, staged through LDS like k_pre_bwd's), the record and the d colour / d direction values of (view, Gaussian), the
// projection chain rule for that view's camera -- and keeps the sums over the views in registers: dL/dmean, dL/dSigma3D
// (its push to scale and rotation is linear: done once, behind the view loop), dL/dopacity, and per lane group the 12
// dL/dSH floats it owns.  The 59 gradient floats of a Gaussian are written ONCE per batch (B single-view launches in
// accumulate mode read and rewrite 236 bytes per Gaussian and view), zeros when no view sees it; dmeans2D [B,P,3] is per
// view.  ACC: the sums are added to what the caller's buffers hold (Gaussians that no view sees are left alone).
// Dynamic LDS: 3 * 64 * B floats (the views' clamped dL/drgb for the dL/dSH phase).
// ------------------------------------------------------------------------------------------------
struct PreBwdBatchArgs {
  int P, g0;              // Gaussians [g0, P) (g0 a multiple of 64); block b owns g0 + 64 b ...
  int B, Ppad;
  const ViewDev* vpack;
  int H, W, deg;
  float mod;
  const uint32_t* offg;   // [B * Ppad + 1]
  const float4* G0;
  const float4* G1;
  const float4* G2;
  const float4* part;
  uint32_t tag_lo, tag_hi, nsub;
  const float* means;
  const float* scales;    // raw: log scales
  const float* rots;      // raw: un-normalised quaternions
  const float* D;         // [B * Ppad, 9]
  float* dmeans3D;
  float* dmeans2D;        // [B, P, 3] or null
  float* dsh;             // gradient of _features_rest
  float* dsh_dc;          // gradient of _features_dc
  float* dopac;
  float* dscales;
  float* drots;
  float* sumsq;           // ACC = false only, or null: [workgroups][SUMSQ_W] sums of squares of what is written
};

// A workgroup is BATCH_K9_WAVES waves that share 64 Gaussians: wave w walks the views v = w, w + W, ... (the walk is a chain of
// dependent memory phases per view; W chains of B / W views finish sooner than one of B), the waves' sums meet in LDS, wave 0
// finishes and stores the geometry gradients, and the dL/dSH phase is dealt over the waves by groups of 16 Gaussians.
constexpr int BATCH_K9_WAVES = 2;
// Waves per SIMD the geometry variant is compiled for: 3 = 164 VGPRs without scratch (left alone the allocator takes 172 and
// the kernel drops to 2 waves: 0.085 -> 0.115 ms per view at B = 8); 4 = 128 VGPRs and ~100 bytes of scratch per lane.
#ifndef GSR_BATCH_K9_OCC
#define GSR_BATCH_K9_OCC 3
#endif
template <bool GEOM, bool ACC>
__global__ void __launch_bounds__(64 * BATCH_K9_WAVES, GEOM ? GSR_BATCH_K9_OCC : 1) k_pre_bwd_batch(PreBwdBatchArgs a) {
  __shared__ float4 srow_all[BATCH_K9_WAVES][ROW_CHUNK * PART_F4];
  __shared__ float spos[64 * 3];
  __shared__ uint32_t sany[BATCH_K9_WAVES][64];
  __shared__ float sacc[BATCH_K9_WAVES > 1 ? BATCH_K9_WAVES - 1 : 1][12][64];     // the other waves' sums: [value][lane]
  extern __shared__ float shrgb[];                       // [B][64][3]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float4* const srow = srow_all[wave];
  const int gw0 = a.g0 + blockIdx.x * 64;                // first Gaussian of this workgroup
  const int g = gw0 + lane;
  const int nw = min(64, a.P - gw0);                     // (the grid covers [g0, P): nw >= 1)
  const bool mine = g < a.P;
  float p[3] = {0.f, 0.f, 0.f};
  if (mine) { p[0] = a.means[3 * g]; p[1] = a.means[3 * g + 1]; p[2] = a.means[3 * g + 2]; }
  if (wave == 0) { spos[3 * lane] = p[0]; spos[3 * lane + 1] = p[1]; spos[3 * lane + 2] = p[2]; }
  // view-independent: activated scale / rotation and the 3D covariance
  // (the activated scale and rotation themselves are formed again behind the view loop, where dL/dSigma3D is pushed through
  // them: eight registers less across the loop)
  float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto activated = [&](float sc[3], float q[4], float& inv_qn) {
    sc[0] = expf(a.scales[3 * g]); sc[1] = expf(a.scales[3 * g + 1]); sc[2] = expf(a.scales[3 * g + 2]);
    const float4 q4 = reinterpret_cast<const float4*>(a.rots)[g];
    q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    act_normalize4(q, q, inv_qn);
  };
  if (GEOM && mine) {
    float sc[3], q[4], inv_qn;
    activated(sc, q, inv_qn);
    cov3d_from_scale_rot(sc, a.mod, q, c6);
  }
  float dp[3] = {0.f, 0.f, 0.f}, dc6s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dops = 0.f, opv = 0.f;
  bool any = false;
#pragma unroll 1
  for (int v = wave; v < a.B; v += BATCH_K9_WAVES) {
    const size_t o = (size_t)v * (size_t)a.Ppad;
    const uint32_t* offg = a.offg + o;
    uint32_t o0 = 0, o1 = 0;
    if (mine) { o0 = offg[g] * a.nsub; o1 = offg[g + 1] * a.nsub; }
    const bool has = o1 != o0;
    float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0, e2 = e0;
    if (has) { e0 = a.G0[REC * (o + g)]; e1 = a.G1[REC * (o + g)]; e2 = a.G2[REC * (o + g)]; }
    // ---- this view's partial rows of the wave's Gaussians (one contiguous span), as in k_pre_bwd -----------------------
    float dop = 0.f, dr = 0.f, dg = 0.f, db = 0.f;
    double mx = 0.0, my = 0.0, mxx = 0.0, mxy = 0.0, myy = 0.0;
    {
      const uint32_t S = offg[gw0] * a.nsub, E = offg[gw0 + nw] * a.nsub;
      const bool big = (o1 - o0) > (uint32_t)ROW_CHUNK;
      for (uint32_t c0 = S; c0 < E; c0 += (uint32_t)ROW_CHUNK) {
        const uint32_t rows = min((uint32_t)ROW_CHUNK, E - c0);
        const float4* src = a.part + (size_t)c0 * PART_F4;
        for (uint32_t i = lane; i < rows * PART_F4; i += 64) srow[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!big) {
          const uint32_t lo = max(o0, c0), hi = min(o1, c0 + rows);
          for (uint32_t e = lo; e < hi; ++e) {
            const float4* r = &srow[(e - c0) * PART_F4];
            const float4 p0 = r[0], p1 = r[1], p2 = r[2];
            if (__float_as_uint(p2.y) != a.tag_lo || __float_as_uint(p2.z) != a.tag_hi) continue;
            if (GEOM) { mx += (double)p0.x; my += (double)p0.y; mxx += (double)p0.z; mxy += (double)p0.w; myy += (double)p1.x; dop += p1.y; }
            dr += p1.z; dg += p1.w; db += p2.x;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      uint64_t bm = __ballot(big);
      while (bm) {                                         // a Gaussian with more rows than a chunk: the whole wave sums it
        const int L = __ffsll((unsigned long long)bm) - 1;
        bm &= bm - 1;
        const uint32_t b0 = __shfl(o0, L, 64), b1 = __shfl(o1, L, 64);
        float t[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        double t2[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (uint32_t e = b0 + lane; e < b1; e += 64) {
          const float4 p0 = a.part[(size_t)e * PART_F4], p1 = a.part[(size_t)e * PART_F4 + 1], p2 = a.part[(size_t)e * PART_F4 + 2];
          if (__float_as_uint(p2.y) != a.tag_lo || __float_as_uint(p2.z) != a.tag_hi) continue;
          if (GEOM) { t2[3] += (double)p0.x; t2[4] += (double)p0.y; t2[0] += (double)p0.z; t2[1] += (double)p0.w; t2[2] += (double)p1.x; t[5] += p1.y; }
          t[6] += p1.z; t[7] += p1.w; t[8] += p2.x;
        }
#pragma unroll
        for (int i = 5; i < 9; ++i) t[i] = __shfl(wave_sum_to_hi(t[i]), 63, 64);
        if (GEOM) {
#pragma unroll
          for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) t2[i] += __shfl_xor(t2[i], sft, 64);
          }
        }
        if (lane == L) { mx = t2[3]; my = t2[4]; mxx = t2[0]; mxy = t2[1]; myy = t2[2]; dop = t[5]; dr = t[6]; dg = t[7]; db = t[8]; }
      }
    }
    // ---- chain rule of this view -------------------------------------------------------------------------------------
    float hr[3] = {0.f, 0.f, 0.f};
    if (has) {
      any = true;
      const ViewDev& vd = a.vpack[v];
      const uint32_t cl = clamp_bits_of(e1.z, e1.w, e2.x);
      hr[0] = (cl & 1u) ? 0.f : dr; hr[1] = (cl & 2u) ? 0.f : dg; hr[2] = (cl & 4u) ? 0.f : db;
      if (GEOM) {
        View vw;
        make_view(vw, vd.vm, vd.pm, vd.cam, a.H, a.W, vd.tanfovx, vd.tanfovy, a.mod, a.deg);
        const float A = e0.z, Bc = e0.w, C = e1.x;
        const float dndcx = (float)(-((double)A * mx + (double)Bc * my) * (0.5 * (double)vw.W));
        const float dndcy = (float)(-((double)Bc * mx + (double)C * my) * (0.5 * (double)vw.H));
        const double dA = -0.5 * mxx, dB = -mxy, dC = -0.5 * myy;
        if (a.dmeans2D) {
          float* m2 = a.dmeans2D + 3 * ((size_t)v * (size_t)a.P + (size_t)g);
          m2[0] = dndcx; m2[1] = dndcy; m2[2] = 0.f;
        }
        dops += dop;
        opv = e1.y;
        // view-direction path of dL/dmean (d colour / d direction left by the colour kernel)
        const float vx = p[0] - vw.cam[0], vy = p[1] - vw.cam[1], vz = p[2] - vw.cam[2];
        const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
        const float hx = vx * inv, hy = vy * inv, hz = vz * inv;
        const float* Dg = a.D + 9 * (o + (size_t)g);
        const float ddx = Dg[0] * hr[0] + Dg[3] * hr[1] + Dg[6] * hr[2];
        const float ddy = Dg[1] * hr[0] + Dg[4] * hr[1] + Dg[7] * hr[2];
        const float ddz = Dg[2] * hr[0] + Dg[5] * hr[1] + Dg[8] * hr[2];
        const float dot = hx * ddx + hy * ddy + hz * ddz;
        dp[0] += (ddx - hx * dot) * inv;
        dp[1] += (ddy - hy * dot) * inv;
        dp[2] += (ddz - hz * dot) * inv;
        float dc6[6];
        project_splat_bwd(vw, p, c6, dA, dB, dC, dndcx, dndcy, dp, dc6);
#pragma unroll
        for (int i = 0; i < 6; ++i) dc6s[i] += dc6[i];
      }
    } else if (GEOM && mine && a.dmeans2D) {
      float* m2 = a.dmeans2D + 3 * ((size_t)v * (size_t)a.P + (size_t)g);
      m2[0] = 0.f; m2[1] = 0.f; m2[2] = 0.f;
    }
    float* hs = shrgb + ((size_t)v * 64 + lane) * 3;
    hs[0] = hr[0]; hs[1] = hr[1]; hs[2] = hr[2];
  }
  sany[wave][lane] = any ? 1u : 0u;
  // ---- the waves' sums meet: waves 1.. park theirs, wave 0 adds them up --------------------------------------------------
  if (BATCH_K9_WAVES > 1) {
    if (GEOM && wave > 0) {
      float (*dst)[64] = sacc[wave - 1];
      dst[0][lane] = dp[0]; dst[1][lane] = dp[1]; dst[2][lane] = dp[2];
#pragma unroll
      for (int i = 0; i < 6; ++i) dst[3 + i][lane] = dc6s[i];
      dst[9][lane] = dops; dst[10][lane] = opv;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < BATCH_K9_WAVES; ++w) any = any || sany[w][lane] != 0u;
    if (GEOM && wave == 0) {
#pragma unroll
      for (int w = 1; w < BATCH_K9_WAVES; ++w) {
        float (*src)[64] = sacc[w - 1];
        dp[0] += src[0][lane]; dp[1] += src[1][lane]; dp[2] += src[2][lane];
#pragma unroll
        for (int i = 0; i < 6; ++i) dc6s[i] += src[3 + i][lane];
        dops += src[9][lane];
        if (sany[w][lane] != 0u) opv = src[10][lane];
      }
    }
  }
  // ---- the sums over the views: scale / rotation / opacity through the activations, then the stores (wave 0) --------------
  float ssq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_ss = !ACC && a.sumsq != nullptr;
  auto put = [](float* ptr, float val) { if (ACC) *ptr += val; else *ptr = val; };
  if (GEOM && wave == 0 && mine && (any || !ACC)) {
    float ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (any && (a.dscales || a.drots)) {
      float sc[3], q[4], inv_qn;
      activated(sc, q, inv_qn);
      cov3d_bwd(sc, a.mod, q, dc6s, ds, dq);
      ds[0] *= sc[0]; ds[1] *= sc[1]; ds[2] *= sc[2];          // d exp(x) = exp(x)
      act_normalize4_bwd(q, inv_qn, dq, dq);
    }
    if (a.dopac) {
      const float dov = dops * opv * (1.f - opv);              // opv = sigmoid(raw opacity)
      put(&a.dopac[g], dov);
      ssq[SUMSQ_OPACITY] = dov * dov;
    }
    if (a.dmeans3D) {
      put(&a.dmeans3D[3 * g], dp[0]); put(&a.dmeans3D[3 * g + 1], dp[1]); put(&a.dmeans3D[3 * g + 2], dp[2]);
      ssq[SUMSQ_XYZ] = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
    }
    if (a.dscales) {
      put(&a.dscales[3 * g], ds[0]); put(&a.dscales[3 * g + 1], ds[1]); put(&a.dscales[3 * g + 2], ds[2]);
      ssq[SUMSQ_SCALING] = ds[0] * ds[0] + ds[1] * ds[1] + ds[2] * ds[2];
    }
    if (a.drots) {
      put(&a.drots[4 * g], dq[0]); put(&a.drots[4 * g + 1], dq[1]); put(&a.drots[4 * g + 2], dq[2]); put(&a.drots[4 * g + 3], dq[3]);
      ssq[SUMSQ_ROTATION] = dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3];
    }
  }
  auto flush_sumsq = [&]() {
    if (!want_ss) return;
#pragma unroll
    for (int k = 0; k < 6; ++k) ssq[k] = wave_sum_to_hi(ssq[k]);
    if (lane == 63) {
      float* dst = a.sumsq + ((size_t)blockIdx.x * BATCH_K9_WAVES + wave) * SUMSQ_W;      // every wave of the grid writes its slot
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = ssq[k];
    }
  };
  if (a.dsh == nullptr) { flush_sumsq(); return; }
  // ---- dL/dSH = sum over the views of basis(direction of the view) x dL/drgb of the view: four lanes per Gaussian, the
  // groups of 16 Gaussians dealt over the waves (the views' dL/drgb of all 64 are in LDS behind the barrier above) -----------
  if (BATCH_K9_WAVES == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const int qd = lane & 3, grp = lane >> 2;
#pragma unroll 1
  for (int r0 = 16 * wave; r0 < nw; r0 += 16 * BATCH_K9_WAVES) {
    const int si = r0 + grp;
    if (si < nw) {
      bool seen = false;
#pragma unroll
      for (int w = 0; w < BATCH_K9_WAVES; ++w) seen = seen || sany[w][si] != 0u;
      float out[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) out[j] = 0.f;
      if (seen) {
        const float px = spos[3 * si], py = spos[3 * si + 1], pz = spos[3 * si + 2];
#pragma unroll 1
        for (int v = 0; v < a.B; ++v) {
          const float* hs = shrgb + ((size_t)v * 64 + si) * 3;
          const float g0 = hs[0], g1 = hs[1], g2 = hs[2];
          const float* cam = a.vpack[v].cam;
          const float vx = px - cam[0], vy = py - cam[1], vz = pz - cam[2];
          const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
          float b[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) b[k] = 0.f;
          sh_basis(a.deg, vx * inv, vy * inv, vz * inv, b);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float bj = pick4(qd, b[j], b[4 + j], b[8 + j], b[12 + j]);
            out[3 * j] = fmaf(bj, g0, out[3 * j]); out[3 * j + 1] = fmaf(bj, g1, out[3 * j + 1]); out[3 * j + 2] = fmaf(bj, g2, out[3 * j + 2]);
          }
        }
      }
      if (ACC) {
        if (seen) {
          float cur[12];
          load_sh12<true>(a.dsh, a.dsh_dc, (uint32_t)(gw0 + si), qd, cur);
#pragma unroll
          for (int j = 0; j < 12; ++j) out[j] += cur[j];
          store_sh12<true>(a.dsh, a.dsh_dc, (uint32_t)(gw0 + si), qd, out);
        }
      } else {
        store_sh12<true>(a.dsh, a.dsh_dc, (uint32_t)(gw0 + si), qd, out);
        if (want_ss) {
          const float head = out[0] * out[0] + out[1] * out[1] + out[2] * out[2];
          float rest = 0.f;
#pragma unroll
          for (int j = 3; j < 12; ++j) rest = fmaf(out[j], out[j], rest);
          ssq[SUMSQ_DC] += qd == 0 ? head : 0.f;
          ssq[SUMSQ_REST] += qd == 0 ? rest : rest + head;
        }
      }
    }
  }
  flush_sumsq();
}

}  // namespace gsr
