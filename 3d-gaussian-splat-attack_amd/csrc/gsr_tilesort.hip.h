// gsr_tilesort.hip.h -- depth order inside every tile's list, made in LDS by the tile's own wave / workgroup.
//
// The reference sorts all N (tile, Gaussian) pairs on a 64-bit (tile << 32 | depth bits) key (SURVEY.md section 2.2,
// K4).  Rounds 1-3 of this build sorted the Gaussians by depth first (three counting passes over V keys = eight dependent
// launches of latency-bound kernels, a rank-order scan behind them) and emitted the pairs in that order.  Here the pairs
// are emitted in STORAGE order straight behind the storage-order scan, sorted globally by tile id only, and each tile's
// list -- a few hundred entries on the benchmark scene, a few thousand on the dense one: it fits in a CU's 160 KB of LDS
// many times over -- is then ordered by (depth key, storage index) where it lies: the composite key IS the order a
// stable depth sort of storage-ordered Gaussians gives, ties included.  One launch instead of ten.
//
// A list arrives sorted by storage index (the tile sort is stable), so a STABLE sort on the 32-bit depth key alone gives
// the composite order: LSD counting passes of 8 bits over (key - min key of the list), as many as the list's key range
// needs (three on the benchmark scene), ranked with wave ballots like the global passes (gsr_sort.hip.h).  Keys and
// values stay in registers between passes; LDS holds one scatter image (8 bytes per entry) and the digit counters.
//   * lists of up to 512 entries (eight 64-element rounds per lane): ONE WAVE, no workgroup barrier (eight such lists per
//     512-thread workgroup);
//   * up to 4094: the eight waves of a workgroup together (same registers per lane, so the two modes share one kernel
//     at eight waves per SIMD);
//   * up to 16384: a 1024-thread workgroup with sixteen rounds per lane (k_tile_depth_sort_huge: a thin persistent grid
//     that exits at once when k_tile_schedule counted no such tile);
//   * beyond: a bitonic network over (depth key, index) in global memory by that workgroup -- slow, correct at any
//     length (such a tile takes milliseconds to composite anyway).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_sort.hip.h"

namespace gsr {

constexpr int TDS_ROUNDS = 8;                           // 64-element rounds per wave
constexpr int TDS_WAVE_CAP = TDS_ROUNDS * 64;           // 512: one wave sorts a list of this length by itself
constexpr int TDS_WAVES = 8;
constexpr int TDS_BLOCK_CAP = TDS_WAVES * TDS_WAVE_CAP; // 4096
constexpr int TDS_HUGE_WAVES = 16;
constexpr int TDS_HUGE_ROUNDS = 16;
constexpr int TDS_HUGE_CAP = TDS_HUGE_WAVES * TDS_HUGE_ROUNDS * 64;   // 16384
// Waves per SIMD the kernel is compiled for: 6 (80 registers, a dozen spilled to scratch; three 512-thread workgroups per
// CU, the LDS limit) or 4 (no spills, two workgroups per CU).  -DGSR_TDS_OCC=4 builds the other one for an A/B.
#ifndef GSR_TDS_OCC
#define GSR_TDS_OCC 6
#endif
constexpr int TDS_GRID = 256 * (GSR_TDS_OCC * 4 / TDS_WAVES);   // persistent workgroups of k_tile_depth_sort: all co-resident
constexpr uint32_t TDS_HUGE_MIN = 4095u;                // = the schedule's last (clamped) length bin: see k_tile_schedule
constexpr uint32_t TDS_RANK_MASK = (1u << 28) - 1u;     // a pair's value = Gaussian | strip mask << 28

struct TileSortArgs {
  const uint2* ranges;        // [ntiles] span of every tile in the tile-sorted pair list
  const uint32_t* sched;      // [ntiles] tiles longest list first (k_tile_schedule), low 28 bits = tile
  const uint32_t* dv;         // DV_NHUGE, DV_NMID
  const uint32_t* dkey;       // [P] float bits of view depth per Gaussian
  uint32_t* vals;             // the pair values, sorted in place
  int ntiles;
};

template <int WAVES>
__device__ __forceinline__ void tds_sync() {
  if (WAVES == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// Sorts vals[0 .. len) (global memory, len <= WAVES * ROUNDS * 64) by (dkey[val & mask], position).  Called by all
// WAVES waves of a group (a wave, or the whole workgroup): `wv` = this wave's index in the group, `gtid` = this thread's.
// s_key / s_val: WAVES * ROUNDS * 64 words each; s_cnt: WAVES * 256 words; s_red: 2 * WAVES + 4 words.
template <int WAVES, int ROUNDS>
__device__ __forceinline__ void tile_list_sort(uint32_t* __restrict__ vals, const uint32_t len,
                                               const uint32_t* __restrict__ dkey, uint32_t* s_key, uint32_t* s_val,
                                               uint32_t* s_cnt, uint32_t* s_red, const int wv, const int gtid) {
  const int lane = gtid & 63;
  // every wave owns a contiguous run of positions (whole rounds): position p = first + 64 j + lane
  const uint32_t per_wave = ((len + WAVES * 64u - 1u) / (WAVES * 64u)) * 64u;
  const int nrounds = (int)(per_wave >> 6);                               // <= ROUNDS, the same for every wave
  const uint32_t first = (uint32_t)wv * per_wave;
  uint32_t key[ROUNDS], val[ROUNDS], rnk[ROUNDS];
  uint32_t live = 0, mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
  for (int j = 0; j < ROUNDS; ++j) {
    key[j] = 0u; val[j] = 0u;
    if (j < nrounds) {
      const uint32_t p = first + 64u * j + lane;
      if (p < len) {
        val[j] = vals[p];
        live |= 1u << j;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < ROUNDS; ++j) {
    if ((live >> j) & 1u) {
      key[j] = dkey[val[j] & TDS_RANK_MASK];
      mn = min(mn, key[j]); mx = max(mx, key[j]);
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, 64));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  }
  if (WAVES > 1) {
    if (lane == 0) { s_red[2 * wv] = mn; s_red[2 * wv + 1] = mx; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { mn = min(mn, s_red[2 * w]); mx = max(mx, s_red[2 * w + 1]); }
  }
  if (mx <= mn) return;                                 // one key value (or an empty list): storage order is the order
  const uint32_t range = mx - mn;
  const int passes = (39 - __clz((int)range)) >> 3;     // ceil(bits(range) / 8), bits = 32 - clz
  volatile uint32_t* myc = s_cnt + 256 * wv;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 8 * pass;
    // ---- this wave's digit counters, then the stable rank of every element among the wave's equal digits -------------
#pragma unroll
    for (int i = 0; i < 4; ++i) myc[lane + 64 * i] = 0u;
    __builtin_amdgcn_wave_barrier();
    wave_rank_rounds<ROUNDS>([&](int j) { return ((key[j] - mn) >> shift) & 255u; }, live, nrounds, myc, rnk);
    tds_sync<WAVES>();
    // ---- counters -> start of (digit, wave) in the sorted list ---------------------------------------------------------
    if (WAVES == 1) {
      // lane l owns digits 4l .. 4l+3
      const uint32_t c0 = myc[4 * lane], c1 = myc[4 * lane + 1], c2 = myc[4 * lane + 2], c3 = myc[4 * lane + 3];
      const uint32_t s = c0 + c1 + c2 + c3;
      const uint32_t ex = wave_incl_scan_u32(s) - s;
      myc[4 * lane] = ex; myc[4 * lane + 1] = ex + c0; myc[4 * lane + 2] = ex + c0 + c1; myc[4 * lane + 3] = ex + c0 + c1 + c2;
    } else {
      // threads 0 .. 255 of the group own one digit each: its count in every wave, an exclusive scan over the digits
      uint32_t tot = 0;
      uint32_t c[WAVES];
      if (gtid < 256) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { c[w] = s_cnt[256 * w + gtid]; tot += c[w]; }
      }
      uint32_t inc = 0;
      if (gtid < 256) {
        inc = wave_incl_scan_u32(tot);
        if (lane == 63) s_red[2 * WAVES + (gtid >> 6)] = inc;           // (four words behind the min / max pairs)
      }
      __syncthreads();
      if (gtid < 256) {
        uint32_t base = inc - tot;
#pragma unroll
        for (int w = 0; w < 4; ++w) if (w < (gtid >> 6)) base += s_red[2 * WAVES + w];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { s_cnt[256 * w + gtid] = base; base += c[w]; }
      }
    }
    tds_sync<WAVES>();
    // ---- scatter into the LDS image, read back in position order -------------------------------------------------------
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
      if ((live >> j) & 1u) {
        const uint32_t p = myc[((key[j] - mn) >> shift) & 255u] + rnk[j];
        s_key[p] = key[j];
        s_val[p] = val[j];
      }
    }
    tds_sync<WAVES>();
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
      if ((live >> j) & 1u) {
        const uint32_t p = first + 64u * j + lane;
        key[j] = s_key[p];
        val[j] = s_val[p];
      }
    }
    tds_sync<WAVES>();                                   // the image and the counters are rewritten by the next pass
  }
#pragma unroll
  for (int j = 0; j < ROUNDS; ++j) {
    if ((live >> j) & 1u) vals[first + 64u * j + lane] = val[j];
  }
}

// Work items in schedule order (longest list first).  Item i < n_mid: the workgroup's eight waves sort tile sched[i]
// together (the huge ones among them are skipped here: k_tile_depth_sort_huge).  Items behind: eight tiles per item, one
// per wave.  A persistent grid (TDS_GRID workgroups, four per CU) loops over the items: no dispatch of thousands of
// workgroups that find nothing to do, and the long lists start first.
__global__ void __launch_bounds__(64 * TDS_WAVES, GSR_TDS_OCC) k_tile_depth_sort(TileSortArgs a) {
  __shared__ uint32_t s_key[TDS_BLOCK_CAP];
  __shared__ uint32_t s_val[TDS_BLOCK_CAP];
  __shared__ uint32_t s_cnt[TDS_WAVES * 256];
  __shared__ uint32_t s_red[2 * TDS_WAVES + 4];
  const uint32_t ntiles = (uint32_t)a.ntiles;
  const uint32_t n_mid = min(a.dv[DV_NMID], ntiles), n_huge = min(a.dv[DV_NHUGE], n_mid);
  const uint32_t n_items = n_mid + (ntiles - n_mid + TDS_WAVES - 1u) / TDS_WAVES;
  const int wv = (int)(threadIdx.x >> 6);
  for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    if (it < n_mid) {
      if (it >= n_huge) {
        const uint32_t tile = a.sched[it] & TDS_RANK_MASK;
        const uint2 rg = a.ranges[tile];
        tile_list_sort<TDS_WAVES, TDS_ROUNDS>(a.vals + rg.x, rg.y - rg.x, a.dkey, s_key, s_val, s_cnt, s_red, wv, (int)threadIdx.x);
      }
      __syncthreads();                                   // the LDS image is reused by this workgroup's next item
    } else {
      const uint32_t i = n_mid + (it - n_mid) * TDS_WAVES + (uint32_t)wv;
      if (i < ntiles) {
        const uint32_t tile = a.sched[i] & TDS_RANK_MASK;
        const uint2 rg = a.ranges[tile];
        const uint32_t len = rg.y - rg.x;
        if (len >= 2u)
          tile_list_sort<1, TDS_ROUNDS>(a.vals + rg.x, len, a.dkey, s_key + TDS_WAVE_CAP * wv, s_val + TDS_WAVE_CAP * wv,
                                        s_cnt + 256 * wv, s_red, 0, (int)(threadIdx.x & 63));
      }
      // (a wave-mode item touches only this wave's slices of the image: a workgroup barrier is needed only where a
      // block-mode item follows, and those all come first in the loop)
    }
  }
}

// (depth key, position-independent tie break = the value's Gaussian index) of element i, +inf beyond the list
__device__ __forceinline__ unsigned long long tds_composite(const uint32_t* vals, const uint32_t* dkey, uint32_t i, uint32_t len) {
  if (i >= len) return ~0ull;
  const uint32_t v = vals[i];
  return ((unsigned long long)dkey[v & TDS_RANK_MASK] << 32) | (v & TDS_RANK_MASK);
}

// The tiles with >= TDS_HUGE_MIN entries (the first DV_NHUGE of the schedule): a thin grid of 1024-thread workgroups loops
// over them.  Up to TDS_HUGE_CAP entries: the same LDS counting passes by sixteen waves.  Beyond: an all-ascending bitonic
// network (flip + half-cleaners, so positions beyond the list behave as +inf and are never touched) in global memory.
__global__ void __launch_bounds__(64 * TDS_HUGE_WAVES) k_tile_depth_sort_huge(TileSortArgs a) {
  __shared__ uint32_t s_key[TDS_HUGE_CAP];
  __shared__ uint32_t s_val[TDS_HUGE_CAP];
  __shared__ uint32_t s_cnt[TDS_HUGE_WAVES * 256];
  __shared__ uint32_t s_red[2 * TDS_HUGE_WAVES + 4];
  const uint32_t n_huge = min(a.dv[DV_NHUGE], (uint32_t)a.ntiles);
  const int wv = (int)(threadIdx.x >> 6);
  for (uint32_t it = blockIdx.x; it < n_huge; it += gridDim.x) {
    const uint32_t tile = a.sched[it] & TDS_RANK_MASK;
    const uint2 rg = a.ranges[tile];
    const uint32_t len = rg.y - rg.x;
    uint32_t* v = a.vals + rg.x;
    if (len <= (uint32_t)TDS_HUGE_CAP) {
      tile_list_sort<TDS_HUGE_WAVES, TDS_HUGE_ROUNDS>(v, len, a.dkey, s_key, s_val, s_cnt, s_red, wv, (int)threadIdx.x);
      __syncthreads();                                   // the LDS image is reused by the next tile of this workgroup
      continue;
    }
    uint32_t n2 = 1;
    while (n2 < len) n2 <<= 1;
    for (uint32_t k = 2; k <= n2; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
          // the t-th compare-exchange of this step: lo has bit j clear; its partner is the mirror image inside the block
          // of k in the first step of a merge (flip), lo + j afterwards (half-cleaner)
          const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
          const uint32_t hi = (j == (k >> 1)) ? (lo ^ (k - 1u)) : (lo | j);
          if (hi < len) {                                // lo < hi always; a partner beyond the list is +inf: no exchange
            const unsigned long long ka = tds_composite(v, a.dkey, lo, len), kb = tds_composite(v, a.dkey, hi, len);
            if (ka > kb) { const uint32_t x = v[lo]; v[lo] = v[hi]; v[hi] = x; }
          }
        }
        __syncthreads();
      }
    }
  }
}

}  // namespace gsr
