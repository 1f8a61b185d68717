// gsr_sort.hip.h -- wave64 scan, exclusive scan over u32 arrays and a stable LSD radix sort of
// (u32 key, u32 value) pairs, hand-written for gfx950.
//
// Replaces what the absent reference extension gets from a vendor scan/sort library (SURVEY.md section
// 2.2, K2/K4).  Design for MI355X rather than a translation: the depth order and the tile binning are
// two SEPARATE small-key sorts (32-bit depth over the visible Gaussians, then ceil(log2 T) bits of tile id over
// the N duplicated pairs) instead of one 64-bit (tile<<32|depth) sort over N pairs -- ~4.5x less HBM
// traffic -- and every pass ranks with wave64 ballots (no per-thread digit counters), stages its
// chunk in LDS and writes digit runs back coalesced.
//
// Round 3: element counts may live on the DEVICE (`n_dev`): the forward never waits for the host to learn how many
// Gaussians survive the culls or how many pairs they emit.  Grids are sized from a host-side upper bound and blocks
// beyond the live count exit.  The depth sort's digit layout is chosen on the device too (DigitSpec::dv): the keys'
// range (max - min of the live depth bits) decides a digit width w = ceil(bits / 3) <= 11, so THREE passes always
// cover the key (27 significant bits on the benchmark scene -> 9-bit digits) where fixed 8-bit digits need four.
// Pass 0 also compacts: keys equal to RS_DROP_KEY (culled Gaussians) are neither counted nor written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace gsr {

constexpr int SCAN_THREADS = 256;

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread block; returns the block total in `total`.
// `tmp` is 4 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* tmp, uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan_u32(v);
  if (lane == 63) tmp[w] = incl;
  __syncthreads();
  const uint32_t t0 = tmp[0], t1 = tmp[1], t2 = tmp[2], t3 = tmp[3];
  total = t0 + t1 + t2 + t3;
  const uint32_t wbase = (w > 0 ? t0 : 0u) + (w > 1 ? t1 : 0u) + (w > 2 ? t2 : 0u);
  __syncthreads();
  return wbase + incl - v;
}

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* tmp64) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  if (lane == 0) tmp64[w] = v;
  __syncthreads();
  const unsigned long long t = tmp64[0] + tmp64[1] + tmp64[2] + tmp64[3];
  __syncthreads();
  return t;
}

__device__ __forceinline__ uint32_t block_min_u32(uint32_t v, uint32_t* tmp) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
  if (lane == 0) tmp[w] = v;
  __syncthreads();
  const uint32_t t = min(min(tmp[0], tmp[1]), min(tmp[2], tmp[3]));
  __syncthreads();
  return t;
}
__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, uint32_t* tmp) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
  if (lane == 0) tmp[w] = v;
  __syncthreads();
  const uint32_t t = max(max(tmp[0], tmp[1]), max(tmp[2], tmp[3]));
  __syncthreads();
  return t;
}

// ---- device-side scalars of one forward ("dv" block, 16 words, lives in the context's kept slab) --------------------
enum {
  DV_N = 0,       // number of (tile, Gaussian) pairs the tile sort / compositors see (0 when the capacity overflowed)
  DV_V = 1,       // Gaussians that emit at least one pair (= live keys of the depth sort)
  DV_KMIN = 2,    // smallest live depth key
  DV_W = 3,       // digit width of the three depth passes
  DV_OVF = 4,     // 1: more pairs than the caller's capacity (asynchronous pair count only)
  DV_NREC = 5,    // boundary records in use (k_tile_schedule)
  DV_N64 = 6,     // words 6,7: exact 64-bit pair count
  DV_WORDS = 16
};

// ITEMS consecutive words of a thread, as 16-byte loads when the array allows it (the scans' threads own ITEMS consecutive
// elements each: sixteen 4-byte loads at a 64-byte stride between lanes cost four times the requests of four 16-byte ones)
template <int ITEMS>
__device__ __forceinline__ void load_items_u32(const uint32_t* __restrict__ in, uint32_t base, uint32_t n, uint32_t v[ITEMS]) {
  static_assert(ITEMS % 4 == 0, "whole 16-byte words");
  if (base + ITEMS <= n && (reinterpret_cast<uintptr_t>(in + base) & 15u) == 0u) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int i = 0; i < ITEMS / 4; ++i) { const uint4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) v[i] = (base + i < n) ? in[base + i] : 0u;
  }
}

// ---- exclusive scan over n u32 (two launches: block sums, then carry + local scan) ------------------
// ITEMS per thread: 8 (2048 per block) or 16 (4096).  n_dev != nullptr: the live count is min(n, *n_dev).
template <int ITEMS>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_partial(const uint32_t* __restrict__ in, uint32_t n,
                                                               const uint32_t* __restrict__ n_dev,
                                                               uint32_t* __restrict__ sums) {
  constexpr int CHUNK = SCAN_THREADS * ITEMS;
  __shared__ uint32_t tmp[4];
  if (n_dev) n = min(n, *n_dev);
  const uint32_t base = blockIdx.x * CHUNK + threadIdx.x * ITEMS;
  if (blockIdx.x * CHUNK >= n) return;
  uint32_t s = 0, v[ITEMS];
  load_items_u32<ITEMS>(in, base, n, v);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) s += v[i];
  uint32_t total;
  block_excl_scan_256(s, tmp, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// Second and last launch of the scan: block b forms its own carry from the raw block sums of k_scan_partial (at most
// a few thousand words, L2 resident) instead of waiting for a single-block scan of them; the last live block also
// writes the total to out[n_live].
// chunk_first != nullptr: the scanned values are lengths of consecutive runs of output slots (element r owns slots
// [out[r], out[r] + in[r])), and chunk_first[c] receives the element that owns slot c * chunk_len, for every chunk
// that starts inside the total -- what a slot-parallel consumer (k_emit) would otherwise find by a binary search of `out`
// in global memory, twenty dependent loads per workgroup before its first useful instruction.
template <int ITEMS>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply_carry(const uint32_t* in, uint32_t* out, uint32_t n,
                                                                   const uint32_t* __restrict__ n_dev,
                                                                   const uint32_t* __restrict__ sums,
                                                                   uint32_t* __restrict__ chunk_first, uint32_t chunk_len,
                                                                   uint32_t chunk_cap) {
  constexpr int CHUNK = SCAN_THREADS * ITEMS;
  __shared__ uint32_t tmp[4];
  if (n_dev) n = min(n, *n_dev);
  const uint32_t b = blockIdx.x;
  if (n == 0) { if (b == 0 && threadIdx.x == 0) out[0] = 0u; return; }
  if (b * CHUNK >= n) return;
  const uint32_t nb = (n + CHUNK - 1) / CHUNK;
  uint32_t part = 0;
  for (uint32_t i = threadIdx.x; i < b; i += SCAN_THREADS) part += sums[i];
  uint32_t carry;
  block_excl_scan_256(part, tmp, carry);                 // carry = sum of the block sums in front of this block
  const uint32_t base = b * CHUNK + threadIdx.x * ITEMS;
  uint32_t v[ITEMS];
  uint32_t s = 0;
  load_items_u32<ITEMS>(in, base, n, v);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) s += v[i];
  uint32_t total;
  uint32_t run = block_excl_scan_256(s, tmp, total) + carry;
  uint32_t r[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    r[i] = run;
    if (chunk_first && v[i] != 0u) {
      // chunk starts c * chunk_len inside [run, run + v): usually none or one
      for (uint32_t c = (run + chunk_len - 1u) / chunk_len;
           c < chunk_cap && (unsigned long long)c * chunk_len < (unsigned long long)run + v[i]; ++c)
        chunk_first[c] = base + i;
    }
    run += v[i];
  }
  if (base + ITEMS <= n && (reinterpret_cast<uintptr_t>(out + base) & 15u) == 0u) {
    uint4* o4 = reinterpret_cast<uint4*>(out + base);
#pragma unroll
    for (int i = 0; i < ITEMS / 4; ++i) o4[i] = make_uint4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) if (base + i < n) out[base + i] = r[i];
  }
  if (b == nb - 1 && threadIdx.x == 0) out[n] = carry + total;
}

// out may alias in; out holds n + 1 words (out[n_live] = total).  sums: ceil(n / 2048) words.
inline void scan_exclusive_u32(const uint32_t* in, uint32_t* out, uint32_t n, const uint32_t* n_dev, uint32_t* sums,
                               hipStream_t st, uint32_t* chunk_first = nullptr, uint32_t chunk_len = 1,
                               uint32_t chunk_cap = 0) {
  if (n == 0) { (void)hipMemsetAsync(out, 0, sizeof(uint32_t), st); return; }
  if (n <= (2u << 20)) {
    const uint32_t nb = (n + 2047) / 2048;
    hipLaunchKernelGGL((k_scan_partial<8>), dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, n_dev, sums);
    hipLaunchKernelGGL((k_scan_apply_carry<8>), dim3(nb), dim3(SCAN_THREADS), 0, st, in, out, n, n_dev, (const uint32_t*)sums,
                       chunk_first, chunk_len, chunk_cap);
  } else {
    const uint32_t nb = (n + 4095) / 4096;
    hipLaunchKernelGGL((k_scan_partial<16>), dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, n_dev, sums);
    hipLaunchKernelGGL((k_scan_apply_carry<16>), dim3(nb), dim3(SCAN_THREADS), 0, st, in, out, n, n_dev, (const uint32_t*)sums,
                       chunk_first, chunk_len, chunk_cap);
  }
}

// ---- stable LSD radix sort pass over (key,val) pairs ------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = 4;
// 64-element rounds per wave: 16 (4096 elements per block: long digit runs, the choice for tens of millions of keys) or
// 8 (2048 per block: twice the blocks -- a 1 M-key pass is latency-bound and 245 blocks do not even cover the 256 CUs)
constexpr int RS_ROUNDS_MAX = 16;
constexpr int RS_ROUNDS_MIN = 8;
constexpr int RS_BINS = 256;            // bins of a host-specified digit (tile sort, test hooks): at most 8 bits
constexpr int RS_BINS_DEV = 2048;       // bins of a device-specified digit (depth sort): at most 11 bits
constexpr uint32_t RS_DROP_KEY = 0xFFFFFFFFu;
constexpr int rs_chunk(int rounds) { return RS_WAVES * rounds * 64; }

// Which bits of the key a pass sorts on.  dv == nullptr: (shift, mask) as given by the host, nothing subtracted.
// dv != nullptr (depth sort): digit width w = dv[DV_W]; pass p sorts bits [p w, (p+1) w) of key - sub, where
// sub = dv[DV_KMIN] in pass 0 (which stores the reduced keys) and 0 afterwards.
struct DigitSpec {
  const uint32_t* dv;
  int pass;
  int shift;
  uint32_t mask;
};
__device__ __forceinline__ void resolve_digit(const DigitSpec& s, int& shift, uint32_t& mask, uint32_t& sub, int& w) {
  if (s.dv) {
    w = (int)s.dv[DV_W];
    shift = s.pass * w;
    mask = (1u << w) - 1u;
    sub = s.pass == 0 ? s.dv[DV_KMIN] : 0u;
  } else {
    shift = s.shift; mask = s.mask; sub = 0u;
    w = 32 - __clz(mask);
  }
}

// The depth keys' digit width from their range: `npass` passes of w bits cover max - min (three passes: w <= 11, the
// 2048-bin kernels; four passes: w <= 8, the 256-bin kernels -- what a sort of several million keys runs faster with:
// longer digit runs per block, a third of the LDS, seven waves per SIMD instead of two).
__device__ __forceinline__ uint32_t depth_digit_width(uint32_t kmin, uint32_t kmax, uint32_t npass = 3u) {
  const uint32_t bits = kmax > kmin ? 32u - (uint32_t)__clz(kmax - kmin) : 1u;
  return max(4u, (bits + npass - 1u) / npass);   // at least 16 bins so that tiny ranges stay sane
}

template <int RS_ROUNDS, int BINS>
__global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n,
                                                           const uint32_t* __restrict__ n_dev, DigitSpec ds,
                                                           uint32_t* __restrict__ table, uint32_t nb) {
  constexpr int RS_CHUNK = rs_chunk(RS_ROUNDS);
  constexpr int PER = RS_CHUNK / RS_THREADS;
  __shared__ uint32_t h[BINS];
  const uint32_t base = blockIdx.x * RS_CHUNK;
  // the keys are requested before the live count is known (both come from memory: one round trip instead of two);
  // positions below the host-side bound n are always readable
  uint32_t k[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const uint32_t idx = base + i * RS_THREADS + threadIdx.x;
    k[i] = idx < n ? keys[idx] : 0u;
  }
  if (n_dev) n = min(n, *n_dev);
  if (base >= n) return;                               // dead block: k_radix_rowscan never reads its column
  int shift, w; uint32_t mask, sub;
  resolve_digit(ds, shift, mask, sub, w);
  const uint32_t nbins = mask + 1u;
  for (uint32_t d = threadIdx.x; d < nbins; d += RS_THREADS) h[d] = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const uint32_t idx = base + i * RS_THREADS + threadIdx.x;
    if (idx < n) atomicAdd(&h[((k[i] - sub) >> shift) & mask], 1u);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < nbins; d += RS_THREADS) table[d * nb + blockIdx.x] = h[d];   // digit-major
}

// One WAVE per digit: exclusive scan of that digit's row of per-block counts, in place; the row total goes to
// rowsum[digit].  Only the columns of live blocks are touched.  A row is a few hundred words: its loads are all in
// flight before the first add, and there is no workgroup barrier anywhere.  Launch: ceil(bins / 4) blocks of 256.
constexpr int RSCAN_BATCH = 8;     // 64-word pieces of a row in flight per wave
__global__ void __launch_bounds__(RS_THREADS) k_radix_rowscan(uint32_t* __restrict__ table, uint32_t nb, uint32_t n,
                                                              const uint32_t* __restrict__ n_dev, uint32_t chunk,
                                                              DigitSpec ds, uint32_t* __restrict__ rowsum) {
  if (n_dev) n = min(n, *n_dev);
  int shift, w; uint32_t mask, sub;
  resolve_digit(ds, shift, mask, sub, w);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t d = blockIdx.x * RS_WAVES + (threadIdx.x >> 6);
  if (d > mask) return;
  const uint32_t nlive = (n + chunk - 1u) / chunk;
  uint32_t* row = table + (size_t)d * nb;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nlive; base += 64u * RSCAN_BATCH) {
    uint32_t v[RSCAN_BATCH];
#pragma unroll
    for (int k = 0; k < RSCAN_BATCH; ++k) {
      const uint32_t i = base + 64u * k + lane;
      v[k] = i < nlive ? row[i] : 0u;
    }
#pragma unroll
    for (int k = 0; k < RSCAN_BATCH; ++k) {
      const uint32_t i = base + 64u * k + lane;
      const uint32_t inc = wave_incl_scan_u32(v[k]);
      if (i < nlive) row[i] = carry + inc - v[k];
      carry += (uint32_t)__shfl((int)inc, 63, 64);
    }
  }
  if (lane == 0) rowsum[d] = carry;
}

// table: per-digit exclusive-scanned rows [bins][nb] + rowsum[bins] (k_radix_rowscan).
// iota != 0: values are the element indices (first pass of an argsort).
// drop != 0: keys equal to RS_DROP_KEY are not ranked and not written -- the output is the compacted list, whose
//            length block 0 writes to *count_out.
// keys_out may be null (the last pass of an argsort needs no keys); aux_src/aux_out: aux_out[g] = aux_src[val].
template <int RS_ROUNDS, int BINS>
__global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, uint32_t n,
                                                              const uint32_t* __restrict__ n_dev, DigitSpec ds,
                                                              const uint32_t* __restrict__ table,
                                                              const uint32_t* __restrict__ rowsum, uint32_t nb,
                                                              int iota, int drop, uint32_t* __restrict__ count_out,
                                                              const uint32_t* __restrict__ aux_src,
                                                              uint32_t* __restrict__ aux_out,
                                                              uint32_t* __restrict__ key_ranges, uint32_t key_limit,
                                                              const uint8_t* __restrict__ aux_src8 = nullptr) {
  constexpr int RS_CHUNK = rs_chunk(RS_ROUNDS);
  constexpr int MAXW = BINS == 256 ? 8 : (BINS == 512 ? 9 : (BINS == 1024 ? 10 : 11));
  static_assert(BINS == (1 << MAXW), "BINS is a power of two between 256 and 2048");
  __shared__ uint32_t wcnt[RS_WAVES][BINS];
  __shared__ uint32_t gbase[BINS];
  __shared__ uint32_t tmp[4];
  __shared__ uint32_t skey[RS_CHUNK];
  __shared__ uint32_t sval[RS_CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t blk0 = blockIdx.x * RS_CHUNK;
  const uint32_t base = blk0 + wv * (RS_ROUNDS * 64);
  // keys and values are requested before the live count is known (one memory round trip instead of two); positions
  // below the host-side bound n are always readable
  uint32_t key[RS_ROUNDS], val[RS_ROUNDS], rnk[RS_ROUNDS];
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const uint32_t idx = base + j * 64 + lane;
    const bool ok = idx < n;
    key[j] = ok ? keys_in[idx] : RS_DROP_KEY;
    val[j] = iota ? idx : (ok ? vals_in[idx] : 0u);
  }
  if (n_dev) n = min(n, *n_dev);
  if (blk0 >= n) {
    if (n == 0 && blockIdx.x == 0 && tid == 0 && count_out) *count_out = 0u;
    return;
  }
  int shift, w; uint32_t mask, sub;
  resolve_digit(ds, shift, mask, sub, w);
  const uint32_t nbins = mask + 1u;
  // This block's column of the scanned table and the row totals (digit d = tid + 256 i) are requested now and used
  // after the ranking loop: they used to be fetched one dependent round trip at a time between the workgroup barriers
  // of the digit phase.
  constexpr int DPT = BINS / RS_THREADS;             // digits per thread
  uint32_t tcol[DPT], rsum[DPT];
#pragma unroll
  for (int i = 0; i < DPT; ++i) {
    const uint32_t d = (uint32_t)tid + (uint32_t)i * RS_THREADS;
    const bool in = d < nbins;
    tcol[i] = in ? table[d * nb + blockIdx.x] : 0u;
    rsum[i] = in ? rowsum[d] : 0u;
  }
  for (uint32_t d = tid; d < nbins; d += RS_THREADS) {
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) wcnt[i][d] = 0;
  }
  __syncthreads();

  uint32_t live = 0;                                 // bit j: element j of this thread takes part
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const uint32_t idx = base + j * 64 + lane;
    const bool keep = idx < n && !(drop && key[j] == RS_DROP_KEY);
    key[j] -= sub;
    live |= (keep ? 1u : 0u) << j;
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  volatile uint32_t* myc = wcnt[wv];
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const bool keep = (live >> j) & 1u;
    const uint32_t d = (key[j] >> shift) & mask;
    uint64_t peers = __ballot(keep);
    // all the bits a digit of this kernel can have, unrolled (bits at and above the digit's width are 0 in every lane
    // and leave `peers` as it is): a loop over the actual width costs more than the two or three idle ballots
#pragma unroll
    for (int b = 0; b < MAXW; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t before = __popcll(peers & lt);
    const uint32_t pre = keep ? myc[d] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (keep && before == 0) myc[d] = pre + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rnk[j] = pre + before;
  }
  __syncthreads();
  uint32_t nvalid = 0;
  {
    // thread = digit (BINS / 256 digits per thread, 256 at a time): start of the digit in the block-local order,
    // per-wave bases, and the digit's global base
    uint32_t carry_l = 0, carry_g = 0;
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const uint32_t d = (uint32_t)tid + (uint32_t)i * RS_THREADS;
      if ((uint32_t)i * RS_THREADS >= nbins) break;      // uniform
      const bool in = d < nbins;
      const uint32_t c0 = in ? wcnt[0][d] : 0u, c1 = in ? wcnt[1][d] : 0u, c2 = in ? wcnt[2][d] : 0u, c3 = in ? wcnt[3][d] : 0u;
      uint32_t total, all;
      const uint32_t dsl = carry_l + block_excl_scan_256(c0 + c1 + c2 + c3, tmp, total);
      const uint32_t dbase = carry_g + block_excl_scan_256(in ? rsum[i] : 0u, tmp, all);   // smaller digits, globally
      if (in) {
        wcnt[0][d] = dsl;
        wcnt[1][d] = dsl + c0;
        wcnt[2][d] = dsl + c0 + c1;
        wcnt[3][d] = dsl + c0 + c1 + c2;
        gbase[d] = dbase + tcol[i] - dsl;
      }
      carry_l += total; carry_g += all;
    }
    nvalid = carry_l;
    if (count_out && blockIdx.x == 0 && tid == 0) *count_out = carry_g;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    if ((live >> j) & 1u) {
      const uint32_t d = (key[j] >> shift) & mask;
      const uint32_t p = wcnt[wv][d] + rnk[j];
      skey[p] = key[j];
      sval[p] = val[j];
    }
  }
  __syncthreads();
  for (uint32_t p = tid; p < nvalid; p += RS_THREADS) {
    const uint32_t k = skey[p];
    const uint32_t g = gbase[(k >> shift) & mask] + p;
    const uint32_t v = sval[p];
    if (keys_out) keys_out[g] = k;
    vals_out[g] = v;
    if (aux_out) {
      // aux_src8 (optional): the same numbers saturated at 255 in a byte array -- a random gather over millions of elements
      // then touches an eighth of the sectors (8 MB instead of 32 MB for a batch of eight 1 M-Gaussian views: most of it
      // stays in the XCDs' L2s); 255 = look the number up in the 32-bit array
      uint32_t ax;
      if (aux_src8) { ax = aux_src8[v]; if (ax == 255u) ax = aux_src[v]; }
      else ax = aux_src[v];
      aux_out[g] = ax;
    }
    if (key_ranges && k < key_limit) {
      // LAST pass of a sort on the whole key: equal keys end up contiguous, [key_ranges[2k], key_ranges[2k+1]) is key
      // k's span of the output.  A run of equal keys inside this block's sorted chunk is a piece of that span (pieces of
      // other blocks lie in front of or behind it): its first element lowers the start, its last raises the end.  The
      // caller initialises every span to (0xFFFFFFFF, 0).  (A separate pass over the sorted keys did this before.)
      if (p == 0 || skey[p - 1] != k) atomicMin(&key_ranges[2 * k], g);
      if (p + 1 == nvalid || skey[p + 1] != k) atomicMax(&key_ranges[2 * k + 1], g + 1u);
    }
  }
}

inline uint32_t radix_table_words(uint32_t n, uint32_t bins = RS_BINS) {
  return bins * ((n + rs_chunk(RS_ROUNDS_MIN) - 1) / rs_chunk(RS_ROUNDS_MIN));
}

inline int radix_rounds_for(uint32_t n) {
  static const int env = [] { const char* e = getenv("GSR_RS_ROUNDS"); int v = e ? atoi(e) : 0; return (v == 8 || v == 16) ? v : 0; }();
  if (env) return env;
  // measured on MI355X: 1 M keys 20 vs 22 us per pass, 3.2 M keys 40 vs 38 (round 2); the smaller chunk also means 256-thread
  // workgroups of 21 KB LDS, which find room beside other streams' compositing waves sooner than 37 KB ones
  return n <= (4u << 20) ? RS_ROUNDS_MIN : RS_ROUNDS_MAX;
}

// One pass with a host-specified digit, all three launches (hist may be skipped when the caller has filled `table`
// itself -- k_emit does for the first tile pass).
template <int BINS>
inline void radix_pass(const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, uint32_t n,
                       const uint32_t* n_dev, const DigitSpec& ds, int rounds, uint32_t* table, uint32_t* sums, bool hist,
                       int iota, int drop, uint32_t* count_out, const uint32_t* aux_src, uint32_t* aux_out,
                       hipStream_t st, uint32_t* key_ranges = nullptr, uint32_t key_limit = 0, const uint8_t* aux_src8 = nullptr) {
  const uint32_t chunk = (uint32_t)rs_chunk(rounds);
  const uint32_t nb = (n + chunk - 1) / chunk;
  if (rounds == RS_ROUNDS_MIN) {
    if (hist) hipLaunchKernelGGL((k_radix_hist<RS_ROUNDS_MIN, BINS>), dim3(nb), dim3(RS_THREADS), 0, st, ki, n, n_dev, ds, table, nb);
    hipLaunchKernelGGL(k_radix_rowscan, dim3(BINS / RS_WAVES), dim3(RS_THREADS), 0, st, table, nb, n, n_dev, chunk, ds, sums);
    hipLaunchKernelGGL((k_radix_scatter<RS_ROUNDS_MIN, BINS>), dim3(nb), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, n, n_dev, ds,
                       (const uint32_t*)table, (const uint32_t*)sums, nb, iota, drop, count_out, aux_src, aux_out,
                       key_ranges, key_limit, aux_src8);
  } else {
    if (hist) hipLaunchKernelGGL((k_radix_hist<RS_ROUNDS_MAX, BINS>), dim3(nb), dim3(RS_THREADS), 0, st, ki, n, n_dev, ds, table, nb);
    hipLaunchKernelGGL(k_radix_rowscan, dim3(BINS / RS_WAVES), dim3(RS_THREADS), 0, st, table, nb, n, n_dev, chunk, ds, sums);
    hipLaunchKernelGGL((k_radix_scatter<RS_ROUNDS_MAX, BINS>), dim3(nb), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, n, n_dev, ds,
                       (const uint32_t*)table, (const uint32_t*)sums, nb, iota, drop, count_out, aux_src, aux_out,
                       key_ranges, key_limit, aux_src8);
  }
}

// Sorts on key bits [begin_bit, end_bit) with host-specified 8-bit (or narrower) digits.  Buffers ping-pong; returns
// 0 if the result is in (k0,v0), 1 if in (k1,v1).  iota_first: the values of the first pass are the element indices
// (v0 is then never read).  first_hist_done: `table` already holds the first pass's per-block histogram.
// key_ranges != nullptr (and begin_bit == 0, end_bit covering every key below key_limit): the last pass also leaves the
// span [start, end) of every key value below key_limit in key_ranges[2 key .. 2 key + 1], which the caller has set to
// (0xFFFFFFFF, 0).
// table: radix_table_words(n) words; sums: at least 256 words.
inline int radix_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t n, const uint32_t* n_dev,
                            int begin_bit, int end_bit, bool iota_first, uint32_t* table, uint32_t* sums, hipStream_t st,
                            bool first_hist_done = false, uint32_t* key_ranges = nullptr, uint32_t key_limit = 0) {
  if (n == 0 || end_bit <= begin_bit) return 0;
  const int bits = end_bit - begin_bit;
  const int passes = (bits + 7) / 8;
  const int rounds = radix_rounds_for(n);
  int cur = 0, bit = begin_bit;
  for (int p = 0; p < passes; ++p) {
    // spread the bits evenly over the passes (13 bits -> 7 + 6): longer digit runs per block
    const int w = (bits - (bit - begin_bit) + (passes - p) - 1) / (passes - p);
    DigitSpec ds{nullptr, p, bit, (1u << w) - 1u};
    uint32_t* ki = cur ? k1 : k0; uint32_t* vi = cur ? v1 : v0;
    uint32_t* ko = cur ? k0 : k1; uint32_t* vo = cur ? v0 : v1;
    const bool last = p == passes - 1;
    radix_pass<RS_BINS>(ki, vi, ko, vo, n, n_dev, ds, rounds, table, sums, !(p == 0 && first_hist_done),
                        (iota_first && p == 0) ? 1 : 0, 0, nullptr, nullptr, nullptr, st, last ? key_ranges : nullptr,
                        key_limit);
    cur ^= 1;
    bit += w;
  }
  return cur;
}

// first-pass digit of radix_sort_pairs(begin_bit = 0): what a caller that fills `table` itself must count
inline void radix_first_digit(int end_bit, int& shift, uint32_t& mask) {
  const int passes = (end_bit + 7) / 8;
  const int w = (end_bit + passes - 1) / passes;
  shift = 0; mask = (1u << w) - 1u;
}

}  // namespace gsr
