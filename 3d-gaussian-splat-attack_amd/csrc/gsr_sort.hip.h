// gsr_sort.hip.h -- wave64 scan, exclusive scan over u32 arrays and a stable LSD radix sort of
// (u32 key, u32 value) pairs, hand-written for gfx950.
//
// Replaces what the absent reference extension gets from a vendor scan/sort library (SURVEY.md section
// 2.2, K2/K4).  Design for MI355X rather than a translation: the depth order and the tile binning are
// two SEPARATE small-key sorts (32-bit depth over P Gaussians, then ceil(log2 T) bits of tile id over
// the N duplicated pairs) instead of one 64-bit (tile<<32|depth) sort over N pairs -- ~4.5x less HBM
// traffic -- and every pass ranks with wave64 ballots (no per-thread digit counters), stages its
// 4096-element chunk in LDS and writes digit runs back coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace gsr {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;   // 4096

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

// ---- exclusive scan over n u32 (two launches: block sums, then carry + local scan) ------------------
// The prefix sums are 32-bit (they number at most 2^31 pairs); next to them the exact 64-bit total is accumulated so
// that the host can tell a wrapped scan from a valid one.
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

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_partial(const uint32_t* __restrict__ in, uint32_t n,
                                                               uint32_t* __restrict__ sums,
                                                               unsigned long long* __restrict__ sums64) {
  __shared__ uint32_t tmp[4];
  __shared__ unsigned long long tmp64[4];
  const uint32_t base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  unsigned long long s64 = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const uint32_t v = (base + i < n) ? in[base + i] : 0u;
    s += v; s64 += v;
  }
  uint32_t total;
  block_excl_scan_256(s, tmp, total);
  const unsigned long long t64 = block_sum_u64(s64, tmp64);
  if (threadIdx.x == 0) { sums[blockIdx.x] = total; if (sums64) sums64[blockIdx.x] = t64; }
}

// Second and last launch of the scan: block b forms its own carry from the raw block sums of k_scan_partial (at most
// a few thousand words, L2 resident) instead of waiting for a single-block scan of them; the last block also writes
// the totals.  Two dependent launches per scan instead of three.
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply_carry(const uint32_t* in, uint32_t* out, uint32_t n,
                                                                   const uint32_t* __restrict__ sums, uint32_t nb,
                                                                   uint32_t* __restrict__ total_out,
                                                                   const unsigned long long* __restrict__ sums64,
                                                                   unsigned long long* __restrict__ total64_out) {
  __shared__ uint32_t tmp[4];
  __shared__ unsigned long long tmp64[4];
  const uint32_t b = blockIdx.x;
  uint32_t part = 0;
  for (uint32_t i = threadIdx.x; i < b; i += SCAN_THREADS) part += sums[i];
  uint32_t carry;
  block_excl_scan_256(part, tmp, carry);                 // carry = sum of the block sums in front of this block
  const uint32_t base = b * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
  uint32_t total;
  uint32_t run = block_excl_scan_256(s, tmp, total) + carry;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (b == nb - 1) {
    if (total_out && threadIdx.x == 0) *total_out = carry + total;
    if (total64_out) {
      unsigned long long a = 0;
      for (uint32_t i = threadIdx.x; i < nb; i += SCAN_THREADS) a += sums64[i];
      const unsigned long long t64 = block_sum_u64(a, tmp64);
      if (threadIdx.x == 0) *total64_out = t64;
    }
  }
}

// out may alias in.  sums must hold ceil(n/4096) words (and sums64 as many 64-bit words when total64_out is wanted).
// total_out (device) receives the grand total modulo 2^32, total64_out the exact one.
inline void scan_exclusive_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums, uint32_t* total_out,
                               hipStream_t st, unsigned long long* sums64 = nullptr,
                               unsigned long long* total64_out = nullptr) {
  if (n == 0) {
    if (total_out) (void)hipMemsetAsync(total_out, 0, sizeof(uint32_t), st);
    if (total64_out) (void)hipMemsetAsync(total64_out, 0, sizeof(unsigned long long), st);
    return;
  }
  const uint32_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  hipLaunchKernelGGL(k_scan_partial, dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, sums, total64_out ? sums64 : nullptr);
  hipLaunchKernelGGL(k_scan_apply_carry, dim3(nb), dim3(SCAN_THREADS), 0, st, in, out, n, (const uint32_t*)sums, nb,
                     total_out, total64_out ? (const unsigned long long*)sums64 : nullptr, total64_out);
}

// ---- stable LSD radix sort pass over (key,val) pairs ------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = 4;
// 64-element rounds per wave: 16 (4096 elements per block: long digit runs, the choice for tens of millions of keys) or
// 8 (2048 per block: twice the blocks -- a 1 M-key pass is latency-bound and 245 blocks do not even cover the 256 CUs)
constexpr int RS_ROUNDS_MAX = 16;
constexpr int RS_ROUNDS_MIN = 8;
constexpr int RS_BINS = 256;
constexpr int rs_chunk(int rounds) { return RS_WAVES * rounds * 64; }

template <int RS_ROUNDS>
__global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                           uint32_t mask, uint32_t* __restrict__ table, uint32_t nb) {
  constexpr int RS_CHUNK = rs_chunk(RS_ROUNDS);
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * RS_CHUNK;
#pragma unroll 4
  for (int i = threadIdx.x; i < RS_CHUNK; i += RS_THREADS) {
    const uint32_t idx = base + i;
    if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & mask], 1u);
  }
  __syncthreads();
  table[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];   // digit-major: one linear scan gives global bases
}

// One block per digit: exclusive scan of that digit's row of per-block counts, in place; the row total goes to
// rowsum[digit].  (Replaces a generic three-launch scan of the whole table: the scatter kernel adds the digit bases
// itself from the 256 row totals.)
__global__ void __launch_bounds__(RS_THREADS) k_radix_rowscan(uint32_t* __restrict__ table, uint32_t nb,
                                                              uint32_t* __restrict__ rowsum) {
  __shared__ uint32_t tmp[4];
  uint32_t* row = table + (size_t)blockIdx.x * nb;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += RS_THREADS) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = (i < nb) ? row[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan_256(v, tmp, total);
    if (i < nb) row[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) rowsum[blockIdx.x] = carry;
}

// table: per-digit exclusive-scanned rows [256][nb] + rowsum[256] (k_radix_rowscan).  iota != 0: values are the
// element indices (first pass of an argsort).
template <int RS_ROUNDS>
__global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                              uint32_t mask, const uint32_t* __restrict__ table,
                                                              const uint32_t* __restrict__ rowsum, uint32_t nb,
                                                              int iota) {
  constexpr int RS_CHUNK = rs_chunk(RS_ROUNDS);
  __shared__ uint32_t wcnt[RS_WAVES][RS_BINS];
  __shared__ uint32_t gbase[RS_BINS];
  __shared__ uint32_t tmp[4];
  __shared__ uint32_t skey[RS_CHUNK];
  __shared__ uint32_t sval[RS_CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t base = blockIdx.x * RS_CHUNK + w * (RS_ROUNDS * 64);
#pragma unroll
  for (int i = 0; i < RS_WAVES; ++i) wcnt[i][tid] = 0;
  __syncthreads();

  uint32_t key[RS_ROUNDS], val[RS_ROUNDS], rnk[RS_ROUNDS];
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const uint32_t idx = base + j * 64 + lane;
    const bool ok = idx < n;
    key[j] = ok ? keys_in[idx] : 0xFFFFFFFFu;     // padding sorts last inside its digit and is never written
    val[j] = ok ? (iota ? idx : vals_in[idx]) : 0u;
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  volatile uint32_t* myc = wcnt[w];
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const uint32_t d = (key[j] >> shift) & mask;
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t before = __popcll(peers & lt);
    const uint32_t pre = myc[d];
    __builtin_amdgcn_wave_barrier();
    if (before == 0) myc[d] = pre + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rnk[j] = pre + before;
  }
  __syncthreads();
  {
    // thread = digit: start of this digit in the block-local order, then per-wave bases
    const uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
    uint32_t total;
    const uint32_t ds = block_excl_scan_256(c0 + c1 + c2 + c3, tmp, total);
    wcnt[0][tid] = ds;
    wcnt[1][tid] = ds + c0;
    wcnt[2][tid] = ds + c0 + c1;
    wcnt[3][tid] = ds + c0 + c1 + c2;
    uint32_t all;
    const uint32_t dbase = block_excl_scan_256(rowsum[tid], tmp, all);   // elements with a smaller digit, globally
    gbase[tid] = dbase + table[tid * nb + blockIdx.x] - ds;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const uint32_t d = (key[j] >> shift) & mask;
    const uint32_t p = wcnt[w][d] + rnk[j];
    skey[p] = key[j];
    sval[p] = val[j];
  }
  __syncthreads();
  const uint32_t blk0 = blockIdx.x * RS_CHUNK;
  const uint32_t nvalid = (n - blk0 < (uint32_t)RS_CHUNK) ? (n - blk0) : (uint32_t)RS_CHUNK;
  for (uint32_t p = tid; p < nvalid; p += RS_THREADS) {
    const uint32_t k = skey[p];
    const uint32_t g = gbase[(k >> shift) & mask] + p;
    keys_out[g] = k;
    vals_out[g] = sval[p];
  }
}

inline uint32_t radix_table_words(uint32_t n) { return RS_BINS * ((n + rs_chunk(RS_ROUNDS_MIN) - 1) / rs_chunk(RS_ROUNDS_MIN)); }

inline int radix_rounds_for(uint32_t n) {
  static const int env = [] { const char* e = getenv("GSR_RS_ROUNDS"); int v = e ? atoi(e) : 0; return (v == 8 || v == 16) ? v : 0; }();
  if (env) return env;
  return n <= (2u << 20) ? RS_ROUNDS_MIN : RS_ROUNDS_MAX;     // measured on MI355X: 1 M keys 20 vs 22 us per pass, 3.2 M keys 40 vs 38
}

// Sorts on key bits [begin_bit, end_bit).  Buffers ping-pong; returns 0 if the result is in (k0,v0), 1 if in
// (k1,v1).  iota_first: the values of the first pass are the element indices (v0 is then never read).
// table: radix_table_words(n) words; sums: at least 256 words.
inline int radix_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t n, int begin_bit,
                            int end_bit, bool iota_first, uint32_t* table, uint32_t* sums, hipStream_t st) {
  if (n == 0 || end_bit <= begin_bit) return 0;
  const int bits = end_bit - begin_bit;
  const int passes = (bits + 7) / 8;
  const int rounds = radix_rounds_for(n);
  const uint32_t chunk = (uint32_t)rs_chunk(rounds);
  const uint32_t nb = (n + chunk - 1) / chunk;
  int cur = 0, bit = begin_bit;
  for (int p = 0; p < passes; ++p) {
    // spread the bits evenly over the passes (13 bits -> 7 + 6): longer digit runs per block
    const int w = (bits - (bit - begin_bit) + (passes - p) - 1) / (passes - p);
    const uint32_t mask = (1u << w) - 1u;
    uint32_t* ki = cur ? k1 : k0; uint32_t* vi = cur ? v1 : v0;
    uint32_t* ko = cur ? k0 : k1; uint32_t* vo = cur ? v0 : v1;
    const int iota = (iota_first && p == 0) ? 1 : 0;
    if (rounds == RS_ROUNDS_MIN) {
      hipLaunchKernelGGL((k_radix_hist<RS_ROUNDS_MIN>), dim3(nb), dim3(RS_THREADS), 0, st, ki, n, bit, mask, table, nb);
      hipLaunchKernelGGL(k_radix_rowscan, dim3(RS_BINS), dim3(RS_THREADS), 0, st, table, nb, sums);
      hipLaunchKernelGGL((k_radix_scatter<RS_ROUNDS_MIN>), dim3(nb), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, n, bit, mask,
                         table, sums, nb, iota);
    } else {
      hipLaunchKernelGGL((k_radix_hist<RS_ROUNDS_MAX>), dim3(nb), dim3(RS_THREADS), 0, st, ki, n, bit, mask, table, nb);
      hipLaunchKernelGGL(k_radix_rowscan, dim3(RS_BINS), dim3(RS_THREADS), 0, st, table, nb, sums);
      hipLaunchKernelGGL((k_radix_scatter<RS_ROUNDS_MAX>), dim3(nb), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, n, bit, mask,
                         table, sums, nb, iota);
    }
    cur ^= 1;
    bit += w;
  }
  return cur;
}

}  // namespace gsr
