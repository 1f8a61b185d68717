// gsr_pgd.hip.h -- the projected-gradient update rules on one raw attribute tensor [rows, cols] (SURVEY.md section
// 8a row a11; reference attack.py:25-173), fused into two launches per tensor instead of the ~15 elementwise /
// reduction passes the PyTorch formulation makes over the same 236 MB:
//   L-inf:  x += -alpha * sign(g);  x = clamp(x - x0, -eps, eps) + x0                     (attack.py:25-51, 121-136)
//   L2:     x += -alpha * g / ||g||_2 (norm over the WHOLE tensor; no step if it is 0);
//           d = x - x0, each ROW clipped to the eps ball: d *= eps / (||d_row|| + 1e-7) when ||d_row|| > eps
//           (torch.renorm(p=2, dim=0, maxnorm=eps));  x = x0 + d                          (attack.py:53-119, 138-173)
// Rows are whole Gaussians (cols = 3, 4, 1 or 45 floats).  A wave owns 64 consecutive rows (32 of the wide ones) = one
// contiguous span of the three tensors: it is copied through LDS with coalesced 16-byte loads (cols is odd or small, so lane r walking row r
// is bank-conflict free for the 45-float rows) and written back the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

constexpr int PGD_MAX_COLS = 48;
// Rows per wave: 64 for narrow rows; 32 for rows wider than PGD_NARROW floats, so that a wave's deltas take 6 KB of LDS
// instead of 12 and 16 waves fit a CU where 12 did (the kernel is bandwidth-bound: more waves = more loads in flight).
constexpr int PGD_NARROW = 24;
constexpr int PGD_LDS_FLOATS = 64 * PGD_NARROW > 32 * PGD_MAX_COLS ? 64 * PGD_NARROW : 32 * PGD_MAX_COLS;
__host__ __device__ inline int pgd_rows_per_wave(int cols) { return cols > PGD_NARROW ? 32 : 64; }

// per-block partial sums of squares (float per thread, double across the block: the order is fixed => reproducible).
// 16-byte loads when the tensor starts on a 16-byte boundary (n / 4 float4 + a scalar tail), 4-byte loads otherwise.
__global__ void __launch_bounds__(256) k_pgd_sumsq(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  __shared__ double wsum[4];
  float acc = 0.f;
  // (a gradient that is a slice of a flat bucket starts at a multiple of P floats: 16-byte aligned only for some P)
  const size_t n4 = (reinterpret_cast<uintptr_t>(g) & 15u) == 0u ? (n >> 2) : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = g4[i];
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
  double d = (double)acc;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) d += __shfl_xor(d, s, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// L2 == true: `partial[nb]` holds the block sums of squares of g.
// A wave's 64 (32) rows are one contiguous, 128-byte aligned span of 64 (32) * cols floats: it is moved with 16-byte accesses
// (a full wave's span is a whole number of float4; the last, partial wave of a tensor falls back to 4-byte accesses).
// (the body of both kernels below: `blk` = the wave's index among the tensor's groups of 64 rows)
template <bool L2>
__device__ __forceinline__ void pgd_step_wave(float* __restrict__ x, const float* __restrict__ g,
                                              const float* __restrict__ x0, size_t rows, int cols, float alpha,
                                              float eps, const double* __restrict__ partial, int nb, unsigned blk) {
  __shared__ __attribute__((aligned(16))) float sd[PGD_LDS_FLOATS];        // the wave's deltas x_new - x0
  __shared__ float sf[64];                    // per-row clip factors
  const int lane = threadIdx.x;
  const int rpw = pgd_rows_per_wave(cols);
  const size_t r0 = (size_t)blk * (size_t)rpw;
  const int nr = (int)min((size_t)rpw, rows - r0);
  const size_t base = r0 * (size_t)cols;
  const int n = nr * cols;
  const bool vec = nr == rpw && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x0)) & 15u) == 0u;
  float scale = 0.f;                            // alpha / ||g||  (0 when the gradient is zero: no step)
  if (L2) {
    double t = 0.0;
    for (int i = lane; i < nb; i += 64) t += partial[i];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) t += __shfl_xor(t, s, 64);
    const float nrm = sqrtf((float)t);
    scale = nrm > 0.f ? alpha / nrm : 0.f;
  }
  auto upd = [&](float xv, float gv, float ov, float& dlt) -> float {
    if (L2) {
      const float xn = fmaf(-scale, gv, xv);
      dlt = xn - ov;
      return xn;
    }
    const float sg = gv > 0.f ? 1.f : (gv < 0.f ? -1.f : 0.f);        // torch.sign (NaN -> NaN is not reproduced: 0)
    const float xn = fmaf(-alpha, sg, xv);
    dlt = 0.f;
    return fminf(fmaxf(xn - ov, -eps), eps) + ov;
  };
  if (vec) {
    const int n4 = n >> 2;                      // 16 * cols
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* g4 = reinterpret_cast<const float4*>(g + base);
    const float4* o4 = reinterpret_cast<const float4*>(x0 + base);
    float4* xo4 = reinterpret_cast<float4*>(x + base);
    float4* sd4 = reinterpret_cast<float4*>(sd);
    for (int i = lane; i < n4; i += 64) {
      const float4 xv = x4[i], gv = g4[i], ov = o4[i];
      float4 d, r;
      r.x = upd(xv.x, gv.x, ov.x, d.x); r.y = upd(xv.y, gv.y, ov.y, d.y);
      r.z = upd(xv.z, gv.z, ov.z, d.z); r.w = upd(xv.w, gv.w, ov.w, d.w);
      if (L2) sd4[i] = d; else xo4[i] = r;
    }
  } else {
    for (int i = lane; i < n; i += 64) {
      float d;
      const float r = upd(x[base + i], g[base + i], x0[base + i], d);
      if (L2) sd[i] = d; else x[base + i] = r;
    }
  }
  if (!L2) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float f = 1.f;
  if (lane < nr) {
    float ss = 0.f;
    for (int c = 0; c < cols; ++c) { const float d = sd[lane * cols + c]; ss = fmaf(d, d, ss); }
    const float nrm = sqrtf(ss);
    if (nrm > eps) f = eps / (nrm + 1e-7f);
  }
  // Row r's factor is parked in LDS and read per element.  (A cross-lane read of lane r's register inside the loop
  // below would return 0 for rows whose owner lane has already left the loop in the last, partial iteration:
  // ds_bpermute yields 0 for an inactive source lane.)
  sf[lane] = f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (vec) {
    const int n4 = n >> 2;
    const float4* o4 = reinterpret_cast<const float4*>(x0 + base);
    float4* xo4 = reinterpret_cast<float4*>(x + base);
    const float4* sd4 = reinterpret_cast<const float4*>(sd);
    for (int i = lane; i < n4; i += 64) {
      const float4 ov = o4[i], d = sd4[i];
      const int e = 4 * i;
      float4 r;
      r.x = fmaf(d.x, sf[e / cols], ov.x); r.y = fmaf(d.y, sf[(e + 1) / cols], ov.y);
      r.z = fmaf(d.z, sf[(e + 2) / cols], ov.z); r.w = fmaf(d.w, sf[(e + 3) / cols], ov.w);
      xo4[i] = r;
    }
  } else {
    for (int i = lane; i < n; i += 64) {
      const int r = i / cols;
      x[base + i] = fmaf(sd[i], sf[r], x0[base + i]);
    }
  }
}

template <bool L2>
__global__ void __launch_bounds__(64) k_pgd_step(float* __restrict__ x, const float* __restrict__ g,
                                                 const float* __restrict__ x0, size_t rows, int cols, float alpha,
                                                 float eps, const double* __restrict__ partial, int nb) {
  pgd_step_wave<L2>(x, g, x0, rows, cols, alpha, eps, partial, nb, blockIdx.x);
}

// Several tensors of one model in ONE launch (gsr_pgd_step_multi): the attack steps up to six attribute tensors per
// iteration, each a bandwidth-bound launch of a few tens of microseconds whose tail and whose successor's ramp-up are
// not covered by anything on the stream.  Workgroup b belongs to the tensor t with first[t] <= b < first[t + 1]; the
// arithmetic per tensor is pgd_step_wave's, so the result is bit for bit that of the per-tensor launches.
constexpr int PGD_MAX_TENSORS = 8;
struct PgdMulti {
  float* x[PGD_MAX_TENSORS];
  const float* g[PGD_MAX_TENSORS];
  const float* x0[PGD_MAX_TENSORS];
  const double* partial[PGD_MAX_TENSORS];     // L2: sums of squares of g (nb[t] partial sums)
  unsigned long long rows[PGD_MAX_TENSORS];
  unsigned first[PGD_MAX_TENSORS + 1];        // first workgroup of tensor t; first[n] = the grid
  int cols[PGD_MAX_TENSORS];
  int nb[PGD_MAX_TENSORS];
  float alpha[PGD_MAX_TENSORS];
  float eps[PGD_MAX_TENSORS];
  int n;
};

template <bool L2>
__global__ void __launch_bounds__(64) k_pgd_step_multi(PgdMulti m) {
  int t = 0;
#pragma unroll
  for (int i = 1; i < PGD_MAX_TENSORS; ++i) t += (i < m.n && blockIdx.x >= m.first[i]) ? 1 : 0;
  pgd_step_wave<L2>(m.x[t], m.g[t], m.x0[t], (size_t)m.rows[t], m.cols[t], m.alpha[t], m.eps[t], m.partial[t], m.nb[t],
                    blockIdx.x - m.first[t]);
}

}  // namespace gsr
