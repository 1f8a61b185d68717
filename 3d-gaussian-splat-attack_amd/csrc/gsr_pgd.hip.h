// gsr_pgd.hip.h -- the projected-gradient update rules on one raw attribute tensor [rows, cols] (SURVEY.md section
// 8a row a11; reference attack.py:25-173), fused into two launches per tensor instead of the ~15 elementwise /
// reduction passes the PyTorch formulation makes over the same 236 MB:
//   L-inf:  x += -alpha * sign(g);  x = clamp(x - x0, -eps, eps) + x0                     (attack.py:25-51, 121-136)
//   L2:     x += -alpha * g / ||g||_2 (norm over the WHOLE tensor; no step if it is 0);
//           d = x - x0, each ROW clipped to the eps ball: d *= eps / (||d_row|| + 1e-7) when ||d_row|| > eps
//           (torch.renorm(p=2, dim=0, maxnorm=eps));  x = x0 + d                          (attack.py:53-119, 138-173)
// Rows are whole Gaussians (cols = 3, 4, 1 or 45 floats).  A wave owns 64 consecutive rows = one contiguous span of
// the three tensors: it is copied through LDS with coalesced loads (cols is odd or small, so lane r walking row r is
// bank-conflict free for the 45-float rows) and written back the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

constexpr int PGD_MAX_COLS = 48;

// per-block partial sums of squares (float per thread, double across the block: the order is fixed => reproducible)
__global__ void __launch_bounds__(256) k_pgd_sumsq(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  __shared__ double wsum[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
  double d = (double)acc;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) d += __shfl_xor(d, s, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// L2 == true: `partial[nb]` holds the block sums of squares of g.
template <bool L2>
__global__ void __launch_bounds__(64) k_pgd_step(float* __restrict__ x, const float* __restrict__ g,
                                                 const float* __restrict__ x0, size_t rows, int cols, float alpha,
                                                 float eps, const double* __restrict__ partial, int nb) {
  __shared__ float sd[64 * PGD_MAX_COLS];     // the wave's deltas x_new - x0
  __shared__ float sf[64];                    // per-row clip factors
  const int lane = threadIdx.x;
  const size_t r0 = (size_t)blockIdx.x * 64;
  const int nr = (int)min((size_t)64, rows - r0);
  const size_t base = r0 * (size_t)cols;
  const int n = nr * cols;
  float scale = 0.f;                            // alpha / ||g||  (0 when the gradient is zero: no step)
  if (L2) {
    double t = 0.0;
    for (int i = lane; i < nb; i += 64) t += partial[i];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) t += __shfl_xor(t, s, 64);
    const float nrm = sqrtf((float)t);
    scale = nrm > 0.f ? alpha / nrm : 0.f;
  }
  for (int i = lane; i < n; i += 64) {
    const float xv = x[base + i], gv = g[base + i], ov = x0[base + i];
    float xn;
    if (L2) {
      xn = fmaf(-scale, gv, xv);
      sd[i] = xn - ov;
    } else {
      const float sg = gv > 0.f ? 1.f : (gv < 0.f ? -1.f : 0.f);      // torch.sign (NaN -> NaN is not reproduced: 0)
      xn = fmaf(-alpha, sg, xv);
      x[base + i] = fminf(fmaxf(xn - ov, -eps), eps) + ov;
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
  for (int i = lane; i < n; i += 64) {
    const int r = i / cols;
    x[base + i] = fmaf(sd[i], sf[r], x0[base + i]);
  }
}

}  // namespace gsr
