// gsr_knn.hip.h -- mean squared distance of every point to its 3 nearest neighbours, exact, on gfx950.
//
// Replaces the reference's second native import, `simple_knn._C.distCUDA2` (reference scene/gaussian_model.py:17,
// used at :144 in create_from_pcd to seed the initial scales).  Own algorithm: points are binned into a uniform
// grid (about 3 points per cell), sorted by cell with the library's radix sort, and every point scans cubic shells
// of cells around its own until the third-best distance found so far is no larger than the distance to anything
// outside the scanned block -- which makes the result exact.  A point that has not terminated after MAX_SHELL
// shells (an isolated outlier) finishes with a brute-force sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

namespace gsr {

struct KnnGrid {
  float minx, miny, minz;
  float inv_cx, inv_cy, inv_cz;   // 1 / cell size per axis
  float sx, sy, sz;               // cell size per axis
  int dx, dy, dz;                 // cells per axis
};

__device__ __forceinline__ float atomic_min_f(float* addr, float v) {
  // monotone int mapping of IEEE floats
  int* ia = reinterpret_cast<int*>(addr);
  int old = *ia, assumed;
  do {
    assumed = old;
    if (__int_as_float(assumed) <= v) break;
    old = atomicCAS(ia, assumed, __float_as_int(v));
  } while (assumed != old);
  return __int_as_float(old);
}
__device__ __forceinline__ float atomic_max_f(float* addr, float v) {
  int* ia = reinterpret_cast<int*>(addr);
  int old = *ia, assumed;
  do {
    assumed = old;
    if (__int_as_float(assumed) >= v) break;
    old = atomicCAS(ia, assumed, __float_as_int(v));
  } while (assumed != old);
  return __int_as_float(old);
}

// bbox[0..2] = min, bbox[3..5] = max (initialised to +/-FLT_MAX by the host)
__global__ void __launch_bounds__(256) k_knn_bbox(const float* __restrict__ pts, int P, float* __restrict__ bbox) {
  __shared__ float smin[3][4], smax[3][4];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float v = pts[3 * i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
    }
    if (lane == 0) { smin[a][w] = mn[a]; smax[a][w] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    atomic_min_f(&bbox[a], fminf(fminf(smin[a][0], smin[a][1]), fminf(smin[a][2], smin[a][3])));
    atomic_max_f(&bbox[3 + a], fmaxf(fmaxf(smax[a][0], smax[a][1]), fmaxf(smax[a][2], smax[a][3])));
  }
}

__device__ __forceinline__ int knn_cell_coord(float v, float mn, float inv, int d) {
  const int c = (int)((v - mn) * inv);
  return c < 0 ? 0 : (c >= d ? d - 1 : c);
}

__global__ void __launch_bounds__(256) k_knn_cells(const float* __restrict__ pts, int P, KnnGrid g,
                                                   uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int cx = knn_cell_coord(pts[3 * i], g.minx, g.inv_cx, g.dx);
  const int cy = knn_cell_coord(pts[3 * i + 1], g.miny, g.inv_cy, g.dy);
  const int cz = knn_cell_coord(pts[3 * i + 2], g.minz, g.inv_cz, g.dz);
  keys[i] = (uint32_t)((cz * g.dy + cy) * g.dx + cx);
}

// ranges of equal keys in the sorted key array -> cell_range[cell] = (start, end)
__global__ void __launch_bounds__(256) k_knn_ranges(uint32_t P, const uint32_t* __restrict__ keys,
                                                    uint2* __restrict__ cell_range) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const uint32_t c = keys[i];
  if (i == 0 || keys[i - 1] != c) cell_range[c].x = i;
  if (i == P - 1 || keys[i + 1] != c) cell_range[c].y = i + 1;
}

__device__ __forceinline__ void knn_update(float d2, float best[3]) {
  if (d2 < best[2]) {
    if (d2 < best[1]) {
      best[2] = best[1];
      if (d2 < best[0]) { best[1] = best[0]; best[0] = d2; } else best[1] = d2;
    } else best[2] = d2;
  }
}

constexpr int KNN_MAX_SHELL = 10;

// one thread per point, in cell-sorted order (neighbouring threads scan the same cells)
__global__ void __launch_bounds__(256) k_knn_search(const float* __restrict__ pts, int P, KnnGrid g,
                                                    const uint32_t* __restrict__ sorted_idx,
                                                    const uint2* __restrict__ cell_range, float* __restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= P) return;
  const uint32_t me = sorted_idx[s];
  const float px = pts[3 * me], py = pts[3 * me + 1], pz = pts[3 * me + 2];
  const int cx = knn_cell_coord(px, g.minx, g.inv_cx, g.dx);
  const int cy = knn_cell_coord(py, g.miny, g.inv_cy, g.dy);
  const int cz = knn_cell_coord(pz, g.minz, g.inv_cz, g.dz);
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  bool done = false;
  for (int r = 0; r <= KNN_MAX_SHELL && !done; ++r) {
    for (int z = cz - r; z <= cz + r; ++z) {
      if (z < 0 || z >= g.dz) continue;
      for (int y = cy - r; y <= cy + r; ++y) {
        if (y < 0 || y >= g.dy) continue;
        const bool face = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
        for (int x = cx - r; x <= cx + r; x += (face ? 1 : 2 * r)) {     // interior rows: only the two end cells
          if (x >= 0 && x < g.dx) {
            const uint2 rg = cell_range[(z * g.dy + y) * g.dx + x];
            for (uint32_t j = rg.x; j < rg.y; ++j) {
              const uint32_t o = sorted_idx[j];
              if (o == me) continue;
              const float ddx = pts[3 * o] - px, ddy = pts[3 * o + 1] - py, ddz = pts[3 * o + 2] - pz;
              knn_update(ddx * ddx + ddy * ddy + ddz * ddz, best);
            }
          }
          if (r == 0) break;
        }
      }
    }
    // distance from the point to the nearest face of the scanned (2r+1)^3 block that is not a face of the grid:
    // every point binned outside the block is at least that far away (shaved for the rounding of the binning)
    float safe = FLT_MAX;
    if (cx - r > 0) safe = fminf(safe, px - (g.minx + (float)(cx - r) * g.sx));
    if (cx + r < g.dx - 1) safe = fminf(safe, (g.minx + (float)(cx + r + 1) * g.sx) - px);
    if (cy - r > 0) safe = fminf(safe, py - (g.miny + (float)(cy - r) * g.sy));
    if (cy + r < g.dy - 1) safe = fminf(safe, (g.miny + (float)(cy + r + 1) * g.sy) - py);
    if (cz - r > 0) safe = fminf(safe, pz - (g.minz + (float)(cz - r) * g.sz));
    if (cz + r < g.dz - 1) safe = fminf(safe, (g.minz + (float)(cz + r + 1) * g.sz) - pz);
    if (safe == FLT_MAX) {
      done = true;                      // the block covers the whole grid: nothing left to scan
    } else {
      safe = fmaxf(0.f, safe * 0.9999f);
      done = best[2] <= safe * safe;
    }
  }
  if (!done) {   // isolated point: exact answer by a full sweep
    best[0] = best[1] = best[2] = FLT_MAX;
    for (int o = 0; o < P; ++o) {
      if ((uint32_t)o == me) continue;
      const float ddx = pts[3 * o] - px, ddy = pts[3 * o + 1] - py, ddz = pts[3 * o + 2] - pz;
      knn_update(ddx * ddx + ddy * ddy + ddz * ddz, best);
    }
  }
  out[me] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace gsr
