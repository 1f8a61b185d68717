// gsr_math.h -- per-Gaussian and per-(pixel,Gaussian) arithmetic of the splat rasteriser.
//
// Pure scalar float32 functions, usable from HIP kernels (hipcc) and from a host C++ test harness
// (g++; tests/host_math) so that the exact source the kernels run is checked against the oracle on
// CPU before any GPU time is spent.  No reference code: formulas follow SURVEY.md section 8(a)
// (rows a4, a6, a8, a9) and the in-tree Python twins cited there
// (utils/sh_utils.py:57-112, utils/general_utils.py:78-110, utils/graphics_utils.py:38-71).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GSR_HD __host__ __device__ __forceinline__
#else
#define GSR_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define GSR_EXP(x) __expf(x)
#define GSR_LOG_FAST(x) __logf(x)     // v_log_f32 (1 ulp of log2) x ln 2: callers widen their bound by 1e-4 relative + 1e-3
#else
#define GSR_EXP(x) expf(x)
#define GSR_LOG_FAST(x) logf(x)
#endif

namespace gsr {

constexpr int TILE = 16;            // 16x16-pixel tiles
constexpr int NUM_OBJ = 16;         // object-feature channels (scene/gaussian_model.py:52)
constexpr float NEAR_Z = 0.2f;      // view-space near cull
constexpr float FOV_CLAMP = 1.3f;   // t.x/t.z clamp, in units of tanfov
constexpr float DILATE = 0.3f;      // px^2 added to the 2D covariance diagonal
constexpr float ALPHA_CAP = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 1e-4f;
constexpr int RADIUS_MAX = 1 << 24;   // pixels: a larger footprint covers every image this library accepts (65520 px per side)
// the published backward of the 2D covariance inversion: 1 / (det^2 + 0.0000001f); -DGSR_DET_GUARD=0.0f: the exact derivative
#ifndef GSR_DET_GUARD
#define GSR_DET_GUARD 0.0000001f
#endif

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// Per-view constants, filled on the device from the settings' device tensors.
// The projection chain of a splat -- view transform, J W, Sigma3D from scale and rotation, the 2D covariance -- is
// evaluated by the forward (k_pre_geom) and AGAIN by every backward kernel (k_pre_bwd, k_pre_bwd_batch, k_preprocess_bwd),
// which differentiate the inversion of the 2D covariance at the point they recompute.  For an elongated splat that
// derivative cancels by the eigenvalue ratio (project_splat_bwd), so the recomputed (a, b, c) must be THE SAME BITS in
// every kernel: left to the compiler, a*b + c is contracted to an fma or not depending on what a function is inlined
// into, and two kernels that differ in the last bit of (a, b, c) differ by 1e-7 x the eigenvalue ratio in the gradient
// (round 6: the batch kernel's scale gradient of a 175:1 needle was 0.5 % from the single-view kernel's and 30x
// further from the float64 oracle).  GSR_FP_STRICT at the top of a function body switches contraction off for that
// function: every product and sum in it is rounded once, in source order, in every kernel and on the host.
#if defined(__clang__)
#define GSR_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define GSR_FP_STRICT
#endif

struct View {
  float V[16];    // viewmatrix, row-major as given: p_view = [p,1] * V (row-vector convention)
  float PV[16];   // full projection, same convention
  float cam[3];
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  int W, H, gridx, gridy, sh_degree;
};

GSR_HD void make_view(View& v, const float* vm, const float* pm, const float* campos, int H, int W, float tanfovx,
                      float tanfovy, float scale_modifier, int sh_degree) {
  for (int i = 0; i < 16; ++i) { v.V[i] = vm[i]; v.PV[i] = pm[i]; }
  v.cam[0] = campos[0]; v.cam[1] = campos[1]; v.cam[2] = campos[2];
  v.tanfovx = tanfovx; v.tanfovy = tanfovy;
  v.focal_x = (float)W / (2.0f * tanfovx);
  v.focal_y = (float)H / (2.0f * tanfovy);
  v.scale_modifier = scale_modifier;
  v.W = W; v.H = H; v.gridx = (W + TILE - 1) / TILE; v.gridy = (H + TILE - 1) / TILE;
  v.sh_degree = sh_degree;
}

// ---------------------------------------------------------------------------------------------
// 3D covariance from scale + quaternion (q = (r,x,y,z) used as given), packed xx,xy,xz,yy,yz,zz
// ---------------------------------------------------------------------------------------------
GSR_HD void quat_to_R(const float q[4], float R[9]) {
  GSR_FP_STRICT
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

GSR_HD void cov3d_from_scale_rot(const float s_in[3], float mod, const float q[4], float c6[6]) {
  GSR_FP_STRICT
  float R[9];
  quat_to_R(q, R);
  const float s0 = mod * s_in[0], s1 = mod * s_in[1], s2 = mod * s_in[2];
  // L = R * diag(s);  Sigma = L L^T
  const float L00 = R[0] * s0, L01 = R[1] * s1, L02 = R[2] * s2;
  const float L10 = R[3] * s0, L11 = R[4] * s1, L12 = R[5] * s2;
  const float L20 = R[6] * s0, L21 = R[7] * s1, L22 = R[8] * s2;
  c6[0] = L00 * L00 + L01 * L01 + L02 * L02;
  c6[1] = L00 * L10 + L01 * L11 + L02 * L12;
  c6[2] = L00 * L20 + L01 * L21 + L02 * L22;
  c6[3] = L10 * L10 + L11 * L11 + L12 * L12;
  c6[4] = L10 * L20 + L11 * L21 + L12 * L22;
  c6[5] = L20 * L20 + L21 * L21 + L22 * L22;
}

// ---------------------------------------------------------------------------------------------
// Geometry of one Gaussian on screen (K1 steps 1-8, 10)
// ---------------------------------------------------------------------------------------------
struct Splat {
  float px, py;      // pixel-space centre, the published float32 arithmetic: every INTEGER decision (radius, tile rect, culls)
                     // is taken from these, like the reference's
  float ca, cb, cc;  // the dilated 2D covariance of the float32 chain (what A, B, C were inverted from)
  double pxd, pyd;   // the same centre from double-precision dot products of the float32 inputs: what the compositors measure
                     // distances from (round 5).  A float32 coordinate beyond 2048 resolves 2.4e-4 px, and d ln(alpha) / d centre
                     // of a one-pixel splat is O(1): at 4K the published float32 centre alone costs up to 2e-4 in a pixel's
                     // colour (profiles/r05_fullsize_sweep.txt, tests/diag_cfg5_pixel.py).  Stored relative to the rect's
                     // first tile (a small number: full float32 resolution), see k_pre_geom / stage_splat.
  float depth;       // view-space z
  float A, B, C;     // conic = inverse of the dilated 2D covariance
  int radius;        // 0 => culled
  int rminx, rminy, rmaxx, rmaxy;   // tile rect [min,max)
};

GSR_HD int trunc_clamp(float v, int hi) {
  // (int) truncation toward zero, then clamp to [0,hi]; NaN/inf safe
  if (!(v > 0.f)) return 0;
  if (v >= (float)hi) return hi;
  return (int)v;
}

// M = J * Wr (2x3): the affine map from world-space covariance to screen space.
// Also returns the clamped view-space point and whether each axis was clamped.
struct ProjLin {
  float M[6];
  float tx, ty, tz;
  bool clx, cly;
};

// p_view = [p, 1] V
GSR_HD void view_transform(const View& v, const float p[3], float t[3]) {
  GSR_FP_STRICT
  for (int j = 0; j < 3; ++j) t[j] = p[0] * v.V[j] + p[1] * v.V[4 + j] + p[2] * v.V[8 + j] + v.V[12 + j];
}

GSR_HD void proj_linear(const View& v, const float t[3], ProjLin& o) {
  GSR_FP_STRICT
  const float limx = FOV_CLAMP * v.tanfovx, limy = FOV_CLAMP * v.tanfovy;
  const float tz = t[2];
  const float txtz = t[0] / tz, tytz = t[1] / tz;
  o.clx = (txtz < -limx) || (txtz > limx);
  o.cly = (tytz < -limy) || (tytz > limy);
  o.tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
  o.ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
  o.tz = tz;
  const float J00 = v.focal_x / tz, J02 = -(v.focal_x * o.tx) / (tz * tz);
  const float J11 = v.focal_y / tz, J12 = -(v.focal_y * o.ty) / (tz * tz);
  // Wr[j][i] = V[i*4+j] (column-vector view rotation); M[a][i] = sum_j J[a][j] Wr[j][i]
  for (int i = 0; i < 3; ++i) {
    o.M[i] = J00 * v.V[i * 4 + 0] + J02 * v.V[i * 4 + 2];
    o.M[3 + i] = J11 * v.V[i * 4 + 1] + J12 * v.V[i * 4 + 2];
  }
}

GSR_HD void cov2d_from_M(const float M[6], const float c6[6], float& a, float& b, float& c) {
  GSR_FP_STRICT
  // u = Sigma * M0^T, w = Sigma * M1^T
  const float u0 = c6[0] * M[0] + c6[1] * M[1] + c6[2] * M[2];
  const float u1 = c6[1] * M[0] + c6[3] * M[1] + c6[4] * M[2];
  const float u2 = c6[2] * M[0] + c6[4] * M[1] + c6[5] * M[2];
  const float w0 = c6[0] * M[3] + c6[1] * M[4] + c6[2] * M[5];
  const float w1 = c6[1] * M[3] + c6[3] * M[4] + c6[4] * M[5];
  const float w2 = c6[2] * M[3] + c6[4] * M[4] + c6[5] * M[5];
  a = M[0] * u0 + M[1] * u1 + M[2] * u2 + DILATE;
  b = M[3] * u0 + M[4] * u1 + M[5] * u2;
  c = M[3] * w0 + M[4] * w1 + M[5] * w2 + DILATE;
}

GSR_HD bool project_splat(const View& v, const float p[3], const float c6[6], Splat& s) {
  s.radius = 0;
  float t[3];
  view_transform(v, p, t);
  if (!(t[2] > NEAR_Z)) return false;
  float h[4];
  for (int j = 0; j < 4; ++j) h[j] = p[0] * v.PV[j] + p[1] * v.PV[4 + j] + p[2] * v.PV[8 + j] + v.PV[12 + j];
  const float w = 1.0f / (h[3] + 1e-7f);
  const float ndcx = h[0] * w, ndcy = h[1] * w;
  ProjLin pl;
  proj_linear(v, t, pl);
  float a, b, c;
  cov2d_from_M(pl.M, c6, a, b, c);
  const float det = a * c - b * b;
  if (det == 0.0f) return false;
  const float dinv = 1.0f / det;
  s.A = c * dinv; s.B = -b * dinv; s.C = a * dinv;
  s.ca = a; s.cb = b; s.cc = c;
  const float mid = 0.5f * (a + c);
  const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  s.px = ((ndcx + 1.0f) * (float)v.W - 1.0f) * 0.5f;
  s.py = ((ndcy + 1.0f) * (float)v.H - 1.0f) * 0.5f;
  // Non-finite and extreme inputs (include/gsraster.h, "Non-finite inputs"): a splat whose conic, centre or depth is not a
  // finite number -- NaN or +-inf in its mean, scale, rotation or covariance, or finite values whose products overflow -- is
  // culled like one behind the camera (radius 0, no pairs, zero gradients): x * 0 is 0 for every finite x and NaN
  // otherwise.  A finite but enormous footprint keeps the published behaviour (a rect clamped to the image) with the
  // radius saturated at RADIUS_MAX instead of a float -> int conversion that is undefined beyond 2^31.
  if (!((s.A + s.B + s.C) * 0.0f + (s.px + s.py + t[2]) * 0.0f == 0.0f)) return false;
  const float r3 = 3.0f * sqrtf(lam);                      // (NaN lam: mid^2 overflowed although the conic is finite)
  if (!(r3 == r3)) return false;
  const int radius = r3 < (float)RADIUS_MAX ? (int)ceilf(r3) : RADIUS_MAX;
  {
    const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
    const double h0 = x * (double)v.PV[0] + y * (double)v.PV[4] + z * (double)v.PV[8] + (double)v.PV[12];
    const double h1 = x * (double)v.PV[1] + y * (double)v.PV[5] + z * (double)v.PV[9] + (double)v.PV[13];
    const double h3 = x * (double)v.PV[3] + y * (double)v.PV[7] + z * (double)v.PV[11] + (double)v.PV[15];
    const double wd = 1.0 / (h3 + (double)1e-7f);
    s.pxd = ((h0 * wd + 1.0) * (double)v.W - 1.0) * 0.5;
    s.pyd = ((h1 * wd + 1.0) * (double)v.H - 1.0) * 0.5;
  }
  const float rf = (float)radius;
  s.rminx = trunc_clamp((s.px - rf) / (float)TILE, v.gridx);
  s.rminy = trunc_clamp((s.py - rf) / (float)TILE, v.gridy);
  s.rmaxx = trunc_clamp((s.px + rf + (float)(TILE - 1)) / (float)TILE, v.gridx);
  s.rmaxy = trunc_clamp((s.py + rf + (float)(TILE - 1)) / (float)TILE, v.gridy);
  if ((s.rmaxx - s.rminx) * (s.rmaxy - s.rminy) <= 0) return false;
  s.depth = t[2];
  s.radius = radius;
  return true;
}

// An opacity that is not a number emits nothing (a NaN alpha would pass min(0.99, .) as 0.99 on this hardware as in the
// published kernels -- garbage in, a fully opaque splat out; here the Gaussian is culled instead, with and without the
// footprint cull): include/gsraster.h, "Non-finite inputs".
GSR_HD bool opacity_ok(float o) { return o == o; }

// A splat is a NEEDLE for the purposes below when the eigenvalues of its dilated 2D covariance are more than
// NEEDLE_RATIO apart ((a + c)^2 / det = r + 2 + 1/r): the float32 chain leaves about 1e-7 x that ratio in every conic
// entry -- 3e-5 at the threshold, where the two chains therefore agree to five digits --, the exponent of a needle sums
// terms of (radius)^2 that cancel to O(1), and its gradients cancel once more (EXPERIMENTS.md, rounds 3-5: a 1500:1 needle's
// dL/dmean2D 2.3 % off in float32).  Ordinary splats keep the published float32 conic bit for bit.
constexpr float NEEDLE_RATIO = 256.0f;
GSR_HD bool is_needle(float a, float b, float c) { const float tr = a + c; return tr * tr > (NEEDLE_RATIO + 2.0f) * (a * c - b * b); }

// The conic of a visible splat in double (round 5): the published chain -- Sigma = (R diag(mod s)) (...)^T from the
// quaternion AS GIVEN, or the precomputed covariance; t = p V with t.x/t.z, t.y/t.z clamped to +-1.3 tan(fov/2); J; cov2D =
// J W Sigma W^T J^T + 0.3 I; conic = cov2D^-1 -- with every product in double, from the same float32 inputs the float32
// chain of project_splat() reads.  Nothing INTEGER comes from it: radius, rect and culls stay project_splat()'s.
// (cov2d_accurate: the dilated 2D covariance (a, b, c) of that chain; needle_conic_to_float: its inverse as the record's three floats)
GSR_HD void cov2d_accurate(const View& v, const float p[3], const float* sc, float mod, const float* q, const float* c6pre,
                           double& a, double& b, double& c) {
  double S[6];
  if (c6pre) {
    for (int i = 0; i < 6; ++i) S[i] = (double)c6pre[i];
  } else {
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    const double R[9] = {1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - r * z), 2.0 * (x * z + r * y),
                         2.0 * (x * y + r * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - r * x),
                         2.0 * (x * z - r * y), 2.0 * (y * z + r * x), 1.0 - 2.0 * (x * x + y * y)};
    const double s0 = (double)mod * sc[0], s1 = (double)mod * sc[1], s2 = (double)mod * sc[2];
    const double L[9] = {R[0] * s0, R[1] * s1, R[2] * s2, R[3] * s0, R[4] * s1, R[5] * s2, R[6] * s0, R[7] * s1, R[8] * s2};
    S[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    S[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
    S[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
    S[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
    S[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
    S[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
  }
  double t[3];
  for (int j = 0; j < 3; ++j)
    t[j] = (double)p[0] * v.V[j] + (double)p[1] * v.V[4 + j] + (double)p[2] * v.V[8 + j] + (double)v.V[12 + j];
  const double limx = (double)FOV_CLAMP * v.tanfovx, limy = (double)FOV_CLAMP * v.tanfovy;
  const double tz = t[2];
  double tx = t[0] / tz, ty = t[1] / tz;
  tx = (tx < -limx ? -limx : (tx > limx ? limx : tx)) * tz;
  ty = (ty < -limy ? -limy : (ty > limy ? limy : ty)) * tz;
  const double fx = (double)v.W / (2.0 * (double)v.tanfovx), fy = (double)v.H / (2.0 * (double)v.tanfovy);
  const double J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  double M[6];
  for (int i = 0; i < 3; ++i) {
    M[i] = J00 * v.V[i * 4 + 0] + J02 * v.V[i * 4 + 2];
    M[3 + i] = J11 * v.V[i * 4 + 1] + J12 * v.V[i * 4 + 2];
  }
  const double u0 = S[0] * M[0] + S[1] * M[1] + S[2] * M[2], u1 = S[1] * M[0] + S[3] * M[1] + S[4] * M[2],
               u2 = S[2] * M[0] + S[4] * M[1] + S[5] * M[2];
  const double w0 = S[0] * M[3] + S[1] * M[4] + S[2] * M[5], w1 = S[1] * M[3] + S[3] * M[4] + S[4] * M[5],
               w2 = S[2] * M[3] + S[4] * M[4] + S[5] * M[5];
  a = M[0] * u0 + M[1] * u1 + M[2] * u2 + (double)DILATE;
  b = M[3] * u0 + M[4] * u1 + M[5] * u2;
  c = M[3] * w0 + M[4] * w1 + M[5] * w2 + (double)DILATE;
}

// A needle's conic, double -> the three float32 numbers the records carry.  The small eigenvalue k_b of the conic (1 / the
// long axis's variance; 2e-3 where the entries are ~1 for a 440:1 needle) is what a float32 triple cannot hold: independent
// rounding of the entries leaves up to 6e-8 in u^T K u along the long axis u -- 3e-5 of k_b -- and a needle's rotation
// gradient amplifies that by ~250 (0.6 % measured, against 0.04 % for the float32 chain's error, which is a COMMON FACTOR
// on the conic (the determinant) and harmless in that direction: EXPERIMENTS.md round 5, tests/diag_needle_sensitivity.py).
// So among the float32 neighbours (+-2 ulps per entry) of the rounded triple, take the one whose quadratic form along u is
// closest to the double conic's: 125 candidates, residual ~1/9 of plain rounding's.  (a, b, c): the dilated 2D covariance.
GSR_HD float gsr_step_ulps(float x, int n) {
  if (x == 0.0f || n == 0) return x;
  union { float f; int32_t i; } w;
  w.f = x;
  w.i += n;               // sign-magnitude: +n moves away from zero; the search is symmetric, so direction does not matter
  return w.f;
}
GSR_HD void needle_conic_to_float(double a, double b, double c, float& A, float& B, float& C) {
  const double dinv = 1.0 / (a * c - b * b);     // det >= 0.09 in exact arithmetic: a PSD matrix + 0.3 I
  const double Ad = c * dinv, Bd = -b * dinv, Cd = a * dinv;
  // u: eigenvector of the covariance's LARGE eigenvalue
  const double hd = 0.5 * (a - c), lam = 0.5 * (a + c) + sqrt(hd * hd + b * b);
  double ux, uy;
  if (a >= c) { ux = lam - c; uy = b; } else { ux = b; uy = lam - a; }
  const double un = 1.0 / (ux * ux + uy * uy);
  const double wA = ux * ux * un, wB = 2.0 * ux * uy * un, wC = uy * uy * un;
  const float A0 = (float)Ad, B0 = (float)Bd, C0 = (float)Cd;
  double eA[5], eB[5], eC[5];
  for (int i = 0; i < 5; ++i) {
    eA[i] = wA * ((double)gsr_step_ulps(A0, i - 2) - Ad);
    eB[i] = wB * ((double)gsr_step_ulps(B0, i - 2) - Bd);
    eC[i] = wC * ((double)gsr_step_ulps(C0, i - 2) - Cd);
  }
  double best = fabs(eA[2] + eB[2] + eC[2]);
  int bi = 2, bj = 2, bk = 2;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      const double eab = eA[i] + eB[j];
      for (int k = 0; k < 5; ++k) {
        const double e = fabs(eab + eC[k]);
        if (e < best) { best = e; bi = i; bj = j; bk = k; }
      }
    }
  A = gsr_step_ulps(A0, bi - 2); B = gsr_step_ulps(B0, bj - 2); C = gsr_step_ulps(C0, bk - 2);
}

// ---------------------------------------------------------------------------------------------
// SH -> RGB (K1 step 9) and its backward
// ---------------------------------------------------------------------------------------------
GSR_HD int sh_count(int deg) { return (deg + 1) * (deg + 1); }

GSR_HD void sh_basis(int deg, float x, float y, float z, float b[16]) {
  b[0] = SH_C0;
  if (deg > 0) {
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy);
      b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
      if (deg > 2) {
        b[9] = SH_C3_0 * y * (3.f * xx - yy);
        b[10] = SH_C3_1 * xy * z;
        b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
        b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
        b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
        b[14] = SH_C3_5 * z * (xx - yy);
        b[15] = SH_C3_6 * x * (xx - 3.f * yy);
      }
    }
  }
}

// d(basis_k)/d(x,y,z) for unit-direction components
GSR_HD void sh_basis_grad(int deg, float x, float y, float z, float gx[16], float gy[16], float gz[16]) {
  for (int k = 0; k < 16; ++k) { gx[k] = 0.f; gy[k] = 0.f; gz[k] = 0.f; }
  if (deg > 0) {
    gy[1] = -SH_C1; gz[2] = SH_C1; gx[3] = -SH_C1;
    if (deg > 1) {
      gx[4] = SH_C2_0 * y; gy[4] = SH_C2_0 * x;
      gy[5] = SH_C2_1 * z; gz[5] = SH_C2_1 * y;
      gx[6] = -2.f * SH_C2_2 * x; gy[6] = -2.f * SH_C2_2 * y; gz[6] = 4.f * SH_C2_2 * z;
      gx[7] = SH_C2_3 * z; gz[7] = SH_C2_3 * x;
      gx[8] = 2.f * SH_C2_4 * x; gy[8] = -2.f * SH_C2_4 * y;
      if (deg > 2) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        gx[9] = 6.f * SH_C3_0 * xy;            gy[9] = SH_C3_0 * (3.f * xx - 3.f * yy);
        gx[10] = SH_C3_1 * yz;                 gy[10] = SH_C3_1 * xz;                      gz[10] = SH_C3_1 * xy;
        gx[11] = -2.f * SH_C3_2 * xy;          gy[11] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); gz[11] = 8.f * SH_C3_2 * yz;
        gx[12] = -6.f * SH_C3_3 * xz;          gy[12] = -6.f * SH_C3_3 * yz;               gz[12] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
        gx[13] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); gy[13] = -2.f * SH_C3_4 * xy;       gz[13] = 8.f * SH_C3_4 * xz;
        gx[14] = 2.f * SH_C3_5 * xz;           gy[14] = -2.f * SH_C3_5 * yz;               gz[14] = SH_C3_5 * (xx - yy);
        gx[15] = SH_C3_6 * (3.f * xx - 3.f * yy); gy[15] = -6.f * SH_C3_6 * xy;
      }
    }
  }
}

// sh: this Gaussian's coefficients, [K][3] (coefficient-major, then channel).  Returns clamp bits.
GSR_HD uint32_t sh_to_rgb(int deg, const float* sh, const float p[3], const float cam[3], float rgb[3]) {
  float dx = p[0] - cam[0], dy = p[1] - cam[1], dz = p[2] - cam[2];
  const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  dx *= inv; dy *= inv; dz *= inv;
  float b[16];
  sh_basis(deg, dx, dy, dz, b);
  const int n = sh_count(deg);
  float r = 0.f, g = 0.f, bl = 0.f;
  for (int k = 0; k < n; ++k) { r += b[k] * sh[3 * k]; g += b[k] * sh[3 * k + 1]; bl += b[k] * sh[3 * k + 2]; }
  r += 0.5f; g += 0.5f; bl += 0.5f;
  uint32_t cl = (r < 0.f ? 1u : 0u) | (g < 0.f ? 2u : 0u) | (bl < 0.f ? 4u : 0u);
  rgb[0] = fmaxf(r, 0.f); rgb[1] = fmaxf(g, 0.f); rgb[2] = fmaxf(bl, 0.f);
  return cl;
}

// dL/drgb (already zeroed where clamped) -> dL/dsh [K][3] (written for k < n, zero for the rest up to Kstore)
// and the view-direction path's contribution to dL/dmean (added into dp).
GSR_HD void sh_to_rgb_bwd(int deg, int Kstore, const float* sh, const float p[3], const float cam[3],
                          const float drgb[3], float* dsh, float dp[3]) {
  const float vx = p[0] - cam[0], vy = p[1] - cam[1], vz = p[2] - cam[2];
  const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
  const float x = vx * inv, y = vy * inv, z = vz * inv;
  float b[16], gx[16], gy[16], gz[16];
  sh_basis(deg, x, y, z, b);
  sh_basis_grad(deg, x, y, z, gx, gy, gz);
  const int n = sh_count(deg);
  float ddx = 0.f, ddy = 0.f, ddz = 0.f;
  for (int k = 0; k < n; ++k) {   // coefficient k is read before gradient k is written: dsh may alias sh
    const float s = sh[3 * k] * drgb[0] + sh[3 * k + 1] * drgb[1] + sh[3 * k + 2] * drgb[2];
    dsh[3 * k] = b[k] * drgb[0]; dsh[3 * k + 1] = b[k] * drgb[1]; dsh[3 * k + 2] = b[k] * drgb[2];
    ddx += gx[k] * s; ddy += gy[k] * s; ddz += gz[k] * s;
  }
  for (int k = n; k < Kstore; ++k) { dsh[3 * k] = 0.f; dsh[3 * k + 1] = 0.f; dsh[3 * k + 2] = 0.f; }
  // d = v/|v| :  dL/dv = (dL/dd - d (d . dL/dd)) / |v|
  const float dot = x * ddx + y * ddy + z * ddz;
  dp[0] += (ddx - x * dot) * inv;
  dp[1] += (ddy - y * dot) * inv;
  dp[2] += (ddz - z * dot) * inv;
}

// ---------------------------------------------------------------------------------------------
// Backward of the per-Gaussian geometry (K8 + K9)
// ---------------------------------------------------------------------------------------------
// Inputs: true partials dL/dA, dL/dB, dL/dC of the conic (B counted once), dL/d(ndc.xy) (= the
// extension's dL_dmean2D), and the forward inputs.  Outputs: dL/dmean (added into dp), dL/dcov3D
// packed (dc6), to be pushed further to scale/rotation by cov3d_bwd when those were the inputs.
GSR_HD void project_splat_bwd(const View& v, const float p[3], const float c6[6], double dA, double dB, double dC,
                              float dndcx, float dndcy, float dp[3], float dc6[6]) {
  float t[3];
  view_transform(v, p, t);
  ProjLin pl;
  proj_linear(v, t, pl);
  float a, b, c;
  cov2d_from_M(pl.M, c6, a, b, c);
  // conic = (c, -b, a) / det.  The three sums below cancel to first order for an elongated splat (cov2D ~ l1 u u^T, and
  // dL/dconic ~ K (u_x^2, 2 u_x u_y, u_y^2) from the pixels along its axis: every term is ~ l1^2 K, their sum ~ l1 l2 K),
  // so they are formed in double: float32 products lose l1 / l2 (1e4 for a 100:1 needle) times 6e-8 here, on top of
  // what the summed dL/dconic already carries.  A few dozen double operations per Gaussian, in a memory-bound kernel.
  const double ad = (double)a, bd = (double)b, cd = (double)c;
  const double det = ad * cd - bd * bd;
  // the published backward divides by det^2 + 0.0000001f (GSR_DET_GUARD; 0 = the exact derivative of the inversion); with
  // det >= DILATE^2 = 0.09 the guard changes dL/dcov2D by at most 1.2e-5 relative
  const double d2 = 1.0 / (det * det + (double)GSR_DET_GUARD);
  const float da = (float)((-cd * cd * dA + bd * cd * dB - bd * bd * dC) * d2);
  const float db = (float)((2.0 * bd * cd * dA - (det + 2.0 * bd * bd) * dB + 2.0 * ad * bd * dC) * d2);
  const float dc = (float)((-bd * bd * dA + ad * bd * dB - ad * ad * dC) * d2);
  // D = [[da, db/2],[db/2, dc]];  dL/dSigma3 = M^T D M (symmetric 3x3)
  const float hb = 0.5f * db;
  const float* M = pl.M;
  float DM[6];   // D*M (2x3)
  for (int i = 0; i < 3; ++i) { DM[i] = da * M[i] + hb * M[3 + i]; DM[3 + i] = hb * M[i] + dc * M[3 + i]; }
  float Gs[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Gs[i * 3 + j] = M[i] * DM[j] + M[3 + i] * DM[3 + j];
  dc6[0] = Gs[0]; dc6[1] = 2.f * Gs[1]; dc6[2] = 2.f * Gs[2]; dc6[3] = Gs[4]; dc6[4] = 2.f * Gs[5]; dc6[5] = Gs[8];
  // dL/dM = 2 D M Sigma3  (2x3)
  float dM[6];
  {
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    for (int r = 0; r < 2; ++r)
      for (int j = 0; j < 3; ++j)
        dM[r * 3 + j] = 2.f * (DM[r * 3 + 0] * S[0 * 3 + j] + DM[r * 3 + 1] * S[1 * 3 + j] + DM[r * 3 + 2] * S[2 * 3 + j]);
  }
  // M[a][i] = sum_j J[a][j] Wr[j][i], Wr[j][i] = V[i*4+j]  =>  dL/dJ[a][j] = sum_i dM[a][i] V[i*4+j]
  const float dJ00 = dM[0] * v.V[0] + dM[1] * v.V[4] + dM[2] * v.V[8];
  const float dJ02 = dM[0] * v.V[2] + dM[1] * v.V[6] + dM[2] * v.V[10];
  const float dJ11 = dM[3] * v.V[1] + dM[4] * v.V[5] + dM[5] * v.V[9];
  const float dJ12 = dM[3] * v.V[2] + dM[4] * v.V[6] + dM[5] * v.V[10];
  const float tz = pl.tz, itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
  // clamped axes: t.x (t.y) is treated as a constant
  const float dtx = pl.clx ? 0.f : -v.focal_x * itz2 * dJ02;
  const float dty = pl.cly ? 0.f : -v.focal_y * itz2 * dJ12;
  const float dtz = -v.focal_x * itz2 * dJ00 - v.focal_y * itz2 * dJ11 + 2.f * v.focal_x * pl.tx * itz3 * dJ02 +
                    2.f * v.focal_y * pl.ty * itz3 * dJ12;
  // t_j = sum_i p_i V[i*4+j] + V[12+j]
  for (int i = 0; i < 3; ++i) dp[i] += v.V[i * 4 + 0] * dtx + v.V[i * 4 + 1] * dty + v.V[i * 4 + 2] * dtz;
  // screen position: ndc = hom.xy / (hom.w + 1e-7)
  float h[4];
  for (int j = 0; j < 4; ++j) h[j] = p[0] * v.PV[j] + p[1] * v.PV[4 + j] + p[2] * v.PV[8 + j] + v.PV[12 + j];
  const float w = 1.0f / (h[3] + 1e-7f);
  const float dh0 = dndcx * w, dh1 = dndcy * w;
  const float dh3 = -(dndcx * h[0] + dndcy * h[1]) * w * w;
  for (int i = 0; i < 3; ++i) dp[i] += v.PV[i * 4 + 0] * dh0 + v.PV[i * 4 + 1] * dh1 + v.PV[i * 4 + 3] * dh3;
}

// dL/dcov3D packed -> dL/dscale (3), dL/dq (4, w.r.t. the quaternion as given)
GSR_HD void cov3d_bwd(const float s_in[3], float mod, const float q[4], const float dc6[6], float ds[3], float dq[4]) {
  float R[9];
  quat_to_R(q, R);
  const float s[3] = {mod * s_in[0], mod * s_in[1], mod * s_in[2]};
  // symmetric dL/dSigma as a full matrix (off-diagonals halved back)
  const float G[9] = {dc6[0], 0.5f * dc6[1], 0.5f * dc6[2], 0.5f * dc6[1], dc6[3], 0.5f * dc6[4],
                      0.5f * dc6[2], 0.5f * dc6[4], dc6[5]};
  // L = R diag(s); dL/dL = 2 G L
  float dL[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k)
      dL[i * 3 + k] = 2.f * (G[i * 3 + 0] * R[0 * 3 + k] + G[i * 3 + 1] * R[1 * 3 + k] + G[i * 3 + 2] * R[2 * 3 + k]) * s[k];
  float dR[9];
  for (int k = 0; k < 3; ++k) {
    ds[k] = (dL[0 * 3 + k] * R[0 * 3 + k] + dL[1 * 3 + k] * R[1 * 3 + k] + dL[2 * 3 + k] * R[2 * 3 + k]) * mod;
    for (int i = 0; i < 3; ++i) dR[i * 3 + k] = dL[i * 3 + k] * s[k];
  }
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  dq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
  dq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
  dq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
  dq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}

// ---------------------------------------------------------------------------------------------
// GSR_FLAG_NEEDLE_DOUBLE, backward: project_splat_bwd + cov3d_bwd for ONE needle in double, end to end.  Differentiating
// the double chain's conic through float32 M / Sigma / quaternion products is worse than either consistent chain (a 440:1
// needle's rotation gradient 0.6 % off where the all-float32 backward is 0.05 % off: tests/diag_aniso_elem.py, seed 41),
// so a needle's whole per-Gaussian chain rule runs in double on the float32 inputs, like its forward (cov2d_accurate).
// sc / q: the scales and the quaternion AS USED by the forward (activated); c6pre: the precomputed covariance instead.
// dp: dL/dmean, added to; ds / dq: dL/d(sc), dL/d(q as used); dS6: dL/dSigma (packed, off-diagonals doubled) for c6pre.
// ---------------------------------------------------------------------------------------------
GSR_HD void needle_bwd_d(const View& v, const float p[3], const float* sc, float mod, const float* q, const float* c6pre,
                         double dA, double dB, double dC, double dndcx, double dndcy, double dp[3], double ds[3], double dq[4],
                         double dS6[6]) {
  double S[6], R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, s3[3] = {0, 0, 0};
  if (c6pre) {
    for (int i = 0; i < 6; ++i) S[i] = (double)c6pre[i];
  } else {
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - r * z);       R[2] = 2.0 * (x * z + r * y);
    R[3] = 2.0 * (x * y + r * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - r * x);
    R[6] = 2.0 * (x * z - r * y);       R[7] = 2.0 * (y * z + r * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int k = 0; k < 3; ++k) s3[k] = (double)mod * sc[k];
    double L[9];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) L[i * 3 + k] = R[i * 3 + k] * s3[k];
    S[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2]; S[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
    S[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8]; S[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
    S[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8]; S[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
  }
  double t[3];
  for (int j = 0; j < 3; ++j)
    t[j] = (double)p[0] * v.V[j] + (double)p[1] * v.V[4 + j] + (double)p[2] * v.V[8 + j] + (double)v.V[12 + j];
  const double limx = (double)FOV_CLAMP * v.tanfovx, limy = (double)FOV_CLAMP * v.tanfovy;
  const double tz = t[2], txtz = t[0] / tz, tytz = t[1] / tz;
  const bool clx = (txtz < -limx) || (txtz > limx), cly = (tytz < -limy) || (tytz > limy);
  const double tx = (txtz < -limx ? -limx : (txtz > limx ? limx : txtz)) * tz;
  const double ty = (tytz < -limy ? -limy : (tytz > limy ? limy : tytz)) * tz;
  const double fx = (double)v.W / (2.0 * (double)v.tanfovx), fy = (double)v.H / (2.0 * (double)v.tanfovy);
  const double J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  double M[6];
  for (int i = 0; i < 3; ++i) {
    M[i] = J00 * v.V[i * 4 + 0] + J02 * v.V[i * 4 + 2];
    M[3 + i] = J11 * v.V[i * 4 + 1] + J12 * v.V[i * 4 + 2];
  }
  const double Sf[9] = {S[0], S[1], S[2], S[1], S[3], S[4], S[2], S[4], S[5]};
  double SM0[3], SM1[3];                                       // Sigma M0^T, Sigma M1^T
  for (int i = 0; i < 3; ++i) {
    SM0[i] = Sf[i * 3] * M[0] + Sf[i * 3 + 1] * M[1] + Sf[i * 3 + 2] * M[2];
    SM1[i] = Sf[i * 3] * M[3] + Sf[i * 3 + 1] * M[4] + Sf[i * 3 + 2] * M[5];
  }
  const double a = M[0] * SM0[0] + M[1] * SM0[1] + M[2] * SM0[2] + (double)DILATE;
  const double b = M[3] * SM0[0] + M[4] * SM0[1] + M[5] * SM0[2];
  const double c = M[3] * SM1[0] + M[4] * SM1[1] + M[5] * SM1[2] + (double)DILATE;
  const double det = a * c - b * b, d2 = 1.0 / (det * det + (double)GSR_DET_GUARD);
  const double da = (-c * c * dA + b * c * dB - b * b * dC) * d2;
  const double db = (2.0 * b * c * dA - (det + 2.0 * b * b) * dB + 2.0 * a * b * dC) * d2;
  const double dc = (-b * b * dA + a * b * dB - a * a * dC) * d2;
  const double hb = 0.5 * db;
  double DM[6];
  for (int i = 0; i < 3; ++i) { DM[i] = da * M[i] + hb * M[3 + i]; DM[3 + i] = hb * M[i] + dc * M[3 + i]; }
  double G[9];                                                 // dL/dSigma3 = M^T D M (full symmetric matrix)
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) G[i * 3 + j] = M[i] * DM[j] + M[3 + i] * DM[3 + j];
  dS6[0] = G[0]; dS6[1] = 2.0 * G[1]; dS6[2] = 2.0 * G[2]; dS6[3] = G[4]; dS6[4] = 2.0 * G[5]; dS6[5] = G[8];
  double dM[6];                                                // dL/dM = 2 D M Sigma3
  for (int r = 0; r < 2; ++r)
    for (int j = 0; j < 3; ++j)
      dM[r * 3 + j] = 2.0 * (DM[r * 3 + 0] * Sf[0 * 3 + j] + DM[r * 3 + 1] * Sf[1 * 3 + j] + DM[r * 3 + 2] * Sf[2 * 3 + j]);
  const double dJ00 = dM[0] * v.V[0] + dM[1] * v.V[4] + dM[2] * v.V[8];
  const double dJ02 = dM[0] * v.V[2] + dM[1] * v.V[6] + dM[2] * v.V[10];
  const double dJ11 = dM[3] * v.V[1] + dM[4] * v.V[5] + dM[5] * v.V[9];
  const double dJ12 = dM[3] * v.V[2] + dM[4] * v.V[6] + dM[5] * v.V[10];
  const double itz = 1.0 / tz, itz2 = itz * itz, itz3 = itz2 * itz;
  const double dtx = clx ? 0.0 : -fx * itz2 * dJ02;            // clamped axes: t.x (t.y) is treated as a constant
  const double dty = cly ? 0.0 : -fy * itz2 * dJ12;
  const double dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + 2.0 * fx * tx * itz3 * dJ02 + 2.0 * fy * ty * itz3 * dJ12;
  for (int i = 0; i < 3; ++i) dp[i] += v.V[i * 4 + 0] * dtx + v.V[i * 4 + 1] * dty + v.V[i * 4 + 2] * dtz;
  double h[4];
  for (int j = 0; j < 4; ++j)
    h[j] = (double)p[0] * v.PV[j] + (double)p[1] * v.PV[4 + j] + (double)p[2] * v.PV[8 + j] + (double)v.PV[12 + j];
  const double w = 1.0 / (h[3] + (double)1e-7f);
  const double dh0 = dndcx * w, dh1 = dndcy * w, dh3 = -(dndcx * h[0] + dndcy * h[1]) * w * w;
  for (int i = 0; i < 3; ++i) dp[i] += v.PV[i * 4 + 0] * dh0 + v.PV[i * 4 + 1] * dh1 + v.PV[i * 4 + 3] * dh3;
  ds[0] = ds[1] = ds[2] = 0.0; dq[0] = dq[1] = dq[2] = dq[3] = 0.0;
  if (c6pre) return;
  // Sigma = L L^T, L = R diag(s): dL/dL = 2 G L; ds_k = sum_i dL[i][k] R[i][k] * mod; dR[i][k] = dL[i][k] s_k
  double dR[9];
  for (int k = 0; k < 3; ++k) {
    double acc = 0.0;
    for (int i = 0; i < 3; ++i) {
      const double dL = 2.0 * (G[i * 3 + 0] * R[0 * 3 + k] + G[i * 3 + 1] * R[1 * 3 + k] + G[i * 3 + 2] * R[2 * 3 + k]) * s3[k];
      acc += dL * R[i * 3 + k];
      dR[i * 3 + k] = dL * s3[k];
    }
    ds[k] = acc * (double)mod;
  }
  const double r = q[0], x = q[1], y = q[2], z = q[3];
  dq[0] = 2.0 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
  dq[1] = 2.0 * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0 * x * dR[8]);
  dq[2] = 2.0 * (-2.0 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0 * y * dR[8]);
  dq[3] = 2.0 * (-2.0 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}

// ---------------------------------------------------------------------------------------------
// Activation getters of the reference's GaussianModel (scene/gaussian_model.py:31-39) and their chain rule,
// for the fused raw-parameter entry points: sigmoid (opacity), exp (scaling, inline at the call sites) and
// torch.nn.functional.normalize (rotation: v / max(||v||, 1e-12)).
// ---------------------------------------------------------------------------------------------
GSR_HD float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

GSR_HD void act_normalize4(const float r[4], float q[4], float& inv_n) {
  GSR_FP_STRICT
  const float n = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  inv_n = 1.0f / fmaxf(n, 1e-12f);
  const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
  q[0] = r0 * inv_n; q[1] = r1 * inv_n; q[2] = r2 * inv_n; q[3] = r3 * inv_n;
}

// q = normalised quaternion, inv_n = 1/||raw||, dq = dL/dq  ->  dr = dL/d(raw) = (dq - q (q.dq)) / ||raw||
GSR_HD void act_normalize4_bwd(const float q[4], float inv_n, const float dq[4], float dr[4]) {
  const float dot = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
  const float d0 = dq[0], d1 = dq[1], d2 = dq[2], d3 = dq[3];
  dr[0] = (d0 - q[0] * dot) * inv_n; dr[1] = (d1 - q[1] * dot) * inv_n;
  dr[2] = (d2 - q[2] * dot) * inv_n; dr[3] = (d3 - q[3] * dot) * inv_n;
}

// ---------------------------------------------------------------------------------------------
// Exact-footprint test for one (tile, Gaussian) pair.  A pixel contributes only if
// alpha = o*exp(-Q/2) >= 1/255 with Q = A dx^2 + 2 B dx dy + C dy^2, i.e. Q <= tau = 2 ln(255 o).
// Returns false only when NO pixel centre of the rectangle [x0,x1]x[y0,y1] can satisfy that: the minimum
// of the convex Q over the continuous rectangle (0 if the centre is inside, else attained on one of the
// four edges) is compared with tau widened by a margin far above float32 rounding of the per-pixel test.
// ---------------------------------------------------------------------------------------------
GSR_HD float gsr_clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

GSR_HD bool tile_can_contribute(float cx, float cy, float A, float B, float C, float o, float x0, float y0, float x1,
                                float y1) {
  if (!(A > 0.f) || !(C > 0.f)) return true;          // not a proper conic: leave the pair alone
  const float tau = 2.0f * logf(255.0f * o);          // -inf / NaN for o <= 0
  const float bound = tau + 1e-4f * fabsf(tau) + 1e-3f;
  if (!(bound >= 0.f)) return false;                  // opacity below 1/255: alpha never reaches the floor
  const float dxl = x0 - cx, dxh = x1 - cx, dyl = y0 - cy, dyh = y1 - cy;
  if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return true;
  float qmin = 3.0e38f;
  const float ex[2] = {dxl, dxh}, ey[2] = {dyl, dyh};
  for (int i = 0; i < 2; ++i) {
    const float dx = ex[i];
    const float dy = gsr_clampf(-B * dx / C, dyl, dyh);
    qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * dy + C * dy * dy);
    const float dy2 = ey[i];
    const float dx2 = gsr_clampf(-B * dy2 / A, dxl, dxh);
    qmin = fminf(qmin, A * dx2 * dx2 + 2.f * B * dx2 * dy2 + C * dy2 * dy2);
  }
  return qmin <= bound;
}

// Shrinks a splat's tile rect [rmin, rmax) -- the reference's square of ceil(3 sigma) around the centre -- to the tiles
// that the axis-aligned bounding box of its alpha >= 1/255 footprint touches: Q(d) <= tau bounds |dx| by
// sqrt(tau C / (A C - B^2)) and |dy| by sqrt(tau A / (A C - B^2)) (the extents of the ellipse Q = tau), with the same
// widened tau as the per-tile tests above plus a relative and an absolute margin on the extents.  Every (tile, Gaussian)
// pair this drops would have been dropped by strip_masks4 at emission anyway (its mask is 0): the pair list, the image and
// the gradients are unchanged, only fewer slots are emitted, sorted and numbered.  An opacity below 1/255 empties the rect.
GSR_HD void tighten_rect(float px, float py, float A, float B, float C, float o, int gridx, int gridy, int& rminx,
                         int& rminy, int& rmaxx, int& rmaxy) {
  const float tau = 2.0f * logf(255.0f * o);
  const float bound = tau + 1e-4f * fabsf(tau) + 1e-3f;
  if (!(bound >= 0.f)) { rmaxx = rminx; rmaxy = rminy; return; }    // alpha never reaches the floor (also NaN opacity)
  const float det = A * C - B * B;
  if (!(A > 0.f) || !(C > 0.f) || !(det > 0.f)) return;            // not a proper conic: leave the rect alone
  const float ex = sqrtf(bound * C / det) * 1.001f + 0.01f, ey = sqrtf(bound * A / det) * 1.001f + 0.01f;
  if (!(ex < 1.0e9f) || !(ey < 1.0e9f)) return;
  const float inv = 1.0f / (float)TILE;
  const int x0 = trunc_clamp((px - ex) * inv, gridx), y0 = trunc_clamp((py - ey) * inv, gridy);
  const int x1 = (px + ex < 0.f) ? 0 : trunc_clamp((px + ex) * inv + 1.0f, gridx);
  const int y1 = (py + ey < 0.f) ? 0 : trunc_clamp((py + ey) * inv + 1.0f, gridy);
  rminx = x0 > rminx ? x0 : rminx; rminy = y0 > rminy ? y0 : rminy;
  rmaxx = x1 < rmaxx ? x1 : rmaxx; rmaxy = y1 < rmaxy ? y1 : rmaxy;
  if (rmaxx < rminx) rmaxx = rminx;
  if (rmaxy < rminy) rmaxy = rminy;
}

// The same test for the four 16x4 strips of a tile at once (strip k = pixel rows y0 + 4k .. y0 + 4k + 3, clipped to
// the image height): bit k of the result is set when strip k can be reached.  What is shared between the strips --
// the threshold, the two column offsets, the two reciprocals that place the minimum on an edge -- is computed once
// (k_emit calls this per (tile, Gaussian) pair: four independent calls cost four logf and sixteen divisions).
GSR_HD uint32_t strip_masks4(float cx, float cy, float A, float B, float C, float o, float x0, float x1, float y0,
                             float ymax) {
  if (!(A > 0.f) || !(C > 0.f)) {                       // not a proper conic: leave the pair alone
    uint32_t m = 0;
    for (int k = 0; k < 4; ++k) if (y0 + 4.f * (float)k <= ymax) m |= 1u << k;
    return m;
  }
  const float tau = 2.0f * GSR_LOG_FAST(255.0f * o);
  const float bound = tau + 1e-4f * fabsf(tau) + 1e-3f;
  if (!(bound >= 0.f)) return 0u;
  const float dxl = x0 - cx, dxh = x1 - cx;
  const bool inx = dxl <= 0.f && dxh >= 0.f;
  const float invC = 1.0f / C, invA = 1.0f / A;
  const float tyl = -B * dxl * invC, tyh = -B * dxh * invC;     // unconstrained minimiser of Q along the two columns
  const float qxl = A * dxl * dxl, qxh = A * dxh * dxh;
  uint32_t mask = 0;
  for (int k = 0; k < 4; ++k) {
    const float ya = y0 + 4.f * (float)k;
    if (!(ya <= ymax)) break;
    const float yb = fminf(ya + 3.0f, ymax);
    const float dyl = ya - cy, dyh = yb - cy;
    bool hit = inx && dyl <= 0.f && dyh >= 0.f;
    if (!hit) {
      const float d0 = gsr_clampf(tyl, dyl, dyh), d1 = gsr_clampf(tyh, dyl, dyh);
      float qmin = fminf(qxl + (2.f * B * dxl + C * d0) * d0, qxh + (2.f * B * dxh + C * d1) * d1);
      const float e0 = gsr_clampf(-B * dyl * invA, dxl, dxh), e1 = gsr_clampf(-B * dyh * invA, dxl, dxh);
      qmin = fminf(qmin, fminf((A * e0 + 2.f * B * dyl) * e0 + C * dyl * dyl, (A * e1 + 2.f * B * dyh) * e1 + C * dyh * dyh));
      hit = qmin <= bound;
    }
    if (hit) mask |= 1u << k;
  }
  return mask;
}

// ---------------------------------------------------------------------------------------------
// Per-(pixel, Gaussian) blend weight (K6 inner step).  Returns false when the entry is skipped.
// ---------------------------------------------------------------------------------------------
GSR_HD bool splat_alpha(float dx, float dy, float A, float B, float C, float o, float& alpha, float& G) {
  const float power = -0.5f * (A * dx * dx + C * dy * dy) - B * dx * dy;
  if (power > 0.0f) return false;
  G = GSR_EXP(power);
  alpha = fminf(ALPHA_CAP, o * G);
  return alpha >= ALPHA_MIN;
}

}  // namespace gsr
