// TEST HARNESS ONLY (built by tests with g++): runs the per-Gaussian arithmetic of csrc/gsr_math.h on the
// host so that the exact source the HIP kernels compile is checked against oracle/oracle_r.py without a GPU.
// Not part of the product: nothing in the package loads this library.
#include "gsr_math.h"

using namespace gsr;

extern "C" {

// geom out [P,12]: px,py,depth,A,B,C,radius,rminx,rminy,rmaxx,rmaxy,clampbits ; rgb out [P,3]
void hm_preprocess(int P, int K, int H, int W, float tanfovx, float tanfovy, float mod, int deg, const float* vm,
                   const float* pm, const float* campos, const float* means, const float* scales, const float* rots,
                   const float* cov3d_precomp, const float* sh, float* geom, float* rgb) {
  View v;
  make_view(v, vm, pm, campos, H, W, tanfovx, tanfovy, mod, deg);
  for (int g = 0; g < P; ++g) {
    float c6[6];
    if (cov3d_precomp) for (int i = 0; i < 6; ++i) c6[i] = cov3d_precomp[6 * g + i];
    else cov3d_from_scale_rot(scales + 3 * g, mod, rots + 4 * g, c6);
    Splat s;
    bool ok = project_splat(v, means + 3 * g, c6, s);
    float* o = geom + 12 * g;
    for (int i = 0; i < 12; ++i) o[i] = 0.f;
    rgb[3 * g] = rgb[3 * g + 1] = rgb[3 * g + 2] = 0.f;
    if (!ok) continue;
    uint32_t cl = sh_to_rgb(deg, sh + (size_t)g * K * 3, means + 3 * g, v.cam, rgb + 3 * g);
    o[0] = s.px; o[1] = s.py; o[2] = s.depth; o[3] = s.A; o[4] = s.B; o[5] = s.C; o[6] = (float)s.radius;
    o[7] = (float)s.rminx; o[8] = (float)s.rminy; o[9] = (float)s.rmaxx; o[10] = (float)s.rmaxy; o[11] = (float)cl;
  }
}

// upstream [P,8]: dA,dB,dC (true partials),dndcx,dndcy,drgb[3]; valid[P]; clamp bits [P]
void hm_preprocess_bwd(int P, int K, int H, int W, float tanfovx, float tanfovy, float mod, int deg, const float* vm,
                       const float* pm, const float* campos, const float* means, const float* scales,
                       const float* rots, const float* cov3d_precomp, const float* sh, const float* upstream,
                       const int* valid, const int* clampbits, float* dmeans, float* dscales, float* drots,
                       float* dcov3d, float* dsh) {
  View v;
  make_view(v, vm, pm, campos, H, W, tanfovx, tanfovy, mod, deg);
  for (int g = 0; g < P; ++g) {
    float dp[3] = {0, 0, 0};
    float* dshg = dsh + (size_t)g * K * 3;
    if (!valid[g]) {
      for (int i = 0; i < 3; ++i) dmeans[3 * g + i] = 0.f;
      if (dscales) { for (int i = 0; i < 3; ++i) dscales[3 * g + i] = 0.f; for (int i = 0; i < 4; ++i) drots[4 * g + i] = 0.f; }
      if (dcov3d) for (int i = 0; i < 6; ++i) dcov3d[6 * g + i] = 0.f;
      for (int i = 0; i < K * 3; ++i) dshg[i] = 0.f;
      continue;
    }
    const float* u = upstream + 8 * g;
    float c6[6];
    if (cov3d_precomp) for (int i = 0; i < 6; ++i) c6[i] = cov3d_precomp[6 * g + i];
    else cov3d_from_scale_rot(scales + 3 * g, mod, rots + 4 * g, c6);
    float drgb[3] = {u[5], u[6], u[7]};
    for (int c = 0; c < 3; ++c) if (clampbits[g] & (1 << c)) drgb[c] = 0.f;
    sh_to_rgb_bwd(deg, K, sh + (size_t)g * K * 3, means + 3 * g, v.cam, drgb, dshg, dp);
    float dc6[6];
    project_splat_bwd(v, means + 3 * g, c6, u[0], u[1], u[2], u[3], u[4], dp, dc6);
    if (dcov3d) for (int i = 0; i < 6; ++i) dcov3d[6 * g + i] = dc6[i];
    if (dscales) cov3d_bwd(scales + 3 * g, mod, rots + 4 * g, dc6, dscales + 3 * g, drots + 4 * g);
    for (int i = 0; i < 3; ++i) dmeans[3 * g + i] = dp[i];
  }
}

// geo [n,6]: cx,cy,A,B,C,opacity ; out[n] = tile_can_contribute over the pixel-centre rectangle
void hm_tile_can_contribute(int n, const float* geo, float x0, float y0, float x1, float y1, int* out) {
  for (int i = 0; i < n; ++i) {
    const float* g = geo + 6 * i;
    out[i] = tile_can_contribute(g[0], g[1], g[2], g[3], g[4], g[5], x0, y0, x1, y1) ? 1 : 0;
  }
}

// geo [n,6] as above; out[n] = strip_masks4 of the tile whose first pixel centre is (x0, y0), image height ymax + 1
void hm_strip_masks4(int n, const float* geo, float x0, float x1, float y0, float ymax, int* out) {
  for (int i = 0; i < n; ++i) {
    const float* g = geo + 6 * i;
    out[i] = (int)strip_masks4(g[0], g[1], g[2], g[3], g[4], g[5], x0, x1, y0, ymax);
  }
}

// geo [n,6] as above; rect_in [4] = rminx, rminy, rmaxx, rmaxy (tiles); out [n,4] = the tightened rect
void hm_tighten_rect(int n, const float* geo, int gridx, int gridy, const int* rect_in, int* out) {
  for (int i = 0; i < n; ++i) {
    const float* g = geo + 6 * i;
    int x0 = rect_in[0], y0 = rect_in[1], x1 = rect_in[2], y1 = rect_in[3];
    tighten_rect(g[0], g[1], g[2], g[3], g[4], g[5], gridx, gridy, x0, y0, x1, y1);
    out[4 * i] = x0; out[4 * i + 1] = y0; out[4 * i + 2] = x1; out[4 * i + 3] = y1;
  }
}

// GSR_FLAG_NEEDLE_DOUBLE, one Gaussian (scales + quaternion as used): conic_d [3] = the double chain's conic,
// conic_f [3] = needle_conic_to_float's triple, needle [1], and for dL/dconic = (dA, dB, dC): out_f [10] = dp, ds, dq of the
// float32 chain rule (project_splat_bwd + cov3d_bwd), out_d [10] = needle_bwd_d's.
void hm_needle(int H, int W, float tanfovx, float tanfovy, float mod, const float* vm, const float* pm, const float* campos,
               const float* mean, const float* scale, const float* rot, double dA, double dB, double dC, double* conic_d,
               float* conic_f, int* needle, float* out_f, double* out_d) {
  View v;
  make_view(v, vm, pm, campos, H, W, tanfovx, tanfovy, mod, 0);
  float c6[6];
  cov3d_from_scale_rot(scale, mod, rot, c6);
  Splat s;
  project_splat(v, mean, c6, s);
  needle[0] = is_needle(s.ca, s.cb, s.cc) ? 1 : 0;
  double a, b, c;
  cov2d_accurate(v, mean, scale, mod, rot, nullptr, a, b, c);
  const double dinv = 1.0 / (a * c - b * b);
  conic_d[0] = c * dinv; conic_d[1] = -b * dinv; conic_d[2] = a * dinv;
  needle_conic_to_float(a, b, c, conic_f[0], conic_f[1], conic_f[2]);
  float dp[3] = {0.f, 0.f, 0.f}, dc6[6], ds[3], dq[4];
  project_splat_bwd(v, mean, c6, dA, dB, dC, 0.f, 0.f, dp, dc6);
  cov3d_bwd(scale, mod, rot, dc6, ds, dq);
  for (int i = 0; i < 3; ++i) { out_f[i] = dp[i]; out_f[3 + i] = ds[i]; }
  for (int i = 0; i < 4; ++i) out_f[6 + i] = dq[i];
  double dpd[3] = {0.0, 0.0, 0.0}, dsd[3], dqd[4], dS6[6];
  needle_bwd_d(v, mean, scale, mod, rot, nullptr, dA, dB, dC, 0.0, 0.0, dpd, dsd, dqd, dS6);
  for (int i = 0; i < 3; ++i) { out_d[i] = dpd[i]; out_d[3 + i] = dsd[i]; }
  for (int i = 0; i < 4; ++i) out_d[6 + i] = dqd[i];
}

// needle_conic_to_float on a given dilated 2D covariance (a, b, c): out [3] = the float32 triple, outd [3] = the double conic
void hm_conic_to_float(double a, double b, double c, float* out, double* outd) {
  needle_conic_to_float(a, b, c, out[0], out[1], out[2]);
  const double dinv = 1.0 / (a * c - b * b);
  outd[0] = c * dinv; outd[1] = -b * dinv; outd[2] = a * dinv;
}

}  // extern "C"
