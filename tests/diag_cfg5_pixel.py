"""Diagnostic (CPU only): why a solid pixel of S-airport-4K is further from float64 than the float32
oracle is.  For one tile, the kernels' own float32 per-Gaussian arithmetic (csrc/gsr_math.h through tests/host_math) and
oracle-R's float32 arithmetic are each compared with oracle-R float64 -- pixel centre, conic, and what their errors do to
alpha at one pixel -- and the pixel is composited in float64 from each of the three geometries.

    python tests/diag_cfg5_pixel.py [px py]      (default: the worst solid pixel of profiles/r05_fullsize_sweep.txt)
"""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splat-attack_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_r as O  # noqa: E402
from util import settings_for  # noqa: E402
from gsplat_attack.scenes import make_scene  # noqa: E402

PX, PY = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3484, 437)
HM = os.path.join(ROOT, "tests", "host_math")
so = os.path.join(HM, "libhostmath.so")
subprocess.run(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "3d-gaussian-splat-attack_amd", "csrc"),
                os.path.join(HM, "host_math.cpp"), "-o", so], check=True)
lib = ctypes.CDLL(so)
ref, cams, _ = make_scene("airport-4K", device="cpu", n_views=1)
cam = cams[0]
H, W = cam.image_height, cam.image_width
st = settings_for(cam, torch.tensor([0.2, 0.1, 0.0]), 3, 1.0)
with torch.no_grad():
    xyz, sc, ro, op = ref.get_xyz.detach(), ref.get_scaling.detach(), ref.get_rotation.detach(), ref.get_opacity.detach().view(-1)
    g64 = O.preprocess(xyz.double(), sc.double(), ro.double(), None, st)
    g32 = O.preprocess(xyz.float(), sc.float(), ro.float(), None, st)
    tx, ty = PX // 16, PY // 16
    gid, ranges, ghost = O.build_tile_lists(g64, H, W, [(tx, ty, tx + 1, ty + 1)])
    gx = (W + 15) // 16
    s, e = (int(v) for v in ranges[ty * gx + tx])
    ids = gid[s:e]
    n = ids.numel()
    f32 = lambda t: np.ascontiguousarray(t.numpy().astype(np.float32))
    geom = np.zeros((n, 12), np.float32)
    rgbh = np.zeros((n, 3), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    means, scs, rts, shs = f32(xyz[ids]), f32(sc[ids]), f32(ro[ids]), f32(ref.get_features.detach()[ids])
    lib.hm_preprocess(n, 16, H, W, ctypes.c_float(st.tanfovx), ctypes.c_float(st.tanfovy), ctypes.c_float(1.0), 3,
                      vp(f32(st.viewmatrix)), vp(f32(st.projmatrix)), vp(f32(st.campos)), vp(means), vp(scs), vp(rts), None,
                      vp(shs), vp(geom), vp(rgbh))
    rgb64, _ = O.sh_to_rgb(3, ref.get_features.detach().double()[ids], xyz.double()[ids], st.campos.double())
    o = op[ids].double()

    def composite(pxy, conic, tag):
        dx, dy = pxy[:, 0] - PX, pxy[:, 1] - PY
        power = -0.5 * (conic[:, 0] * dx * dx + conic[:, 2] * dy * dy) - conic[:, 1] * dx * dy
        alpha = torch.clamp_max(o * torch.exp(power), 0.99)
        ok = (power <= 0) & (alpha >= 1.0 / 255.0)
        a = torch.where(ok, alpha, torch.zeros_like(alpha))
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - a[:-1]]), 0)
        w = a * T
        col = (w[:, None] * rgb64).sum(0) + (T[-1] * (1 - a[-1])) * st.bg.double()[:3]
        return col, a, w
    xy64, con64 = g64.xy[ids], g64.conic[ids]
    xy32, con32 = g32.xy[ids].double(), g32.conic[ids].double()
    xyh = torch.from_numpy(geom[:, 0:2].astype(np.float64))
    conh = torch.from_numpy(geom[:, 3:6].astype(np.float64))
    c64, a64, w64 = composite(xy64, con64, "f64")
    c32, a32, _ = composite(xy32, con32, "oracle f32 geometry")
    ch, ah, _ = composite(xyh, conh, "kernel f32 geometry")
    print(f"pixel ({PX},{PY}), tile ({tx},{ty}), list {n}; float64 colour {c64.tolist()}")
    print(f"  composited in float64 from the float32 ORACLE's geometry: err {float((c32 - c64).abs().max()):.2e}")
    print(f"  composited in float64 from the KERNELS' float32 geometry:  err {float((ch - c64).abs().max()):.2e}")
    print(f"  pixel-centre error (px), max over the list: oracle f32 {float((xy32 - xy64).abs().max()):.2e}, kernels {float((xyh - xy64).abs().max()):.2e}; "
          f"ulp of float32 at x = {PX}: {np.spacing(np.float32(PX)):.2e}")
    print(f"  conic relative error, max: oracle f32 {float(((con32 - con64).abs() / con64.abs().clamp_min(1e-30)).max()):.2e}, "
          f"kernels {float(((conh - con64).abs() / con64.abs().clamp_min(1e-30)).max()):.2e}")
    k = int((w64 * (ah - a64).abs() / a64.clamp_min(1e-12)).argmax())
    big = torch.argsort(-(ah - a64).abs() * w64 / a64.clamp_min(1e-12))[:4]
    for k in big.tolist():
        print(f"  entry {k}: weight {float(w64[k]):.3f}, alpha64 {float(a64[k]):.5f}, alpha from kernel geometry {float(ah[k]):.5f} "
              f"(oracle f32 {float(a32[k]):.5f}); centre error kernels ({float(xyh[k, 0] - xy64[k, 0]):+.1e}, {float(xyh[k, 1] - xy64[k, 1]):+.1e}) "
              f"oracle f32 ({float(xy32[k, 0] - xy64[k, 0]):+.1e}, {float(xy32[k, 1] - xy64[k, 1]):+.1e}); conic A,B,C {[round(float(v), 3) for v in con64[k]]}, "
              f"dx, dy ({float(xy64[k, 0] - PX):+.2f}, {float(xy64[k, 1] - PY):+.2f}); view depth {float(g64.depth[ids][k]):.1f}")
