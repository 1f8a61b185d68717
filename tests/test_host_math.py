"""The scalar math the HIP kernels compile (csrc/gsr_math.h) run on the HOST through a g++ test harness
(tests/host_math) and compared with oracle-R: forward geometry / colour, the hand-derived backward, and the
conservativeness of the per-tile footprint test.  No GPU, no product path involved."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from gsplat_attack.cameras import look_at_camera
from gsplat_attack.scenes import make_scene
from oracle import oracle_r as O
from util import settings_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HM = os.path.join(ROOT, "tests", "host_math")


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(HM, "libhostmath.so")
    src = os.path.join(HM, "host_math.cpp")
    hdr = os.path.join(ROOT, "3d-gaussian-splat-attack_amd", "csrc", "gsr_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.dirname(hdr), src, "-o", so],
                       check=True)
    return ctypes.CDLL(so)


def f32(x):
    return np.ascontiguousarray(x.detach().numpy().astype(np.float32))


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def cf(v):
    return ctypes.c_float(float(v))


def _case(precomp_cov: bool):
    model, _, _ = make_scene("hydrant-1k", n_views=1)
    # a close camera: near-plane culls, the 1.3*tanfov clamp and image-border rect clamping all occur
    cam = look_at_camera((0.3, -0.1, -0.6), (0.0, 0.0, 0.0), fovx=0.8, fovy=0.8, width=128, height=112)
    st = settings_for(cam, torch.zeros(3), scale_modifier=1.0 if precomp_cov else 1.3)
    return model, cam, st


@pytest.mark.parametrize("precomp_cov", [False, True])
def test_per_gaussian_forward_and_backward(lib, precomp_cov):
    model, cam, st = _case(precomp_cov)
    P, K, H, W = 1000, 16, cam.image_height, cam.image_width
    means, scales, rots, sh = f32(model.get_xyz), f32(model.get_scaling), f32(model.get_rotation), f32(model.get_features)
    cov = f32(model.get_covariance(1.0)) if precomp_cov else None
    vm, pm, cp = f32(st.viewmatrix), f32(st.projmatrix), f32(st.campos)
    geom = np.zeros((P, 12), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    lib.hm_preprocess(P, K, H, W, cf(st.tanfovx), cf(st.tanfovy), cf(st.scale_modifier), 3, ptr(vm), ptr(pm), ptr(cp),
                      ptr(means), None if precomp_cov else ptr(scales), None if precomp_cov else ptr(rots), ptr(cov),
                      ptr(sh), ptr(geom), ptr(rgb))
    dt = torch.float64
    m3 = torch.tensor(means, dtype=dt, requires_grad=True)
    shs = torch.tensor(sh, dtype=dt, requires_grad=True)
    m2 = torch.zeros(P, 3, dtype=dt, requires_grad=True)
    if precomp_cov:
        c6 = torch.tensor(cov, dtype=dt, requires_grad=True)
        sc = ro = None
    else:
        sc = torch.tensor(scales, dtype=dt, requires_grad=True)
        ro = torch.tensor(rots, dtype=dt, requires_grad=True)
        c6 = None
    g = O.preprocess(m3, sc, ro, c6, st, m2)
    col, clamped = O.sh_to_rgb(3, shs, m3, st.campos.to(dt))
    valid = g.valid.numpy()
    solid = valid & ~g.fragile.numpy()
    assert 200 < valid.sum() < P                       # the case really culls some and keeps many
    assert ((geom[:, 6] > 0) == valid)[~g.fragile.numpy()].all()
    assert (g.radii.numpy()[solid] == geom[solid, 6]).all()
    assert np.abs(g.xy.detach().numpy()[solid] - geom[solid, 0:2]).max() < 2e-3
    conic = g.conic.detach().numpy()[solid]
    assert (np.abs(conic - geom[solid, 3:6]) / np.abs(conic).max(axis=1, keepdims=True)).max() < 1e-4
    rect = torch.cat([g.rect_min, g.rect_max], 1).numpy()
    assert (rect[solid] == geom[solid, 7:11]).all()
    assert np.abs(col.detach().numpy()[solid] - rgb[solid]).max() < 2e-5
    bits = (clamped.numpy() * np.array([1, 2, 4])).sum(1)
    near0 = (np.abs(col.detach().numpy()) < 1e-6).any(axis=1) & (bits == 0)
    assert (bits[solid & ~near0] == geom[solid & ~near0, 11]).all()

    # ---- backward: random upstream gradients on (conic, ndc, rgb) -------------------------------------
    u = torch.randn(P, 8, generator=torch.Generator().manual_seed(5), dtype=dt)
    vm_ = torch.tensor(valid)
    ndcx = (2 * g.xy[:, 0] + 1) / W - 1
    ndcy = (2 * g.xy[:, 1] + 1) / H - 1
    loss = ((g.conic * u[:, 0:3]).sum(1)[vm_]).sum() + ((col * u[:, 5:8]).sum(1)[vm_]).sum() \
        + ((ndcx * u[:, 3] + ndcy * u[:, 4])[vm_]).sum()
    loss.backward()
    up = f32(u)
    dmeans = np.zeros((P, 3), np.float32)
    dsh = np.zeros((P, 16, 3), np.float32)
    dsc = np.zeros((P, 3), np.float32)
    dro = np.zeros((P, 4), np.float32)
    dcov = np.zeros((P, 6), np.float32)
    va = np.ascontiguousarray((geom[:, 6] > 0).astype(np.int32))
    cb = np.ascontiguousarray(geom[:, 11].astype(np.int32))
    lib.hm_preprocess_bwd(P, K, H, W, cf(st.tanfovx), cf(st.tanfovy), cf(st.scale_modifier), 3, ptr(vm), ptr(pm), ptr(cp),
                          ptr(means), None if precomp_cov else ptr(scales), None if precomp_cov else ptr(rots), ptr(cov),
                          ptr(sh), ptr(up), ptr(va), ptr(cb), ptr(dmeans), None if precomp_cov else ptr(dsc),
                          None if precomp_cov else ptr(dro), ptr(dcov) if precomp_cov else None, ptr(dsh))
    pairs = [("means", dmeans, m3.grad), ("sh", dsh, shs.grad)]
    pairs += [("cov3d", dcov, c6.grad)] if precomp_cov else [("scales", dsc, sc.grad), ("rots", dro, ro.grad)]
    same = torch.tensor(solid | ~valid)
    for name, a, b in pairs:
        b = b.numpy()
        sel = same.numpy()
        rel = np.abs(a[sel] - b[sel]).max() / np.abs(b[sel]).max()
        assert rel < 2e-5, (name, rel)
    assert torch.allclose(m2.grad[:, :2][vm_], u[:, 3:5][vm_])      # means2D receives dL/d(ndc.xy) unchanged


def test_backward_of_needle_splats_survives_the_cancellation(lib):
    """A 100:1 needle seen side-on has cov2D ~ l1 u u^T, and the upstream dL/dconic that its own pixels produce is
    ~ K (u_x^2, 2 u_x u_y, u_y^2): in dL/dcov2D = (-c^2 dA + b c dB - b^2 dC) / det^2 (and its two siblings) terms of
    size l1^2 K cancel down to l1 l2 K.  gsr_math.h forms those sums in double; in float32 the scale / rotation
    gradients of such splats came out 1e-3 .. 1e-2 off (EXPERIMENTS.md, round 3)."""
    P, K, H, W = 12, 16, 96, 128
    g = torch.Generator().manual_seed(3)
    cam = look_at_camera((0.0, 0.0, -3.3), (0.0, 0.0, 0.0), fovx=0.9, fovy=0.7, width=W, height=H)
    st = settings_for(cam, torch.zeros(3), scale_modifier=1.0)
    means = (torch.randn(P, 3, generator=g) * 0.3).float()
    scales = torch.tensor([2.0, 0.02, 0.03]).repeat(P, 1) * torch.exp(torch.randn(P, 3, generator=g) * 0.1)
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    sh = torch.zeros(P, K, 3)
    dt = torch.float64
    m3 = means.to(dt).requires_grad_(True)
    sc = scales.float().to(dt).requires_grad_(True)
    ro = rots.float().to(dt).requires_grad_(True)
    geo = O.preprocess(m3, sc, ro, None, st, torch.zeros(P, 3, dtype=dt))
    assert bool(geo.valid.all())
    A, B, C = geo.conic[:, 0].detach(), geo.conic[:, 1].detach(), geo.conic[:, 2].detach()
    assert float(((A * C - B * B) * 1.0).max()) < 5e-3, "the case is made of needles (conic determinant ~ 1 / (l1 l2))"
    # long axis u of every splat on screen = eigenvector of the conic's SMALL eigenvalue
    w, V = torch.linalg.eigh(torch.stack([torch.stack([A, B], 1), torch.stack([B, C], 1)], 1))
    ux, uy = V[:, 0, 0], V[:, 1, 0]
    Kc = torch.rand(P, generator=g, dtype=dt) * 50.0 + 10.0
    u = torch.zeros(P, 8, dtype=dt)
    u[:, 0], u[:, 1], u[:, 2] = Kc * ux * ux, 2.0 * Kc * ux * uy, Kc * uy * uy
    u[:, 0:3] += torch.randn(P, 3, generator=g, dtype=dt) * 1e-3 * Kc[:, None]      # what the pixels off the axis add
    u = torch.tensor(f32(u), dtype=dt)                       # both sides start from the same float32 upstream values
    (geo.conic * u[:, 0:3]).sum().backward()
    vm, pm, cp = f32(st.viewmatrix), f32(st.projmatrix), f32(st.campos)
    mf, sf, rf, shf, up = f32(m3), f32(sc), f32(ro), f32(sh), f32(u)
    dmeans = np.zeros((P, 3), np.float32)
    dsh = np.zeros((P, K, 3), np.float32)
    dsc = np.zeros((P, 3), np.float32)
    dro = np.zeros((P, 4), np.float32)
    va = np.ones(P, np.int32)
    cb = np.zeros(P, np.int32)
    lib.hm_preprocess_bwd(P, K, H, W, cf(st.tanfovx), cf(st.tanfovy), cf(1.0), 3, ptr(vm), ptr(pm), ptr(cp), ptr(mf), ptr(sf),
                          ptr(rf), None, ptr(shf), ptr(up), ptr(va), ptr(cb), ptr(dmeans), ptr(dsc), ptr(dro), None, ptr(dsh))
    for name, a, b in (("scales", dsc, sc.grad), ("rots", dro, ro.grad), ("means", dmeans, m3.grad)):
        b = b.numpy()
        rel = np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)          # per splat: every one of them is a needle
        print(f"needle backward on the host, {name}: worst per-splat relative error {rel.max():.2e}")
        # float32 cov2D entries (6e-8 each) times l1 / l2 bound what is left: 3e-4 on the scales; the float32 sums gave 9e-3
        assert rel.max() < 1e-3, (name, rel)


def test_tile_footprint_test_is_conservative(lib):
    """tile_can_contribute must never reject a (tile, Gaussian) pair in which some pixel passes the
    reference's alpha >= 1/255 test (brute force over the 256 pixel centres), and should reject most that do not."""
    rng = np.random.default_rng(0)
    n = 4000
    # random PSD conics, centres around a 16x16 tile at the origin, opacities across the whole range
    th = rng.uniform(0, math.pi, n)
    s1, s2 = np.exp(rng.uniform(-1.5, 2.5, n)), np.exp(rng.uniform(-1.5, 2.5, n))
    c, s = np.cos(th), np.sin(th)
    a = (c * c) * s1 * s1 + (s * s) * s2 * s2 + 0.3
    b = c * s * (s1 * s1 - s2 * s2)
    cc = (s * s) * s1 * s1 + (c * c) * s2 * s2 + 0.3
    det = a * cc - b * b
    A, B, C = cc / det, -b / det, a / det
    cx, cy = rng.uniform(-24, 40, n), rng.uniform(-24, 40, n)
    o = np.concatenate([rng.uniform(0, 1, n // 2), rng.uniform(0, 0.02, n - n // 2)])
    geo = np.ascontiguousarray(np.stack([cx, cy, A, B, C, o], 1).astype(np.float32))
    out = np.zeros(n, np.int32)
    lib.hm_tile_can_contribute(n, ptr(geo), cf(0.0), cf(0.0), cf(15.0), cf(15.0), ptr(out))
    xs, ys = np.meshgrid(np.arange(16.0), np.arange(16.0))
    g64 = geo.astype(np.float64)
    dx = g64[:, 0, None, None] - xs[None]
    dy = g64[:, 1, None, None] - ys[None]
    power = -0.5 * (g64[:, 2, None, None] * dx * dx + g64[:, 4, None, None] * dy * dy) - g64[:, 3, None, None] * dx * dy
    alpha = np.minimum(0.99, g64[:, 5, None, None] * np.exp(np.minimum(power, 0)))
    touches = ((power <= 0) & (alpha >= 1.0 / 255.0)).any(axis=(1, 2))
    assert not (touches & (out == 0)).any(), "a contributing pair was culled"
    kept_useless = (~touches & (out == 1)).sum()
    assert kept_useless < 0.35 * (~touches).sum()        # the test is tight, not just safe
    assert 0.1 < touches.mean() < 0.9


def test_strip_masks_are_conservative_and_tight(lib):
    """strip_masks4 (what k_emit stores in a pair's top bits, and K6 / K7 skip strips on WITHOUT re-checking) must have
    the bit of every 16x4 strip in which some pixel passes the reference's alpha test, for whole and clipped tiles, and
    must agree with four tile_can_contribute calls on all but a sliver of cases."""
    rng = np.random.default_rng(1)
    n = 6000
    th = rng.uniform(0, math.pi, n)
    s1, s2 = np.exp(rng.uniform(-1.5, 2.5, n)), np.exp(rng.uniform(-1.5, 2.5, n))
    c, s = np.cos(th), np.sin(th)
    a = (c * c) * s1 * s1 + (s * s) * s2 * s2 + 0.3
    b = c * s * (s1 * s1 - s2 * s2)
    cc = (s * s) * s1 * s1 + (c * c) * s2 * s2 + 0.3
    det = a * cc - b * b
    A, B, C = cc / det, -b / det, a / det
    cx, cy = rng.uniform(-24, 40, n), rng.uniform(-24, 40, n)
    o = np.concatenate([rng.uniform(0, 1, n // 2), rng.uniform(0, 0.02, n - n // 2)])
    geo = np.ascontiguousarray(np.stack([cx, cy, A, B, C, o], 1).astype(np.float32))
    g64 = geo.astype(np.float64)
    lib.hm_strip_masks4.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_float] * 4 + [ctypes.c_void_p]
    for ymax, x1 in ((15.0, 15.0), (9.0, 15.0), (15.0, 6.0), (2.0, 11.0)):      # whole tile, clipped below / right
        out = np.zeros(n, np.int32)
        lib.hm_strip_masks4(n, ptr(geo), cf(0.0), cf(x1), cf(0.0), cf(ymax), ptr(out))
        xs, ys = np.meshgrid(np.arange(x1 + 1), np.arange(ymax + 1))
        dx = g64[:, 0, None, None] - xs[None]
        dy = g64[:, 1, None, None] - ys[None]
        power = -0.5 * (g64[:, 2, None, None] * dx * dx + g64[:, 4, None, None] * dy * dy) - g64[:, 3, None, None] * dx * dy
        alpha = np.minimum(0.99, g64[:, 5, None, None] * np.exp(np.minimum(power, 0)))
        ok = (power <= 0) & (alpha >= 1.0 / 255.0)
        same = 0
        for k in range(4):
            rows = slice(4 * k, min(4 * k + 4, int(ymax) + 1))
            if 4 * k > ymax:
                assert not ((out >> k) & 1).any(), "a strip outside the image was marked"
                continue
            touches = ok[:, rows, :].any(axis=(1, 2))
            bit = ((out >> k) & 1).astype(bool)
            assert not (touches & ~bit).any(), f"strip {k}: a contributing strip was dropped (ymax {ymax}, x1 {x1})"
            ref = np.zeros(n, np.int32)
            lib.hm_tile_can_contribute(n, ptr(geo), cf(0.0), cf(4.0 * k), cf(x1), cf(min(4.0 * k + 3.0, ymax)), ptr(ref))
            same += int((ref.astype(bool) == bit).sum())
            assert (~touches & bit).sum() < 0.35 * max((~touches).sum(), 1)
        assert same >= 0.999 * n * (int(ymax) // 4 + 1)


def test_tightened_rect_keeps_every_contributing_tile(lib):
    """tighten_rect (K1, default flags) shrinks a splat's 3-sigma tile rect to the bounding box of its alpha >= 1/255
    footprint.  No tile in which some pixel passes the reference's alpha test may fall outside the shrunk rect (brute
    force over an 8x8-tile image), an opacity below 1/255 must empty it, and it must actually shrink most rects."""
    rng = np.random.default_rng(5)
    n = 3000
    th = rng.uniform(0, math.pi, n)
    s1, s2 = np.exp(rng.uniform(-1.5, 3.0, n)), np.exp(rng.uniform(-1.5, 3.0, n))
    c, s = np.cos(th), np.sin(th)
    a = (c * c) * s1 * s1 + (s * s) * s2 * s2 + 0.3
    b = c * s * (s1 * s1 - s2 * s2)
    cc = (s * s) * s1 * s1 + (c * c) * s2 * s2 + 0.3
    det = a * cc - b * b
    A, B, C = cc / det, -b / det, a / det
    G = 8                                                    # 8x8 tiles = 128x128 pixels
    cx, cy = rng.uniform(-40, 168, n), rng.uniform(-40, 168, n)
    o = np.concatenate([rng.uniform(0, 1, n // 2), rng.uniform(0, 0.02, n - n // 2)])
    geo = np.ascontiguousarray(np.stack([cx, cy, A, B, C, o], 1).astype(np.float32))
    rect = np.array([0, 0, G, G], np.int32)
    out = np.zeros((n, 4), np.int32)
    lib.hm_tighten_rect.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.hm_tighten_rect(n, ptr(geo), G, G, ptr(rect), ptr(out))
    xs, ys = np.meshgrid(np.arange(16.0 * G), np.arange(16.0 * G))
    g64 = geo.astype(np.float64)
    shrunk = 0
    for i0 in range(0, n, 250):
        gi = g64[i0:i0 + 250]
        dx = gi[:, 0, None, None] - xs[None]
        dy = gi[:, 1, None, None] - ys[None]
        power = -0.5 * (gi[:, 2, None, None] * dx * dx + gi[:, 4, None, None] * dy * dy) - gi[:, 3, None, None] * dx * dy
        alpha = np.minimum(0.99, gi[:, 5, None, None] * np.exp(np.minimum(power, 0)))
        ok = (power <= 0) & (alpha >= 1.0 / 255.0)
        tiles = ok.reshape(len(gi), G, 16, G, 16).any(axis=(2, 4))              # [i, ty, tx]
        for j in range(len(gi)):
            x0, y0, x1, y1 = out[i0 + j]
            assert 0 <= x0 <= x1 <= G and 0 <= y0 <= y1 <= G
            inside = np.zeros((G, G), bool)
            inside[y0:y1, x0:x1] = True
            assert not (tiles[j] & ~inside).any(), f"splat {i0 + j}: a contributing tile lies outside the tightened rect"
            if g64[i0 + j, 5] < 0.99 / 255.0:
                assert (x1 - x0) * (y1 - y0) == 0
            shrunk += int((x1 - x0) * (y1 - y0) < G * G)
    assert shrunk > 0.5 * n


def _needle_case(seed):
    """One strongly elongated Gaussian (0.024 x 0.021 x 1.56, the 440:1 needle of tests/diag_aniso_elem.py seed 41) under a
    random rotation, a realistic dL/dconic (signed pixel weights along its footprint) and the float64 truth by autograd."""
    g = torch.Generator().manual_seed(seed)
    cam = look_at_camera((0.2, -0.3, -9.0), (0.0, 0.0, 0.0), fovx=0.9, fovy=0.9, width=256, height=256)
    st = settings_for(cam, torch.zeros(3), scale_modifier=1.0)
    # (the kernels receive tan(fov/2) as float32: the float64 truth starts from the same numbers)
    st = st._replace(tanfovx=float(np.float32(st.tanfovx)), tanfovy=float(np.float32(st.tanfovy)))
    mean = (torch.rand(1, 3, generator=g) - 0.5).float()
    scale = torch.tensor([[0.0239, 0.0207, 1.5583]])
    rot = torch.nn.functional.normalize(torch.randn(1, 4, generator=g), dim=1).float()
    leaf = [t.double().clone().requires_grad_(True) for t in (mean, scale, rot)]
    K = O._project(leaf[0], leaf[1], leaf[2], None, st)["conic"][0]
    Kd = K.detach()
    ys, xs = torch.meshgrid(torch.arange(-70, 71, dtype=torch.float64), torch.arange(-70, 71, dtype=torch.float64), indexing="ij")
    dx, dy = xs + 0.3, ys - 0.2
    power = -0.5 * (Kd[0] * dx * dx + Kd[2] * dy * dy) - Kd[1] * dx * dy
    w = torch.where(power > -6.0, (torch.rand(power.shape, generator=g, dtype=torch.float64) * 2 - 1) * power.exp(), torch.zeros_like(power))
    dK = torch.stack([(w * -0.5 * dx * dx).sum(), (w * -dx * dy).sum(), (w * -0.5 * dy * dy).sum()])
    (K * dK).sum().backward()
    truth = torch.cat([leaf[0].grad[0], leaf[1].grad[0], leaf[2].grad[0]])
    return st, cam, mean, scale, rot, Kd, dK, truth


def _hm_needle(lib, st, cam, mean, scale, rot, dK):
    vm, pm, cp = f32(st.viewmatrix), f32(st.projmatrix), f32(st.campos)
    kd, kf, nd = np.zeros(3, np.float64), np.zeros(3, np.float32), np.zeros(1, np.int32)
    of, od = np.zeros(10, np.float32), np.zeros(10, np.float64)
    lib.hm_needle(cam.image_height, cam.image_width, cf(st.tanfovx), cf(st.tanfovy), cf(1.0), ptr(vm), ptr(pm), ptr(cp),
                  ptr(f32(mean)), ptr(f32(scale)), ptr(f32(rot)), ctypes.c_double(dK[0]), ctypes.c_double(dK[1]),
                  ctypes.c_double(dK[2]), ptr(kd), ptr(kf), ptr(nd), ptr(of), ptr(od))
    return kd, kf, int(nd[0]), of, od


@pytest.mark.parametrize("seed", range(6))
def test_needle_chain_rule_in_double(lib, seed):
    """GSR_FLAG_NEEDLE_DOUBLE: needle_bwd_d (gsr_math.h) is the float64 chain rule of the conic -- to 1e-9 of autograd through
    oracle-R's projection in float64 -- and the float32 chain rule it replaces for needles is good to a few 1e-5."""
    st, cam, mean, scale, rot, Kd, dK, truth = _needle_case(seed)
    kd, kf, needle, of, od = _hm_needle(lib, st, cam, mean, scale, rot, dK.tolist())
    assert needle == 1
    # (the dilation is the published float32 constant 0.3f = 0.3 + 1.2e-8 in the kernels' double chain, 0.3 in oracle-R's)
    np.testing.assert_allclose(kd, Kd.numpy(), rtol=1e-7)
    t = truth.numpy()
    for sl in (slice(0, 3), slice(3, 6), slice(6, 10)):             # dL/dmean, dL/dscale, dL/dq
        scale_ = np.abs(t[sl]).max()
        ed, ef = np.abs(od[sl] - t[sl]).max() / scale_, np.abs(of[sl].astype(np.float64) - t[sl]).max() / scale_
        print(f"seed {seed} {sl}: double chain {ed:.2e}, float32 chain {ef:.2e}")
        assert ed <= 2e-7 and ef <= 3e-4


@pytest.mark.parametrize("seed", range(6))
def test_needle_conic_keeps_the_long_axis_in_float32(lib, seed):
    """needle_conic_to_float: each float32 entry within 2 ulps of the double conic's, and the quadratic form along the long axis
    (the conic's small eigenvalue -- what a needle's gradients amplify) no further from the double conic's than plain rounding
    leaves it, and within 1e-8 (plain rounding: up to 6e-8)."""
    st, cam, mean, scale, rot, Kd, dK, truth = _needle_case(seed)
    kd, kf, needle, of, od = _hm_needle(lib, st, cam, mean, scale, rot, dK.tolist())
    w, V = np.linalg.eigh(np.array([[kd[0], kd[1]], [kd[1], kd[2]]]))
    u = V[:, 0]                                                      # small conic eigenvalue = the long axis

    def along(k):
        e = np.asarray(k, np.float64) - kd
        return abs(u[0] ** 2 * e[0] + 2 * u[0] * u[1] * e[1] + u[1] ** 2 * e[2])
    plain = kd.astype(np.float32)
    assert np.all(np.abs(kf.astype(np.float64) - kd) <= 2.5 * np.spacing(np.abs(plain)).astype(np.float64))
    assert along(kf) <= along(plain) + 1e-18
    assert along(kf) <= 1e-8, (along(kf), along(plain), w[0])


@pytest.mark.parametrize("a,b,c", [(4000.3, 0.0, 0.31), (0.31, 0.0, 4000.3), (2000.0, 1999.0, 2000.0), (2000.0, -1999.0, 2000.0),
                                   (900.7, 300.1, 100.4), (100.4, -300.1, 900.7), (0.3001, 1e-9, 1500.0)])
def test_needle_conic_rounding_on_axis_aligned_diagonal_and_mirrored_covariances(lib, a, b, c):
    """needle_conic_to_float on hand-picked covariances: the long axis along x, along y (b = 0: the off-diagonal entry is not
    stepped), along either diagonal, mirrored, and with a vanishing off-diagonal -- entries within 2 ulps, finite, and the
    form along the long axis no worse than plain rounding's."""
    kf, kd = np.zeros(3, np.float32), np.zeros(3, np.float64)
    lib.hm_conic_to_float(ctypes.c_double(a), ctypes.c_double(b), ctypes.c_double(c), ptr(kf), ptr(kd))
    assert np.all(np.isfinite(kf))
    plain = kd.astype(np.float32)
    assert np.all(np.abs(kf.astype(np.float64) - kd) <= 2.5 * np.spacing(np.abs(plain)).astype(np.float64) + 1e-300)
    if b == 0.0:
        assert kf[1] == 0.0
    w, V = np.linalg.eigh(np.array([[kd[0], kd[1]], [kd[1], kd[2]]]))
    u = V[:, 0]

    def along(k):
        e = np.asarray(k, np.float64) - kd
        return abs(u[0] ** 2 * e[0] + 2 * u[0] * u[1] * e[1] + u[1] ** 2 * e[2])
    assert along(kf) <= along(plain) * (1 + 1e-9) + 1e-22


_UBSAN_CHILD = r'''
import ctypes, sys
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
rng = np.random.default_rng(0)
P, K, H, W = 64, 16, 112, 128
f = np.float32
nan, inf = f("nan"), f("inf")
means = (rng.standard_normal((P, 3)) * 0.3).astype(f)
scales = np.exp(rng.standard_normal((P, 3)) * 0.4 - 3.5).astype(f)
rots = rng.standard_normal((P, 4)).astype(f)
rots /= np.linalg.norm(rots, axis=1, keepdims=True)
sh = (rng.standard_normal((P, K, 3)) * 0.2).astype(f)
# poisons: one row each (the rest of the batch is ordinary)
means[0] = (nan, 0, 0); means[1] = (inf, 0, 1); means[2] = (0, -inf, 0); means[3] = (1e30, 1e30, 1e30); means[4] = (-1e38, 1e38, 3e38)
scales[5] = (nan, 1, 1); scales[6] = (inf, 1, 1); scales[7] = (1e30, 1e30, 1e30); scales[8] = (3e38, 3e38, 3e38); scales[9] = (0, 0, 0)
scales[10] = (1e19, 1e-19, 1.0); scales[11] = (1e9, 1e9, 1e9); scales[12] = (1e4, 1e4, 1e4)
rots[13] = (nan, 0, 0, 1); rots[14] = (inf, 0, 0, 0); rots[15] = (0, 0, 0, 0); rots[16] = (1e30, -1e30, 1e30, 1e30); rots[17] = (3e38, 3e38, 3e38, 3e38)
sh[18] = nan; sh[19] = inf; sh[20] = 3e38
means[21] = (0, 0, -0.6 + 0.2)          # exactly on the near plane from the camera at z = -0.6
vm = np.eye(4, dtype=f); vm[3, 2] = 0.6   # row-vector convention: p_view = [p, 1] V  -> z + 0.6
n_, f_ = 0.01, 100.0
t = 0.4227932187  # tan(0.8 / 2)
pm = np.zeros((4, 4), dtype=f)
pm[0, 0] = 1 / t; pm[1, 1] = 1 / t; pm[2, 2] = f_ / (f_ - n_); pm[3, 2] = -(f_ * n_) / (f_ - n_); pm[2, 3] = 1.0
pm = (vm @ pm).astype(f)
cam = np.array([0, 0, -0.6], dtype=f)
geom = np.zeros((P, 12), dtype=f); rgb = np.zeros((P, 3), dtype=f)
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
cf = ctypes.c_float
for mod in (1.0, 1e10, 3e38):
    lib.hm_preprocess(P, K, H, W, cf(t), cf(t), cf(mod), 3, vp(vm), vp(pm), vp(cam), vp(means), vp(scales), vp(rots), None, vp(sh),
                      vp(geom), vp(rgb))
    assert np.isfinite(geom[:, :6][geom[:, 6] > 0]).all(), "a kept splat has a finite centre, depth and conic"
    assert (geom[:, 6] <= 2 ** 24).all() and (geom[:, 7:11] >= 0).all() and (geom[:, 9] <= (W + 15) // 16).all()
    for bad in (0, 1, 2, 5, 6, 7, 8, 13, 14):
        assert geom[bad, 6] == 0, (mod, bad, geom[bad])
# the backward on the same inputs, every Gaussian treated as valid (what K9 would do if one slipped through), with large
# and non-finite upstream gradients: no undefined conversion or shift anywhere (values may be non-finite)
up = (rng.standard_normal((P, 8)) * 10).astype(f); up[30] = nan; up[31] = inf; up[32] = 3e38
valid = np.ones(P, dtype=np.int32); cl = np.zeros(P, dtype=np.int32)
dm = np.zeros((P, 3), dtype=f); ds = np.zeros((P, 3), dtype=f); dr = np.zeros((P, 4), dtype=f); dsh = np.zeros((P, K, 3), dtype=f)
lib.hm_preprocess_bwd(P, K, H, W, cf(t), cf(t), cf(1.0), 3, vp(vm), vp(pm), vp(cam), vp(means), vp(scales), vp(rots), None, vp(sh),
                      vp(up), vp(valid), vp(cl), vp(dm), vp(ds), vp(dr), None, vp(dsh))
# footprint tests with non-finite conics / opacities / centres
geo = np.array([[nan, 0, 1, 0, 1, 0.5], [0, 0, nan, 0, 1, 0.5], [0, 0, 1, 0, 1, nan], [inf, inf, 1, 0, 1, 0.5], [0, 0, inf, 0, inf, 0.9],
                [0, 0, 1e-38, 0, 1e-38, 0.99], [0, 0, 3e38, 3e38, 3e38, 1.0], [1e30, -1e30, 1, 0.5, 1, 0.5], [8, 8, 1, 0, 1, 0.0],
                [8, 8, 1, 0, 1, -1.0], [8, 8, 1, 0, 1, inf]], dtype=f)
out = np.zeros(len(geo), dtype=np.int32)
lib.hm_tile_can_contribute(len(geo), vp(geo), cf(0), cf(0), cf(15), cf(15), vp(out))
lib.hm_strip_masks4(len(geo), vp(geo), cf(0), cf(15), cf(0), cf(111), vp(out))
rect = np.array([0, 0, 8, 7], dtype=np.int32); out4 = np.zeros((len(geo), 4), dtype=np.int32)
lib.hm_tighten_rect(len(geo), vp(geo), 8, 7, vp(rect), vp(out4))
assert (out4[:, 0] >= 0).all() and (out4[:, 2] <= 8).all() and (out4[:, 1] >= 0).all() and (out4[:, 3] <= 7).all()
print("ubsan child ok")
'''


def test_scalar_math_under_the_undefined_behaviour_sanitizer(tmp_path):
    """csrc/gsr_math.h compiled for the host with -fsanitize=undefined,float-cast-overflow (no recovery) and fed NaN, +-inf,
    1e30 / 3e38, zero scales and quaternions through every float -> int conversion of the path (radius, tile rect, rect
    tightening, strip masks): any undefined conversion aborts the child process.  (SURVEY.md section 5 asked for the host
    code under sanitizers; VERDICT r05 item 6: `(int)ceilf(3 sqrtf(lam))` was undefined for lam = inf.)"""
    import sys
    so = str(tmp_path / "libhostmath_ubsan.so")
    src = os.path.join(HM, "host_math.cpp")
    hdr = os.path.join(ROOT, "3d-gaussian-splat-attack_amd", "csrc")
    subprocess.run(["g++", "-O1", "-g", "-ffp-contract=off", "-fsanitize=undefined,float-cast-overflow", "-fno-sanitize-recover=all",
                    "-shared", "-fPIC", "-I", hdr, src, "-o", so], check=True)
    ub = subprocess.run(["g++", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    if os.path.isabs(ub) and os.path.exists(ub):
        env["LD_PRELOAD"] = os.path.realpath(ub)          # the runtime must be in the process before the library is loaded
    out = subprocess.run([sys.executable, "-c", _UBSAN_CHILD, so], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ubsan child ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
