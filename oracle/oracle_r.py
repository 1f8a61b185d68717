"""oracle-R: CPU restatement of the differentiable 3D-Gaussian-splat rasteriser.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the reported CPU
baseline.  The product (``diff_gaussian_rasterization`` -> ``libgsraster.so``)
never falls back to this code.

PARITY STATUS: **parity unpinned** for the rasteriser arithmetic itself.  The
reference imports the third-party CUDA extension ``diff_gaussian_rasterization``
(Gaussian-Grouping variant; ``gaussian_renderer/__init__.py:14``) whose source
is not vendored under /root/reference and for which no commit is pinned, and the
reference holds no tests / golden vectors for this path.
What IS pinned (tests/golden, made by tests/golden/make_golden.py from the
reference's importable Python twins): the SH basis (``utils/sh_utils.py:57-112``
+ the ``+0.5 / clamp_min`` of ``gaussian_renderer/__init__.py:74-78``), the
camera matrices (``utils/graphics_utils.py:38-71``, ``scene/cameras.py:54-57``),
the 3D covariance construction (``utils/general_utils.py:78-110``,
``scene/gaussian_model.py:25-29``) and the PGD step functions
(``attack.py:25-173``).  The remainder follows the published 3DGS algorithm
(Kerbl et al. 2023) with the constants listed in SURVEY.md section 8(a):
0.2 near cull, 1.3*tanfov clamp, +0.3 px^2 dilation, ceil(3*sqrt(lambda)) radius
with max(0.1, .), 16x16 tiles, alpha cap 0.99, alpha floor 1/255, T stop 1e-4,
+1e-7 w-guard, pixel centre ((ndc+1)*S-1)/2.

Gradients come from autograd over this restatement, with three deliberate
constructions so that they equal what the reference extension's hand-written
backward produces rather than the "true" derivative:
  * ``alpha = min(0.99, o*G)``: the backward treats the cap as transparent
    (dalpha/dG = o even when capped);
  * the 1.3*tanfov clamp of t.x/t.z: when clamped, t.x is treated as a constant
    (zero gradient to t.x, and no t.z dependence through the clamp);
  * the inversion of the 2D covariance: the published backward divides by
    ``det^2 + 1e-7`` (a float32 literal) where the exact derivative of
    conic = (c, -b, a) / det divides by ``det^2`` (``_Conic`` below;
    ``DET_GUARD = 0.0`` gives the exact derivative).  With the 0.3 px^2
    dilation det >= 0.09, so the two differ by at most 1.2e-5 relative in
    dL/d(cov2D) -- measured on BASELINE configs 3 and 5 in
    profiles/r06_parity_notes.txt.
Everything else is the exact derivative.

Works in float32 or float64 (``dtype=``); float64 is the parity reference.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

TILE = 16
NUM_OBJECTS = 16  # scene/gaussian_model.py:52

# utils/sh_utils.py:26-54 (values; deg <= 3 is what the rasteriser supports)
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


class Settings(NamedTuple):
    """Same 12 fields, same order, as the call site gaussian_renderer/__init__.py:36-49."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# --------------------------------------------------------------------------
# per-Gaussian stages
# --------------------------------------------------------------------------

def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    """Real SH basis values [P,(deg+1)^2] for unit directions d [P,3].

    Sign/ordering convention of utils/sh_utils.py:72-101 (deg<=3)."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    cols = [torch.full_like(x, SH_C0)]
    if deg > 0:
        cols += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        cols += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy),
                 SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        cols += [SH_C3[0] * y * (3.0 * xx - yy), SH_C3[1] * xy * z,
                 SH_C3[2] * y * (4.0 * zz - xx - yy),
                 SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy),
                 SH_C3[4] * x * (4.0 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                 SH_C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(cols, dim=1)


def sh_to_rgb(deg: int, sh: torch.Tensor, means3D: torch.Tensor, campos: torch.Tensor):
    """sh [P,K,3] (coefficient-major, then channel: scene/gaussian_model.py:113-116)
    -> (rgb [P,3] after +0.5 and clamp at 0, clamped flags [P,3])."""
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    B = sh_basis(deg, d)                                   # [P,k]
    k = B.shape[1]
    raw = (B[:, :, None] * sh[:, :k, :]).sum(dim=1) + 0.5  # gaussian_renderer/__init__.py:78
    clamped = raw < 0
    return torch.clamp_min(raw, 0.0), clamped


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """utils/general_utils.py:85-98 but WITHOUT the re-normalisation: the rasteriser
    consumes the quaternion as given (get_rotation normalised it upstream,
    scene/gaussian_model.py:39,102-103)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def cov3d_from_scale_rot(scales, rotations, mod: float) -> torch.Tensor:
    """L = R diag(mod*s); Sigma = L L^T (utils/general_utils.py:101-110,
    scene/gaussian_model.py:25-29) -> full [P,3,3]."""
    R = quat_to_rot(rotations)
    L = R * (mod * scales)[:, None, :]
    return L @ L.transpose(1, 2)


def cov3d_pack(S: torch.Tensor) -> torch.Tensor:
    """strip_lowerdiag order (utils/general_utils.py:64-73): xx,xy,xz,yy,yz,zz."""
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def cov3d_unpack(c6: torch.Tensor) -> torch.Tensor:
    return torch.stack([c6[:, 0], c6[:, 1], c6[:, 2],
                        c6[:, 1], c6[:, 3], c6[:, 4],
                        c6[:, 2], c6[:, 4], c6[:, 5]], dim=1).view(-1, 3, 3)


class Geom(NamedTuple):
    valid: torch.Tensor      # [P] bool   radius > 0 after all culls
    radii: torch.Tensor      # [P] int32
    xy: torch.Tensor         # [P,2]      pixel-space centre
    depth: torch.Tensor      # [P]        view-space z
    conic: torch.Tensor      # [P,3]      (A,B,C) of the inverse dilated 2D covariance
    rect_min: torch.Tensor   # [P,2] int  tile rect [min,max)
    rect_max: torch.Tensor   # [P,2] int
    fragile: torch.Tensor    # [P] bool   an integer decision sits within rounding of its edge
    ghost_min: torch.Tensor  # [P,2] int  tile rect a float32 implementation might use instead (= rect unless fragile)
    ghost_max: torch.Tensor  # [P,2] int
    maybe: torch.Tensor      # [P] bool   culled here, but a float32 implementation might keep it (fragile cull)
    e_conic: torch.Tensor    # [P,3]      error bars (see preprocess) of the conic,
    e_xy: torch.Tensor       # [P,2]      the pixel centre
    e_depth: torch.Tensor    # [P]        and the view depth


# The published backward of the 2D covariance inversion: 1 / (det^2 + 0.0000001f).  0.0: the exact derivative.
DET_GUARD = 1.0000000116860974e-07       # = float32(1e-7), the literal as the float32 kernels see it


class _Conic(torch.autograd.Function):
    """conic = (c, -b, a) / det of the dilated 2D covariance [[a, b], [b, c]]; the backward is the exact derivative with
    1 / det^2 replaced by 1 / (det^2 + DET_GUARD) -- what the published hand-written backward computes (dL/dB counted once
    for the single off-diagonal entry, as everywhere in this file)."""

    @staticmethod
    def forward(ctx, a, b, c, det_safe):
        ctx.save_for_backward(a, b, c, det_safe)
        return torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=1)

    @staticmethod
    def backward(ctx, g):
        a, b, c, det = ctx.saved_tensors
        dA, dB, dC = g[:, 0], g[:, 1], g[:, 2]
        d2 = 1.0 / (det * det + DET_GUARD)
        da = (-c * c * dA + b * c * dB - b * b * dC) * d2
        db = (2.0 * b * c * dA - (det + 2.0 * b * b) * dB + 2.0 * a * b * dC) * d2
        dc = (-b * b * dA + a * b * dB - a * a * dC) * d2
        return da, db, dc, None


def _project(means3D, scales, rotations, cov3D_precomp, st: Settings, means2D=None):
    """The arithmetic of K1 in the dtype of its inputs.
    -> dict(tz, in_front, det_ok, conic [P,3], rr (un-rounded radius 3 sqrt(lambda)), px, py)."""
    dt = means3D.dtype
    H, W = int(st.image_height), int(st.image_width)
    V = st.viewmatrix.to(dt)
    PV = st.projmatrix.to(dt)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], dim=1)
    p_view = ph @ V                                        # row-vector convention (scene/cameras.py:54)
    p_hom = ph @ PV
    w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :3] * w[:, None]
    if means2D is not None:                                # receives dL/d(ndc.xy) like the extension's dL_dmean2D
        ndc = torch.cat([ndc[:, :2] + means2D[:, :2], ndc[:, 2:]], dim=1)
    tz = p_view[:, 2]
    in_front = tz > 0.2

    if cov3D_precomp is not None:
        S3 = cov3d_unpack(cov3D_precomp)
    else:
        S3 = cov3d_from_scale_rot(scales, rotations, float(st.scale_modifier))

    fx = W / (2.0 * st.tanfovx)
    fy = H / (2.0 * st.tanfovy)
    limx = 1.3 * st.tanfovx
    limy = 1.3 * st.tanfovy
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    txtz = p_view[:, 0] / tz_safe
    tytz = p_view[:, 1] / tz_safe
    # straight-through-as-constant when clamped (see module docstring)
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz_safe).detach(), p_view[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz_safe).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], dim=1).view(-1, 2, 3)
    Wr = V[:3, :3].t()                                     # column-vector view rotation
    M = J @ Wr[None]                                       # [P,2,3]
    cov2 = M @ S3 @ M.transpose(1, 2)                      # [P,2,2]
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_ok = det != 0
    det_safe = torch.where(det_ok, det, torch.ones_like(det))
    conic = _Conic.apply(a, b, c, det_safe.detach())
    mid = 0.5 * (a + c)
    disc = torch.clamp_min(mid * mid - det, 0.1)
    lam = mid + torch.sqrt(disc)
    rr = 3.0 * torch.sqrt(lam.detach())
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    return dict(tz=tz, in_front=in_front, det_ok=det_ok, conic=conic, rr=rr, px=px, py=py)


DEBUG_COUNTS = None
ERR_SAFETY = 3.0   # error bar = ERR_SAFETY * |float32 run - float64 run| + a few ulps


def preprocess(means3D, scales, rotations, cov3D_precomp, st: Settings, means2D=None):
    """K1 of SURVEY.md section 8(a) row a4, steps (1)-(8) and (10), plus error bars.

    The implementation under test computes in float32.  To know where ITS threshold decisions may legitimately differ
    from this restatement's, the same arithmetic is run a second time in the other precision (float32 when the
    caller asked for float64 and vice versa, no gradients): ERR_SAFETY times the difference between the two runs, plus
    a few float32 ulps, is the per-Gaussian error bar of the radius, the pixel centre, the conic and the depth.
    `fragile` marks Gaussians whose integer decisions (ceil of the radius, tile rect, near-plane cull) sit inside their
    bar; rasterize() uses the bars of conic / centre / depth for the per-pixel tests."""
    dt = means3D.dtype
    H, W = int(st.image_height), int(st.image_width)
    q = _project(means3D, scales, rotations, cov3D_precomp, st, means2D)
    other = torch.float32 if dt == torch.float64 else torch.float64
    with torch.no_grad():
        def cv(t):
            return None if t is None else t.detach().to(other)
        q2 = _project(cv(means3D), cv(scales), cv(rotations), cv(cov3D_precomp), st, cv(means2D))
    tz, in_front, det_ok, conic, rr = q["tz"], q["in_front"], q["det_ok"], q["conic"], q["rr"]
    px, py = q["px"], q["py"]
    xy = torch.stack([px, py], dim=1)
    radius = torch.ceil(rr)
    f32eps = 1.2e-7

    def bar(x, x2, floor):
        d = (x.detach().double() - x2.double()).abs()
        d = torch.nan_to_num(d, nan=0.0, posinf=0.0, neginf=0.0)
        return ERR_SAFETY * d + floor
    e_rr = bar(rr, q2["rr"], 8 * f32eps * rr.detach().double().abs())
    e_px = bar(px, q2["px"], 0.5 * f32eps * max(W, 1))
    e_py = bar(py, q2["py"], 0.5 * f32eps * max(H, 1))
    # depth: the oracle orders by float32(depth); an implementation's own float32 depth differs from that by its
    # rounding, which the shadow run measures (3x: it is an order statistic, not a continuous quantity) plus two ulps
    e_z = bar(tz, q2["tz"], 0.0) + 2 * f32eps * tz.detach().double().abs()
    e_con = bar(conic, q2["conic"], 8 * f32eps * conic.detach().double().abs())

    gx = (W + TILE - 1) // TILE
    gy = (H + TILE - 1) // TILE
    pxd, pyd = px.detach(), py.detach()

    def _trunc_clamp(v, hi):
        v = torch.nan_to_num(v, nan=0.0, posinf=1e9, neginf=-1e9)
        return torch.clamp(torch.trunc(v).to(torch.int64), 0, hi)
    rminx = _trunc_clamp((pxd - radius) / TILE, gx)
    rminy = _trunc_clamp((pyd - radius) / TILE, gy)
    rmaxx = _trunc_clamp((pxd + radius + (TILE - 1)) / TILE, gx)
    rmaxy = _trunc_clamp((pyd + radius + (TILE - 1)) / TILE, gy)
    area = (rmaxx - rminx) * (rmaxy - rminy)
    valid = in_front & det_ok & (area > 0)
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)

    # fragile integer decisions: the decision variable sits inside its error bar of the edge
    def _near_int(v, tol):
        v = torch.nan_to_num(v.double(), nan=0.0, posinf=0.0, neginf=0.0)
        return (v - torch.round(v)).abs() < tol
    fragile = _near_int(rr.detach(), e_rr)
    for qv, e in (((pxd - radius) / TILE, e_px), ((pyd - radius) / TILE, e_py),
                  ((pxd + radius + (TILE - 1)) / TILE, e_px), ((pyd + radius + (TILE - 1)) / TILE, e_py)):
        fragile = fragile | _near_int(qv, e / TILE + 1e-7)
    fragile = fragile | ((tz.detach().double() - 0.2).abs() < e_z)
    fragile = fragile & (tz.detach() > 0.1)

    # The rect a float32 implementation might legitimately arrive at for a fragile Gaussian: one tile more on every
    # side.  rasterize() walks these extra (tile, Gaussian) pairs as GHOSTS -- they never contribute, they only mark
    # the pixels where they would have -- so that the fragile-pixel flags are per pixel, not per tile.
    fr_i = fragile.to(torch.int64)
    gminx = torch.clamp(rminx - fr_i, 0, gx)
    gminy = torch.clamp(rminy - fr_i, 0, gy)
    gmaxx = torch.clamp(rmaxx + fr_i, 0, gx)
    gmaxy = torch.clamp(rmaxy + fr_i, 0, gy)
    maybe = fragile & ~valid & (tz > 0.1) & det_ok & ((gmaxx - gminx) * (gmaxy - gminy) > 0)

    return Geom(valid, radii, xy, tz, conic,
                torch.stack([rminx, rminy], dim=1), torch.stack([rmaxx, rmaxy], dim=1), fragile,
                torch.stack([gminx, gminy], dim=1), torch.stack([gmaxx, gmaxy], dim=1), maybe,
                e_con.to(dt), torch.stack([e_px, e_py], dim=1).to(dt), e_z.to(dt))


# --------------------------------------------------------------------------
# binning (K2-K5) and compositing (K6)
# --------------------------------------------------------------------------

def _tiles_in_windows(tx, ty, windows):
    keep = torch.zeros_like(tx, dtype=torch.bool)
    for (x0, y0, x1, y1) in windows:
        keep |= (tx >= x0) & (tx < x1) & (ty >= y0) & (ty < y1)
    return keep


def build_tile_lists(g: Geom, H: int, W: int, tile_windows=None, ghosts: bool = False, depth_key=None):
    """(tile id, Gaussian id) pairs sorted by (tile, depth) with ties in Gaussian-index
    order, i.e. a stable sort of row-major-emitted pairs on key (tile<<32 | depth bits).
    Returns (sorted gaussian ids [N], ranges [T,2], ghost flags [N]).

    tile_windows: list of (tx0, ty0, tx1, ty1) half-open tile rectangles -- only pairs of those tiles are
    kept (the lists of the kept tiles are complete).  ghosts: also emit the pairs of the widened rects of fragile
    Gaussians (Geom.ghost_min/max, Geom.maybe), flagged so that the compositor skips them."""
    gx = (W + TILE - 1) // TILE
    gy = (H + TILE - 1) // TILE
    T = gx * gy
    sel = (g.valid | g.maybe) if ghosts else g.valid
    ids = torch.nonzero(sel).flatten()
    if ids.numel() == 0:
        return (torch.zeros(0, dtype=torch.int64), torch.zeros(T, 2, dtype=torch.int64),
                torch.zeros(0, dtype=torch.bool))
    if depth_key is not None:
        depth = depth_key.detach().to(torch.float32)[ids]  # the implementation's own float32 keys (see rasterize)
    else:
        depth = g.depth.detach()[ids].to(torch.float32)    # the key holds float32 depth bits
    order = torch.argsort(depth, stable=True)
    ids = ids[order]
    if ghosts:
        rmin, rmax = g.ghost_min[ids], g.ghost_max[ids]
    else:
        rmin, rmax = g.rect_min[ids], g.rect_max[ids]
    wx = rmax[:, 0] - rmin[:, 0]
    cnt = wx * (rmax[:, 1] - rmin[:, 1])
    owner = torch.repeat_interleave(torch.arange(ids.numel()), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(int(cnt.sum())) - start[owner]
    ty = rmin[owner, 1] + local // wx[owner]
    tx = rmin[owner, 0] + local % wx[owner]
    gid = ids[owner]
    if ghosts:
        real_min, real_max = g.rect_min[gid], g.rect_max[gid]
        ghost = ~g.valid[gid] | (tx < real_min[:, 0]) | (tx >= real_max[:, 0]) | (ty < real_min[:, 1]) | (ty >= real_max[:, 1])
    else:
        ghost = torch.zeros(gid.numel(), dtype=torch.bool)
    if tile_windows is not None:
        keep = _tiles_in_windows(tx, ty, tile_windows)
        tx, ty, gid, ghost = tx[keep], ty[keep], gid[keep], ghost[keep]
    tile = ty * gx + tx
    o2 = torch.argsort(tile, stable=True)                  # stable => depth order kept inside each tile
    tile_s = tile[o2]
    gid_s = gid[o2]
    counts = torch.bincount(tile_s, minlength=T)
    ends = torch.cumsum(counts, 0)
    ranges = torch.stack([ends - counts, ends], dim=1)
    return gid_s, ranges, ghost[o2]


class RenderOut(NamedTuple):
    color: torch.Tensor        # [3,H,W]
    radii: torch.Tensor        # [P] int32
    objects: torch.Tensor      # [16,H,W]
    final_T: torch.Tensor      # [H,W]
    n_contrib: torch.Tensor    # [H,W] int   (1-based position of the last contributor in the tile list)
    fragile_px: torch.Tensor   # [H,W] bool  a threshold test sat within rounding of its edge
    num_rendered: int
    fragile_gauss: torch.Tensor
    window_px: Optional[torch.Tensor] = None   # [H,W] bool: pixels that were composited (None = all of them)


def rasterize(means3D, means2D, opacities, st: Settings, shs=None, sh_objs=None, colors_precomp=None,
              scales=None, rotations=None, cov3D_precomp=None, dtype=torch.float64,
              frag_tol: Optional[float] = None, tile_windows=None, depth_key=None) -> RenderOut:
    """Full forward (differentiable).  Argument names follow the rasteriser call site
    gaussian_renderer/__init__.py:86-95.

    tile_windows: optional list of half-open tile rectangles (tx0, ty0, tx1, ty1).  Only those tiles are
    composited (everything per Gaussian still runs over all P); the other pixels of the outputs are zero and
    ``window_px`` marks the composited ones.  Lets full-size scenes (1M Gaussians @1080p, 2M @4K) be checked on the
    tiles a test picks, with dL/dC zero elsewhere.

    depth_key: optional [P] float32 view depths AS COMPUTED BY THE IMPLEMENTATION UNDER TEST.  The per-tile order is a
    sort on float32 depth bits; two Gaussians whose depths differ by an ulp or two may legitimately sort either way in
    two float32 implementations, and in a 1000-entry list such near-ties are the rule, not the exception.  Given the
    keys, the oracle composites in the order THEY define (the caller checks them against `Geom.depth` to a few ulps:
    check_depth_keys) instead of flagging every pixel two near-tied splats share as fragile."""
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

    def cv(t):
        return None if t is None else t.to(dtype)
    means3D, means2D, opacities = cv(means3D), cv(means2D), cv(opacities)
    shs, sh_objs, colors_precomp = cv(shs), cv(sh_objs), cv(colors_precomp)
    scales, rotations, cov3D_precomp = cv(scales), cv(rotations), cv(cov3D_precomp)

    H, W = int(st.image_height), int(st.image_width)
    P = means3D.shape[0]
    bg = st.bg.to(dtype).flatten()[:3]                     # bg may hold 4 elements (attack.py:396)
    g = preprocess(means3D, scales, rotations, cov3D_precomp, st, means2D)
    if shs is not None:
        rgb, _ = sh_to_rgb(int(st.sh_degree), shs, means3D, st.campos.to(dtype).flatten()[:3])
    else:
        rgb = colors_precomp
    opac = opacities.view(P)
    objs = sh_objs.reshape(P, NUM_OBJECTS) if sh_objs is not None else None

    gx = (W + TILE - 1) // TILE
    gy = (H + TILE - 1) // TILE
    if tile_windows is not None:
        tile_windows = [(max(0, x0), max(0, y0), min(gx, x1), min(gy, y1)) for (x0, y0, x1, y1) in tile_windows]
    gid, ranges, ghost_all = build_tile_lists(g, H, W, tile_windows, ghosts=True, depth_key=depth_key)
    num_rendered = int((~ghost_all).sum())
    # relative half-width of the "a float32 implementation may legitimately flip this test" band
    tol = frag_tol if frag_tol is not None else 2e-5

    if tile_windows is None:
        todo = [(tx, ty) for ty in range(gy) for tx in range(gx)]
    else:
        seen = set()
        for (x0, y0, x1, y1) in tile_windows:
            for ty in range(y0, y1):
                for tx in range(x0, x1):
                    seen.add((tx, ty))
        todo = sorted(seen, key=lambda q: (q[1], q[0]))

    c_t, o_t, T_t, n_t, f_t = [], [], [], [], []
    yy, xx = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    for (tx, ty) in todo:
        s, e = int(ranges[ty * gx + tx, 0]), int(ranges[ty * gx + tx, 1])
        ids = gid[s:e]
        ghost = ghost_all[s:e]
        L = ids.numel()
        pxs = (tx * TILE + xx).reshape(-1, 1).to(dtype)       # [256,1]
        pys = (ty * TILE + yy).reshape(-1, 1).to(dtype)
        if L == 0:
            c_t.append(bg[:, None].expand(3, TILE * TILE).reshape(3, TILE, TILE))
            o_t.append(torch.zeros(NUM_OBJECTS, TILE, TILE, dtype=dtype))
            T_t.append(torch.ones(TILE, TILE, dtype=dtype))
            n_t.append(torch.zeros(TILE, TILE, dtype=torch.int64))
            f_t.append(torch.zeros(TILE, TILE, dtype=torch.bool))
            continue
        real = ~ghost
        dx = g.xy[ids, 0][None, :] - pxs                       # [256,L]
        dy = g.xy[ids, 1][None, :] - pys
        A, B, C = g.conic[ids, 0][None], g.conic[ids, 1][None], g.conic[ids, 2][None]
        power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        Gv = torch.exp(torch.clamp_max(power, 0.0))
        a_raw = opac[ids][None, :] * Gv
        alpha = a_raw - torch.relu(a_raw - 0.99).detach()      # min(0.99, .) with transparent backward
        would = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)  # passes the reference's two per-pixel tests
        valid = would & real[None, :]
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - a_eff
        T_incl = torch.cumprod(one_m, dim=1)
        T_excl = torch.cat([torch.ones(TILE * TILE, 1, dtype=dtype), T_incl[:, :-1]], dim=1)
        stop = valid & (T_incl.detach() < 1e-4)
        alive = torch.cumsum(stop.to(torch.int64), dim=1) == 0  # entries strictly before the stopping one
        wgt = torch.where(alive & valid, a_eff * T_excl, torch.zeros_like(a_eff))   # [256,L]
        col = wgt @ rgb[ids]                                    # [256,3]
        # final T = T before the stopping entry, or the full product
        contrib = alive & valid
        T_alive = torch.where(alive, one_m, torch.ones_like(one_m))
        T_fin = torch.prod(T_alive, dim=1)
        pos = torch.cumsum(real.to(torch.int64), 0)[None, :].expand_as(contrib)   # list position without the ghosts
        n_c = torch.where(contrib, pos, torch.zeros_like(pos)).max(dim=1).values
        col = col + T_fin[:, None] * bg[None, :]
        if objs is not None:
            ob = wgt @ objs[ids]
        else:
            ob = torch.zeros(TILE * TILE, NUM_OBJECTS, dtype=dtype)
        with torch.no_grad():
            reach = alive | stop                               # entries the loop actually evaluated
            # error bar of `power` at every (pixel, entry): first-order propagation of the conic / centre bars of
            # preprocess(), plus the rounding of the float32 evaluation of the quadratic form itself
            eA, eB, eC = g.e_conic[ids, 0][None], g.e_conic[ids, 1][None], g.e_conic[ids, 2][None]
            ex, ey = g.e_xy[ids, 0][None], g.e_xy[ids, 1][None]
            adx, ady = dx.abs(), dy.abs()
            e_pow = (0.5 * eA * dx * dx + 0.5 * eC * dy * dy + eB * adx * ady
                     + (A * dx + B * dy).abs() * ex + (C * dy + B * dx).abs() * ey
                     + 6e-7 * (0.5 * A.abs() * dx * dx + 0.5 * C.abs() * dy * dy + B.abs() * adx * ady))
            e_al = e_pow + tol                                 # relative bar of alpha = o exp(power): d ln(alpha) = d power
            near_floor = (alpha * 255.0 - 1.0).abs() < e_al    # alpha ~ 1/255
            fr = (reach & real[None] & (power <= e_pow) & near_floor).any(dim=1)
            if DEBUG_COUNTS is not None:
                DEBUG_COUNTS["alpha"] = DEBUG_COUNTS.get("alpha", 0) + int(fr.sum())
                DEBUG_COUNTS["e_al_med"] = float(e_al[reach].median()) if bool(reach.any()) else 0
            fr |= (reach & real[None] & (power.abs() < e_pow) & would & ((dx != 0) | (dy != 0))).any(dim=1)   # power ~ 0
            # T' = prod (1 - alpha): relative bar = sum over the contributing entries of alpha/(1-alpha) * bar(alpha)
            # (an alpha held at the 0.99 cap has no error of its own)
            capped = a_raw >= 0.99 * (1.0 + e_al)
            # (the entries of a list are different Gaussians with independent errors: root of the sum of squares)
            term = torch.where(valid & ~capped, a_eff / one_m * e_al, torch.zeros_like(e_al))
            e_T = 3.0 * torch.sqrt(torch.cumsum(term * term, dim=1)) \
                + 3e-7 * torch.sqrt(torch.cumsum(valid.to(dtype), dim=1)) + tol
            frT = (reach & valid & ((T_incl * 1e4 - 1.0).abs() < e_T)).any(dim=1)           # T' ~ 1e-4
            if DEBUG_COUNTS is not None:
                DEBUG_COUNTS["T"] = DEBUG_COUNTS.get("T", 0) + int(frT.sum())
            fr |= frT
            # Gaussians whose integer decisions (radius, tile rect, near-plane cull) are fragile: a float32
            # implementation may drop them from this tile (real entries) or add them to it (ghost entries).  Only the
            # pixels where such an entry passes -- or all but passes -- the alpha test can differ.
            loose = (power <= e_pow) & (alpha * 255.0 >= 1.0 - e_al)
            flip = (g.fragile[ids] & real)[None] | ghost[None]
            frG = (reach & flip & loose).any(dim=1)
            if DEBUG_COUNTS is not None:
                DEBUG_COUNTS["gauss"] = DEBUG_COUNTS.get("gauss", 0) + int(frG.sum())
            fr |= frG
            # depth near-ties: list neighbours (up to two apart) whose float32 depth keys may sort the other way round;
            # only pixels that both of them reach can differ
            if L > 1 and depth_key is None:
                z = g.depth.detach()[ids].double()
                ez = g.e_depth[ids].double()
                for sft in (1, 2):
                    if L > sft:
                        tie = (z[sft:] - z[:-sft]) <= (ez[sft:] + ez[:-sft])
                        if bool(tie.any()):
                            both = loose[:, sft:] & loose[:, :-sft] & reach[:, :-sft] & tie[None, :]
                            if DEBUG_COUNTS is not None:
                                DEBUG_COUNTS["ties"] = DEBUG_COUNTS.get("ties", 0) + int(both.any(dim=1).sum())
                            fr |= both.any(dim=1)
        c_t.append(col.t().reshape(3, TILE, TILE))
        o_t.append(ob.t().reshape(NUM_OBJECTS, TILE, TILE))
        T_t.append(T_fin.reshape(TILE, TILE))
        n_t.append(n_c.reshape(TILE, TILE))
        f_t.append(fr.reshape(TILE, TILE))

    # assemble: one differentiable scatter of the composited tiles into the (tile-padded) image
    tys = torch.tensor([q[1] for q in todo], dtype=torch.int64)
    txs = torch.tensor([q[0] for q in todo], dtype=torch.int64)

    def assemble(tiles, ch, dt, fill=0):
        buf = torch.full((gy, gx) + ((ch,) if ch else ()) + (TILE, TILE), fill, dtype=dt)
        if len(tiles):
            buf = buf.index_put((tys, txs), torch.stack(tiles))
        if ch:
            return buf.permute(2, 0, 3, 1, 4).reshape(ch, gy * TILE, gx * TILE)[:, :H, :W]
        return buf.permute(0, 2, 1, 3).reshape(gy * TILE, gx * TILE)[:H, :W]
    color = assemble(c_t, 3, dtype)
    objects = assemble(o_t, NUM_OBJECTS, dtype)
    final_T = assemble(T_t, 0, dtype)
    n_contrib = assemble(n_t, 0, torch.int64)
    fragile_px = assemble(f_t, 0, torch.bool, False)
    window_px = None
    if tile_windows is not None:
        window_px = assemble([torch.ones(TILE, TILE, dtype=torch.bool)] * len(todo), 0, torch.bool, False)
    return RenderOut(color, g.radii, objects, final_T, n_contrib, fragile_px, num_rendered, g.fragile, window_px)


def mark_visible(means3D, st: Settings) -> torch.Tensor:
    """K10: frustum test only (view-space z > 0.2)."""
    dt = means3D.dtype
    ph = torch.cat([means3D, torch.ones(means3D.shape[0], 1, dtype=dt)], dim=1)
    return (ph @ st.viewmatrix.to(dt))[:, 2] > 0.2


# --------------------------------------------------------------------------
# convenience: forward + backward for a fixed dL/dC (what tests and bench use)
# --------------------------------------------------------------------------

def check_depth_keys(depth_key, means3D, st: Settings, radii, ulps: float = 6.0) -> float:
    """Largest deviation, in float32 ulps, of an implementation's depth keys from the float64 view depth over the
    Gaussians it kept (radii > 0); raises if it exceeds `ulps`.  What licenses rasterize(depth_key=...)."""
    m = means3D.detach().double()
    z = (torch.cat([m, torch.ones(m.shape[0], 1, dtype=torch.float64)], dim=1) @ st.viewmatrix.double())[:, 2]
    vis = radii.to(torch.int64) > 0
    if not bool(vis.any()):
        return 0.0
    # the three products and the offset of p . V[:, 2] may cancel: measure in ulps of the largest partial sum
    terms = torch.cat([m, torch.ones(m.shape[0], 1, dtype=torch.float64)], dim=1) * st.viewmatrix.double()[:, 2][None]
    scale = terms.abs().sum(dim=1).clamp_min(1e-30)
    dev = ((depth_key.detach().double() - z).abs() / (scale * 1.1920929e-07))[vis].max().item()
    if dev > ulps:
        raise AssertionError(f"depth keys deviate from the float64 view depth by {dev:.1f} float32 ulps (> {ulps})")
    return dev


def solid_grads(out: RenderOut, grad_color, grad_objects=None):
    """dL/dC (and dL/dobjects) with the fragile pixels -- and, for a windowed render, the pixels outside the windows --
    zeroed.  A parity test feeds THESE to both sides: a pixel where a float32 threshold test may legitimately flip then
    owns no gradient, so the gradient comparison needs no outlier allowance."""
    keep = ~out.fragile_px
    if out.window_px is not None:
        keep = keep & out.window_px
    gc = grad_color * keep.to(grad_color.dtype)
    go = None if grad_objects is None else grad_objects * keep.to(grad_objects.dtype)
    return gc, go


def forward_backward(inputs: dict, st: Settings, grad_color, grad_objects=None, dtype=torch.float64,
                     tile_windows=None, drop_fragile: bool = False, depth_key=None):
    """inputs: dict of float tensors (means3D, shs, opacities, scales, rotations[, sh_objs, ...]).
    Returns (RenderOut, grads dict incl. 'means2D').  drop_fragile: the loss ignores the fragile pixels
    (solid_grads(); the caller gets the effective dL/dC by calling solid_grads on the returned RenderOut)."""
    leaf = {}
    for k, v in inputs.items():
        leaf[k] = None if v is None else v.detach().to(dtype).clone().requires_grad_(True)
    P = leaf["means3D"].shape[0]
    m2d = torch.zeros(P, 3, dtype=dtype, requires_grad=True)
    out = rasterize(leaf["means3D"], m2d, leaf["opacities"], st, shs=leaf.get("shs"),
                    sh_objs=leaf.get("sh_objs"), colors_precomp=leaf.get("colors_precomp"),
                    scales=leaf.get("scales"), rotations=leaf.get("rotations"),
                    cov3D_precomp=leaf.get("cov3D_precomp"), dtype=dtype, tile_windows=tile_windows,
                    depth_key=depth_key)
    if drop_fragile:
        grad_color, grad_objects = solid_grads(out, grad_color, grad_objects)
    loss = (out.color * grad_color.to(dtype)).sum()
    if grad_objects is not None:
        loss = loss + (out.objects * grad_objects.to(dtype)).sum()
    if loss.requires_grad:          # nothing visible: the image is the background, every gradient is zero
        loss.backward()
    grads = {k: (v.grad if v is not None and v.grad is not None else
                 (torch.zeros_like(v) if v is not None else None)) for k, v in leaf.items()}
    grads["means2D"] = m2d.grad if m2d.grad is not None else torch.zeros_like(m2d)
    return out, grads
