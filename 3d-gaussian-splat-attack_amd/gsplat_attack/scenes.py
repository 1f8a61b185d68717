"""Synthetic stand-ins for the reference's scenes.

Every asset the reference names (assets/hydrant, assets/nyc_block, assets/airport_scene) is a
git-LFS pointer stub, so all benchmark / parity inputs are generated here from fixed seeds with the
distributions SURVEY.md section 8(d) lists.  Tensors are raw (pre-activation) attributes in the
``load_ply`` layout (scene/gaussian_model.py:418-467).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import torch

from .cameras import Camera, look_at_camera
from .gaussian_model import GaussianModel, NUM_OBJECTS


@dataclass
class SceneSpec:
    name: str
    seed: int
    P: int
    width: int
    height: int
    fovx: float


SPECS = {
    "hydrant-1k": SceneSpec("S-hydrant-1k", 1, 1000, 128, 128, 0.8),
    "hydrant-full": SceneSpec("S-hydrant-full", 2, 300_000, 800, 800, 0.8),
    "nyc-1M": SceneSpec("S-nyc-1M", 3, 1_000_000, 1920, 1080, 1.0),
    "airport-4K": SceneSpec("S-airport-4K", 5, 2_000_000, 3840, 2160, 1.0),
}


def _common_attrs(g: torch.Generator, P: int, log_scale_mean: float, log_scale_std: float,
                  opa_mean: float, opa_std: float, aniso: bool, sh_degree: int):
    K = (sh_degree + 1) ** 2
    scaling = torch.randn(P, 3, generator=g) * log_scale_std + log_scale_mean
    if aniso:
        axis = torch.randint(0, 3, (P,), generator=g)
        fac = torch.rand(P, generator=g) * 0.9 + 0.1
        scaling[torch.arange(P), axis] += torch.log(fac)
    rotation = torch.randn(P, 4, generator=g)
    opacity = torch.randn(P, 1, generator=g) * opa_std + opa_mean
    f_dc = torch.randn(P, 1, 3, generator=g)
    f_rest = torch.randn(P, K - 1, 3, generator=g) * 0.15
    objects = torch.randn(P, 1, NUM_OBJECTS, generator=g) * 0.5
    return scaling, rotation, opacity, f_dc, f_rest, objects


def hydrant(P: int = 1000, seed: int = 1, log_scale_mean: float = math.log(0.03), sh_degree: int = 3):
    """A capsule-shaped blob around the origin (configs 1 and 2)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(P, 3, generator=g) * 0.3
    # clip to a capsule: radius 0.45 around the segment y in [-0.4, 0.4]
    yc = xyz[:, 1].clamp(-0.4, 0.4)
    off = xyz - torch.stack([torch.zeros(P), yc, torch.zeros(P)], dim=1)
    n = off.norm(dim=1, keepdim=True).clamp_min(1e-9)
    off = off * torch.clamp(n, max=0.45) / n
    xyz = off + torch.stack([torch.zeros(P), yc, torch.zeros(P)], dim=1)
    scaling, rotation, opacity, f_dc, f_rest, objects = _common_attrs(
        g, P, log_scale_mean, 0.4, 0.0, 1.5, False, sh_degree)
    return dict(xyz=xyz, features_dc=f_dc, features_rest=f_rest, scaling=scaling, rotation=rotation,
                opacity=opacity, objects_dc=objects)


def _box_surface_points(g: torch.Generator, n: int, nboxes: int, extent: float, hmin: float, hmax: float):
    cx = (torch.rand(nboxes, generator=g) - 0.5) * 2 * extent * 0.8
    cy = (torch.rand(nboxes, generator=g) - 0.5) * 2 * extent * 0.8
    sx = torch.rand(nboxes, generator=g) * 4 + 2
    sy = torch.rand(nboxes, generator=g) * 4 + 2
    hz = torch.rand(nboxes, generator=g) * (hmax - hmin) + hmin
    # face areas: 2 walls of sx*hz, 2 walls of sy*hz, roof sx*sy
    areas = torch.stack([sx * hz, sx * hz, sy * hz, sy * hz, sx * sy], dim=1)   # [B,5]
    flat = areas.flatten()
    pick = torch.multinomial(flat / flat.sum(), n, replacement=True, generator=g)
    b, f = pick // 5, pick % 5
    u = torch.rand(n, generator=g)
    v = torch.rand(n, generator=g)
    x = torch.empty(n)
    y = torch.empty(n)
    z = torch.empty(n)
    bx, by, bsx, bsy, bh = cx[b], cy[b], sx[b], sy[b], hz[b]
    m = f == 0
    x[m], y[m], z[m] = (bx + (u - 0.5) * bsx)[m], (by - 0.5 * bsy)[m], (v * bh)[m]
    m = f == 1
    x[m], y[m], z[m] = (bx + (u - 0.5) * bsx)[m], (by + 0.5 * bsy)[m], (v * bh)[m]
    m = f == 2
    x[m], y[m], z[m] = (bx - 0.5 * bsx)[m], (by + (u - 0.5) * bsy)[m], (v * bh)[m]
    m = f == 3
    x[m], y[m], z[m] = (bx + 0.5 * bsx)[m], (by + (u - 0.5) * bsy)[m], (v * bh)[m]
    m = f == 4
    x[m], y[m], z[m] = (bx + (u - 0.5) * bsx)[m], (by + (v - 0.5) * bsy)[m], bh[m]
    return torch.stack([x, y, z], dim=1)


def city(P: int, seed: int, extent: float, nboxes: int, sh_degree: int = 3,
         log_scale_mean: float = math.log(0.04)):
    """Ground slab (60 %), box surfaces (35 %), diffuse (5 %); world z is up (configs 3-5)."""
    g = torch.Generator().manual_seed(seed)
    n_ground = int(P * 0.60)
    n_box = int(P * 0.35)
    n_diff = P - n_ground - n_box
    ground = torch.stack([(torch.rand(n_ground, generator=g) - 0.5) * 2 * extent,
                          (torch.rand(n_ground, generator=g) - 0.5) * 2 * extent,
                          torch.rand(n_ground, generator=g) * 0.2], dim=1)
    boxes = _box_surface_points(g, n_box, nboxes, extent, 3.0, 25.0)
    diffuse = torch.stack([(torch.rand(n_diff, generator=g) - 0.5) * 2 * extent,
                           (torch.rand(n_diff, generator=g) - 0.5) * 2 * extent,
                           torch.rand(n_diff, generator=g) * 25.0], dim=1)
    xyz = torch.cat([ground, boxes, diffuse], dim=0)
    perm = torch.randperm(P, generator=g)          # storage order carries no spatial meaning
    xyz = xyz[perm]
    scaling, rotation, opacity, f_dc, f_rest, objects = _common_attrs(
        g, P, log_scale_mean, 0.5, 1.0, 2.0, True, sh_degree)
    return dict(xyz=xyz, features_dc=f_dc, features_rest=f_rest, scaling=scaling, rotation=rotation,
                opacity=opacity, objects_dc=objects)


def ring_cameras(n: int, radius: float, height: float, target, fovx: float, width: int, height_px: int,
                 device="cpu") -> List[Camera]:
    cams = []
    for i in range(n):
        th = 2.0 * math.pi * i / n + 0.3
        eye = (radius * math.cos(th), radius * math.sin(th), height)
        cams.append(look_at_camera(eye, target, up=(0.0, 0.0, 1.0), fovx=fovx, width=width,
                                   height=height_px, uid=i, device=device))
    return cams


def make_scene(key: str, device="cpu", P: int | None = None, width: int | None = None,
               height: int | None = None, n_views: int = 8):
    """-> (GaussianModel, [Camera], SceneSpec).  ``P`` / ``width`` / ``height`` override the spec
    (used for scaled-down parity cases of the same distribution)."""
    spec = SPECS[key]
    P = P or spec.P
    W = width or spec.width
    H = height or spec.height
    if key.startswith("hydrant"):
        attrs = hydrant(P, spec.seed, math.log(0.03) if key == "hydrant-1k" else math.log(0.008))
        cams = []
        for i in range(n_views):
            th = 2.0 * math.pi * i / max(n_views, 1)
            eye = (2.0 * math.sin(th), -0.3, -2.0 * math.cos(th))
            cams.append(look_at_camera(eye, (0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0), fovx=spec.fovx, fovy=spec.fovx
                                       if W == H else None, width=W, height=H, uid=i, device=device))
    elif key == "nyc-1M":
        attrs = city(P, spec.seed, 20.0, 60)
        cams = ring_cameras(n_views, 30.0, 12.0, (0.0, 0.0, 3.0), spec.fovx, W, H, device)
    elif key == "airport-4K":
        attrs = city(P, spec.seed, 100.0, 200)
        cams = ring_cameras(n_views, 120.0, 40.0, (0.0, 0.0, 3.0), spec.fovx, W, H, device)
    else:
        raise KeyError(key)
    model = GaussianModel.from_tensors(**attrs, device=device)
    return model, cams, spec
