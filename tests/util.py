"""Shared helpers for the parity tests (oracle side + HIP side)."""
import math

import torch

from oracle import oracle_r as O


def settings_for(cam, bg, sh_degree=3, scale_modifier=1.0, cls=O.Settings, device=None):
    def dev(t):
        return t if device is None else t.to(device)
    return cls(int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
               dev(bg), scale_modifier, dev(cam.world_view_transform), dev(cam.full_proj_transform), sh_degree,
               dev(cam.camera_center), False, False)


def model_inputs(model, with_objs=True):
    """Activated attributes exactly as render() hands them to the rasteriser."""
    d = dict(means3D=model.get_xyz.detach(), shs=model.get_features.detach(), opacities=model.get_opacity.detach(),
             scales=model.get_scaling.detach(), rotations=model.get_rotation.detach())
    if with_objs:
        d["sh_objs"] = model.get_objects.detach()
    return d


def grad_error(g, ref, elem_tol=5e-3):
    """(normwise max error / max|ref|,  fraction of significant elements whose relative error exceeds
    elem_tol).  Significant = within 3 decades of the group's largest magnitude."""
    g = g.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    scale = ref.abs().max().item()
    if scale == 0.0:
        return g.abs().max().item(), 0.0
    err = (g - ref).abs()
    norm = (err.max() / scale).item()
    big = ref.abs() > 1e-3 * scale
    frac = ((err[big] / ref.abs()[big]) > elem_tol).double().mean().item() if big.any() else 0.0
    return norm, frac
