"""render(): the boundary between scene state and the rasteriser.

Counterpart of the reference's ``gaussian_renderer/__init__.py:18-103`` with the same signature, the
same settings construction (:33-49), the same keyword call into ``GaussianRasterizer`` (:86-95) and the
same returned dict (:99-103; image NOT clamped).  The zero ``screenspace_points`` tensor is created on
the model's device instead of the hard-coded "cuda".
"""
from __future__ import annotations

import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_raw

from .sh import eval_sh


class PipelineParams:
    """The three switches render() reads (reference attack.py:254-256, configs/config.yaml:61-63)."""

    def __init__(self, convert_SHs_python: bool = False, compute_cov3D_python: bool = False, debug: bool = False,
                 skip_objects: bool = False, fused_activations: bool = True, viewspace_grad: bool = True):
        self.convert_SHs_python = convert_SHs_python
        self.compute_cov3D_python = compute_cov3D_python
        self.debug = debug
        # extension (results unchanged): hand the model's RAW parameters to the rasteriser and let its kernels apply
        # exp / sigmoid / normalize / cat and their chain rule, instead of materialising activated copies in HBM
        self.fused_activations = fused_activations
        # extension (default off = reference behaviour): do not composite the 16 object-feature channels,
        # which the attack never reads (``render_object`` is then all zeros)
        self.skip_objects = skip_objects
        # extension (default on = reference behaviour): `viewspace_points` receives the screen-space gradient.  A
        # colour-only attack turns it off (and freezes the geometry parameters): the backward then runs without the
        # geometry sums and the projection chain rule (BASELINE configs 2 and 3)
        self.viewspace_grad = viewspace_grad


def _has_raw_layout(pc) -> bool:
    """The fused path needs the reference model's storage: degree-3 SH split in _features_dc / _features_rest."""
    try:
        P = pc._xyz.shape[0]
        return (pc._xyz.is_cuda and tuple(pc._features_dc.shape) == (P, 1, 3)
                and tuple(pc._features_rest.shape) == (P, 15, 3) and pc._opacity.numel() == P)
    except AttributeError:
        return False


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    # The reference builds `zeros_like(...) + 0` and calls retain_grad() on that non-leaf (:25-29); a zero LEAF gives
    # callers the same thing (values 0, .grad filled by backward) without an add kernel and a gradient copy per view.
    screenspace_points = torch.zeros_like(pc.get_xyz, requires_grad=bool(getattr(pipe, "viewspace_grad", True)))

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=pipe.debug,
    )
    if (getattr(pipe, "fused_activations", False) and override_color is None and not pipe.convert_SHs_python
            and not pipe.compute_cov3D_python and _has_raw_layout(pc)):
        rendered_image, radii, rendered_objects = rasterize_gaussians_raw(
            pc._xyz, screenspace_points, pc._features_dc, pc._features_rest,
            None if getattr(pipe, "skip_objects", False) else pc._objects_dc, pc._opacity, pc._scaling, pc._rotation,
            raster_settings)
        return {"render": rendered_image,
                "viewspace_points": screenspace_points,
                "visibility_filter": radii > 0,
                "radii": radii,
                "render_object": rendered_objects}

    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = pc.get_xyz
    means2D = screenspace_points
    opacity = pc.get_opacity

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation

    # The reference leaves `sh_objs` unbound on the two Python-colour branches;
    # here the object features are always passed, which is what its only working branch does.
    shs = colors_precomp = None
    sh_objs = None if getattr(pipe, "skip_objects", False) else pc.get_objects
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii, rendered_objects = rasterizer(
        means3D=means3D,
        means2D=means2D,
        shs=shs,
        sh_objs=sh_objs,
        colors_precomp=colors_precomp,
        opacities=opacity,
        scales=scales,
        rotations=rotations,
        cov3D_precomp=cov3D_precomp)

    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii,
            "render_object": rendered_objects}
