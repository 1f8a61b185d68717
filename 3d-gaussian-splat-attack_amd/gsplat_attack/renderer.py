"""render(): the boundary between scene state and the rasteriser.

Counterpart of the reference's ``gaussian_renderer/__init__.py:18-103`` with the same signature, the
same settings construction (:33-49), the same keyword call into ``GaussianRasterizer`` (:86-95) and the
same returned dict (:99-103; image NOT clamped).  The zero ``screenspace_points`` tensor is created on
the model's device instead of the hard-coded "cuda".
"""
from __future__ import annotations

import math

import torch

from diff_gaussian_rasterization import (MAX_BATCH, GaussianRasterizationSettings, GaussianRasterizer, _zero_scalar,
                                         rasterize_gaussians_raw, rasterize_gaussians_raw2, rasterize_gaussians_raw2_batch,
                                         rasterize_gaussians_raw_batch)

from .sh import eval_sh


class PipelineParams:
    """The three switches render() reads (reference attack.py:254-256, configs/config.yaml:61-63)."""

    def __init__(self, convert_SHs_python: bool = False, compute_cov3D_python: bool = False, debug: bool = False,
                 skip_objects: bool = False, fused_activations: bool = True, viewspace_grad: bool = True,
                 grad_bucket=None, render_cache=None, grad_norms=None):
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
        # extension (default None = autograd fills .grad): a diff_gaussian_rasterization.GradBucket -- or a callable
        # returning the bucket to use for this call (one per stream) -- that receives the 59 attribute gradients of the
        # fused path directly; see GradBucket
        self.grad_bucket = grad_bucket
        # extension (default None = every render runs the whole forward): a diff_gaussian_rasterization.RenderCache.  A
        # render of a camera whose geometry inputs (means, opacities, scales, rotations -- the same tensors at the same
        # versions) and camera tensors are unchanged since the last render of that camera through this cache re-uses
        # that render's projection, sorts and tile lists and runs the colour kernel and the compositor only: what every
        # iteration after the first of a colour attack is (reference attack.py:25-49).  Same bits either way.
        self.render_cache = render_cache
        self.cache_tag = "view"       # namespace of this pipe's keys in the cache (the success check renders under its own)
        # extension (default None): a diff_gaussian_rasterization.GradNorms -- the fused path's backward leaves the sums of
        # squares of the gradients it writes there, for the L2 step rules (no second pass over the gradient)
        self.grad_norms = grad_norms
        # extension (default on; same bits either way): the success renders of two or more cameras (attack.render_combined)
        # go through one launch chain (render_batch / render_pair_batch) instead of one forward per camera
        self.batched_checks = True


def _has_raw_layout(pc) -> bool:
    """The fused path needs the reference model's storage: degree-3 SH split in _features_dc / _features_rest."""
    try:
        P = pc._xyz.shape[0]
        return (pc._xyz.is_cuda and tuple(pc._features_dc.shape) == (P, 1, 3)
                and tuple(pc._features_rest.shape) == (P, 15, 3) and pc._opacity.numel() == P)
    except AttributeError:
        return False


def takes_fused_path(pc, pipe, override_color=None) -> bool:
    """True when render() hands the model's RAW parameters to the rasteriser (gsr_forward_raw): the only path that fills a
    GradBucket.  The reference's two Python switches (gaussian_renderer/__init__.py:62-63,70-78) and an override colour
    take the classic activated-tensor surface."""
    return bool(getattr(pipe, "fused_activations", True) and override_color is None
                and not getattr(pipe, "convert_SHs_python", False) and not getattr(pipe, "compute_cov3D_python", False)
                and _has_raw_layout(pc))


def _settings(cam, pc, pipe, bg_color, scaling_modifier) -> GaussianRasterizationSettings:
    """The 12 fields in call-site order (reference gaussian_renderer/__init__.py:33-49): tan(FoV/2), integer image size,
    the camera's transposed matrices, the model's ACTIVE SH degree, prefiltered always False."""
    half_x, half_y = 0.5 * cam.FoVx, 0.5 * cam.FoVy
    return GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), math.tan(half_x), math.tan(half_y),
                                         bg_color, scaling_modifier, cam.world_view_transform, cam.full_proj_transform,
                                         pc.active_sh_degree, cam.camera_center, False, pipe.debug)


class RenderResult(dict):
    """The dict of reference gaussian_renderer/__init__.py:99-103, with one entry made on demand: `visibility_filter`
    (= radii > 0) costs a compare kernel per view that the attack never looks at (training code does), so it is
    computed -- on the stream current at that moment -- the first time it is read.  The key is present from the start
    (`in`, keys(), len() see it); items() / values() / get() / [] / pop() resolve it."""
    _LAZY = "visibility_filter"

    def _resolve(self):
        v = dict.__getitem__(self, self._LAZY)
        if v is None:
            v = dict.__getitem__(self, "radii") > 0
            dict.__setitem__(self, self._LAZY, v)
        return v

    def __getitem__(self, key):
        return self._resolve() if key == self._LAZY else dict.__getitem__(self, key)

    # dict(result), {**result}, f(**result) and other.update(result) copy a dict SUBCLASS through PyDict_Merge's fast path
    # (raw stored values, no __getitem__) unless the subclass has its own __iter__: with these two they go through
    # keys() + __getitem__ and see the resolved entry
    def __iter__(self):
        return dict.__iter__(self)

    def keys(self):
        return dict.keys(self)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        if key == self._LAZY and key in self:
            self._resolve()
        return dict.pop(self, key, *default)

    def items(self):
        self._resolve()
        return dict.items(self)

    def values(self):
        self._resolve()
        return dict.values(self)

    def copy(self):
        self._resolve()
        return dict(self)

    def __repr__(self):
        self._resolve()
        return dict.__repr__(self)


def _result(image, screenspace_points, radii, objects) -> dict:
    """The dict of reference gaussian_renderer/__init__.py:99-103 (the image is NOT clamped)."""
    return RenderResult(render=image, viewspace_points=screenspace_points, visibility_filter=None, radii=radii,
                        render_object=objects)


def _python_colours(cam, pc):
    """convert_SHs_python branch (reference :70-78): SH evaluated with tensor ops, +0.5, clamped at 0."""
    n = pc.get_features.shape[0]
    coeffs = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    rays = pc.get_xyz - cam.camera_center.repeat(n, 1)
    rays = rays / rays.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, coeffs, rays) + 0.5, 0.0)


_ZEROS = {}      # (device, shape, dtype) -> (zero tensor, its _version): the storage every view's screenspace_points aliases


def _zero_points(like: torch.Tensor, requires_grad: bool) -> torch.Tensor:
    """A fresh LEAF of zeros shaped like `like` without a fill kernel per view: the values are never written by the
    rasteriser (only .grad is), so all views alias one zero buffer per (device, shape).  Should a caller write into
    one of them in place, the buffer's version changes and it is replaced."""
    if not like.is_cuda:
        return torch.zeros_like(like, requires_grad=requires_grad)
    key = (like.device, tuple(like.shape), like.dtype)
    hit = _ZEROS.get(key)
    if hit is None or hit[0]._version != hit[1]:
        z = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
        if len(_ZEROS) > 64:
            _ZEROS.clear()
        hit = _ZEROS[key] = (z, z._version)
        # the fill runs on the current stream; views rendered on other streams must not read ahead of it
        torch.cuda.current_stream(like.device).synchronize()
    return hit[0].detach().requires_grad_(requires_grad)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    # The reference builds `zeros_like(...) + 0` and calls retain_grad() on that non-leaf (:25-29); a zero LEAF gives
    # callers the same thing (values 0, .grad filled by backward) without an add kernel and a gradient copy per view.
    screenspace_points = _zero_points(pc.get_xyz, bool(getattr(pipe, "viewspace_grad", True)))
    st = _settings(viewpoint_camera, pc, pipe, bg_color, scaling_modifier)
    no_objects = bool(getattr(pipe, "skip_objects", False))

    if takes_fused_path(pc, pipe, override_color):
        bucket = getattr(pipe, "grad_bucket", None)
        if callable(bucket):
            bucket = bucket()
        image, radii, objects = rasterize_gaussians_raw(
            pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, None if no_objects else pc._objects_dc,
            pc._opacity, pc._scaling, pc._rotation, st, grad_bucket=bucket,
            cache=getattr(pipe, "render_cache", None), cache_key=(getattr(pipe, "cache_tag", "view"), id(viewpoint_camera)),
            grad_norms=getattr(pipe, "grad_norms", None))
        return _result(image, screenspace_points, radii, objects)

    # classic surface: activated tensors through the keyword call of reference :86-95
    kw = dict(means3D=pc.get_xyz, means2D=screenspace_points, opacities=pc.get_opacity, shs=None, colors_precomp=None,
              scales=None, rotations=None, cov3D_precomp=None,
              # the reference leaves `sh_objs` unbound on its two Python-colour branches;
              # here the object features are always passed, which is what its only working branch does
              sh_objs=None if no_objects else pc.get_objects)
    if pipe.compute_cov3D_python:
        kw["cov3D_precomp"] = pc.get_covariance(scaling_modifier)
    else:
        kw["scales"], kw["rotations"] = pc.get_scaling, pc.get_rotation
    if override_color is not None:
        kw["colors_precomp"] = override_color
    elif pipe.convert_SHs_python:
        kw["colors_precomp"] = _python_colours(viewpoint_camera, pc)
    else:
        kw["shs"] = pc.get_features
    image, radii, objects = GaussianRasterizer(raster_settings=st)(**kw)
    return _result(image, screenspace_points, radii, objects)


def can_batch(cameras, pc, pipe, override_color=None) -> bool:
    """True when render_batch() can take `cameras` through one launch chain: the fused raw-parameter path, no object
    channels, one image size, at most MAX_BATCH views."""
    cams = list(cameras)
    return bool(1 <= len(cams) <= MAX_BATCH and takes_fused_path(pc, pipe, override_color)
                and bool(getattr(pipe, "skip_objects", False))
                and len({(int(c.image_height), int(c.image_width)) for c in cams}) == 1)


def render_batch(cameras, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0):
    """render() of a BATCH of cameras of one model through ONE launch chain (gsr_forward_raw_batch).  The reference's batch is
    a Python loop of render() calls whose backward passes add up in .grad (attack.py:476-494); here the B views are one
    virtual scene: one scan, one depth sort, one emission, one tile sort, one schedule, one forward and one backward
    composite over all B images, and the 59 attribute gradients per Gaussian are written once for the batch.
    -> dict like render()'s with a leading view axis: render [B,3,H,W], viewspace_points [B,P,3] (a zero leaf whose .grad
    receives every view's screen-space gradient), radii [B,P], visibility_filter (lazy) [B,P]; render_object is a
    broadcast zero (no object channels: needs pipe.skip_objects).  Every image and radius is bit for bit render()'s for that
    camera; the gradients are those of the B render() calls summed in view order.  With pipe.render_cache (a RenderCache)
    the batch's context is kept under the tuple of cameras: a later batch of the same cameras whose geometry tensors are
    unchanged (a colour attack) runs the batch's colour kernel and one compositor launch over the kept lists."""
    cams = list(cameras)
    if not can_batch(cams, pc, pipe):
        raise ValueError("render_batch needs the fused raw-parameter path without object channels (PipelineParams("
                         f"skip_objects=True)), one image size and 1..{MAX_BATCH} cameras")
    B, P = len(cams), int(pc.get_xyz.shape[0])
    want_vs = bool(getattr(pipe, "viewspace_grad", True))
    # one zero leaf [B,P,3] (no allocation per call: _zero_points keeps one zero buffer per shape)
    screenspace_points = _zero_points(pc.get_xyz.detach().unsqueeze(0).expand(B, P, 3), True) if want_vs else None
    sts = [_settings(cam, pc, pipe, bg_color, scaling_modifier) for cam in cams]
    bucket = getattr(pipe, "grad_bucket", None)
    if callable(bucket):
        bucket = bucket()
    image, radii = rasterize_gaussians_raw_batch(pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity,
                                                 pc._scaling, pc._rotation, sts, grad_bucket=bucket,
                                                 grad_norms=getattr(pipe, "grad_norms", None),
                                                 cache=getattr(pipe, "render_cache", None),
                                                 cache_key=(getattr(pipe, "cache_tag", "view"), "batch") + tuple(id(c) for c in cams))
    objects = _zero_scalar(image.device).unsqueeze(0).expand(B, 16, image.shape[2], image.shape[3])
    return _result(image, screenspace_points, radii, objects)


@torch.no_grad()
def render_pair(viewpoint_camera, pc_a, pc_b, pipe, bg_color: torch.Tensor, scaling_modifier=1.0):
    """render() of the scene "pc_a followed by pc_b" without building it: the attacked target plus the frozen
    background, which the reference re-renders after every PGD step from a deep copy with all seven tensors
    concatenated (attack.py:513-530).  Forward only (the reference never differentiates that render); same dict as
    render(), `viewspace_points` None, the Gaussian-indexed entries cover pc_a then pc_b."""
    if not (_has_raw_layout(pc_a) and _has_raw_layout(pc_b)):
        raise ValueError("render_pair needs two models in the reference's raw storage layout on a HIP device")
    st = _settings(viewpoint_camera, pc_a, pipe, bg_color, scaling_modifier)

    def raw(pc):
        return (pc._xyz, pc._features_dc, pc._features_rest, pc._objects_dc, pc._opacity, pc._scaling, pc._rotation)
    image, radii, objects = rasterize_gaussians_raw2(raw(pc_a), raw(pc_b), st,
                                                     objects=not bool(getattr(pipe, "skip_objects", False)),
                                                     cache=getattr(pipe, "render_cache", None),
                                                     cache_key=("pair", id(viewpoint_camera)))
    return _result(image, None, radii, objects)


@torch.no_grad()
def render_pair_batch(cameras, pc_a, pc_b, pipe, bg_color: torch.Tensor, scaling_modifier=1.0):
    """render_pair() of a BATCH of cameras through one launch chain (gsr_forward_raw2_batch): the success renders of the
    attack's batch (reference attack.py:513-530, once per camera of :476-485).  -> dict with a leading view axis like
    render_batch()'s, `viewspace_points` None; every image bit for bit render_pair()'s for that camera.  Needs
    pipe.skip_objects (no object channels in a batch), one image size, 1..MAX_BATCH cameras."""
    cams = list(cameras)
    if not (_has_raw_layout(pc_a) and _has_raw_layout(pc_b)):
        raise ValueError("render_pair_batch needs two models in the reference's raw storage layout on a HIP device")
    if not can_batch(cams, pc_a, pipe):
        raise ValueError("render_pair_batch needs PipelineParams(skip_objects=True), one image size and "
                         f"1..{MAX_BATCH} cameras")
    sts = [_settings(cam, pc_a, pipe, bg_color, scaling_modifier) for cam in cams]

    def raw(pc):
        return (pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation)
    image, radii = rasterize_gaussians_raw2_batch(raw(pc_a), raw(pc_b), sts, cache=getattr(pipe, "render_cache", None),
                                                  cache_key=("pair", "batch") + tuple(id(c) for c in cams))
    objects = _zero_scalar(image.device).unsqueeze(0).expand(len(cams), 16, image.shape[2], image.shape[3])
    return _result(image, None, radii, objects)
