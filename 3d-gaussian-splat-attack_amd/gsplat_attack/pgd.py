"""Projected-gradient steps on raw Gaussian attributes (the DAGGER update rules).

Counterpart of the reference's ``attack.py:25-173``: one L-inf and one L2 rule, applied to position,
rotation, opacity, scaling or SH colour.  Semantics kept exactly:
  * steps are TARGETED (the perturbation is multiplied by -1: attack.py:28,63,127-128,159-160);
  * L-inf: ``x += -alpha*sign(grad)`` then ``clamp(x - x0, -eps, eps) + x0``;
  * L2: ``x += -alpha*grad/||grad||_2`` with the norm over the WHOLE tensor (zero step when the norm is
    0), then ``(x - x0).renorm(p=2, dim=0, maxnorm=eps)`` -- i.e. each Gaussian's row is clipped to an
    eps-ball on its own;
  * colour steps treat ``_features_rest`` and ``_features_dc`` separately (attack.py:138-173);
  * updates act in place on the raw (pre-activation) parameters and read ``.grad``.
"""
from __future__ import annotations

import torch


_WARNED = set()


def _hip_step(x: torch.Tensor, grad, alpha: float, epsilon: float, x0: torch.Tensor, l2: bool, sumsq=None) -> bool:
    """Device tensors the fused HIP update can take -- contiguous float32 [rows, <= 48 columns], which is every tensor
    the reference attacks -- take it (libgsraster.so: gsr_pgd_step, two launches per tensor) and True is returned.
    False = the caller runs the tensor formulation below: host tensors (the CPU statement the golden fixtures pin),
    and device tensors in another layout (float64, non-contiguous, wider rows, 0-dim), with one warning per layout --
    the same arithmetic through tensor ops, slower.  Mismatched shapes or devices raise."""
    if not x.is_cuda:
        return False
    if grad is None:
        # no gradient reached this tensor (a rank without views, a frozen group): zero step, projection only
        grad = torch.zeros_like(x)
    if not (grad.is_cuda and x0.is_cuda) or x.shape != grad.shape or x.shape != x0.shape:
        raise ValueError(f"pgd step: x {tuple(x.shape)} on {x.device}, grad {tuple(grad.shape)} on {grad.device}, "
                         f"x0 {tuple(x0.shape)} on {x0.device} must agree")
    if x.numel() == 0:
        return True
    rows = x.shape[0] if x.dim() >= 1 else 0
    cols = x.numel() // rows if rows else 0
    if x.dim() < 1 or x.dtype != torch.float32 or not x.is_contiguous() or cols > 48:
        key = (x.dtype, x.is_contiguous(), x.dim() < 1, cols > 48)
        if key not in _WARNED:
            _WARNED.add(key)
            import warnings
            warnings.warn("gsplat_attack.pgd: gsr_pgd_step updates contiguous float32 [rows, <=48] tensors; this one "
                          f"({x.dtype}, contiguous={x.is_contiguous()}, shape {tuple(x.shape)}) takes the tensor-op "
                          "formulation instead", stacklevel=3)
        return False
    grad = grad.to(torch.float32).contiguous()
    x0 = x0.to(torch.float32).contiguous()
    import ctypes
    import diff_gaussian_rasterization as D
    lib = D._load()
    with torch.cuda.device(x.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if l2 and sumsq is not None:
            # ||grad||^2 is already on the device (left by the raster backward that wrote `grad`: GradNorms): one launch,
            # one read of the gradient
            assert sumsq.dtype == torch.float64 and sumsq.is_cuda and sumsq.numel() == 1
            rc = lib.gsr_pgd_step_normed(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(grad.data_ptr()),
                                         ctypes.c_void_p(x0.data_ptr()), ctypes.c_int64(rows), ctypes.c_int32(cols),
                                         ctypes.c_float(alpha), ctypes.c_float(epsilon), ctypes.c_void_p(sumsq.data_ptr()), stream)
        else:
            rc = lib.gsr_pgd_step(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(grad.data_ptr()), ctypes.c_void_p(x0.data_ptr()),
                                  ctypes.c_int64(rows), ctypes.c_int32(cols), ctypes.c_float(alpha), ctypes.c_float(epsilon),
                                  ctypes.c_int32(1 if l2 else 0), stream)
    if rc != 0:
        raise RuntimeError(lib.gsr_last_error().decode())
    # the kernel wrote through the raw pointer: tell autograd (a rasteriser forward that saved this tensor and has not
    # run its backward yet must see the change, diff_gaussian_rasterization checks the versions)
    torch.autograd.graph.increment_version(x)
    return True


def multi_step_(items, alpha: float, epsilon: float, l2: bool) -> bool:
    """The update of several tensors of one model in ONE launch (libgsraster.so: gsr_pgd_step_multi; bit for bit what
    l2_step_ / linf_step_ give tensor by tensor).  items: [(x, grad, x0, sumsq or None), ...], at most 8.  True = done;
    False = some tensor is not one the fused update takes (host tensors, another dtype or layout, no gradient): nothing
    was touched and the caller steps tensor by tensor."""
    import ctypes
    items = list(items)
    if not items or len(items) > 8:
        return False
    dev = items[0][0].device
    prepared = []
    for x, grad, x0, ss in items:
        x = x.detach()
        if grad is None or not (x.is_cuda and grad.is_cuda and x0.is_cuda) or x.device != dev:
            return False
        if x.shape != grad.shape or x.shape != x0.shape or x.dim() < 1 or x.numel() == 0:
            return False
        rows = x.shape[0]
        cols = x.numel() // rows
        if cols > 48 or any(t.dtype != torch.float32 or not t.is_contiguous() for t in (x, grad, x0)):
            return False
        if ss is not None and not (ss.dtype == torch.float64 and ss.is_cuda and ss.numel() == 1):
            return False
        prepared.append((x, grad, x0, ss, rows, cols))
    import diff_gaussian_rasterization as D
    lib = D._load()
    n = len(prepared)
    ptrs = lambda k: (ctypes.c_void_p * n)(*[(None if p[k] is None else p[k].data_ptr()) for p in prepared])   # noqa: E731
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.gsr_pgd_step_multi(ctypes.c_int32(n), ptrs(0), ptrs(1), ptrs(2), (ctypes.c_int64 * n)(*[p[4] for p in prepared]),
                                    (ctypes.c_int32 * n)(*[p[5] for p in prepared]), (ctypes.c_float * n)(*([alpha] * n)),
                                    (ctypes.c_float * n)(*([epsilon] * n)), ctypes.c_int32(1 if l2 else 0), ptrs(3), stream)
    if rc != 0:
        raise RuntimeError(lib.gsr_last_error().decode())
    for p in prepared:
        torch.autograd.graph.increment_version(p[0])
    return True


def linf_step_(x: torch.Tensor, grad: torch.Tensor, alpha: float, epsilon: float, x0: torch.Tensor) -> None:
    if _hip_step(x.detach(), grad, alpha, epsilon, x0, False):
        return
    if grad is None:
        grad = torch.zeros_like(x)
    with torch.no_grad():
        x.add_(torch.sign(grad), alpha=-alpha)
        x.sub_(x0).clamp_(-epsilon, epsilon).add_(x0)


def l2_step_(x: torch.Tensor, grad: torch.Tensor, alpha: float, epsilon: float, x0: torch.Tensor, sumsq=None) -> None:
    """sumsq (device tensors only): a one-element float64 device tensor holding ||grad||^2, when the caller already has
    it (diff_gaussian_rasterization.GradNorms); None: the norm is summed here."""
    if _hip_step(x.detach(), grad, alpha, epsilon, x0, True, sumsq=sumsq if grad is not None else None):
        return
    if grad is None:
        grad = torch.zeros_like(x)
    with torch.no_grad():
        norm = torch.linalg.vector_norm(grad.reshape(-1), ord=2)
        # branch-free form of "if norm > 0 ... else zero step" (no host sync on the device path)
        step = torch.where(norm > 0, grad / norm.clamp_min(torch.finfo(grad.dtype).tiny), torch.zeros_like(grad))
        x.add_(step, alpha=-alpha)
        d = x - x0
        delta = d.renorm(p=2, dim=0, maxnorm=epsilon) if d.dim() >= 2 else d.reshape(-1, 1).renorm(p=2, dim=0, maxnorm=epsilon).reshape(d.shape)
        x.copy_(x0 + delta)


def _single(attr):
    def linf(gaussian, alpha, epsilon, original):
        t = getattr(gaussian, attr)
        linf_step_(t, t.grad, alpha, epsilon, original)

    def l2(gaussian, alpha, epsilon, original):
        t = getattr(gaussian, attr)
        l2_step_(t, t.grad, alpha, epsilon, original)
    return linf, l2


gaussian_position_linf_attack, gaussian_position_l2_attack = _single("_xyz")
gaussian_rotation_linf_attack, gaussian_rotation_l2_attack = _single("_rotation")
gaussian_opacity_linf_attack, gaussian_opacity_l2_attack = _single("_opacity")
gaussian_scaling_linf_attack, gaussian_scaling_l2_attack = _single("_scaling")


def gaussian_color_linf_attack(gaussian, alpha, epsilon, features_rest, features_dc):
    linf_step_(gaussian._features_rest, gaussian._features_rest.grad, alpha, epsilon, features_rest)
    linf_step_(gaussian._features_dc, gaussian._features_dc.grad, alpha, epsilon, features_dc)


def gaussian_color_l2_attack(gaussian, alpha, epsilon, features_rest, features_dc):
    l2_step_(gaussian._features_rest, gaussian._features_rest.grad, alpha, epsilon, features_rest)
    l2_step_(gaussian._features_dc, gaussian._features_dc.grad, alpha, epsilon, features_dc)
