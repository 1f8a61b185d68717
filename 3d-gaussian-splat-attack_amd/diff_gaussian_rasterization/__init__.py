"""diff_gaussian_rasterization -- MI355X-native drop-in for the rasteriser the reference imports.

The reference does ``from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`` (reference gaussian_renderer/__init__.py:14), builds the settings with the 12
keyword fields of :36-49, constructs ``GaussianRasterizer(raster_settings=...)`` (:51) and calls it with
the keyword arguments of :86-95, getting ``(color[3,H,W], radii[P] int32, objects[16,H,W])``.  This
package provides exactly that surface on top of ``libgsraster.so`` (hand-written HIP for gfx950, C ABI
declared in include/gsraster.h), bound with ctypes: PyTorch only supplies device memory, the current
stream and autograd plumbing.

There is NO fallback: without the compiled library or without a HIP device every entry point raises.
"""
from __future__ import annotations

import contextlib
import ctypes
import weakref
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

NUM_OBJECTS = 16
_LIB_NAME = "libgsraster.so"
_lib = None

FLAG_NO_CULL = 1          # GSR_FLAG_NO_CULL (include/gsraster.h)
FLAG_NO_SEGMENTS = 1 << 16
FLAG_FWD_SHARED = 1 << 17   # the forward waves of a tile share one staging of the list (opt-in; include/gsraster.h)
FLAG_ASYNC_COUNT = 1 << 18  # GSR_FLAG_ASYNC_COUNT: never wait for the pair count (capacity guess + overflow flag)
FLAG_NO_SIDE_STREAM = 1 << 19   # GSR_FLAG_NO_SIDE_STREAM: SH -> RGB on the caller's stream instead of the side stream
FLAG_NEEDLE_DOUBLE = 1 << 20    # GSR_FLAG_NEEDLE_DOUBLE: needles' conic (and its backward) from the double chain (opt-in)
FLAG_OBJECTS_FOR_BACKWARD_ONLY = 1 << 21    # GSR_FLAG_OBJECTS_FOR_BACKWARD_ONLY: sh_objs kept for the backward, not composited
_FLAGS = int(os.environ.get("GSR_FLAGS", "0"), 0)


def flag_fwd_split(npx: int) -> int:
    """GSR_FLAG_FWD_SPLIT: pixels per lane (1|2|4) of the forward compositor, 0 = library's choice."""
    return {0: 0, 1: 1, 2: 2, 4: 3}[npx] << 4


def flag_bwd_split(npx: int) -> int:
    """GSR_FLAG_BWD_SPLIT: pixels per lane (2|4) of the backward compositor, 0 = library's choice."""
    return {0: 0, 2: 1, 4: 2}[npx] << 8


def flag_tile_map(mode: int) -> int:
    """GSR_FLAG_TILE_MAP: block -> tile map 0..3 (3 = longest list first, the default)."""
    return ((mode & 3) + 1) << 12


def set_flags(flags: int) -> None:
    """Extension flags passed in GsrSettings.flags by every later call (0 = default behaviour)."""
    global _FLAGS
    _FLAGS = int(flags)



@contextlib.contextmanager
def extra_flags(flags: int):
    """`with extra_flags(FLAG_NO_CULL): ...` -- the given bits are OR-ed into every call's flags inside the block."""
    global _FLAGS
    old = _FLAGS
    _FLAGS = old | int(flags)
    try:
        yield
    finally:
        _FLAGS = old


GSR_STAGES = ("preprocess", "depth_sort", "bin", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd")


_CHUNK_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64)   # gsr_chunk_fn


class _CSettings(ctypes.Structure):
    # field order and types mirror `struct GsrSettings` in include/gsraster.h
    _fields_ = [
        ("image_height", ctypes.c_int32),
        ("image_width", ctypes.c_int32),
        ("tanfovx", ctypes.c_float),
        ("tanfovy", ctypes.c_float),
        ("bg", ctypes.c_void_p),
        ("scale_modifier", ctypes.c_float),
        ("viewmatrix", ctypes.c_void_p),
        ("projmatrix", ctypes.c_void_p),
        ("sh_degree", ctypes.c_int32),
        ("campos", ctypes.c_void_p),
        ("prefiltered", ctypes.c_int32),
        ("debug", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
    ]


def library_path() -> str:
    # GSR_LIBRARY: another build of the same library (A/B runs of kernel variants on one box); default: the in-tree one
    return os.environ.get("GSR_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def _load():
    """Load libgsraster.so once; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"diff_gaussian_rasterization: {path} is missing. Build it with "
            f"`make -C {os.path.join(os.path.dirname(os.path.dirname(path)), 'csrc')}` (hipcc, gfx950); "
            "there is no CPU or PyTorch fallback for the raster path.")
    lib = ctypes.CDLL(path)
    vp, i32, i64p = ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)
    lib.gsr_forward.restype = ctypes.c_int
    lib.gsr_forward.argtypes = [ctypes.POINTER(_CSettings), i32, i32] + [vp] * 8 + [vp, vp, vp,
                                ctypes.POINTER(vp), i64p, vp]
    lib.gsr_backward.restype = ctypes.c_int
    lib.gsr_backward.argtypes = [vp] * 13
    lib.gsr_forward_raw.restype = ctypes.c_int
    lib.gsr_forward_raw.argtypes = [ctypes.POINTER(_CSettings), i32] + [vp] * 7 + [vp, vp, vp, ctypes.POINTER(vp), i64p, vp]
    lib.gsr_backward_raw.restype = ctypes.c_int
    lib.gsr_backward_raw.argtypes = [vp] * 12
    lib.gsr_backward_raw_into.restype = ctypes.c_int
    lib.gsr_backward_raw_into.argtypes = [vp] * 11 + [i32, vp]
    lib.gsr_backward_raw_chunked.restype = ctypes.c_int
    lib.gsr_backward_raw_chunked.argtypes = [vp] * 11 + [i32, i32, _CHUNK_FN, vp, vp]
    lib.gsr_forward_raw2.restype = ctypes.c_int
    lib.gsr_forward_raw2.argtypes = [ctypes.POINTER(_CSettings), i32] + [vp] * 7 + [i32] + [vp] * 7 + [vp, vp, vp, i64p, vp]
    lib.gsr_forward_raw2_keep.restype = ctypes.c_int
    lib.gsr_forward_raw2_keep.argtypes = ([ctypes.POINTER(_CSettings), i32] + [vp] * 7 + [i32] + [vp] * 7
                                          + [vp, vp, vp, ctypes.POINTER(vp), i64p, vp])
    if hasattr(lib, "gsr_forward_raw_batch"):              # (absent from older builds selected through GSR_LIBRARY for A/B runs)
        lib.gsr_forward_raw_batch.restype = ctypes.c_int
        lib.gsr_forward_raw_batch.argtypes = ([ctypes.POINTER(_CSettings), i32, i32] + [vp] * 6 + [vp, vp, ctypes.POINTER(vp), i64p, vp])
        lib.gsr_backward_raw_batch_into.restype = ctypes.c_int
        lib.gsr_backward_raw_batch_into.argtypes = [vp] * 9 + [i32, vp]
        lib.gsr_backward_raw_batch_views.restype = ctypes.c_int
        lib.gsr_backward_raw_batch_views.argtypes = [vp] * 9 + [ctypes.c_int64, vp]
    if hasattr(lib, "gsr_forward_raw2_batch"):
        lib.gsr_forward_raw2_batch.restype = ctypes.c_int
        lib.gsr_forward_raw2_batch.argtypes = ([ctypes.POINTER(_CSettings), i32, i32] + [vp] * 6 + [i32] + [vp] * 6
                                               + [vp, vp, ctypes.POINTER(vp), i64p, vp])
    lib.gsr_ctx_rerender.restype = ctypes.c_int
    lib.gsr_ctx_rerender.argtypes = [vp] * 8 + [ctypes.c_uint32, vp]
    lib.gsr_ctx_free.restype = None
    lib.gsr_ctx_free.argtypes = [vp]
    lib.gsr_mark_visible.restype = ctypes.c_int
    lib.gsr_mark_visible.argtypes = [ctypes.POINTER(_CSettings), i32, vp, vp, vp]
    lib.gsr_query.restype = ctypes.c_int
    lib.gsr_query.argtypes = [i32, i64p]
    lib.gsr_ctx_info.restype = ctypes.c_int
    lib.gsr_ctx_info.argtypes = [vp, i32, i64p]
    lib.gsr_ctx_export.restype = ctypes.c_int
    lib.gsr_ctx_export.argtypes = [vp, i32, vp, ctypes.c_int64, vp]
    lib.gsr_trim_pool.restype = None
    lib.gsr_trim_pool.argtypes = []
    lib.gsr_profile.restype = None
    lib.gsr_profile.argtypes = [i32]
    lib.gsr_profile_read.restype = ctypes.c_int
    lib.gsr_profile_read.argtypes = [ctypes.POINTER(ctypes.c_float), i64p]
    if hasattr(lib, "gsr_ctx_request_sumsq"):              # (absent from older builds selected through GSR_LIBRARY for A/B runs)
        lib.gsr_ctx_request_sumsq.restype = ctypes.c_int
        lib.gsr_ctx_request_sumsq.argtypes = [vp, vp]
        lib.gsr_pgd_step_normed.restype = ctypes.c_int
        lib.gsr_pgd_step_normed.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_float, vp, vp]
        lib.gsr_pgd_step_multi.restype = ctypes.c_int
        lib.gsr_pgd_step_multi.argtypes = [ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp, vp]
    lib.gsr_pgd_step.restype = ctypes.c_int
    lib.gsr_pgd_step.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_int32, vp]
    lib.gsr_test_scan.restype = ctypes.c_int
    lib.gsr_test_scan.argtypes = [vp, vp, ctypes.c_uint32, vp]
    lib.gsr_test_sort_pairs.restype = ctypes.c_int
    lib.gsr_test_sort_pairs.argtypes = [vp, vp, ctypes.c_uint32, i32, i32, i32, vp]
    lib.gsr_last_error.restype = ctypes.c_char_p
    lib.gsr_last_error.argtypes = []
    _lib = lib
    return lib


GSR_ERR_OVERFLOW = 5          # include/gsraster.h


class PairCapacityExceeded(RuntimeError):
    """FLAG_ASYNC_COUNT only: a forward emitted more (tile, Gaussian) pairs than the capacity guessed from earlier views;
    its image is NaN.  Render again (the library counts synchronously on the next forward of that size)."""


def _err(lib) -> str:
    msg = lib.gsr_last_error()
    return msg.decode("utf-8", "replace") if msg else "unknown error"


class GaussianRasterizationSettings(NamedTuple):
    """The 12 fields of the call site, in its order (reference gaussian_renderer/__init__.py:36-49)."""
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


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else ctypes.c_void_p(t.data_ptr())


_SMALL = {}      # (id(tensor), device) -> (weakref to it, its _version, dense float32 device copy, copy event, stream)


def _small_dense(t: torch.Tensor, device) -> torch.Tensor:
    """Dense float32 device form of a tiny settings tensor.  The reference's Camera keeps world_view_transform as a
    transposed VIEW and camera_center as a row of a column-major inverse (scene/cameras.py:54-57): both would cost a copy
    kernel on every render call.  The copy is made once per tensor object and version (weak reference: a dead or
    modified tensor is never served from here)."""
    if t.device == device and t.dtype == torch.float32 and t.is_contiguous():
        return t
    key = (id(t), str(device))
    hit = _SMALL.get(key)
    cur = torch.cuda.current_stream(device) if device.type == "cuda" else None
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        if cur is not None and hit[4] != cur.cuda_stream:
            cur.wait_event(hit[3])          # the copy was enqueued on another stream: order this stream behind it
        return hit[2]
    d = _f32c(t.detach(), device)
    if d is t or d.data_ptr() == t.data_ptr():
        return d
    if len(_SMALL) > 512:
        _SMALL.clear()
    try:
        ev = None
        if cur is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
        _SMALL[key] = (weakref.ref(t), t._version, d, ev, cur.cuda_stream if cur is not None else 0)
    except TypeError:
        pass
    return d


class _SettingsPack:
    """C settings + the device tensors it points into (kept alive as long as the pack lives)."""

    def __init__(self, rs: GaussianRasterizationSettings, device):
        self.device = device
        self.bg = _f32c(rs.bg.detach().flatten(), device)
        if self.bg.numel() < 3:
            raise ValueError("bg must hold at least 3 values")
        self.vm = _small_dense(rs.viewmatrix, device)
        self.pm = _small_dense(rs.projmatrix, device)
        self.cam = _small_dense(rs.campos, device).flatten()
        if self.vm.numel() != 16 or self.pm.numel() != 16 or self.cam.numel() < 3:
            raise ValueError("viewmatrix / projmatrix must be 4x4 and campos must hold 3 values")
        self.c = _CSettings(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                            self.bg.data_ptr(), float(rs.scale_modifier), self.vm.data_ptr(), self.pm.data_ptr(),
                            int(rs.sh_degree), self.cam.data_ptr(), int(bool(rs.prefiltered)), int(bool(rs.debug)),
                            _FLAGS)


class _CtxHolder:
    """Owns one GsrCtx; the workspace returns to the library's pool when the autograd graph dies."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def info(self, what: int) -> int:
        out = ctypes.c_int64(0)
        self.lib.gsr_ctx_info(self.handle, what, ctypes.byref(out))
        return out.value

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                self.lib.gsr_ctx_free(self.handle)
            except Exception:
                pass
            self.handle = None


def _versions(tensors):
    return tuple(None if t is None else t._version for t in tensors)


def _check_versions(tensors, versions):
    """The kept buffers are aliases of the caller's tensors (no copy when they are already dense float32): the library
    re-reads them in backward, so an in-place modification in between would silently change the gradients."""
    for t, v in zip(tensors, versions):
        if t is not None and t._version != v:
            raise RuntimeError("one of the tensors given to the rasteriser's forward was modified in place before its "
                               f"backward ran (version {t._version}, expected {v}); clone it or step after backward()")


def _empty_like_or_none(t):
    return None if t is None else torch.empty_like(t)


_ZERO = {}


def _zero_scalar(device):
    """One cached 1x1x1 zero per device: the object map handed back when no object features were composited."""
    z = _ZERO.get(device)
    if z is None:
        z = _ZERO[device] = torch.zeros(1, 1, 1, dtype=torch.float32, device=device)
    return z


_OBJ_ZERO = {}        # data_ptr -> (numel, _version, all zero?) of object-feature tensors that have been looked at


def _objects_all_zero(t: torch.Tensor, src: torch.Tensor) -> bool:
    """True when the object features `t` (dense float32 device tensor made from the caller's `src`) are all zero: their 16
    channels then composite to exactly zero, and the forward can run without them (the rasteriser's object variant of the
    forward compositor costs 0.27 ms against 0.16 at 1 M Gaussians / 1080p, plus 133 MB of output).  That is the attack's
    case: `combine_splats` gives every Gaussian of a combined scene zero object features (reference
    scene/gaussian_model.py:528), and the reference's render() passes them all the same (gaussian_renderer/__init__.py:81).
    The answer is cached per storage and autograd version: one reduction and one host read the first time a tensor (version)
    is seen.  A tensor found non-zero is not looked at again while it keeps its storage (a training loop that steps the
    features every iteration never pays a second read); a zero one is re-checked when its version changes."""
    if not _OBJ_SHORTCUT:
        return False
    key = t.data_ptr()
    ver = src._version
    hit = _OBJ_ZERO.get(key)
    if hit is not None and hit[0] == t.numel():
        if not hit[2]:
            return False                    # seen non-zero before: assume it still is
        if hit[1] == ver:
            return True
    zero = not bool(torch.any(t != 0).item())
    if len(_OBJ_ZERO) > 256:
        _OBJ_ZERO.clear()
    _OBJ_ZERO[key] = (t.numel(), ver, zero)
    return zero


_OBJ_SHORTCUT = os.environ.get("GSR_ZERO_OBJECT_SHORTCUT", "1") != "0"


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, sh_objs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, keep=True):
        lib = _load()
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization: tensors must live on a HIP device (got "
                               f"{means3D.device}); there is no CPU path")
        device = means3D.device
        P = int(means3D.shape[0])

        def prep(t):
            return None if t is None or t.numel() == 0 else _f32c(t.detach(), device)
        m3, shc, shoc, colc = prep(means3D), prep(sh), prep(sh_objs), prep(colors_precomp)
        opc, scc, roc, covc = prep(opacities), prep(scales), prep(rotations), prep(cov3Ds_precomp)
        # all-zero object features composite to exactly zero: the forward runs without them and hands back a broadcast
        # zero; the context keeps the features, so a backward that IS given dL/dobjects still produces dL/dsh_objs
        obj_zero = shoc is not None and shoc.numel() == P * NUM_OBJECTS and _objects_all_zero(shoc, sh_objs)
        K = 0
        if shc is not None:
            if shc.dim() != 3 or shc.shape[2] != 3 or shc.shape[0] != P:
                raise ValueError(f"shs must be [P,K,3], got {tuple(shc.shape)}")
            K = int(shc.shape[1])
        if shoc is not None and shoc.numel() != P * NUM_OBJECTS:
            raise ValueError(f"sh_objs must hold P*{NUM_OBJECTS} values, got {tuple(shoc.shape)}")
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        pack = _SettingsPack(raster_settings, device)
        color = torch.empty(3, H, W, dtype=torch.float32, device=device)
        # without object features the 16 object channels are identically zero: hand back a broadcast zero instead of
        # writing 16*H*W floats per view
        with_obj = shoc is not None and not obj_zero
        if obj_zero:
            pack.c.flags |= FLAG_OBJECTS_FOR_BACKWARD_ONLY
        objects = (torch.empty(NUM_OBJECTS, H, W, dtype=torch.float32, device=device) if with_obj
                   else _zero_scalar(device).expand(NUM_OBJECTS, H, W))
        radii = torch.empty(P, dtype=torch.int32, device=device)
        handle = ctypes.c_void_p(None)
        nren = ctypes.c_int64(0)
        # `keep` was decided by the caller of apply() (inside forward grad mode is always off and needs_input_grad is
        # True for a requires_grad tensor even under torch.no_grad()): a forward-only call keeps no backward state
        # (segment boundaries, d colour / d direction)
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            rc = lib.gsr_forward(ctypes.byref(pack.c), P, K, _ptr(m3), _ptr(shc), _ptr(shoc), _ptr(colc), _ptr(opc),
                                 _ptr(scc), _ptr(roc), _ptr(covc), _ptr(color), _ptr(objects) if with_obj else None,
                                 _ptr(radii),
                                 ctypes.byref(handle) if keep else None, ctypes.byref(nren), stream)
        if rc != 0:
            msg = _err(lib)
            if raster_settings.debug:
                torch.save({"means3D": m3, "sh": shc, "sh_objs": shoc, "colors_precomp": colc, "opacities": opc,
                            "scales": scc, "rotations": roc, "cov3D_precomp": covc,
                            "settings": raster_settings._asdict()}, "snapshot_fw.dump")
                msg += " (inputs saved to snapshot_fw.dump)"
            if rc == 1:
                raise Exception(msg)
            raise RuntimeError(msg)
        ctx.holder = _CtxHolder(lib, handle) if keep else None
        ctx.pack = pack
        ctx._nren = nren.value
        ctx.shapes = (means3D.shape, means2D.shape if means2D is not None else None,
                      None if sh is None else sh.shape, None if sh_objs is None else sh_objs.shape,
                      None if colors_precomp is None else colors_precomp.shape, opacities.shape,
                      None if scales is None else scales.shape, None if rotations is None else rotations.shape,
                      None if cov3Ds_precomp is None else cov3Ds_precomp.shape)
        # the library reads these again in backward: keep the exact (contiguous fp32) buffers alive, and remember
        # their versions -- an in-place edit between forward and backward must raise, as a saved tensor would
        ctx.kept = (m3, shc, shoc, colc, opc, scc, roc, covc)
        ctx.versions = _versions(ctx.kept)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return color, radii, objects

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_objects):
        lib = ctx.holder.lib
        _check_versions(ctx.kept, ctx.versions)
        m3, shc, shoc, colc, opc, scc, roc, covc = ctx.kept
        device = m3.device
        P = int(m3.shape[0])
        H, W = ctx.pack.c.image_height, ctx.pack.c.image_width
        if grad_color is None:
            grad_color = torch.zeros(3, H, W, dtype=torch.float32, device=device)
        gcol = _f32c(grad_color, device)
        gobj = None if (grad_objects is None or shoc is None) else _f32c(grad_objects, device)
        need = ctx.needs_input_grad

        def out(cond, *shape):
            return torch.empty(*shape, dtype=torch.float32, device=device) if cond else None
        d_m3 = out(need[0], P, 3)
        d_m2 = out(need[1], P, 3)
        d_sh = out(need[2] and shc is not None, *(shc.shape if shc is not None else (0,)))
        d_obj = out(need[3] and shoc is not None, P, NUM_OBJECTS)
        d_col = out(need[4] and colc is not None, P, 3)
        d_op = out(need[5], P)
        d_sc = out(need[6] and scc is not None, P, 3)
        d_ro = out(need[7] and roc is not None, P, 4)
        d_cov = out(need[8] and covc is not None, P, 6)
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            rc = lib.gsr_backward(ctx.holder.handle, _ptr(gcol), _ptr(gobj), _ptr(d_m3), _ptr(d_m2), _ptr(d_sh),
                                  _ptr(d_obj), _ptr(d_col), _ptr(d_op), _ptr(d_sc), _ptr(d_ro), _ptr(d_cov), stream)
        if rc != 0:
            msg = _err(lib)
            if ctx.pack.c.debug:
                torch.save({"grad_color": gcol, "grad_objects": gobj}, "snapshot_bw.dump")
                msg += " (gradients saved to snapshot_bw.dump)"
            raise (PairCapacityExceeded if rc == 5 else RuntimeError)(msg)
        if P == 0:
            for t in (d_m3, d_m2, d_sh, d_obj, d_col, d_op, d_sc, d_ro, d_cov):
                if t is not None:
                    t.zero_()
        s = ctx.shapes

        def shaped(t, shape):
            return None if t is None else t.reshape(shape)
        return (shaped(d_m3, s[0]), shaped(d_m2, s[1]), shaped(d_sh, s[2]), shaped(d_obj, s[3]), shaped(d_col, s[4]),
                shaped(d_op, s[5]), shaped(d_sc, s[6]), shaped(d_ro, s[7]), shaped(d_cov, s[8]), None, None)


class _RenderToken:
    """Held by the autograd node of a differentiable render through a kept context: while it is alive and not `done`,
    that render's backward may still come and the context must not be rendered again."""
    __slots__ = ("done", "__weakref__")

    def __init__(self):
        self.done = False


class _CacheEntry:
    __slots__ = ("holder", "sig", "refs", "pack", "gen", "event", "stream", "nren", "token")

    def __init__(self, holder, sig, refs, pack, nren):
        self.holder, self.sig, self.refs, self.pack, self.nren = holder, sig, refs, pack, nren
        self.gen = 0          # bumped by every render through this entry: a backward belongs to ONE generation
        self.event = None     # recorded after the entry's last use, on `stream`
        self.stream = None
        self.token = None     # weak reference to the _RenderToken of the entry's last differentiable render

    def busy(self) -> bool:
        t = self.token() if self.token is not None else None
        return t is not None and not t.done


class RenderCache:
    """Kept rasteriser contexts for renders whose GEOMETRY does not change from call to call -- a colour attack
    (reference attack.py:25-49: only _features_dc / _features_rest are stepped; BASELINE configs 2 and 3) renders the
    same cameras iteration after iteration with the same means, scales, rotations and opacities.  The first render of a
    key (one per camera) is an ordinary forward whose context is kept here; later renders of that key whose geometry
    inputs are the SAME tensor objects at the SAME autograd versions, under the same camera tensors / image size /
    scale modifier / SH degree / flags, run gsr_ctx_rerender: the colour half of K1 and the compositor K6 over the kept
    lists -- projection, both sorts, the emission and the schedule are not redone.  Anything else (a stepped geometry
    tensor, another camera behind the key, a non-dense input) silently takes the full forward and replaces the entry.
    Image and gradients are bit for bit those of the uncached call (tests/test_gpu_rerender.py).  "Unchanged" is what
    autograd itself goes by: an edit that bypasses a tensor's version counter (through `.data`, or a kernel writing
    through the raw pointer without bumping it) is not seen -- the reference's step functions (attack.py:25-173) and
    this package's fused step both go through the counter.

    One context per key: a render overwrites the per-pixel state the previous render's backward reads.  While the
    previous differentiable render of a key is still waiting for its backward (its graph is alive and has not been
    differentiated), another render of that key takes the full forward with a context of its own and leaves the entry
    alone; a second backward through a graph whose context has been rendered again since (retain_graph) raises.
    About 250 MB of HBM per entry
    at 1 M Gaussians and 1080p; the least recently used entry goes when `max_entries` is exceeded."""

    def __init__(self, max_entries: int = 64):
        import collections
        self.max_entries = int(max_entries)
        self.entries = collections.OrderedDict()
        self.hits = self.misses = self.bypassed = 0
        self.dropped_overflow = 0        # entries dropped because their (asynchronously counted) forward had overflowed

    def clear(self):
        self.entries.clear()

    def _lookup(self, key, sig, tensors):
        """-> (entry or None, may_store): a busy entry (see the class comment) is neither used nor replaced."""
        e = self.entries.get(key)
        if e is not None and e.busy():
            self.bypassed += 1
            return None, False
        if e is None or e.sig != sig or any(r() is not t for r, t in zip(e.refs, tensors)):
            self.misses += 1
            return None, True
        self.entries.move_to_end(key)
        self.hits += 1
        return e, True

    def _store(self, key, entry):
        self.entries[key] = entry
        self.entries.move_to_end(key)
        while len(self.entries) > self.max_entries:
            self.entries.popitem(last=False)


def _cache_sig(tensors, rs: "GaussianRasterizationSettings", extra=()):
    """What must be unchanged for a kept context to be re-rendered: storage and autograd version of the geometry and
    the camera tensors (their identity is checked through weak references besides: same object + same version means same
    shape and contents), and the scalar settings.  One flat tuple: this runs on every cached render."""
    sig = [int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
           int(rs.sh_degree), _FLAGS]
    for t in tensors:
        if t is None:
            sig.append(None)
        else:
            sig.append(t.data_ptr())
            sig.append(t._version)
    for t in (rs.viewmatrix, rs.projmatrix, rs.campos):
        sig.append(t.data_ptr())
        sig.append(t._version)
    sig.extend(extra)
    return tuple(sig)


def _entry_enter(entry: _CacheEntry, device):
    """Orders the current stream behind the entry's last use (a no-op on the same stream)."""
    cur = torch.cuda.current_stream(device)
    if entry.event is not None and entry.stream != cur.cuda_stream:
        cur.wait_event(entry.event)


def _entry_leave(entry: _CacheEntry, device):
    cur = torch.cuda.current_stream(device)
    if entry.event is None:
        entry.event = torch.cuda.Event()
    entry.event.record(cur)
    entry.stream = cur.cuda_stream


class GradNorms:
    """The six sums of squares the reference's L2 steps divide by (attack.py:53-119, 138-173: one global norm per raw
    attribute tensor, _features_dc and _features_rest separately), taken from the raster backward that WRITES the
    gradients instead of from a second pass over 236 MB of them (gsr_ctx_request_sumsq).  A rasterise call that is given
    one arms its backward when that backward overwrites its outputs; `sumsq_of(name)` hands the step the device scalar
    only while exactly ONE backward has contributed since begin() -- a second view's gradients added on top (autograd
    accumulation, a GradBucket in add mode, a folded or all-reduced bucket) make the sums stale, and the caller falls
    back to summing the gradient itself."""
    NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def __init__(self, device):
        self.sumsq = torch.zeros(6, dtype=torch.float64, device=device)
        self.writes = 0            # backwards that contributed since begin()
        self.names = ()            # tensors whose sums the one arming backward wrote

    def begin(self):
        self.writes, self.names = 0, ()

    def invalidate(self):
        self.writes = 2

    def sumsq_of(self, name: str):
        if self.writes != 1 or name not in self.names:
            return None
        i = self.NAMES.index(name)
        return self.sumsq[i:i + 1]


class GradBucket:
    """Caller-owned gradient bucket of a reference-style GaussianModel: ONE flat float32 buffer of 59 floats per Gaussian
    in the order xyz | f_dc | f_rest | opacity | scaling | rotation (the layout of the flat buffer the fused backward
    hands to autograd, and of the all-reduce bucket of gsplat_attack.dist).  A rasterise call that is given a bucket
    writes its attribute gradients straight into it -- the first backward after reset() overwrites, later ones add
    (gsr_backward_raw_into) -- and returns no gradient for those inputs to autograd: the per-view 236-byte-per-Gaussian
    gradient buffer and the framework's read-modify-write accumulation into .grad both disappear (reference
    attack.py:476-494 accumulates B views' gradients in .grad).  One bucket per stream: concurrent backwards must not
    share one."""
    CUTS = (0, 3, 6, 51, 52, 55, 59)
    NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    SHAPES = ((3,), (1, 3), (15, 3), (1,), (3,), (4,))

    def __init__(self, P: int, device):
        self.P = int(P)
        self.flat = torch.empty(59 * self.P, dtype=torch.float32, device=device)
        self.fresh = True          # the next backward overwrites instead of adding
        self.used = False          # some backward has written since reset()
        # The next backward runs its per-Gaussian stage in `chunks` launches over ranges of Gaussians and calls
        # on_chunk(chunk, g_begin, g_end) after each one is enqueued (gsr_backward_raw_chunked): the hook of a
        # multi-GPU caller that all-reduces range by range while the rest is still being computed.  One-shot:
        # both are cleared by that backward.
        self.chunks = 1
        self.on_chunk = None

    def reset(self):
        self.fresh, self.used = True, False

    def slices(self):
        P = self.P
        return [self.flat[c0 * P:c1 * P] for c0, c1 in zip(self.CUTS[:-1], self.CUTS[1:])]

    def range_slices(self, g_begin: int, g_end: int):
        """The six pieces of the flat buffer that hold the gradients of Gaussians [g_begin, g_end)."""
        P, W = self.P, (3, 3, 45, 1, 3, 4)
        return [self.flat[c0 * P + g_begin * w:c0 * P + g_end * w] for c0, w in zip(self.CUTS[:-1], W)]

    def views(self) -> dict:
        """name -> tensor view shaped like the model's parameter (e.g. _features_rest: [P,15,3])."""
        return {n: sl.view(self.P, *shp) for n, sl, shp in zip(self.NAMES, self.slices(), self.SHAPES)}

    def add_(self, other: "GradBucket"):
        """Fold another (stream's) bucket into this one.  Buckets no backward wrote to hold nothing."""
        if other.used:
            if self.used:
                self.flat.add_(other.flat)
            else:
                self.flat.copy_(other.flat)
                self.used, self.fresh = True, False
        return self

    def assign_to(self, model):
        """model.<param>.grad = the bucket's view of it (no copy).  An unused bucket means zero gradients."""
        if not self.used:
            self.flat.zero_()
            self.used, self.fresh = True, False
        for n, v in self.views().items():
            p = getattr(model, n)
            p.grad = v.view(p.shape)


class GradBucketSet:
    """B gradient buckets of one model, one per view of a batch, in ONE flat buffer [B, 59 P] (each bucket laid out like a
    GradBucket: xyz | f_dc | f_rest | opacity | scaling | rotation).  Handed to rasterize_gaussians_raw_batch / render_batch
    (PipelineParams.grad_bucket) in place of a GradBucket, it receives every view's OWN attribute gradients
    (gsr_backward_raw_batch_views): the views share one launch chain and one backward composite, and view v's bucket holds
    bit for bit what the single-view backward of that view writes.  For callers that need per-view gradients -- independent
    views, per-view clipping or statistics; the attack's batch wants their sum and takes a GradBucket."""

    def __init__(self, B: int, P: int, device):
        self.B, self.P = int(B), int(P)
        self.flat = torch.empty(self.B * 59 * self.P, dtype=torch.float32, device=device)
        self.used = 0              # views written by the last backward

    def bucket(self, v: int) -> "GradBucket":
        """View v's gradients as a GradBucket over this set's memory (no copy)."""
        b = GradBucket.__new__(GradBucket)
        b.P = self.P
        b.flat = self.flat[v * 59 * self.P:(v + 1) * 59 * self.P]
        b.fresh, b.used, b.chunks, b.on_chunk = False, v < self.used, 1, None
        return b


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Same path with the activation getters fused into the kernels (gsr_forward_raw / gsr_backward_raw): takes the
    RAW parameter tensors of a reference-style GaussianModel."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, objects_dc, opacity, scaling, rotation, raster_settings,
                keep=True, bucket=None, cache_slot=None, color_only=False, norms=None):
        lib = _load()
        ctx.bucket = bucket
        ctx.norms = norms
        ctx.entry = None
        if not xyz.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization: tensors must live on a HIP device (got "
                               f"{xyz.device}); there is no CPU path")
        device = xyz.device
        P = int(xyz.shape[0])
        if tuple(features_dc.shape) != (P, 1, 3) or tuple(features_rest.shape) != (P, 15, 3):
            raise ValueError("fused path needs _features_dc [P,1,3] and _features_rest [P,15,3] (SH degree 3 storage)")

        def prep(t):
            return None if t is None or t.numel() == 0 else _f32c(t.detach(), device)
        x, dc, rest, obj = prep(xyz), prep(features_dc), prep(features_rest), prep(objects_dc)
        op, sc, ro = prep(opacity), prep(scaling), prep(rotation)
        if obj is not None and obj.numel() != P * NUM_OBJECTS:
            raise ValueError(f"objects_dc must hold P*{NUM_OBJECTS} values, got {tuple(obj.shape)}")
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        pack = _SettingsPack(raster_settings, device)
        color = torch.empty(3, H, W, dtype=torch.float32, device=device)
        # all-zero object features (the attack's combined scenes, reference scene/gaussian_model.py:528) composite to exactly
        # zero: as on the classic surface, the forward then runs without the 16 object channels and hands back a broadcast
        # zero; the context keeps the features for a backward that is given dL/dobjects
        obj_zero = obj is not None and _objects_all_zero(obj, objects_dc)
        with_obj = obj is not None and not obj_zero
        if obj_zero:
            pack.c.flags |= FLAG_OBJECTS_FOR_BACKWARD_ONLY
        objects = (torch.empty(NUM_OBJECTS, H, W, dtype=torch.float32, device=device) if with_obj
                   else _zero_scalar(device).expand(NUM_OBJECTS, H, W))
        radii = None
        handle = ctypes.c_void_p(None)
        nren = ctypes.c_int64(0)
        entry = None
        if cache_slot is not None and P > 0:
            cache, key = cache_slot
            geo = (xyz, opacity, scaling, rotation, objects_dc if obj is not None else None)
            dense = all(a is None or a.data_ptr() == b.data_ptr() for a, b in zip(geo, (x, op, sc, ro, obj)))
            sig = _cache_sig(geo, raster_settings, extra=(obj_zero,)) if dense else None
            # identity (weak references) of the geometry AND the camera tensors: a new camera object that happens to get
            # the id, the addresses and the versions of a dead one is never mistaken for it
            geo = geo + (raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.campos)
            if dense:
                entry, may_store = cache._lookup(key, sig, geo)
                if not may_store:
                    sig = None                      # the key's context is waiting for a backward: plain forward, not stored
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            if entry is not None:
                # the geometry of this key is what its kept context was built from: colour kernel + compositor only
                _entry_enter(entry, device)
                radii = entry.pack.radii.detach()      # geometry is unchanged: so are the radii of the key's first render
                rc = lib.gsr_ctx_rerender(entry.holder.handle, _ptr(dc), _ptr(rest), None, None, _ptr(pack.bg), _ptr(color),
                                          _ptr(objects) if with_obj else None, 1 if color_only else 0, stream)
                entry.gen += 1
                nren.value = entry.nren
                if rc == GSR_ERR_OVERFLOW:
                    # the kept context's forward (GSR_FLAG_ASYNC_COUNT) turned out to have overflowed its guessed pair
                    # capacity: it can never be re-rendered.  Drop the entry and take the full forward in this very call
                    # (which counts synchronously after an overflow) instead of poisoning the key.
                    cache.entries.pop(key, None)
                    cache.dropped_overflow += 1
                    entry = None
            if entry is None:
                want_ctx = keep or (cache_slot is not None and P > 0 and sig is not None)
                radii = torch.empty(P, dtype=torch.int32, device=device)
                rc = lib.gsr_forward_raw(ctypes.byref(pack.c), P, _ptr(x), _ptr(dc), _ptr(rest), _ptr(obj), _ptr(op), _ptr(sc),
                                         _ptr(ro), _ptr(color), _ptr(objects) if with_obj else None, _ptr(radii),
                                         ctypes.byref(handle) if want_ctx else None,
                                         ctypes.byref(nren), stream)
        if rc != 0:
            msg = _err(lib)
            if raster_settings.debug:
                torch.save({"xyz": x, "features_dc": dc, "features_rest": rest, "objects_dc": obj, "opacity": op,
                            "scaling": sc, "rotation": ro, "settings": raster_settings._asdict()}, "snapshot_fw.dump")
                msg += " (raw parameters saved to snapshot_fw.dump)"
            raise Exception(msg) if rc == 1 else (PairCapacityExceeded if rc == GSR_ERR_OVERFLOW else RuntimeError)(msg)
        if entry is not None:
            ctx.holder = entry.holder
        else:
            ctx.holder = _CtxHolder(lib, handle) if handle.value else None
            if cache_slot is not None and P > 0 and sig is not None and ctx.holder is not None:
                pack.radii = radii         # the kept context does not recompute them: later renders of the key return these
                entry = _CacheEntry(ctx.holder, sig, tuple(weakref.ref(t) if t is not None else (lambda: None) for t in geo),
                                    pack, nren.value)
                cache._store(key, entry)
        if entry is not None:
            _entry_leave(entry, device)
            ctx.entry, ctx.entry_gen = entry, entry.gen
            if keep:
                ctx.token = _RenderToken()
                entry.token = weakref.ref(ctx.token)
        if not keep:
            ctx.holder = None
        ctx.pack = pack
        ctx._nren = nren.value
        ctx.shapes = (xyz.shape, means2D.shape, features_dc.shape, features_rest.shape,
                      None if objects_dc is None else objects_dc.shape, opacity.shape, scaling.shape, rotation.shape)
        ctx.kept = (x, dc, rest, obj, op, sc, ro)
        ctx.versions = _versions(ctx.kept)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return color, radii, objects

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_objects):
        lib = ctx.holder.lib
        _check_versions(ctx.kept, ctx.versions)
        x, dc, rest, obj, op, sc, ro = ctx.kept
        device = x.device
        if ctx.entry is not None:
            if ctx.entry.gen != ctx.entry_gen:
                raise RuntimeError("diff_gaussian_rasterization: this render's kept context (RenderCache) was rendered again "
                                   "before its backward ran; call backward() first, or render without the cache")
            _entry_enter(ctx.entry, device)
        P = int(x.shape[0])
        H, W = ctx.pack.c.image_height, ctx.pack.c.image_width
        if grad_color is None:
            grad_color = torch.zeros(3, H, W, dtype=torch.float32, device=device)
        gcol = _f32c(grad_color, device)
        gobj = None if (grad_objects is None or obj is None) else _f32c(grad_objects, device)
        need = ctx.needs_input_grad

        def out(cond, *shape):
            return torch.empty(*shape, dtype=torch.float32, device=device) if cond else None
        want_sh = need[2] or need[3]
        all59 = need[0] and want_sh and need[5] and need[6] and need[7]
        bucket = ctx.bucket if all59 else None
        if bucket is not None:
            # the caller's bucket receives the 59 attribute gradients directly (overwritten by the first backward after
            # reset(), added to by the others); autograd gets no gradient for those inputs
            if bucket.P != P or bucket.flat.device != device:
                raise ValueError("grad bucket does not match the model (P or device)")
            d_x, d_dc, d_rest, d_op, d_sc, d_ro = bucket.slices()
        elif all59:
            # All 59 attack-relevant floats per Gaussian are wanted (the normal case): carve them out of ONE flat
            # buffer, in the order xyz | f_dc | f_rest | opacity | scaling | rotation.  autograd hands these views to
            # .grad as they are, so a data-parallel caller can sum the whole gradient with a single collective
            # (gsplat_attack.dist.allreduce_attribute_grads) instead of one per tensor.
            flat = torch.empty(59 * P, dtype=torch.float32, device=device)
            cuts = [0, 3 * P, 6 * P, 51 * P, 52 * P, 55 * P, 59 * P]
            d_x, d_dc, d_rest, d_op, d_sc, d_ro = (flat[cuts[i]:cuts[i + 1]] for i in range(6))
        else:
            d_x = out(need[0], P, 3)
            d_dc = out(want_sh, P, 1, 3)
            d_rest = out(want_sh, P, 15, 3)
            d_op = out(need[5], P)
            d_sc = out(need[6], P, 3)
            d_ro = out(need[7], P, 4)
        d_m2 = out(need[1], P, 3)
        d_obj = out(need[4] and obj is not None, P, NUM_OBJECTS)
        norms = getattr(ctx, "norms", None)
        if norms is not None:
            norms.writes += 1
            overwrites = (bucket is None or bucket.fresh) and not (bucket is not None and bucket.chunks > 1)
            if norms.writes == 1 and overwrites and P > 0:
                if lib.gsr_ctx_request_sumsq(ctx.holder.handle, ctypes.c_void_p(norms.sumsq.data_ptr())) != 0:
                    raise RuntimeError(_err(lib))
                norms.names = tuple(n for n, w in zip(GradNorms.NAMES, (need[0], want_sh, want_sh, need[5], need[6], need[7])) if w)
            elif not overwrites:
                norms.invalidate()
        if P > 0:
            with torch.cuda.device(device):
                stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                if bucket is not None and bucket.chunks > 1:
                    hook, bucket.on_chunk = bucket.on_chunk, None
                    errs = []

                    def _done(_user, chunk, g0, g1):
                        try:
                            if hook is not None:
                                hook(int(chunk), int(g0), int(g1))
                        except BaseException as e:       # an exception must not unwind through the C frames
                            errs.append(e)
                    cb = _CHUNK_FN(_done)
                    rc = lib.gsr_backward_raw_chunked(ctx.holder.handle, _ptr(gcol), _ptr(gobj), _ptr(d_x), _ptr(d_m2),
                                                      _ptr(d_dc), _ptr(d_rest), _ptr(d_obj), _ptr(d_op), _ptr(d_sc), _ptr(d_ro),
                                                      0 if bucket.fresh else 1, int(bucket.chunks), cb, None, stream)
                    bucket.chunks = 1
                    if errs:
                        raise errs[0]
                elif bucket is not None:
                    rc = lib.gsr_backward_raw_into(ctx.holder.handle, _ptr(gcol), _ptr(gobj), _ptr(d_x), _ptr(d_m2), _ptr(d_dc),
                                                   _ptr(d_rest), _ptr(d_obj), _ptr(d_op), _ptr(d_sc), _ptr(d_ro),
                                                   0 if bucket.fresh else 1, stream)
                else:
                    rc = lib.gsr_backward_raw(ctx.holder.handle, _ptr(gcol), _ptr(gobj), _ptr(d_x), _ptr(d_m2), _ptr(d_dc),
                                              _ptr(d_rest), _ptr(d_obj), _ptr(d_op), _ptr(d_sc), _ptr(d_ro), stream)
            if rc != 0:
                raise (PairCapacityExceeded if rc == 5 else RuntimeError)(_err(lib))
        if ctx.entry is not None:
            _entry_leave(ctx.entry, device)
            ctx.token.done = True
        if bucket is not None:
            bucket.fresh, bucket.used = False, True
            s = ctx.shapes
            return (None, None if d_m2 is None else d_m2.reshape(s[1]), None, None,
                    None if d_obj is None else d_obj.reshape(s[4]), None, None, None, None, None, None, None, None, None)
        s = ctx.shapes

        def shaped(t, shape, wanted=True):
            return None if (t is None or not wanted) else t.reshape(shape)
        return (shaped(d_x, s[0]), shaped(d_m2, s[1]), shaped(d_dc, s[2], need[2]), shaped(d_rest, s[3], need[3]),
                shaped(d_obj, s[4]), shaped(d_op, s[5]), shaped(d_sc, s[6]), shaped(d_ro, s[7]), None, None, None, None, None, None)


MAX_BATCH = 16                # views per gsr_forward_raw_batch call (csrc/gsr_kernels.hip.h)


class _BatchPack:
    """What a kept BATCH context's cache entry holds on to: the views' settings packs (their device tensors are what the
    context's pointers point into), the radii of the key's first render, and -- once a render came with other background
    tensors than the forward's -- the [B,3] tensor of background values the context reads instead."""
    __slots__ = ("packs", "radii", "bg_src", "bgt", "b_sig", "last_inputs")

    def __init__(self, packs, radii):
        self.packs, self.radii = packs, radii
        self.bg_src = tuple(pk.bg.data_ptr() for pk in packs)
        self.bgt = None
        self.b_sig = self.last_inputs = None      # (pair batches: the second model's coefficient versions, the inputs in flight)


class _RasterizeGaussiansRawBatch(torch.autograd.Function):
    """B views of one set of RAW parameters through ONE launch chain (gsr_forward_raw_batch / gsr_backward_raw_batch_into):
    what the reference's batch loop does with B render() calls and one accumulated .grad (attack.py:476-494).
    -> (color[B,3,H,W], radii[B,P]); the attribute gradients are the SUM over the views of dL/dC[v] pulled back."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, settings_list, keep=True,
                bucket=None, norms=None, cache_slot=None, color_only=False):
        lib = _load()
        ctx.bucket = bucket
        ctx.norms = norms
        ctx.entry = None
        if not xyz.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization: tensors must live on a HIP device (got "
                               f"{xyz.device}); there is no CPU path")
        device = xyz.device
        P = int(xyz.shape[0])
        B = len(settings_list)
        if not 1 <= B <= MAX_BATCH:
            raise ValueError(f"a batch holds 1..{MAX_BATCH} views, got {B}")
        if tuple(features_dc.shape) != (P, 1, 3) or tuple(features_rest.shape) != (P, 15, 3):
            raise ValueError("fused path needs _features_dc [P,1,3] and _features_rest [P,15,3] (SH degree 3 storage)")

        def prep(t):
            return None if t is None or t.numel() == 0 else _f32c(t.detach(), device)
        x, dc, rest = prep(xyz), prep(features_dc), prep(features_rest)
        op, sc, ro = prep(opacity), prep(scaling), prep(rotation)
        H, W = int(settings_list[0].image_height), int(settings_list[0].image_width)
        packs = [_SettingsPack(rs, device) for rs in settings_list]
        carr = (_CSettings * B)()
        for v, pk in enumerate(packs):
            carr[v] = pk.c
        color = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
        radii = torch.empty(B, P, dtype=torch.int32, device=device)
        handle = ctypes.c_void_p(None)
        nren = ctypes.c_int64(0)
        # a kept batch context (RenderCache): the key's geometry tensors and ALL its views' camera tensors unchanged since the
        # key's last render -> the batch's colour kernel + one compositor launch over the kept lists (gsr_ctx_rerender)
        entry, sig, refs = None, None, None
        if cache_slot is not None and P > 0:
            cache, key = cache_slot
            geo = (xyz, opacity, scaling, rotation)
            if all(a.data_ptr() == b.data_ptr() for a, b in zip(geo, (x, op, sc, ro))):
                extra = [B]
                for rs in settings_list[1:]:
                    extra += [float(rs.tanfovx), float(rs.tanfovy)]
                    for t in (rs.viewmatrix, rs.projmatrix, rs.campos):
                        extra += [t.data_ptr(), t._version]
                sig = _cache_sig(geo, settings_list[0], extra=extra)
                refs = geo + tuple(t for rs in settings_list for t in (rs.viewmatrix, rs.projmatrix, rs.campos))
                entry, may_store = cache._lookup(key, sig, refs)
                if not may_store:
                    sig = None                      # the key's context is waiting for a backward: plain forward, not stored
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            if entry is not None:
                _entry_enter(entry, device)
                bp = entry.pack
                radii = bp.radii.detach()              # geometry is unchanged: so are the radii of the key's first render
                ptrs = tuple(pk.bg.data_ptr() for pk in packs)
                bgt = None
                if ptrs != bp.bg_src or bp.bgt is not None:
                    # other background tensors than the context's forward saw: their values as one [B,3] tensor
                    bgt = torch.stack([pk.bg[:3] for pk in packs]).contiguous()
                    bp.bgt, bp.bg_src = bgt, ptrs       # (bp.packs stays: the context's camera pointers point into them)
                rc = lib.gsr_ctx_rerender(entry.holder.handle, _ptr(dc), _ptr(rest), None, None, _ptr(bgt), _ptr(color), None,
                                          1 if color_only else 0, stream)
                entry.gen += 1
                nren.value = entry.nren
                if rc == GSR_ERR_OVERFLOW:             # (see _RasterizeGaussiansRaw: the entry can never be re-rendered)
                    cache.entries.pop(key, None)
                    cache.dropped_overflow += 1
                    entry = None
            if entry is None:
                want_ctx = keep or sig is not None
                rc = lib.gsr_forward_raw_batch(carr, B, P, _ptr(x), _ptr(dc), _ptr(rest), _ptr(op), _ptr(sc), _ptr(ro), _ptr(color),
                                               _ptr(radii), ctypes.byref(handle) if want_ctx else None, ctypes.byref(nren), stream)
        if rc != 0:
            raise Exception(_err(lib)) if rc == 1 else (PairCapacityExceeded if rc == GSR_ERR_OVERFLOW else RuntimeError)(_err(lib))
        if entry is not None:
            ctx.holder = entry.holder
        else:
            ctx.holder = _CtxHolder(lib, handle) if handle.value else None
            if sig is not None and ctx.holder is not None:
                bp = _BatchPack(packs, radii)
                entry = _CacheEntry(ctx.holder, sig, tuple(weakref.ref(t) for t in refs), bp, nren.value)
                cache_slot[0]._store(cache_slot[1], entry)
        if entry is not None:
            _entry_leave(entry, device)
            ctx.entry, ctx.entry_gen = entry, entry.gen
            if keep:
                ctx.token = _RenderToken()
                entry.token = weakref.ref(ctx.token)
        if not keep:
            ctx.holder = None
        ctx.packs = packs
        ctx.pack = packs[0]
        ctx.B = B
        ctx._nren = nren.value
        ctx.shapes = (xyz.shape, None if means2D is None else means2D.shape, features_dc.shape, features_rest.shape,
                      opacity.shape, scaling.shape, rotation.shape)
        ctx.kept = (x, dc, rest, op, sc, ro)
        ctx.versions = _versions(ctx.kept)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, grad_radii):
        _check_versions(ctx.kept, ctx.versions)
        x, dc, rest, op, sc, ro = ctx.kept
        if x is None:                                  # an empty scene: empty gradients
            need = ctx.needs_input_grad
            s = ctx.shapes
            dev0 = ctx.packs[0].device
            z = lambda i, shp: torch.zeros(shp, dtype=torch.float32, device=dev0) if (need[i] and shp is not None) else None
            return (z(0, s[0]), z(1, s[1]), z(2, s[2]), z(3, s[3]), z(4, s[4]), z(5, s[5]), z(6, s[6]), None, None, None, None,
                    None, None)
        lib = ctx.holder.lib
        device = x.device
        if ctx.entry is not None:
            if ctx.entry.gen != ctx.entry_gen:
                raise RuntimeError("diff_gaussian_rasterization: this batch's kept context (RenderCache) was rendered again "
                                   "before its backward ran; call backward() first, or render without the cache")
            _entry_enter(ctx.entry, device)
        P, B = int(x.shape[0]), ctx.B
        H, W = ctx.pack.c.image_height, ctx.pack.c.image_width
        if grad_color is None:
            grad_color = torch.zeros(B, 3, H, W, dtype=torch.float32, device=device)
        gcol = _f32c(grad_color, device)
        need = ctx.needs_input_grad

        def out(cond, *shape):
            return torch.empty(*shape, dtype=torch.float32, device=device) if cond else None
        want_sh = need[2] or need[3]
        all59 = need[0] and want_sh and need[4] and need[5] and need[6]
        bucket = ctx.bucket if all59 else None
        bset = bucket if isinstance(bucket, GradBucketSet) else None
        if bset is not None:
            # per-view gradients: the pointers are view 0's bucket, view v's lie 59 P floats further
            if bset.P != P or bset.B < B or bset.flat.device != device:
                raise ValueError("grad bucket set does not match the batch (views, P or device)")
            d_x, d_dc, d_rest, d_op, d_sc, d_ro = bset.bucket(0).slices()
        elif bucket is not None:
            if bucket.P != P or bucket.flat.device != device:
                raise ValueError("grad bucket does not match the model (P or device)")
            d_x, d_dc, d_rest, d_op, d_sc, d_ro = bucket.slices()
        elif all59:
            flat = torch.empty(59 * P, dtype=torch.float32, device=device)
            cuts = [0, 3 * P, 6 * P, 51 * P, 52 * P, 55 * P, 59 * P]
            d_x, d_dc, d_rest, d_op, d_sc, d_ro = (flat[cuts[i]:cuts[i + 1]] for i in range(6))
        else:
            d_x = out(need[0], P, 3)
            d_dc = out(want_sh, P, 1, 3)
            d_rest = out(want_sh, P, 15, 3)
            d_op = out(need[4], P)
            d_sc = out(need[5], P, 3)
            d_ro = out(need[6], P, 4)
        d_m2 = out(need[1], B, P, 3)
        norms = getattr(ctx, "norms", None)
        if norms is not None:
            # the batch's ONE backward writes the summed gradient: its sums of squares are the L2 steps' norms (GradNorms)
            norms.writes += 1
            overwrites = bset is None and (bucket is None or bucket.fresh) and not (bucket is not None and bucket.chunks > 1)
            if norms.writes == 1 and overwrites and P > 0:
                if lib.gsr_ctx_request_sumsq(ctx.holder.handle, ctypes.c_void_p(norms.sumsq.data_ptr())) != 0:
                    norms.invalidate()              # (a build or a mode that cannot serve it: the step sums the gradient itself)
                else:
                    norms.names = tuple(n for n, w in zip(GradNorms.NAMES, (need[0], want_sh, want_sh, need[4], need[5], need[6])) if w)
            elif not overwrites:
                norms.invalidate()
        if P > 0:
            with torch.cuda.device(device):
                stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                acc = 1 if (bucket is not None and bset is None and not bucket.fresh) else 0
                if bset is not None:
                    rc = lib.gsr_backward_raw_batch_views(ctx.holder.handle, _ptr(gcol), _ptr(d_x), _ptr(d_m2), _ptr(d_dc),
                                                          _ptr(d_rest), _ptr(d_op), _ptr(d_sc), _ptr(d_ro), 59 * P, stream)
                    bset.used = B
                elif bucket is not None and bucket.chunks > 1:
                    # the per-Gaussian stage in ranges, each announced as soon as its launch is enqueued (all-reduce overlap)
                    hook, bucket.on_chunk = bucket.on_chunk, None
                    errs = []

                    def _done(_user, chunk, g0, g1):
                        try:
                            if hook is not None:
                                hook(int(chunk), int(g0), int(g1))
                        except BaseException as e:       # an exception must not unwind through the C frames
                            errs.append(e)
                    cb = _CHUNK_FN(_done)
                    rc = lib.gsr_backward_raw_chunked(ctx.holder.handle, _ptr(gcol), None, _ptr(d_x), _ptr(d_m2), _ptr(d_dc),
                                                      _ptr(d_rest), None, _ptr(d_op), _ptr(d_sc), _ptr(d_ro), acc,
                                                      int(bucket.chunks), cb, None, stream)
                    bucket.chunks = 1
                    if errs:
                        raise errs[0]
                else:
                    rc = lib.gsr_backward_raw_batch_into(ctx.holder.handle, _ptr(gcol), _ptr(d_x), _ptr(d_m2), _ptr(d_dc),
                                                         _ptr(d_rest), _ptr(d_op), _ptr(d_sc), _ptr(d_ro), acc, stream)
            if rc != 0:
                raise (PairCapacityExceeded if rc == 5 else RuntimeError)(_err(lib))
        else:
            for t in (d_x, d_dc, d_rest, d_op, d_sc, d_ro, d_m2):
                if t is not None and bucket is None:
                    t.zero_()
        if ctx.entry is not None:
            _entry_leave(ctx.entry, device)
            ctx.token.done = True
        s = ctx.shapes
        if bset is not None:
            return (None, None if d_m2 is None else d_m2.reshape(s[1]), None, None, None, None, None, None, None, None, None,
                    None, None)
        if bucket is not None:
            bucket.fresh, bucket.used = False, True
            return (None, None if d_m2 is None else d_m2.reshape(s[1]), None, None, None, None, None, None, None, None, None,
                    None, None)

        def shaped(t, shape, wanted=True):
            return None if (t is None or not wanted) else t.reshape(shape)
        return (shaped(d_x, s[0]), shaped(d_m2, s[1]), shaped(d_dc, s[2], need[2]), shaped(d_rest, s[3], need[3]),
                shaped(d_op, s[4]), shaped(d_sc, s[5]), shaped(d_ro, s[6]), None, None, None, None, None, None)


def rasterize_gaussians_raw_batch(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, settings_list,
                                  grad_bucket: Optional["GradBucket"] = None, grad_norms: Optional["GradNorms"] = None,
                                  cache: Optional["RenderCache"] = None, cache_key=None):
    """(color[B,3,H,W], radii[B,P]) of B views (a list of GaussianRasterizationSettings that agree in image size, scale
    modifier and SH degree) of one set of RAW parameters, through one launch chain: every image and radius is bit for bit
    what rasterize_gaussians_raw gives for that view alone, and the backward leaves the SUM over the views of the
    attribute gradients, written once.  means2D: a [B,P,3] tensor whose .grad receives the per-view screen-space gradient
    (viewspace_points.grad of the reference, one slice per view), or None.  No object channels."""
    keep = _wants_backward(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation)
    # `cache` (a RenderCache) + `cache_key` (one per tuple of cameras): a batch whose geometry inputs and cameras are unchanged
    # since the key's last render re-uses that render's binning for all B views (gsr_ctx_rerender on the batch context)
    slot = (cache, cache_key) if cache is not None else None
    color_only = not (torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                       for t in (xyz, means2D, opacity, scaling, rotation)))
    return _RasterizeGaussiansRawBatch.apply(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation,
                                             list(settings_list), keep, grad_bucket, grad_norms, slot, color_only)


def _wants_backward(*tensors) -> bool:
    """A forward keeps backward state only if autograd can reach it: grad mode on and an input that requires grad.
    Decided here, before Function.apply (inside forward() grad mode is off and needs_input_grad ignores no_grad())."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, objects_dc, opacity, scaling, rotation,
                            raster_settings, grad_bucket: Optional["GradBucket"] = None, cache: Optional["RenderCache"] = None,
                            cache_key=None, grad_norms: Optional["GradNorms"] = None):
    """(color[3,H,W], radii[P], objects[16,H,W]) from the RAW parameters of a reference-style GaussianModel
    (_xyz, _features_dc, _features_rest, _objects_dc or None, _opacity, _scaling, _rotation): equal to the
    getters (scene/gaussian_model.py:97-124) followed by GaussianRasterizer.forward, in one fused pass."""
    keep = _wants_backward(xyz, means2D, features_dc, features_rest, objects_dc, opacity, scaling, rotation)
    # `cache` (a RenderCache) + `cache_key` (one per camera): a render whose geometry inputs are unchanged since the key's
    # last render re-uses that render's binning (gsr_ctx_rerender)
    slot = (cache, cache_key) if cache is not None else None
    color_only = not (torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                       for t in (xyz, means2D, opacity, scaling, rotation)))
    return _RasterizeGaussiansRaw.apply(xyz, means2D, features_dc, features_rest, objects_dc, opacity, scaling, rotation,
                                        raster_settings, keep, grad_bucket, slot, color_only, grad_norms)


@torch.no_grad()
def rasterize_gaussians_raw2(params_a, params_b, raster_settings, objects: bool = True,
                             cache: Optional["RenderCache"] = None, cache_key=None):
    """Forward-only render of two reference-style parameter sets as ONE scene -- `params_a` followed by `params_b`,
    each (xyz, features_dc, features_rest, objects_dc or None, opacity, scaling, rotation), RAW tensors -- without
    concatenating them (gsr_forward_raw2; reference attack.py:513-530 deep-copies the model and concatenates all seven
    tensors for this).  -> (color[3,H,W], radii[Pa+Pb], objects[16,H,W]); bitwise equal to rasterize_gaussians_raw on
    the concatenated tensors.  Not differentiable (the reference never calls backward on this render)."""
    lib = _load()
    xa = params_a[0]
    if not xa.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization: tensors must live on a HIP device; there is no CPU path")
    device = xa.device

    def prep(t):
        return None if t is None or t.numel() == 0 else _f32c(t.detach(), device)
    a = [prep(t) for t in params_a]
    b = [prep(t) for t in params_b]
    Pa, Pb = int(params_a[0].shape[0]), int(params_b[0].shape[0])
    for P, (x, dc, rest, obj, op, sc, ro) in ((Pa, a), (Pb, b)):
        if P and (tuple(dc.shape) != (P, 1, 3) or tuple(rest.shape) != (P, 15, 3)):
            raise ValueError("fused path needs _features_dc [P,1,3] and _features_rest [P,15,3] (SH degree 3 storage)")
    with_obj = objects and (Pa == 0 or a[3] is not None) and (Pb == 0 or b[3] is not None)
    H, W = int(raster_settings.image_height), int(raster_settings.image_width)
    pack = _SettingsPack(raster_settings, device)
    color = torch.empty(3, H, W, dtype=torch.float32, device=device)
    objs = (torch.empty(NUM_OBJECTS, H, W, dtype=torch.float32, device=device) if with_obj
            else _zero_scalar(device).expand(NUM_OBJECTS, H, W))
    nren = ctypes.c_int64(0)
    if not with_obj:
        a[3] = b[3] = None
    # cache: with both models' geometry unchanged since the key's last render, only the colour kernel and the compositor
    # run (gsr_ctx_rerender on a context kept by gsr_forward_raw2_keep) -- the success re-render of a colour attack
    entry = sig = None
    geo = ()
    if cache is not None and Pa > 0 and Pb > 0:
        geo = tuple(params_a[i] for i in (0, 4, 5, 6)) + tuple(params_b[i] for i in (0, 4, 5, 6)) + \
            ((params_a[3], params_b[3]) if with_obj else (None, None))
        used = (a[0], a[4], a[5], a[6], b[0], b[4], b[5], b[6], a[3], b[3])
        if all(t is None or t.data_ptr() == u.data_ptr() for t, u in zip(geo, used)):
            sig = _cache_sig(geo, raster_settings, extra=("pair", with_obj))
            geo = geo + (raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.campos)
            entry, _ = cache._lookup(cache_key, sig, geo)        # forward-only contexts are never busy
    with torch.cuda.device(device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        if entry is not None:
            _entry_enter(entry, device)
            radii = entry.pack.radii.detach()
            # the second model is the frozen background (reference attack.py:513-530): while its coefficient tensors are the
            # ones of the entry's last render, unmodified, the colour kernel runs over the first model's Gaussians only
            b_sig = tuple((t.data_ptr(), t._version) for t in (params_b[1], params_b[2]))
            b_same = (entry.pack.b_sig == b_sig and params_b[1].data_ptr() == b[1].data_ptr()
                      and params_b[2].data_ptr() == b[2].data_ptr())
            if b_same:
                rc = lib.gsr_ctx_rerender(entry.holder.handle, _ptr(a[1]), _ptr(a[2]), None, None, _ptr(pack.bg),
                                          _ptr(color), _ptr(objs) if with_obj else None, 1 | 2, stream)
            else:
                rc = lib.gsr_ctx_rerender(entry.holder.handle, _ptr(a[1]), _ptr(a[2]), _ptr(b[1]), _ptr(b[2]), _ptr(pack.bg),
                                          _ptr(color), _ptr(objs) if with_obj else None, 1, stream)
            entry.pack.b_sig = b_sig if (params_b[1].data_ptr() == b[1].data_ptr()
                                         and params_b[2].data_ptr() == b[2].data_ptr()) else None
            entry.gen += 1
            # the context reads the coefficient tensors of THIS call on the stream: they stay referenced until the next render
            entry.pack.last_inputs = (a, b, pack)
            if rc == GSR_ERR_OVERFLOW:                 # see _RasterizeGaussiansRaw.forward: drop the entry, full forward now
                cache.entries.pop(cache_key, None)
                cache.dropped_overflow += 1
                entry = None
        if entry is None:
            radii = torch.empty(Pa + Pb, dtype=torch.int32, device=device)
            handle = ctypes.c_void_p(None)
            rc = lib.gsr_forward_raw2_keep(ctypes.byref(pack.c), Pa, *[_ptr(t) for t in a], Pb, *[_ptr(t) for t in b],
                                           _ptr(color), _ptr(objs) if with_obj else None, _ptr(radii),
                                           ctypes.byref(handle) if sig is not None else None, ctypes.byref(nren), stream)
            if rc == 0 and handle.value:
                pack.radii = radii
                pack.last_inputs = (a, b)
                pack.b_sig = (tuple((t.data_ptr(), t._version) for t in (params_b[1], params_b[2]))
                              if (params_b[1].data_ptr() == b[1].data_ptr() and params_b[2].data_ptr() == b[2].data_ptr())
                              else None)
                entry = _CacheEntry(_CtxHolder(lib, handle), sig,
                                    tuple(weakref.ref(t) if t is not None else (lambda: None) for t in geo), pack, nren.value)
                cache._store(cache_key, entry)
        if entry is not None and rc == 0:
            _entry_leave(entry, device)
    if rc != 0:
        raise (Exception if rc == 1 else PairCapacityExceeded if rc == GSR_ERR_OVERFLOW else RuntimeError)(_err(lib))
    return color, radii, objs


@torch.no_grad()
def rasterize_gaussians_raw2_batch(params_a, params_b, settings_list, cache: Optional["RenderCache"] = None, cache_key=None):
    """rasterize_gaussians_raw2 for a BATCH of views through one launch chain (gsr_forward_raw2_batch): the success renders
    of a batch of cameras (reference attack.py:513-530 renders the combined scene once per camera).  `params_*`: (xyz,
    features_dc, features_rest, opacity, scaling, rotation), RAW tensors, both non-empty.  -> (color[B,3,H,W],
    radii[B,Pa+Pb]); every image bit for bit rasterize_gaussians_raw2's for that view.  No object channels, not
    differentiable.  With a cache the batch's context is kept: while both models' geometry and the cameras are unchanged a
    render is the batch's colour kernel -- over the first model's Gaussians only while the second model's coefficient
    tensors are untouched -- and one compositor launch."""
    lib = _load()
    xa = params_a[0]
    if not xa.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization: tensors must live on a HIP device; there is no CPU path")
    device = xa.device
    B = len(settings_list)
    if not 1 <= B <= MAX_BATCH:
        raise ValueError(f"a batch holds 1..{MAX_BATCH} views, got {B}")
    a = [_f32c(t.detach(), device) for t in params_a]
    b = [_f32c(t.detach(), device) for t in params_b]
    Pa, Pb = int(a[0].shape[0]), int(b[0].shape[0])
    if Pa == 0 or Pb == 0:
        raise ValueError("rasterize_gaussians_raw2_batch needs two non-empty models")
    for P, (x, dc, rest, op, sc, ro) in ((Pa, a), (Pb, b)):
        if tuple(dc.shape) != (P, 1, 3) or tuple(rest.shape) != (P, 15, 3):
            raise ValueError("fused path needs _features_dc [P,1,3] and _features_rest [P,15,3] (SH degree 3 storage)")
    H, W = int(settings_list[0].image_height), int(settings_list[0].image_width)
    packs = [_SettingsPack(rs, device) for rs in settings_list]
    carr = (_CSettings * B)()
    for v, pk in enumerate(packs):
        carr[v] = pk.c
    color = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
    nren = ctypes.c_int64(0)
    entry = sig = refs = None
    if cache is not None:
        geo = tuple(params_a[i] for i in (0, 3, 4, 5)) + tuple(params_b[i] for i in (0, 3, 4, 5))
        used = (a[0], a[3], a[4], a[5], b[0], b[3], b[4], b[5])
        if all(t.data_ptr() == u.data_ptr() for t, u in zip(geo, used)):
            extra = ["pair-batch", B]
            for rs in settings_list[1:]:
                extra += [float(rs.tanfovx), float(rs.tanfovy)]
                for t in (rs.viewmatrix, rs.projmatrix, rs.campos):
                    extra += [t.data_ptr(), t._version]
            sig = _cache_sig(geo, settings_list[0], extra=extra)
            refs = geo + tuple(t for rs in settings_list for t in (rs.viewmatrix, rs.projmatrix, rs.campos))
            entry, _ = cache._lookup(cache_key, sig, refs)       # forward-only contexts are never busy
    b_dense = params_b[1].data_ptr() == b[1].data_ptr() and params_b[2].data_ptr() == b[2].data_ptr()
    b_sig = tuple((t.data_ptr(), t._version) for t in (params_b[1], params_b[2])) if b_dense else None
    with torch.cuda.device(device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        if entry is not None:
            _entry_enter(entry, device)
            bp = entry.pack
            radii = bp.radii.detach()
            ptrs = tuple(pk.bg.data_ptr() for pk in packs)
            bgt = None
            if ptrs != bp.bg_src or bp.bgt is not None:
                bgt = torch.stack([pk.bg[:3] for pk in packs]).contiguous()
                bp.bgt, bp.bg_src = bgt, ptrs
            # the second model is the frozen background: while its coefficient tensors are those of the entry's last render,
            # unmodified, the colour kernel covers the first model's Gaussians only
            if b_sig is not None and bp.b_sig == b_sig:
                rc = lib.gsr_ctx_rerender(entry.holder.handle, _ptr(a[1]), _ptr(a[2]), None, None, _ptr(bgt), _ptr(color), None,
                                          1 | 2, stream)
            else:
                rc = lib.gsr_ctx_rerender(entry.holder.handle, _ptr(a[1]), _ptr(a[2]), _ptr(b[1]), _ptr(b[2]), _ptr(bgt),
                                          _ptr(color), None, 1, stream)
            bp.b_sig = b_sig
            bp.last_inputs = (a, b, packs)             # read on the stream: referenced until the next render
            entry.gen += 1
            if rc == GSR_ERR_OVERFLOW:
                cache.entries.pop(cache_key, None)
                cache.dropped_overflow += 1
                entry = None
        if entry is None:
            radii = torch.empty(B, Pa + Pb, dtype=torch.int32, device=device)
            handle = ctypes.c_void_p(None)
            rc = lib.gsr_forward_raw2_batch(carr, B, Pa, *[_ptr(t) for t in a], Pb, *[_ptr(t) for t in b], _ptr(color),
                                            _ptr(radii), ctypes.byref(handle) if sig is not None else None, ctypes.byref(nren),
                                            stream)
            if rc == 0 and handle.value:
                bp = _BatchPack(packs, radii)
                bp.b_sig, bp.last_inputs = b_sig, (a, b, packs)
                entry = _CacheEntry(_CtxHolder(lib, handle), sig, tuple(weakref.ref(t) for t in refs), bp, nren.value)
                cache._store(cache_key, entry)
        if entry is not None and rc == 0:
            _entry_leave(entry, device)
    if rc != 0:
        raise (Exception if rc == 1 else PairCapacityExceeded if rc == GSR_ERR_OVERFLOW else RuntimeError)(_err(lib))
    return color, radii


def rasterize_gaussians(means3D, means2D, sh, sh_objs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    keep = _wants_backward(means3D, means2D, sh, sh_objs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    return _RasterizeGaussians.apply(means3D, means2D, sh, sh_objs, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, keep)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test (view-space z > 0.2) -> bool[P]."""
        lib = _load()
        if not positions.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization: positions must live on a HIP device; there is no CPU path")
        with torch.no_grad():
            device = positions.device
            pos = _f32c(positions, device)
            pack = _SettingsPack(self.raster_settings, device)
            present = torch.empty(pos.shape[0], dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                rc = lib.gsr_mark_visible(ctypes.byref(pack.c), int(pos.shape[0]), _ptr(pos), _ptr(present), stream)
            if rc != 0:
                raise RuntimeError(_err(lib))
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, sh_objs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, sh_objs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)


# ---- introspection used by the benchmark / tests -----------------------------------------------------------
def last_num_rendered(output: torch.Tensor) -> int:
    """Number of (tile, Gaussian) pairs of the forward that produced `output` (its grad_fn's context).  An
    asynchronous-count forward (FLAG_ASYNC_COUNT) learns it here, from the context (waits for the early device copy)."""
    fn = output.grad_fn
    if fn is None:
        return -1
    n = int(getattr(fn, "_nren", -1))
    if n < 0 and getattr(fn, "holder", None) is not None:
        n = fn.holder.info(0)
    return n


_EXPORTS = {"ranges": (0, torch.int32), "pair_rank": (1, torch.int32), "n_contrib": (2, torch.int32),
            "final_T": (3, torch.float32), "order": (4, torch.int32), "off": (5, torch.int32),
            "R": (7, torch.float32), "G": (7, torch.float32), "dv": (8, torch.int32), "offg": (9, torch.int32)}


def export_state(output: torch.Tensor, name: str) -> torch.Tensor:
    """Copy one internal array of the forward that produced `output` (tests / diagnostics only)."""
    fn = output.grad_fn
    holder = fn.holder
    what, dt = _EXPORTS[name]
    P = int(fn.kept[0].shape[0])
    H, W = fn.pack.c.image_height, fn.pack.c.image_width
    T = ((W + 15) // 16) * ((H + 15) // 16)
    nb = max(holder.info(4), 1)                      # a batch context: B views as one virtual scene of B * Ppad Gaussians
    if nb > 1:
        P, T, H = nb * holder.info(5), nb * T, nb * H
    n = {"ranges": 2 * T, "pair_rank": holder.info(0), "n_contrib": H * W, "final_T": H * W, "order": P,
         "off": P + 1, "R": 12 * P, "G": 12 * P, "dv": 16, "offg": P + 1}[name]
    live = None
    if name in ("order", "off") and P > 0:
        # only the Gaussians that emit pairs are ranked: dv[1] = V of them (order[:V], off[:V + 1] are meaningful)
        live = int(export_state(output, "dv")[1].item()) + (1 if name == "off" else 0)
    dst = torch.empty(max(n, 1), dtype=dt, device=output.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(output.device).cuda_stream)
    if holder.lib.gsr_ctx_export(holder.handle, what, dst.data_ptr(), dst.numel() * 4, stream) != 0:
        raise RuntimeError(_err(holder.lib))
    return dst[:n if live is None else live]


def profile(enable, stages=None) -> None:
    """Start (and reset) / stop the library's per-stage HIP-event timing.  `stages`: iterable of names from
    GSR_STAGES to time only those (each timed stage costs two event records per call)."""
    if not enable:
        mask = 0
    elif stages is None:
        mask = (1 << len(GSR_STAGES)) - 1
    else:
        mask = sum(1 << GSR_STAGES.index(s) for s in stages)
    _load().gsr_profile(mask)


def profile_read() -> dict:
    lib = _load()
    ms = (ctypes.c_float * len(GSR_STAGES))()
    calls = (ctypes.c_int64 * len(GSR_STAGES))()
    if lib.gsr_profile_read(ms, calls) != 0:
        raise RuntimeError(_err(lib))
    return {n: (float(ms[i]), int(calls[i])) for i, n in enumerate(GSR_STAGES)}


def pool_bytes() -> int:
    out = ctypes.c_int64(0)
    _load().gsr_query(1, ctypes.byref(out))
    return out.value


def trim_pool() -> None:
    _load().gsr_trim_pool()


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_gaussians_raw",
           "rasterize_gaussians_raw2", "rasterize_gaussians_raw_batch", "MAX_BATCH", "GradBucketSet", "PairCapacityExceeded", "GradBucket", "GradNorms",
           "NUM_OBJECTS",
           "library_path", "profile", "profile_read", "pool_bytes", "trim_pool", "last_num_rendered", "export_state"]


def _auto_patch_reference():
    """GSR_PATCH_REFERENCE=1 in the environment: the reference's ``gaussian_renderer.render`` becomes this package's fused
    ``gsplat_attack.renderer.render`` WITHOUT a line of the reference being edited and without a call to
    ``gsplat_attack.patch_reference()``.  The module object of ``gaussian_renderer`` is given a subclass of
    ``types.ModuleType`` whose attribute lookup answers ``render`` with the fused function once the reference has defined
    its own -- so ``from gaussian_renderer import render`` (reference attack.py:20) and ``gaussian_renderer.render`` both
    bind the fused one.  Two ways in, whichever comes first: the reference imports this package from INSIDE
    ``gaussian_renderer/__init__.py`` (line 14), while that module is still being executed -- it is then in sys.modules
    already and is re-classed on the spot; or this package was imported earlier (``scene/gaussian_model.py:17`` imports
    ``simple_knn._C``, which loads the same library) -- then a meta-path finder re-classes ``gaussian_renderer`` right after
    its own import has run.  Same signature, same returned dict (tests/test_golden_glue.py pins the fused function against
    the EXECUTED reference one); the raw parameters go straight into the kernels: no activated copies, no 192 MB ``cat`` per
    view, no PyTorch backward of the getters.  GSR_PATCH_MODULE names another module than ``gaussian_renderer``.  Off by
    default: an installed package does not rebind a caller's functions unasked."""
    if os.environ.get("GSR_PATCH_REFERENCE", "0") in ("", "0"):
        return None
    import importlib.abc
    import importlib.util
    import sys
    import types
    want = os.environ.get("GSR_PATCH_MODULE", "gaussian_renderer")

    class _FusedRenderModule(types.ModuleType):
        def __getattribute__(self, name):
            if name == "render" and "render" in types.ModuleType.__getattribute__(self, "__dict__"):
                from gsplat_attack.renderer import render as fused
                return fused
            return types.ModuleType.__getattribute__(self, name)

    def reclass(module):
        try:
            if isinstance(module, types.ModuleType) and type(module) is not _FusedRenderModule:
                module.__class__ = _FusedRenderModule
        except TypeError:
            pass

    if want in sys.modules:                    # being imported right now (or imported already): re-class it on the spot
        reclass(sys.modules[want])
        return "reclassed"

    class _Finder(importlib.abc.MetaPathFinder):
        busy = False

        def find_spec(self, fullname, path, target=None):
            if fullname != want or _Finder.busy:
                return None
            _Finder.busy = True
            try:
                spec = importlib.util.find_spec(fullname)
            finally:
                _Finder.busy = False
            if spec is None or spec.loader is None:
                return None
            inner = spec.loader

            class _Loader(importlib.abc.Loader):
                def create_module(self, spec_):
                    return inner.create_module(spec_)

                def exec_module(self, module):
                    inner.exec_module(module)
                    reclass(module)
            spec.loader = _Loader()
            return spec

    sys.meta_path.insert(0, _Finder())
    return "finder"


_auto_patch_reference()
