"""Generate tests/golden/ref_glue.npz by EXECUTING the reference's boundary glue:

  * gaussian_renderer/__init__.py:18-103 render() -- called on a 40-Gaussian CPU GaussianModel (the reference's own class)
    and the reference's own Camera, with `diff_gaussian_rasterization` replaced by a RECORDER: the stand-in
    GaussianRasterizationSettings keeps the keyword arguments it is built with, the stand-in GaussianRasterizer keeps the
    keyword tensors render() passes and returns fixed fake outputs.  Captured per call: the 12 settings (names in call
    order, values), the 9 keyword arguments (None or values, shape, requires_grad, dtype), the returned dict.
    Both Python switches on and off, an override colour, a 4-element background, a scaling modifier, an active SH degree
    below the maximum.  On the branches where the reference leaves `sh_objs` unbound (its convert_SHs_python and
    override_color branches, gaussian_renderer/__init__.py:69-83) the call raises UnboundLocalError before it reaches the
    rasteriser: the exception's name is recorded together with the locals render() had computed by then
    (colors_precomp, cov3D_precomp ... read from the raising frame).
  * scene/cameras.py:17-105 Camera -- constructor (with and without trans / scale), transform(), yaw(7), yaw(-14):
    world_view_transform, projection_matrix, full_proj_transform, camera_center and R after every step.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_glue.py

The reference hard-codes the device "cuda" (`zeros_like(..., device="cuda")` at gaussian_renderer/__init__.py:26,
`.cuda()` at scene/cameras.py:54-55,62-64, `device="cuda"` in utils/general_utils.py:65,83,102); for the duration of this
script tensor.cuda() returns the tensor itself and a "cuda" device argument of torch.zeros / torch.zeros_like is dropped.
Third-party modules this image lacks (plyfile, simple_knn) are empty stand-ins for the import only.
Data only: inputs written by this script, outputs produced by the reference's code.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


CALLS = []          # one dict per GaussianRasterizer.__call__


class RecSettings:
    """Recorder for GaussianRasterizationSettings: keeps the keyword names in call order and their values."""

    def __init__(self, *args, **kw):
        assert not args, "the reference builds the settings with keywords only"
        self.names = list(kw)
        self.kw = kw


class RecRasterizer:
    def __init__(self, raster_settings):
        self.st = raster_settings

    def __call__(self, *args, **kw):
        assert not args, "the reference calls the rasteriser with keywords only"
        P = kw["means3D"].shape[0]
        H, W = self.st.kw["image_height"], self.st.kw["image_width"]
        g = torch.Generator().manual_seed(5)
        image = torch.randn(3, H, W, generator=g) * 2.0          # values outside [0, 1]: render() must not clamp
        radii = (torch.arange(P, dtype=torch.int32) % 3) * 4     # every third Gaussian "invisible"
        objects = torch.randn(16, H, W, generator=g)
        CALLS.append(dict(settings=self.st, kw=kw, out=(image, radii, objects)))
        return image, radii, objects


def _redirect_cuda():
    torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("zeros", "zeros_like", "ones", "tensor"):
        orig = getattr(torch, name)

        def wrapped(*a, _orig=orig, **k):
            if str(k.get("device", "")) .startswith("cuda"):
                k.pop("device")
            return _orig(*a, **k)
        setattr(torch, name, wrapped)


def _pack_call(out, tag, call):
    st, kw = call["settings"], call["kw"]
    out[f"{tag}/settings_names"] = np.array(st.names)
    for n, v in st.kw.items():
        if torch.is_tensor(v):
            out[f"{tag}/settings/{n}"] = v.detach().numpy()
        else:
            out[f"{tag}/settings/{n}"] = np.array(v)
            out[f"{tag}/settings_type/{n}"] = np.array(type(v).__name__)
    out[f"{tag}/kw_names"] = np.array(list(kw))
    for n, v in kw.items():
        out[f"{tag}/kw_none/{n}"] = np.array(v is None)
        if v is not None:
            out[f"{tag}/kw/{n}"] = v.detach().numpy()
            out[f"{tag}/kw_requires_grad/{n}"] = np.array(bool(v.requires_grad))
            out[f"{tag}/kw_is_leaf/{n}"] = np.array(bool(v.is_leaf))
            out[f"{tag}/kw_dtype/{n}"] = np.array(str(v.dtype))


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    _redirect_cuda()
    _stub("plyfile", PlyData=object, PlyElement=object)
    _stub("simple_knn")
    _stub("simple_knn._C", distCUDA2=None)
    _stub("diff_gaussian_rasterization", GaussianRasterizationSettings=RecSettings, GaussianRasterizer=RecRasterizer)
    import gaussian_renderer as GR
    from scene.gaussian_model import GaussianModel
    from scene.cameras import Camera

    out = {}
    # ---- the model: 40 Gaussians, raw parameters written here -----------------------------------------------------
    gen = torch.Generator().manual_seed(2024)
    P = 40
    raw = dict(_xyz=torch.randn(P, 3, generator=gen), _features_dc=torch.randn(P, 1, 3, generator=gen),
               _features_rest=torch.randn(P, 15, 3, generator=gen) * 0.2, _scaling=torch.randn(P, 3, generator=gen) - 3.0,
               _rotation=torch.randn(P, 4, generator=gen), _opacity=torch.randn(P, 1, generator=gen),
               _objects_dc=torch.randn(P, 1, 16, generator=gen))
    pc = GaussianModel(3)
    for n, v in raw.items():
        out["model/" + n] = v.numpy()
        setattr(pc, n, torch.nn.Parameter(v.clone().requires_grad_(True)))

    # ---- the camera: the reference's own class --------------------------------------------------------------------
    def cam_state(c):
        return dict(world_view_transform=c.world_view_transform.numpy().copy(),
                    projection_matrix=c.projection_matrix.numpy().copy(),
                    full_proj_transform=c.full_proj_transform.numpy().copy(),
                    camera_center=c.camera_center.numpy().copy(), R=np.asarray(c.R).copy(), T=np.asarray(c.T).copy())

    th = 0.4
    R0 = np.array([[np.cos(th), 0.0, np.sin(th)], [0.0, 1.0, 0.0], [-np.sin(th), 0.0, np.cos(th)]]) @ \
        np.array([[1.0, 0.0, 0.0], [0.0, np.cos(0.2), -np.sin(0.2)], [0.0, np.sin(0.2), np.cos(0.2)]])
    T0 = np.array([0.3, -0.2, 4.5])
    img = torch.rand(3, 30, 40, generator=gen)
    cam_in = dict(R=R0, T=T0, FoVx=0.9, FoVy=0.7, width=40, height=30)
    for k, v in cam_in.items():
        out["cam_in/" + k] = np.array(v)
    cam = Camera(colmap_id=1, R=R0.copy(), T=T0.copy(), FoVx=0.9, FoVy=0.7, image=img, gt_alpha_mask=None, image_name="v",
                 uid=0, data_device="cpu")
    out["cam_image_size"] = np.array([cam.image_width, cam.image_height])
    steps = [("init", None)]
    for k, v in cam_state(cam).items():
        out["cam/init/" + k] = v
    T1 = np.array([0.5, 0.1, 5.0])
    cam.transform(T1)
    out["cam_in/T1"] = T1
    for k, v in cam_state(cam).items():
        out["cam/transform/" + k] = v
    cam.yaw(7)
    for k, v in cam_state(cam).items():
        out["cam/yaw7/" + k] = v
    cam.yaw(-14)
    for k, v in cam_state(cam).items():
        out["cam/yaw-14/" + k] = v
    # constructor with trans / scale
    tr, sc = np.array([0.5, -0.2, 0.1]), 1.3
    cam2 = Camera(colmap_id=2, R=R0.copy(), T=T0.copy(), FoVx=0.9, FoVy=0.7, image=img, gt_alpha_mask=None, image_name="w",
                  uid=1, trans=tr, scale=sc, data_device="cpu")
    out["cam_in/trans"], out["cam_in/scale"] = tr, np.array(sc)
    for k, v in cam_state(cam2).items():
        out["cam/trans_scale/" + k] = v

    # ---- render(): a fresh camera in its constructed state ----------------------------------------------------------
    cam = Camera(colmap_id=1, R=R0.copy(), T=T0.copy(), FoVx=0.9, FoVy=0.7, image=img, gt_alpha_mask=None, image_name="v",
                 uid=0, data_device="cpu")
    cases = [
        ("plain", dict(conv=False, cov=False), dict(bg=torch.tensor([0.1, 0.2, 0.3]), mod=1.0, deg=3, override=False)),
        ("bg4_mod_deg2", dict(conv=False, cov=False), dict(bg=torch.tensor([1.0, 1.0, 1.0, 0.0]), mod=1.7, deg=2, override=False)),
        ("cov_python", dict(conv=False, cov=True), dict(bg=torch.zeros(3), mod=1.3, deg=3, override=False)),
        ("sh_python", dict(conv=True, cov=False), dict(bg=torch.zeros(3), mod=1.0, deg=3, override=False)),
        ("sh_python_deg1_cov", dict(conv=True, cov=True), dict(bg=torch.zeros(3), mod=0.8, deg=1, override=False)),
        ("override", dict(conv=False, cov=False), dict(bg=torch.zeros(3), mod=1.0, deg=3, override=True)),
    ]
    override = torch.rand(P, 3, generator=gen)
    out["override_color"] = override.numpy()
    names = []
    for tag, sw, o in cases:
        names.append(tag)
        pipe = types.SimpleNamespace(convert_SHs_python=sw["conv"], compute_cov3D_python=sw["cov"], debug=(tag == "plain"))
        pc.active_sh_degree = o["deg"]
        out[f"render/{tag}/in"] = np.array([float(sw["conv"]), float(sw["cov"]), o["mod"], o["deg"], float(o["override"]),
                                           float(pipe.debug)])
        out[f"render/{tag}/bg"] = o["bg"].numpy()
        n0 = len(CALLS)
        try:
            # (scaling_modifier passed only when it differs from the default, as attack.py's call sites do)
            args = (cam, pc, pipe, o["bg"]) + ((o["mod"],) if o["mod"] != 1.0 else ())
            res = GR.render(*args, override_color=override if o["override"] else None)
            out[f"render/{tag}/raised"] = np.array("")
        except Exception as e:                                    # the reference's own failure on this branch
            out[f"render/{tag}/raised"] = np.array(type(e).__name__)
            out[f"render/{tag}/raised_msg"] = np.array(str(e))
            tb = e.__traceback__
            while tb.tb_next is not None:
                tb = tb.tb_next
            loc = tb.tb_frame.f_locals
            assert tb.tb_frame.f_code.co_name == "render", tb.tb_frame.f_code.co_name
            for n in ("colors_precomp", "cov3D_precomp", "scales", "rotations", "shs", "means3D", "opacity"):
                v = loc.get(n)
                out[f"render/{tag}/local_none/{n}"] = np.array(v is None)
                if v is not None:
                    out[f"render/{tag}/local/{n}"] = v.detach().numpy()
            st = loc["raster_settings"]
            out[f"render/{tag}/settings_names"] = np.array(st.names)
            for n, v in st.kw.items():
                out[f"render/{tag}/settings/{n}"] = v.detach().numpy() if torch.is_tensor(v) else np.array(v)
            assert len(CALLS) == n0
            continue
        assert len(CALLS) == n0 + 1
        call = CALLS[-1]
        _pack_call(out, f"render/{tag}", call)
        image, radii, objects = call["out"]
        out[f"render/{tag}/result_keys"] = np.array(list(res))
        out[f"render/{tag}/result_render_is_raster_output"] = np.array(res["render"] is image)
        out[f"render/{tag}/result_render"] = res["render"].detach().numpy()
        out[f"render/{tag}/result_radii"] = res["radii"].numpy()
        out[f"render/{tag}/result_visibility_filter"] = res["visibility_filter"].numpy()
        out[f"render/{tag}/result_render_object_is_raster_output"] = np.array(res["render_object"] is objects)
        vp = res["viewspace_points"]
        out[f"render/{tag}/viewspace_is_means2D"] = np.array(vp is call["kw"]["means2D"])
        out[f"render/{tag}/viewspace"] = vp.detach().numpy()
        out[f"render/{tag}/viewspace_requires_grad"] = np.array(bool(vp.requires_grad))
        # the screen-space gradient reaches viewspace_points.grad (retain_grad on the non-leaf, :27-30)
        (vp * 2.0).sum().backward()
        out[f"render/{tag}/viewspace_grad_after_backward"] = vp.grad.numpy()
    out["render_cases"] = np.array(names)
    path = os.path.join(HERE, "ref_glue.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays;", {t: str(out[f'render/{t}/raised']) for t in names})


if __name__ == "__main__":
    main()
