"""The boundary glue against fixtures made by EXECUTING the reference (tests/golden/make_golden_glue.py ->
tests/golden/ref_glue.npz): the reference's own render() (gaussian_renderer/__init__.py:18-103) ran on its own
GaussianModel and Camera with a recording stand-in for the rasteriser; its own Camera (scene/cameras.py:17-105) ran
constructor / transform / yaw.  Here gsplat_attack.renderer.render's classic branch is handed a recording
GaussianRasterizer and must pass the SAME 12 settings and the SAME 9 keyword tensors; gsplat_attack.cameras.Camera
must reproduce the matrices after every step.  CPU only (a CPU model takes the classic activated-tensor surface)."""
import os
import types

import numpy as np
import pytest
import torch

from gsplat_attack import renderer as RN
from gsplat_attack.cameras import Camera
from gsplat_attack.gaussian_model import GaussianModel

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def glue():
    return np.load(os.path.join(HERE, "golden", "ref_glue.npz"))


def t(a):
    return torch.from_numpy(np.asarray(a))


def our_model(glue, deg):
    m = GaussianModel.from_tensors(t(glue["model/_xyz"]), t(glue["model/_features_dc"]), t(glue["model/_features_rest"]),
                                   t(glue["model/_scaling"]), t(glue["model/_rotation"]), t(glue["model/_opacity"]),
                                   t(glue["model/_objects_dc"]), sh_degree=3)
    m.active_sh_degree = deg
    return m


def our_camera(glue, **kw):
    return Camera(glue["cam_in/R"], glue["cam_in/T"], float(glue["cam_in/FoVx"]), float(glue["cam_in/FoVy"]),
                  int(glue["cam_in/width"]), int(glue["cam_in/height"]), **kw)


class Recorder:
    """Stands where diff_gaussian_rasterization.GaussianRasterizer stands in renderer.py; returns what the fixture's
    recorder returned (same seeds)."""
    calls = []

    def __init__(self, raster_settings):
        self.st = raster_settings

    def __call__(self, *args, **kw):
        assert not args
        P = kw["means3D"].shape[0]
        H, W = self.st.image_height, self.st.image_width
        g = torch.Generator().manual_seed(5)
        image = torch.randn(3, H, W, generator=g) * 2.0
        radii = (torch.arange(P, dtype=torch.int32) % 3) * 4
        objects = torch.randn(16, H, W, generator=g)
        Recorder.calls.append((self.st, kw, (image, radii, objects)))
        return image, radii, objects


def _render(glue, tag, monkeypatch):
    conv, cov, mod, deg, override, debug = (float(v) for v in glue[f"render/{tag}/in"])
    pc = our_model(glue, int(deg))
    cam = our_camera(glue)
    pipe = types.SimpleNamespace(convert_SHs_python=bool(conv), compute_cov3D_python=bool(cov), debug=bool(debug))
    monkeypatch.setattr(RN, "GaussianRasterizer", Recorder)
    Recorder.calls.clear()
    ov = t(glue["override_color"]) if override else None
    res = RN.render(cam, pc, pipe, t(glue[f"render/{tag}/bg"]), mod, override_color=ov)
    assert len(Recorder.calls) == 1
    return res, Recorder.calls[0], pc


def _check_settings(glue, tag, st):
    names = [str(n) for n in glue[f"render/{tag}/settings_names"]]
    assert list(st._fields) == names                       # the 12 fields, call-site order
    for n in names:
        want = glue[f"render/{tag}/settings/{n}"]
        got = getattr(st, n)
        if torch.is_tensor(got):
            assert got.dtype == torch.float32
            np.testing.assert_allclose(got.numpy(), want, rtol=0, atol=1e-6, err_msg=n)
        else:
            tkey = f"render/{tag}/settings_type/{n}"
            if tkey in glue.files:
                assert type(got).__name__ == str(glue[tkey]), (n, type(got).__name__)
            assert got == want.item(), (n, got, want)


@pytest.mark.parametrize("tag", ["plain", "bg4_mod_deg2", "cov_python"])
def test_render_hands_the_rasteriser_what_the_reference_hands_it(glue, tag, monkeypatch):
    assert str(glue[f"render/{tag}/raised"]) == ""
    res, (st, kw, outs), pc = _render(glue, tag, monkeypatch)
    _check_settings(glue, tag, st)
    want_names = [str(n) for n in glue[f"render/{tag}/kw_names"]]
    assert sorted(kw) == sorted(want_names)                 # the 9 keywords of gaussian_renderer/__init__.py:86-95
    for n in want_names:
        none = bool(glue[f"render/{tag}/kw_none/{n}"])
        assert (kw[n] is None) == none, n
        if none:
            continue
        want = glue[f"render/{tag}/kw/{n}"]
        assert tuple(kw[n].shape) == want.shape, n
        assert str(kw[n].dtype) == str(glue[f"render/{tag}/kw_dtype/{n}"]), n
        np.testing.assert_allclose(kw[n].detach().numpy(), want, rtol=1e-6, atol=1e-7, err_msg=n)
        assert bool(kw[n].requires_grad) == bool(glue[f"render/{tag}/kw_requires_grad/{n}"]), n
    # the returned dict (:99-103): same keys in the same order, the rasteriser's image handed through unclamped
    assert list(res.keys()) == [str(k) for k in glue[f"render/{tag}/result_keys"]]
    assert bool(glue[f"render/{tag}/result_render_is_raster_output"]) and res["render"] is outs[0]
    assert float(res["render"].max()) > 1.0 and float(res["render"].min()) < 0.0
    assert bool(glue[f"render/{tag}/result_render_object_is_raster_output"]) and res["render_object"] is outs[2]
    np.testing.assert_array_equal(res["radii"].numpy(), glue[f"render/{tag}/result_radii"])
    np.testing.assert_array_equal(res["visibility_filter"].numpy(), glue[f"render/{tag}/result_visibility_filter"])
    # viewspace_points: the tensor passed as means2D, zeros, requires grad, and its .grad is filled by backward
    assert bool(glue[f"render/{tag}/viewspace_is_means2D"]) and res["viewspace_points"] is kw["means2D"]
    np.testing.assert_array_equal(res["viewspace_points"].detach().numpy(), glue[f"render/{tag}/viewspace"])
    assert res["viewspace_points"].requires_grad == bool(glue[f"render/{tag}/viewspace_requires_grad"])
    (res["viewspace_points"] * 2.0).sum().backward()
    np.testing.assert_array_equal(res["viewspace_points"].grad.numpy(), glue[f"render/{tag}/viewspace_grad_after_backward"])
    # gradients reach the raw parameters through the getters (exp / sigmoid / normalize / cat), as in the reference
    loss = sum((kw[n] ** 2).sum() for n in want_names if kw[n] is not None and n != "means2D")
    loss.backward()
    assert pc._xyz.grad is not None and pc._opacity.grad is not None and pc._features_rest.grad is not None


@pytest.mark.parametrize("tag", ["sh_python", "sh_python_deg1_cov", "override"])
def test_branches_on_which_the_reference_raises(glue, tag, monkeypatch):
    """convert_SHs_python / override_color: the reference's render() leaves `sh_objs` unbound and raises before it reaches
    the rasteriser (recorded: UnboundLocalError).  What it had computed by then -- the settings, colors_precomp,
    cov3D_precomp / scales / rotations -- is in the fixture; the counterpart passes exactly those, plus the object
    features its only working branch passes."""
    assert str(glue[f"render/{tag}/raised"]) == "UnboundLocalError"
    res, (st, kw, outs), pc = _render(glue, tag, monkeypatch)
    _check_settings(glue, tag, st)
    ref2ours = dict(colors_precomp="colors_precomp", cov3D_precomp="cov3D_precomp", scales="scales", rotations="rotations",
                    shs="shs", means3D="means3D", opacity="opacities")
    for rn, on in ref2ours.items():
        none = bool(glue[f"render/{tag}/local_none/{rn}"])
        assert (kw[on] is None) == none, rn
        if not none:
            np.testing.assert_allclose(kw[on].detach().numpy(), glue[f"render/{tag}/local/{rn}"], rtol=1e-5, atol=2e-6,
                                       err_msg=rn)
    assert kw["sh_objs"] is not None and tuple(kw["sh_objs"].shape) == (40, 1, 16)


def _cam_equal(cam, glue, step, center_atol=1e-5):
    np.testing.assert_allclose(cam.world_view_transform.numpy(), glue[f"cam/{step}/world_view_transform"], atol=1e-6)
    np.testing.assert_allclose(cam.projection_matrix.numpy(), glue[f"cam/{step}/projection_matrix"], atol=1e-6)
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), glue[f"cam/{step}/full_proj_transform"], atol=1e-5)
    np.testing.assert_allclose(cam.camera_center.numpy(), glue[f"cam/{step}/camera_center"], atol=center_atol)
    np.testing.assert_allclose(np.asarray(cam.R), glue[f"cam/{step}/R"], atol=1e-12)
    np.testing.assert_allclose(np.asarray(cam.T), glue[f"cam/{step}/T"], atol=0)
    assert cam.world_view_transform.dtype == torch.float32 and cam.full_proj_transform.dtype == torch.float32


def test_camera_reproduces_the_reference_class_step_by_step(glue):
    """scene/cameras.py:17-105 executed: constructor, transform(T), yaw(7), yaw(-14).  camera_center is NOT refreshed by
    transform / yaw in the reference (update_world_view_projection_transforms, :60-69): the fixture shows the stale
    centre and the counterpart keeps it."""
    cam = our_camera(glue)
    assert (cam.image_width, cam.image_height) == tuple(int(v) for v in glue["cam_image_size"])
    assert cam.znear == 0.01 and cam.zfar == 100.0
    _cam_equal(cam, glue, "init")
    cam.transform(glue["cam_in/T1"])
    _cam_equal(cam, glue, "transform")
    np.testing.assert_array_equal(glue["cam/transform/camera_center"], glue["cam/init/camera_center"])   # the quirk itself
    cam.yaw(7)
    _cam_equal(cam, glue, "yaw7")
    cam.yaw(-14)
    _cam_equal(cam, glue, "yaw-14")
    assert not np.allclose(glue["cam/yaw-14/world_view_transform"], glue["cam/init/world_view_transform"])
    cam2 = our_camera(glue, trans=glue["cam_in/trans"], scale=float(glue["cam_in/scale"]))
    _cam_equal(cam2, glue, "trans_scale")
