"""The C-ABI shared library: it loads without a GPU, exports every entry point include/gsraster.h declares,
rejects bad argument combinations before touching the device, and the Python package refuses to run on CPU
(no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gsraster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import diff_gaussian_rasterization as D
    path = D.library_path()
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    return D._load()


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert {"gsr_forward", "gsr_backward", "gsr_ctx_free", "gsr_mark_visible", "gsr_last_error"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gsraster.h but not exported by libgsraster.so"


def test_settings_struct_layout_matches_header():
    import diff_gaussian_rasterization as D
    text = open(os.path.join(ROOT, "include", "gsraster.h")).read()
    body = re.search(r"typedef struct GsrSettings \{(.*?)\} GsrSettings;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [re.findall(r"(\w+)\s*;", line)[0] for line in body.split("\n") if ";" in line]
    assert fields == [f[0] for f in D._CSettings._fields_]
    assert D._CSettings.bg.offset == 16 and D._CSettings.viewmatrix.offset == 32     # natural C alignment
    assert [f for f in D.GaussianRasterizationSettings._fields] == fields[:12]            # the reference's 12 fields, same order


def test_version_and_argument_validation_without_device(lib):
    import diff_gaussian_rasterization as D
    out = ctypes.c_int64(0)
    assert lib.gsr_query(0, ctypes.byref(out)) == 0 and out.value >= 100
    # null settings
    rc = lib.gsr_forward(None, 1, 16, None, None, None, None, None, None, None, None, None, None, None, None, None, None)
    assert rc == 1 and b"null" in lib.gsr_last_error()
    # both SH and precomputed colours -> invalid (checked before any HIP call)
    st = D._CSettings(16, 16, 0.5, 0.5, 1, 1.0, 1, 1, 3, 1, 0, 0, 0)
    one = ctypes.c_void_p(8)
    rc = lib.gsr_forward(ctypes.byref(st), 4, 16, one, one, None, one, one, one, one, None, one, None, one, None, None, None)
    assert rc == 1 and b"exactly one" in lib.gsr_last_error()
    rc = lib.gsr_forward(ctypes.byref(st), 4, 16, one, one, None, None, one, one, None, None, one, None, one, None, None, None)
    assert rc == 1 and b"exactly one" in lib.gsr_last_error()
    rc = lib.gsr_forward(ctypes.byref(st), 4, 9, one, one, None, None, one, one, one, None, one, None, one, None, None, None)
    assert rc == 1 and b"sh_degree" in lib.gsr_last_error()
    assert lib.gsr_backward(None, None, None, None, None, None, None, None, None, None, None, None, None) == 4
    # re-render of a kept context: a null context is a state error, and the version says the entry point exists
    assert out.value >= 400
    assert lib.gsr_ctx_rerender(None, None, None, None, None, None, one, None, 0, None) == 4
    assert b"null context" in lib.gsr_last_error()
    # the multi-tensor step: the tensor count and the column counts are checked before any HIP call
    assert lib.gsr_pgd_step_multi(0, None, None, None, None, None, None, None, 1, None, None) == 0
    assert lib.gsr_pgd_step_multi(9, None, None, None, None, None, None, None, 1, None, None) == 1
    assert b"tensors" in lib.gsr_last_error()
    ptr, rows, cols, f = (ctypes.c_void_p * 1)(8), (ctypes.c_int64 * 1)(4), (ctypes.c_int32 * 1)(49), (ctypes.c_float * 1)(0.5)
    assert lib.gsr_pgd_step_multi(1, ptr, ptr, ptr, rows, cols, f, f, 0, None, None) == 1
    assert b"columns" in lib.gsr_last_error()


def test_package_has_no_cpu_path():
    import diff_gaussian_rasterization as D
    st = D.GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3,
                                         torch.zeros(3), False, False)
    r = D.GaussianRasterizer(raster_settings=st)
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 16, 3), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(z(2, 3))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 16, 3))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import diff_gaussian_rasterization as D
    monkeypatch.setattr(D, "_lib", None)
    monkeypatch.setattr(D, "library_path", lambda: str(tmp_path / "libgsraster.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        D._load()


def test_simple_knn_shim_imports_and_measures():
    from simple_knn._C import distCUDA2
    pts = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [5, 5, 5]])
    d = distCUDA2(pts)
    assert d.shape == (5,)
    assert abs(d[0].item() - (1 + 4 + 9) / 3) < 1e-5


def test_patch_reference_rebinds_render_everywhere():
    """gsplat_attack.patch_reference swaps the reference's render for the fused one in the defining module and in
    modules that imported the name (stand-in modules: the reference is not importable at test time)."""
    import sys
    import types
    import gsplat_attack
    from gsplat_attack.renderer import render as fused
    gr = types.ModuleType("gaussian_renderer_standin")

    def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
        raise AssertionError("the original must not be called any more")
    gr.render = render
    user = types.ModuleType("attack_standin")
    user.render = render                       # `from gaussian_renderer import render`
    other = types.ModuleType("unrelated_standin")
    other.render = lambda: None
    sys.modules.update({gr.__name__: gr, user.__name__: user, other.__name__: other})
    try:
        assert gsplat_attack.patch_reference(gr.__name__) == 2
        assert gr.render is fused and user.render is fused and other.render is not fused
    finally:
        for m in (gr, user, other):
            sys.modules.pop(m.__name__, None)


def test_render_result_makes_the_visibility_filter_on_demand():
    """render()'s dict (reference gaussian_renderer/__init__.py:99-103) has the `visibility_filter` key from the start;
    the tensor (radii > 0) is made the first time it is read."""
    import torch
    from gsplat_attack.renderer import RenderResult
    radii = torch.tensor([0, 3, 0, 7], dtype=torch.int32)
    r = RenderResult(render=torch.zeros(3, 2, 2), viewspace_points=None, visibility_filter=None, radii=radii,
                     render_object=None)
    assert "visibility_filter" in r and len(r) == 5 and list(r.keys())[2] == "visibility_filter"
    assert dict.__getitem__(r, "visibility_filter") is None            # not made yet
    assert r["visibility_filter"].tolist() == [False, True, False, True]
    assert r.get("visibility_filter") is r["visibility_filter"]
    r2 = RenderResult(render=None, viewspace_points=None, visibility_filter=None, radii=radii, render_object=None)
    assert dict(r2.items())["visibility_filter"].tolist() == [False, True, False, True]
    r3 = RenderResult(render=None, viewspace_points=None, visibility_filter=None, radii=radii, render_object=None)
    assert r3.pop("visibility_filter").tolist() == [False, True, False, True] and "visibility_filter" not in r3


def test_render_result_copies_and_splats_see_the_resolved_filter():
    """dict(result), {**result}, f(**result) and other.update(result) go through CPython's PyDict_Merge, whose
    fast path for dict subclasses reads the raw stored None; RenderResult has its own __iter__ / keys so that they
    resolve the lazy entry."""
    import torch
    from gsplat_attack.renderer import RenderResult
    radii = torch.tensor([0, 3, 0, 7], dtype=torch.int32)

    def fresh():
        return RenderResult(render=None, viewspace_points=None, visibility_filter=None, radii=radii, render_object=None)
    want = [False, True, False, True]
    assert dict(fresh())["visibility_filter"].tolist() == want
    assert {**fresh()}["visibility_filter"].tolist() == want
    assert (lambda **kw: kw["visibility_filter"])(**fresh()).tolist() == want
    other = {}
    other.update(fresh())
    assert other["visibility_filter"].tolist() == want
    assert fresh().copy()["visibility_filter"].tolist() == want
    assert [k for k in fresh()] == ["render", "viewspace_points", "visibility_filter", "radii", "render_object"]
