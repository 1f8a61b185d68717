"""Data formats either side of the raster path, through the rasteriser: a model that went through the PLY wire format
renders the same bits; cameras read from a Blender transforms file and from a COLMAP model drive render()."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _render(cam, model, objects=True):
    from gsplat_attack.renderer import PipelineParams, render
    with torch.no_grad():
        return render(cam, model, PipelineParams(skip_objects=not objects), torch.tensor([0.1, 0.2, 0.3], device="cuda"))


def test_ply_round_trip_renders_the_same_bits(tmp_path):
    from gsplat_attack.gaussian_model import GaussianModel
    from gsplat_attack.scenes import make_scene
    model, cams, _ = make_scene("nyc-1M", device="cuda", P=20000, width=320, height=192, n_views=1)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    model.save_ply(path)
    back = GaussianModel.load_ply(path, device="cuda")
    a, b = _render(cams[0], model), _render(cams[0], back)
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"])
    assert torch.equal(a["render_object"], b["render_object"]) and float(a["render_object"].abs().max()) > 0


def test_blender_and_colmap_cameras_drive_the_renderer():
    from gsplat_attack import blender, objects
    from gsplat_attack.colmap import cameras_from_colmap
    from gsplat_attack.scenes import make_scene
    model, _, _ = make_scene("hydrant-1k", device="cuda", n_views=1)        # a blob around the origin
    cams = blender.cameras_from_transforms(os.path.join(HERE, "golden", "blender_sample"), "transforms_train.json",
                                           ".png", resolution=1, device="cuda")
    for cam in cams:                                                        # they look at the origin from 4 units away
        out = _render(cam, model)
        assert tuple(out["render"].shape) == (3, 30, 40)
        assert int((out["radii"] > 0).sum()) > 900
        centre = out["render"][:, 12:18, 16:24].mean(dim=(1, 2))
        assert float((centre - torch.tensor([0.1, 0.2, 0.3], device="cuda")).abs().max()) > 0.02   # not just background
        ids = objects.predict_objects(out["render_object"], objects.ObjectClassifier(5).to("cuda"))
        assert tuple(ids.shape) == (30, 40)
    ccams = cameras_from_colmap(os.path.join(HERE, "golden", "colmap_sample"), device="cuda")
    out = _render(ccams[0], model)
    assert tuple(out["render"].shape) == (3, ccams[0].image_height, ccams[0].image_width)
    assert torch.isfinite(out["render"]).all()
