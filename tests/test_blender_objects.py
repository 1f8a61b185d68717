"""SURVEY.md section 8f rank 4 remainder: the Blender-transforms camera reader, the render-resolution rule and the
consumers of the 16-channel object map, pinned to outputs of the reference's own functions
(tests/golden/ref_f4.npz, made by tests/golden/make_golden_f4.py)."""
import os

import numpy as np
import pytest
import torch

from gsplat_attack import blender, objects

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = os.path.join(HERE, "golden", "blender_sample")


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(HERE, "golden", "ref_f4.npz"))


def test_cameras_from_transforms_match_the_reference_reader(ref):
    cams = blender.cameras_from_transforms(SAMPLE, "transforms_train.json", ".png", resolution=1)
    assert len(cams) == ref["tf_R_b"].shape[0] == 3
    for i, c in enumerate(cams):
        assert np.allclose(c.R, ref["tf_R_b"][i], atol=1e-12) and np.allclose(c.T, ref["tf_T_b"][i], atol=1e-12)
        assert c.FoVx == pytest.approx(ref["tf_fov_b"][i, 0], abs=1e-12)
        assert c.FoVy == pytest.approx(ref["tf_fov_b"][i, 1], abs=1e-12)
        assert (c.image_width, c.image_height) == tuple(ref["tf_size_b"][i])
        assert c.image_name == str(ref["tf_names"][i])
    # the camera looks at the origin from 4 units away: the origin projects to the image centre, in front of it
    c = cams[1]
    o = torch.tensor([[0.0, 0.0, 0.0, 1.0]]) @ c.full_proj_transform
    assert abs(float(o[0, 0] / o[0, 3])) < 1e-5 and abs(float(o[0, 1] / o[0, 3])) < 1e-5
    assert float((torch.tensor([[0.0, 0.0, 0.0, 1.0]]) @ c.world_view_transform)[0, 2]) == pytest.approx(4.0, abs=1e-5)


def test_train_and_test_frames_are_merged_without_eval(tmp_path):
    import json
    import shutil
    shutil.copytree(SAMPLE, tmp_path / "s")
    tf = json.load(open(tmp_path / "s" / "transforms_train.json"))
    json.dump({"camera_angle_x": tf["camera_angle_x"], "frames": tf["frames"][:1]}, open(tmp_path / "s" / "transforms_test.json", "w"))
    train, test = blender.read_nerf_synthetic(str(tmp_path / "s"))
    assert len(train) == 4 and test == [] and [c.uid for c in train] == [0, 1, 2, 3]
    train, test = blender.read_nerf_synthetic(str(tmp_path / "s"), eval=True)
    assert len(train) == 3 and len(test) == 1


def test_ground_truth_compositing_matches_the_reference_bytes(ref):
    from PIL import Image
    for tag, white in (("b", False), ("w", True)):
        for i in range(3):
            rgba = np.array(Image.open(os.path.join(SAMPLE, "train", f"r_{i}.png")).convert("RGBA"))
            assert np.array_equal(blender.blend_on_background(rgba, white), ref[f"tf_image_{tag}"][i])


def test_render_resolution_rule_matches_loadcam(ref):
    for (w, h, r, sc), want in zip(ref["res_cases"], ref["res_out"]):
        r = int(r) if float(r).is_integer() else float(r)
        assert blender.render_resolution(int(w), int(h), r, float(sc)) == tuple(int(v) for v in want), (w, h, r, sc)


def test_object_palette_and_id_map_match_the_reference(ref):
    if "id2rgb" not in ref:
        pytest.skip("fixture made without the viewer module")
    for i in range(257):
        assert np.array_equal(objects.id2rgb(i), ref["id2rgb"][i]), i
    assert np.array_equal(objects.visualize_obj(ref["vis_ids"]), ref["vis_rgb"])
    with pytest.raises(ValueError):
        objects.id2rgb(300)


def test_object_map_consumers():
    g = torch.Generator().manual_seed(0)
    obj = torch.randn(16, 24, 40, generator=g)
    clf = objects.ObjectClassifier(num_classes=7)
    ids = objects.predict_objects(obj, clf)
    assert tuple(ids.shape) == (24, 40) and int(ids.max()) < 7
    assert torch.equal(ids, torch.argmax(clf.conv(obj[None])[0], dim=0))
    rgb = objects.feature_to_rgb(obj)
    assert rgb.shape == (24, 40, 3) and rgb.dtype == np.uint8 and rgb.max() == 255 and rgb.min() == 0
    # principal components: projecting on them reproduces the map's three largest variances
    X = obj.reshape(16, -1).T.double()
    X = X - X.mean(0, keepdim=True)
    s = torch.linalg.svdvals(X)[:3] ** 2 / X.shape[0]
    flat = rgb.reshape(-1, 3).astype(np.float64)
    v = flat.var(axis=0)
    assert v[0] > v[1] > v[2] and np.allclose(v / v[0], (s / s[0]).numpy(), rtol=0.05)
    assert objects.feature_to_rgb(torch.ones(16, 4, 4)).max() == 0
