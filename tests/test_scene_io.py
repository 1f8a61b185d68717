"""Scene directory -> model + cameras (gsplat_attack/scene_io.py) on a synthetic COLMAP scene written by the test."""
import os
import random

import numpy as np
import pytest
import torch

from gsplat_attack.colmap import (ColmapCamera, ColmapImage, cameras_from_colmap, write_model_binary)
from gsplat_attack.gaussian_model import GaussianModel
from gsplat_attack.scene_io import Scene, cameras_extent, search_for_max_iteration

SAMPLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colmap_sample")


def _write_scene(root, n_cams=17, n_pts=200):
    rng = np.random.default_rng(7)
    cams = {1: ColmapCamera(1, "PINHOLE", 160, 120, np.array([150.0, 155.0, 80.0, 60.0]))}
    ims = {}
    for i in range(n_cams):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        ims[i + 1] = ColmapImage(i + 1, q, rng.normal(size=3) * 2.0, 1, f"sub/img_{(i * 7) % n_cams:03d}.png")
    pts = (rng.normal(size=(n_pts, 3)) * 0.4, rng.integers(0, 256, size=(n_pts, 3), dtype=np.uint8), rng.uniform(size=n_pts))
    write_model_binary(cams, ims, os.path.join(root, "sparse", "0"), pts)
    return pts


def test_cameras_extent_matches_reference_getNerfppNorm(golden):
    cams = cameras_from_colmap(SAMPLE)
    assert abs(cameras_extent(cams) - float(golden["colmap_nerfnorm_radius"])) < 1e-5      # the reference goes through float32


def test_scene_from_sparse_points_split_and_shuffle(tmp_path):
    src = str(tmp_path / "src")
    pts = _write_scene(src)
    sc = Scene(src, shuffle=False, eval=True)
    names = [c.image_name for c in cameras_from_colmap(src)]
    assert names == sorted(names)
    assert [c.image_name for c in sc.test_cameras] == names[0::8]
    assert [c.image_name for c in sc.train_cameras] == [n for i, n in enumerate(names) if i % 8 != 0]
    # the shuffle is the reference's: random.seed(42) then random.shuffle on the train list, then on the test list
    sc2 = Scene(src, shuffle=True, eval=True, cam_indices=[3, 0, 5])
    train = [n for i, n in enumerate(names) if i % 8 != 0]
    test = names[0::8]
    random.seed(42); random.shuffle(train); random.shuffle(test)
    assert [c.image_name for c in sc2.train_cameras] == [train[3], train[0], train[5]]
    assert [c.image_name for c in sc2.test_cameras] == test
    # no trained model: initialised from the sparse points
    assert sc.gaussians._xyz.shape == (200, 3)
    assert np.allclose(sc.gaussians._xyz.detach().numpy(), pts[0].astype(np.float32))
    assert sc.cameras_extent > 0


def test_scene_loads_the_highest_iteration_and_saves(tmp_path):
    src, mdl = str(tmp_path / "src"), str(tmp_path / "model")
    _write_scene(src)
    g = torch.Generator().manual_seed(0)
    for it, P in ((7000, 10), (30000, 25)):
        m = GaussianModel.from_tensors(torch.randn(P, 3, generator=g), torch.randn(P, 1, 3, generator=g),
                                       torch.randn(P, 15, 3, generator=g), torch.randn(P, 3, generator=g),
                                       torch.randn(P, 4, generator=g), torch.randn(P, 1, generator=g),
                                       torch.randn(P, 1, 16, generator=g))
        path = os.path.join(mdl, "point_cloud", f"iteration_{it}", "point_cloud.ply")
        os.makedirs(os.path.dirname(path))
        m.save_ply(path)
    assert search_for_max_iteration(os.path.join(mdl, "point_cloud")) == 30000
    sc = Scene(src, mdl, load_iteration=-1)
    assert sc.loaded_iter == 30000 and sc.gaussians._xyz.shape[0] == 25
    assert Scene(src, mdl, load_iteration=7000).gaussians._xyz.shape[0] == 10
    out = sc.save(30001)
    assert os.path.exists(out) and GaussianModel.load_ply(out)._xyz.shape[0] == 25
    with pytest.raises(ValueError):
        Scene(str(tmp_path / "nowhere"))


def test_importing_the_package_asks_for_eight_hardware_queues(monkeypatch):
    """View pipelining deals a batch over four HIP streams; the runtime's default of four hardware queues makes them
    share queues (gsplat_attack/__init__.py).  A value chosen by the user is left alone."""
    import importlib
    import os
    import gsplat_attack
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    importlib.reload(gsplat_attack)
    assert os.environ.get("GPU_MAX_HW_QUEUES") == "8"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    importlib.reload(gsplat_attack)
    assert os.environ.get("GPU_MAX_HW_QUEUES") == "2"


def test_bbox_from_render_is_pils_luma_threshold_and_getbbox():
    """attack.bbox_from_render == the reference's benign-pass rule (attack.py:438-449): clamp, x255, truncate to bytes,
    PIL's convert('L'), p > 20, getbbox() -- on random images with a bright patch, on a dim one (None), at the borders."""
    import numpy as np
    import torch
    from PIL import Image
    from gsplat_attack.attack import bbox_from_render

    def pil_rule(img):
        np_img = (torch.clamp(img, min=0, max=1.0) * 255).byte().permute(1, 2, 0).contiguous().numpy()
        bw = Image.fromarray(np_img).convert('L').point(lambda p: p > 20 and 255)
        return bw.getbbox()
    g = torch.Generator().manual_seed(0)
    for case in range(40):
        H, W = int(torch.randint(5, 70, (1,), generator=g)), int(torch.randint(5, 90, (1,), generator=g))
        img = torch.rand(3, H, W, generator=g) * 0.16 - 0.04        # luma around the threshold (20/255 = 0.078), some < 0
        if case % 4 != 3:
            y0, x0 = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
            img[:, y0:y0 + 1 + case % 5, x0:x0 + 1 + case % 7] += 0.6 + 0.8 * torch.rand(1, generator=g)   # may exceed 1
        if case == 5:
            img = torch.zeros(3, H, W)
        assert bbox_from_render(img) == pil_rule(img), case
    assert bbox_from_render(torch.zeros(3, 8, 8)) is None
    one = torch.zeros(3, 8, 9)
    one[:, 7, 8] = 1.0
    assert bbox_from_render(one) == (8, 7, 9, 8) == pil_rule(one)


def test_augment_cameras_appends_yawed_copies_of_the_first_view():
    """attack.augment_cameras == attack.py:404-415: add_cams - 1 deep copies of view 0, copy i yawed by 7 i degrees
    (R' = Y(7 i deg) R, view and full projection refreshed, camera_center NOT -- the reference's quirk)."""
    import numpy as np
    import torch
    from gsplat_attack.attack import augment_cameras
    from gsplat_attack.cameras import look_at_camera, world_to_view
    cams = [look_at_camera((2.0, 0.5, -1.0), (0, 0, 0), fovx=0.9, width=64, height=48),
            look_at_camera((-1.0, 0.2, 2.5), (0, 0, 0), fovx=0.9, width=64, height=48)]
    assert augment_cameras(cams, 1) == cams and augment_cameras(cams, 0) == cams
    out = augment_cameras(cams, 4)
    assert len(out) == 5 and out[:2] == cams
    R0 = np.array(cams[0].R, dtype=np.float64)
    for i, cam in enumerate(out[2:], start=1):
        th = np.radians(7.0 * i)
        Y = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        assert np.allclose(cam.R, Y @ R0, atol=1e-12) and np.array_equal(cam.T, cams[0].T)
        want_v = torch.tensor(world_to_view(Y @ R0, cams[0].T, cam.trans, cam.scale)).transpose(0, 1)
        assert torch.allclose(cam.world_view_transform, want_v.to(cam.world_view_transform.dtype), atol=1e-6)
        assert torch.allclose(cam.full_proj_transform, cam.world_view_transform @ cam.projection_matrix, atol=1e-6)
        assert torch.equal(cam.camera_center, cams[0].camera_center)           # stale by design (scene/cameras.py:60-69)
        assert not torch.allclose(cam.world_view_transform, cams[0].world_view_transform)
    assert np.array_equal(cams[0].R, R0)                                        # the original is untouched
