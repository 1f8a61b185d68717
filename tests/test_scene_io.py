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
