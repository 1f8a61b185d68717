"""A scene directory -> (GaussianModel, cameras): what the attack needs from the reference's ``Scene``.

Counterpart of ``scene/__init__.py:23-103`` with ``scene/dataset_readers.py:40-177`` and
``utils/system_utils.py:26-28`` for COLMAP scenes (the Blender branch is not built):

  * cameras from ``<source>/sparse/0`` (binary first, text as the fallback), sorted by image name; with ``eval`` every
    ``llffhold``-th camera (index % 8 == 0) goes to the test list;
  * ``cameras_extent`` = 1.1 x the largest distance of a training camera centre from their mean (getNerfppNorm);
  * ``shuffle``: ``random.seed(42); random.shuffle(...)`` on the train and then the test list, exactly as the reference
    (the Python stdlib shuffle, so the order is the reference's); ``cam_indices`` then selects from the shuffled list;
  * the model: ``<model>/point_cloud/iteration_N/point_cloud.ply`` (N = highest found for ``load_iteration=-1``), else
    initialised from the sparse point cloud (``points3D.bin`` / ``.txt``) through ``GaussianModel.create_from_pcd``.

Ground-truth images are not opened (the attack renders; it never reads them); ``Camera.image_name`` carries the name.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Sequence

import numpy as np

from .colmap import cameras_from_colmap, read_points3D_binary, read_points3D_text
from .gaussian_model import GaussianModel


def search_for_max_iteration(folder: str) -> int:
    """Highest N among the ``iteration_N`` entries of ``folder`` (utils/system_utils.py:26-28)."""
    return max(int(name.split("_")[-1]) for name in os.listdir(folder))


def cameras_extent(cameras: Sequence) -> float:
    """getNerfppNorm's radius: 1.1 x max distance of the camera centres from their mean."""
    centres = np.stack([np.asarray(-(np.asarray(c.R) @ np.asarray(c.T)), dtype=np.float64) for c in cameras], axis=1)
    mean = centres.mean(axis=1, keepdims=True)
    return float(np.linalg.norm(centres - mean, axis=0).max() * 1.1)


class Scene:
    def __init__(self, source_path: str, model_path: Optional[str] = None, load_iteration: Optional[int] = None,
                 shuffle: bool = True, eval: bool = False, cam_indices: Optional[Sequence[int]] = None, llffhold: int = 8,
                 sh_degree: int = 3, device="cpu"):
        if not os.path.exists(os.path.join(source_path, "sparse")):
            raise ValueError(f"{source_path}: no sparse/ directory (only COLMAP scenes are handled)")
        self.model_path = model_path
        self.loaded_iter = None
        if load_iteration:
            if model_path is None:
                raise ValueError("load_iteration needs model_path")
            self.loaded_iter = (search_for_max_iteration(os.path.join(model_path, "point_cloud"))
                                if load_iteration == -1 else load_iteration)
        cams = cameras_from_colmap(source_path, device=device)                 # sorted by image name
        if eval:
            train = [c for i, c in enumerate(cams) if i % llffhold != 0]
            test = [c for i, c in enumerate(cams) if i % llffhold == 0]
        else:
            train, test = list(cams), []
        self.cameras_extent = cameras_extent(train)
        if shuffle:
            random.seed(42)
            random.shuffle(train)
            random.shuffle(test)
        if cam_indices:
            train = [train[i] for i in cam_indices]
        self.train_cameras: List = train
        self.test_cameras: List = test
        if self.loaded_iter:
            ply = os.path.join(model_path, "point_cloud", f"iteration_{self.loaded_iter}", "point_cloud.ply")
            self.gaussians = GaussianModel.load_ply(ply, sh_degree=sh_degree, device=device)
        else:
            sparse = os.path.join(source_path, "sparse", "0")
            try:
                xyz, rgb, _ = read_points3D_binary(os.path.join(sparse, "points3D.bin"))
            except (OSError, ValueError):
                xyz, rgb, _ = read_points3D_text(os.path.join(sparse, "points3D.txt"))
            self.gaussians = GaussianModel.create_from_pcd(xyz, rgb.astype(np.float32) / 255.0, sh_degree=sh_degree,
                                                           device=device)

    def save(self, iteration: int) -> str:
        path = os.path.join(self.model_path, "point_cloud", f"iteration_{iteration}", "point_cloud.ply")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        self.gaussians.save_ply(path)
        return path

    def getTrainCameras(self, scale: float = 1.0):
        return self.train_cameras

    def getTestCameras(self, scale: float = 1.0):
        return self.test_cameras
