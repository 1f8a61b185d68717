"""Cameras from a NeRF-synthetic ("Blender") transforms file, and the render resolution rule.

Counterparts of the reference's ``scene/dataset_readers.py:179-259`` (readCamerasFromTransforms /
readNerfSyntheticInfo) and ``utils/camera_utils.py:20-53`` (loadCam): what turns a ``transforms_*.json`` plus its
images into the Camera objects render() consumes, and what decides the image size those cameras render at.
"""
from __future__ import annotations

import json
import math
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .cameras import Camera
from .colmap import focal2fov


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def render_resolution(orig_w: int, orig_h: int, resolution=-1, resolution_scale: float = 1.0) -> Tuple[int, int]:
    """(width, height) a camera renders at -- reference utils/camera_utils.py:23-41.  `resolution` 1|2|4|8 divides the
    original size (rounded); -1 keeps it unless the image is wider than 1600 px, which is then scaled to 1600;
    any other value is the target width."""
    if resolution in (1, 2, 4, 8):
        return (round(orig_w / (resolution_scale * resolution)), round(orig_h / (resolution_scale * resolution)))
    if resolution == -1:
        global_down = orig_w / 1600 if orig_w > 1600 else 1
    else:
        global_down = orig_w / resolution
    scale = float(global_down) * float(resolution_scale)
    return (int(orig_w / scale), int(orig_h / scale))


def transforms_pose(transform_matrix) -> Tuple[np.ndarray, np.ndarray]:
    """camera-to-world 4x4 in OpenGL/Blender axes (Y up, Z back) -> (R, T) as the reference stores them: R transposed
    (camera -> world), T the world -> camera translation (scene/dataset_readers.py:191-199)."""
    c2w = np.array(transform_matrix, dtype=np.float64)
    c2w[:3, 1:3] *= -1                       # to COLMAP axes (Y down, Z forward)
    w2c = np.linalg.inv(c2w)
    return np.transpose(w2c[:3, :3]), w2c[:3, 3]


def cameras_from_transforms(path: str, transformsfile: str = "transforms_train.json", extension: str = ".png",
                            resolution=-1, resolution_scale: float = 1.0, device="cpu",
                            image_size: Optional[Tuple[int, int]] = None) -> List[Camera]:
    """One Camera per frame of ``path/transformsfile``.  FoVx is the file's camera_angle_x, FoVy follows from the image's
    aspect ratio (scene/dataset_readers.py:214-216); the image size comes from the frame's image file (only its
    header is read) or from `image_size` when the images are absent."""
    with open(os.path.join(path, transformsfile)) as f:
        contents = json.load(f)
    fovx = float(contents["camera_angle_x"])
    cams = []
    for idx, frame in enumerate(contents["frames"]):
        image_path = os.path.join(path, frame["file_path"] + extension)
        if image_size is not None:
            w, h = image_size
        else:
            from PIL import Image
            with Image.open(image_path) as im:
                w, h = im.size
        R, T = transforms_pose(frame["transform_matrix"])
        fovy = focal2fov(fov2focal(fovx, w), h)
        rw, rh = render_resolution(w, h, resolution, resolution_scale)
        cam = Camera(R, T, fovx, fovy, rw, rh, uid=idx, device=device)
        cam.image_name = os.path.splitext(os.path.basename(image_path))[0]
        cam.image_path = image_path
        cams.append(cam)
    return cams


def blend_on_background(rgba: np.ndarray, white_background: bool) -> np.ndarray:
    """Ground-truth image of a frame: RGBA in [0,255] composited on white or black, back to bytes the way the reference
    does it (scene/dataset_readers.py:206-212: ``np.array(arr * 255.0, dtype=np.byte)``, i.e. truncation)."""
    bg = np.array([1, 1, 1]) if white_background else np.array([0, 0, 0])
    norm = rgba / 255.0
    arr = norm[:, :, :3] * norm[:, :, 3:4] + bg * (1 - norm[:, :, 3:4])
    return np.array(arr * 255.0, dtype=np.byte).view(np.uint8)


def read_nerf_synthetic(path: str, eval: bool = False, extension: str = ".png", **kw):
    """(train cameras, test cameras) of a Blender scene directory -- the camera part of readNerfSyntheticInfo
    (scene/dataset_readers.py:222-232): without `eval` the test frames join the training set."""
    train = cameras_from_transforms(path, "transforms_train.json", extension, **kw)
    test = cameras_from_transforms(path, "transforms_test.json", extension, **kw)
    if not eval:
        for c in test:
            c.uid += len(train)
        train, test = train + test, []
    return train, test
