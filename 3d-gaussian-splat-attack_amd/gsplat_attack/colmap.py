"""Cameras from a COLMAP text model (``sparse/0/cameras.txt`` + ``images.txt``).

Counterpart of what the reference needs from ``scene/colmap_loader.py:156-271`` (text readers) and
``scene/dataset_readers.py:68-143`` (pose / field-of-view conversion, cameras sorted by image name) to turn a scene
directory into the camera list ``render()`` consumes.  Images themselves are not loaded (the attack only needs their
size, which COLMAP records); only undistorted PINHOLE / SIMPLE_PINHOLE models are accepted, like the reference.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, NamedTuple

import numpy as np

from .cameras import Camera


class ColmapCamera(NamedTuple):
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


class ColmapImage(NamedTuple):
    id: int
    qvec: np.ndarray      # (w, x, y, z), world -> camera
    tvec: np.ndarray
    camera_id: int
    name: str


def _data_lines(path: str):
    with open(path, "r") as f:
        return [ln.rstrip("\n") for ln in f]


def read_cameras_text(path: str) -> Dict[int, ColmapCamera]:
    cams = {}
    for ln in _data_lines(path):
        ln = ln.strip()
        if not ln or ln.startswith("#"):
            continue
        tok = ln.split()
        if tok[1] not in ("PINHOLE", "SIMPLE_PINHOLE"):
            raise ValueError(f"{path}: camera model {tok[1]} not handled: only undistorted PINHOLE / SIMPLE_PINHOLE")
        cams[int(tok[0])] = ColmapCamera(int(tok[0]), tok[1], int(tok[2]), int(tok[3]),
                                         np.array([float(v) for v in tok[4:]]))
    return cams


def read_images_text(path: str) -> Dict[int, ColmapImage]:
    """Two lines per image: the pose line, then the (possibly empty) 2D-point line, which is skipped."""
    lines = _data_lines(path)
    out, i = {}, 0
    while i < len(lines):
        ln = lines[i].strip()
        i += 1
        if not ln or ln.startswith("#"):
            continue
        tok = ln.split()
        out[int(tok[0])] = ColmapImage(int(tok[0]), np.array([float(v) for v in tok[1:5]]),
                                       np.array([float(v) for v in tok[5:8]]), int(tok[8]), tok[9])
        i += 1                                  # the POINTS2D line that belongs to this image
    return out


def quat_to_rotmat(q: np.ndarray) -> np.ndarray:
    """COLMAP (w, x, y, z) -> 3x3 rotation (world -> camera)."""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def focal2fov(focal: float, pixels: float) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def cameras_from_colmap(scene_dir: str, device="cpu") -> List[Camera]:
    """Camera list of ``scene_dir/sparse/0/{cameras,images}.txt``, sorted by image name (reference
    scene/dataset_readers.py:146).  R is stored transposed (camera -> world) like the reference's CameraInfo.R."""
    sparse = os.path.join(scene_dir, "sparse", "0")
    intr = read_cameras_text(os.path.join(sparse, "cameras.txt"))
    extr = read_images_text(os.path.join(sparse, "images.txt"))
    rows = []
    for img in extr.values():
        c = intr[img.camera_id]
        fx = c.params[0]
        fy = c.params[1] if c.model == "PINHOLE" else c.params[0]
        name = os.path.basename(img.name).split(".")[0]
        rows.append((name, Camera(quat_to_rotmat(img.qvec).T, img.tvec, focal2fov(fx, c.width), focal2fov(fy, c.height),
                                  c.width, c.height, uid=c.id, device=device)))
    rows.sort(key=lambda r: r[0])
    cams = [c for _, c in rows]
    for (name, _), c in zip(rows, cams):
        c.image_name = name
    return cams
