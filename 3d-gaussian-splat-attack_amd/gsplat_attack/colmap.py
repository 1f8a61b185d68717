"""Cameras (and the sparse point cloud) from a COLMAP model, text or binary.

Counterpart of what the reference needs from ``scene/colmap_loader.py`` (text readers :83-123, :156-271; binary
readers :125-154, :180-244) and ``scene/dataset_readers.py:68-143`` (pose / field-of-view conversion, cameras sorted
by image name, binary tried first and text as the fallback :133-142) to turn a scene directory into the camera list
``render()`` consumes.  The binary layout is COLMAP's published one (little endian): counts are uint64, ids int32
(point ids uint64), poses and intrinsics float64, image names NUL-terminated.  Only undistorted PINHOLE /
SIMPLE_PINHOLE cameras are accepted, like the reference.  Ground-truth images are optional
(``load_image``: PIL -> float [3,H,W] in [0,1], reference utils/general_utils.py:21-27).
"""
from __future__ import annotations

import math
import os
import struct
from typing import Dict, List, NamedTuple, Optional, Tuple

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


# COLMAP camera model ids -> (name, number of parameters); only the first two are usable here
_CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                  4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                  8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_MODEL_IDS = {v[0]: k for k, v in _CAMERA_MODELS.items()}


def _unpack(f, fmt: str):
    size = struct.calcsize("<" + fmt)
    buf = f.read(size)
    if len(buf) != size:
        raise ValueError("COLMAP binary file ends inside a record")
    return struct.unpack("<" + fmt, buf)


def read_cameras_binary(path: str) -> Dict[int, ColmapCamera]:
    cams = {}
    with open(path, "rb") as f:
        (n,) = _unpack(f, "Q")
        for _ in range(n):
            cid, model_id, width, height = _unpack(f, "iiQQ")
            if model_id not in _CAMERA_MODELS:
                raise ValueError(f"{path}: unknown camera model id {model_id}")
            name, npar = _CAMERA_MODELS[model_id]
            params = np.array(_unpack(f, "d" * npar))
            if name not in ("PINHOLE", "SIMPLE_PINHOLE"):
                raise ValueError(f"{path}: camera model {name} not handled: only undistorted PINHOLE / SIMPLE_PINHOLE")
            cams[cid] = ColmapCamera(cid, name, int(width), int(height), params)
    return cams


def read_images_binary(path: str) -> Dict[int, ColmapImage]:
    out = {}
    with open(path, "rb") as f:
        (n,) = _unpack(f, "Q")
        for _ in range(n):
            vals = _unpack(f, "idddddddi")
            iid, qvec, tvec, cid = vals[0], np.array(vals[1:5]), np.array(vals[5:8]), vals[8]
            name = bytearray()
            while True:
                ch = f.read(1)
                if ch == b"":
                    raise ValueError(f"{path}: unterminated image name")
                if ch == b"\x00":
                    break
                name += ch
            (npts,) = _unpack(f, "Q")
            f.seek(24 * npts, os.SEEK_CUR)              # (x, y: float64, point3D id: int64) per 2D point: not needed
            out[iid] = ColmapImage(iid, qvec, tvec, cid, name.decode("utf-8"))
    return out


def read_points3D_binary(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(xyz float64 [n,3], rgb uint8 [n,3], reprojection error float64 [n]) -- what seeds a scene's Gaussians."""
    with open(path, "rb") as f:
        (n,) = _unpack(f, "Q")
        xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3), dtype=np.uint8), np.empty(n)
        for i in range(n):
            vals = _unpack(f, "QdddBBBdQ")
            xyz[i], rgb[i], err[i] = vals[1:4], vals[4:7], vals[7]
            f.seek(8 * vals[8], os.SEEK_CUR)            # track: (image id, 2D point index) int32 pairs
    return xyz, rgb, err


def read_points3D_text(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    xyz, rgb, err = [], [], []
    for ln in _data_lines(path):
        ln = ln.strip()
        if not ln or ln.startswith("#"):
            continue
        tok = ln.split()
        xyz.append([float(v) for v in tok[1:4]])
        rgb.append([int(v) for v in tok[4:7]])
        err.append(float(tok[7]))
    return (np.array(xyz, dtype=np.float64).reshape(-1, 3), np.array(rgb, dtype=np.uint8).reshape(-1, 3),
            np.array(err, dtype=np.float64))


def write_model_binary(cams: Dict[int, ColmapCamera], images: Dict[int, ColmapImage], sparse_dir: str,
                       points: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None) -> None:
    """cameras.bin / images.bin (/ points3D.bin) in COLMAP's binary layout, without 2D points or tracks."""
    os.makedirs(sparse_dir, exist_ok=True)
    with open(os.path.join(sparse_dir, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for c in cams.values():
            f.write(struct.pack("<iiQQ", c.id, _MODEL_IDS[c.model], c.width, c.height))
            f.write(struct.pack("<" + "d" * len(c.params), *[float(v) for v in c.params]))
    with open(os.path.join(sparse_dir, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images.values():
            f.write(struct.pack("<idddddddi", im.id, *[float(v) for v in im.qvec], *[float(v) for v in im.tvec],
                                im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", 0))
    if points is not None:
        xyz, rgb, err = points
        with open(os.path.join(sparse_dir, "points3D.bin"), "wb") as f:
            f.write(struct.pack("<Q", len(xyz)))
            for i in range(len(xyz)):
                f.write(struct.pack("<QdddBBBdQ", i + 1, *[float(v) for v in xyz[i]], *[int(v) for v in rgb[i]],
                                    float(err[i]), 0))


def load_image(path: str, resolution: Optional[Tuple[int, int]] = None):
    """Ground-truth image as a float tensor [3,H,W] in [0,1] (alpha, if any, as a fourth channel), optionally
    resized to (width, height) first -- reference utils/general_utils.py:21-27 (PILtoTorch)."""
    import torch
    from PIL import Image
    img = Image.open(path)
    if resolution is not None:
        img = img.resize(resolution)
    arr = torch.from_numpy(np.array(img)) / 255.0
    return arr.permute(2, 0, 1) if arr.dim() == 3 else arr.unsqueeze(dim=-1).permute(2, 0, 1)


def quat_to_rotmat(q: np.ndarray) -> np.ndarray:
    """COLMAP (w, x, y, z) -> 3x3 rotation (world -> camera)."""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def focal2fov(focal: float, pixels: float) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def cameras_from_colmap(scene_dir: str, device="cpu") -> List[Camera]:
    """Camera list of ``scene_dir/sparse/0/{cameras,images}.{bin,txt}``, sorted by image name (reference
    scene/dataset_readers.py:146).  R is stored transposed (camera -> world) like the reference's CameraInfo.R."""
    sparse = os.path.join(scene_dir, "sparse", "0")
    if os.path.exists(os.path.join(sparse, "images.bin")) and os.path.exists(os.path.join(sparse, "cameras.bin")):
        intr = read_cameras_binary(os.path.join(sparse, "cameras.bin"))      # binary first, like the reference
        extr = read_images_binary(os.path.join(sparse, "images.bin"))
    else:
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
