"""Camera matrices for the rasteriser boundary.

Counterpart of the reference's ``scene/cameras.py:48-57`` and
``utils/graphics_utils.py:38-71``: same conventions (row-vector matrices stored
transposed, OpenGL-style projection with znear=0.01 / zfar=100, camera centre
read from the inverse view matrix), built on whichever device the caller asks
for instead of hard-coded ``.cuda()``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

ZNEAR = 0.01   # scene/cameras.py:49
ZFAR = 100.0   # scene/cameras.py:48


def world_to_view(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """4x4 world->view (column-vector form). ``R`` is stored camera-to-world like
    COLMAP loaders do, hence the transpose (utils/graphics_utils.py:38-49)."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """utils/graphics_utils.py:51-71 (column-vector form, float32)."""
    ty = math.tan(fovy / 2.0)
    tx = math.tan(fovx / 2.0)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera:
    """The fields ``render()`` reads (gaussian_renderer/__init__.py:33-46):
    FoVx, FoVy, image_height, image_width, world_view_transform,
    full_proj_transform, camera_center."""

    def __init__(self, R, T, FoVx: float, FoVy: float, width: int, height: int, uid: int = 0,
                 trans=(0.0, 0.0, 0.0), scale: float = 1.0, device="cpu"):
        self.uid = uid
        self.R = np.asarray(R, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.FoVx = float(FoVx)
        self.FoVy = float(FoVy)
        self.image_width = int(width)
        self.image_height = int(height)
        self.znear, self.zfar = ZNEAR, ZFAR
        self.trans, self.scale = trans, scale
        self.device = torch.device(device)
        self.projection_matrix = projection_matrix(self.znear, self.zfar, self.FoVx, self.FoVy) \
            .transpose(0, 1).to(self.device)
        self._refresh_view()
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()   # dense: no per-view copy in the binding

    def _refresh_view(self):
        # scene/cameras.py:60-69: view + full projection are refreshed, camera_center is NOT
        #  -- kept, so synthesised cameras shade identically.
        # .contiguous(): the reference keeps the transposed VIEW; a dense copy has the same values and spares the
        # rasteriser binding a 16-element copy kernel per view
        self.world_view_transform = torch.tensor(world_to_view(self.R, self.T, self.trans, self.scale)) \
            .transpose(0, 1).contiguous().to(self.device)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0)
                                    .bmm(self.projection_matrix.unsqueeze(0))).squeeze(0)

    def transform(self, T):
        """scene/cameras.py:72-83."""
        T = np.asarray(T)
        assert T.shape == (3,), "T must be of shape (3,)"
        self.T = T
        self._refresh_view()

    def yaw(self, angle_deg: float):
        """scene/cameras.py:85-105."""
        th = np.radians(angle_deg)
        c, s = np.cos(th), np.sin(th)
        Y = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)
        self.R = (Y @ self.R).astype(np.float64)
        self._refresh_view()

    def to(self, device):
        self.device = torch.device(device)
        for n in ("projection_matrix", "world_view_transform", "full_proj_transform", "camera_center"):
            setattr(self, n, getattr(self, n).to(self.device))
        return self


def look_at_camera(eye, target, up=(0.0, -1.0, 0.0), *, fovx: float, width: int, height: int,
                   fovy: float | None = None, uid: int = 0, device="cpu") -> Camera:
    """Synthetic-scene helper: a camera at ``eye`` looking at ``target`` (view-space +z forward,
    +y down as in COLMAP)."""
    eye = np.asarray(eye, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    f = target - eye
    f /= np.linalg.norm(f)
    down = -np.asarray(up, dtype=np.float64)
    r = np.cross(down, f)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R_c2w = np.stack([r, d, f], axis=1)          # columns = camera axes in world coordinates
    T = -R_c2w.T @ eye
    if fovy is None:
        fovy = 2.0 * math.atan(math.tan(fovx / 2.0) * height / width)
    return Camera(R_c2w, T, fovx, fovy, width, height, uid=uid, device=device)
