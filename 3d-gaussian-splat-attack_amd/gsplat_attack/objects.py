"""Consumers of render()'s 16-channel object map (``render_object``).

The attack itself never reads that output; the reference's evaluation viewer does (``render.py:126-131``): a 1x1
convolution classifies the 16 composited object features per pixel, the arg-max is the object id map, and ids are
painted with a golden-ratio hue palette (``render.py:45-73``).  ``feature_to_rgb`` (``render.py:25-43``) shows the
features themselves through their first three principal components.
"""
from __future__ import annotations

import colorsys

import numpy as np
import torch
from torch import nn

NUM_OBJECTS = 16


class ObjectClassifier(nn.Module):
    """The per-pixel classifier of the object map: Conv2d(16, num_classes, kernel_size=1) (Gaussian-Grouping's;
    render.py:126 applies it to the [16,H,W] map as is)."""

    def __init__(self, num_classes: int = 256, num_objects: int = NUM_OBJECTS):
        super().__init__()
        self.conv = nn.Conv2d(num_objects, num_classes, kernel_size=1)

    def forward(self, render_object: torch.Tensor) -> torch.Tensor:
        return self.conv(render_object)


def predict_objects(render_object: torch.Tensor, classifier: nn.Module) -> torch.Tensor:
    """[16,H,W] object map -> [H,W] object ids (render.py:126-127)."""
    return torch.argmax(classifier(render_object), dim=0)


def id2rgb(id: int, max_num_obj: int = 256) -> np.ndarray:
    """Colour of an object id (render.py:45-63): hue by the golden ratio, saturation alternating, id 0 black."""
    if not 0 <= id <= max_num_obj:
        raise ValueError("ID should be in range(0, max_num_obj)")
    h = (id * 1.6180339887) % 1
    s = 0.5 + (id % 2) * 0.5
    rgb = np.zeros((3,), dtype=np.uint8)
    if id == 0:
        return rgb
    r, g, b = colorsys.hls_to_rgb(h, 0.5, s)
    rgb[0], rgb[1], rgb[2] = int(r * 255), int(g * 255), int(b * 255)
    return rgb


def visualize_obj(objects: np.ndarray) -> np.ndarray:
    """[H,W] ids -> [H,W,3] uint8 (render.py:65-71)."""
    out = np.zeros((*objects.shape[-2:], 3), dtype=np.uint8)
    for i in np.unique(objects):
        out[objects == i] = id2rgb(int(i))
    return out


def feature_to_rgb(features: torch.Tensor) -> np.ndarray:
    """[C,H,W] features -> [H,W,3] uint8: first three principal components, jointly normalised to 0..255
    (render.py:25-43).  Component signs follow the convention of making each component's largest-magnitude loading
    positive; a constant map gives zeros."""
    C, H, W = features.shape
    X = features.detach().reshape(C, -1).T.double().cpu()
    X = X - X.mean(dim=0, keepdim=True)
    _, _, Vt = torch.linalg.svd(X, full_matrices=False)
    Vt = Vt[:3]
    sign = torch.sign(Vt[torch.arange(Vt.shape[0]), Vt.abs().argmax(dim=1)])
    Vt = Vt * torch.where(sign == 0, torch.ones_like(sign), sign)[:, None]
    proj = (X @ Vt.T).reshape(H, W, -1).numpy()
    if proj.shape[2] < 3:
        proj = np.concatenate([proj, np.zeros((H, W, 3 - proj.shape[2]))], axis=2)
    span = proj.max() - proj.min()
    if span == 0:
        return np.zeros((H, W, 3), dtype=np.uint8)
    return (255 * (proj - proj.min()) / span).astype("uint8")
