"""PLY wire format of a Gaussian scene.

Counterpart of the reference's ``scene/gaussian_model.py:377-411`` (save_ply), ``:418-467`` (load_ply) and ``:469-556``
(combine_splats), which go through the third-party ``plyfile`` package (absent here): a minimal reader / writer for the
one layout the reference uses -- a single ``vertex`` element of scalar properties, ``binary_little_endian 1.0`` (ASCII is
read too).  Property names and order are the reference's:

    x y z nx ny nz  f_dc_0..2  f_rest_0..44  opacity  scale_0..2  rot_0..3  obj_dc_0..15

``f_dc`` / ``f_rest`` / ``obj_dc`` are stored CHANNEL-major (``[P,K,3].transpose(1,2).flatten``), i.e. ``f_rest_j`` with
``j = channel*15 + coefficient``; opacity is the logit, scale the log, rot the un-normalised (w,x,y,z) quaternion.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, List, Sequence

import numpy as np
import torch

from .gaussian_model import GaussianModel, NUM_OBJECTS

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def write_ply(path: str, props: "OrderedDict[str, np.ndarray]") -> None:
    """One float32 vertex property per dict entry, in dict order, binary little endian."""
    names = list(props)
    n = len(props[names[0]]) if names else 0
    rec = np.empty(n, dtype=[(k, "<f4") for k in names])
    for k in names:
        col = np.asarray(props[k], dtype=np.float32).reshape(-1)
        if col.shape[0] != n:
            raise ValueError(f"property {k} has {col.shape[0]} values, expected {n}")
        rec[k] = col
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property float {k}" for k in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def read_ply(path: str) -> "OrderedDict[str, np.ndarray]":
    """Vertex properties of a PLY file as float64/int arrays in file order (other elements are ignored)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        elements: List[list] = []          # [name, count, [(type, prop)]]
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                elements[-1][2].append((_PLY_TYPES[tok[1]], tok[2]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("binary_little_endian", "binary_big_endian", "ascii"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        out: "OrderedDict[str, np.ndarray]" = OrderedDict()
        for name, count, plist in elements:
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                data = {p: np.array([r[i] for r in rows], dtype=np.dtype(t)) for i, (t, p) in enumerate(plist)}
            else:
                end = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(p, end + t) for t, p in plist])
                raw = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                data = {p: raw[p] for _, p in plist}
            if name == "vertex":
                for _, p in plist:
                    out[p] = np.asarray(data[p])
        return out


def gaussian_attribute_names(n_dc: int = 3, n_rest: int = 45, n_obj: int = NUM_OBJECTS) -> List[str]:
    """scene/gaussian_model.py:377-391."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
    names += [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)] + [f"obj_dc_{i}" for i in range(n_obj)]
    return names


def save_gaussians(model: GaussianModel, path: str) -> None:
    """scene/gaussian_model.py:393-411 (normals written as zeros)."""
    def chan_major(t):
        return t.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    xyz = model._xyz.detach().cpu().numpy()
    cols = np.concatenate((xyz, np.zeros_like(xyz), chan_major(model._features_dc), chan_major(model._features_rest),
                           model._opacity.detach().cpu().numpy(), model._scaling.detach().cpu().numpy(),
                           model._rotation.detach().cpu().numpy(), chan_major(model._objects_dc)), axis=1)
    names = gaussian_attribute_names(model._features_dc.shape[1] * model._features_dc.shape[2],
                                     model._features_rest.shape[1] * model._features_rest.shape[2],
                                     model._objects_dc.shape[1] * model._objects_dc.shape[2])
    assert cols.shape[1] == len(names)
    write_ply(path, OrderedDict((n, cols[:, i]) for i, n in enumerate(names)))


def _numbered(props, prefix: str) -> List[str]:
    return sorted((p for p in props if p.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))


def _load_attrs(path: str, sh_degree: int, pad_rest: bool, zero_objects: bool) -> Dict[str, torch.Tensor]:
    v = read_ply(path)
    P = len(v["x"])
    f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
    xyz = f32(np.stack((v["x"], v["y"], v["z"]), axis=1))
    opacity = f32(np.asarray(v["opacity"])[:, None])
    f_dc = f32(np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1))[:, :, None]            # [P,3,1]
    rest_names = _numbered(v, "f_rest_")
    expected = 3 * (sh_degree + 1) ** 2 - 3
    rest = np.stack([np.asarray(v[n], dtype=np.float32) for n in rest_names], axis=1) if rest_names \
        else np.zeros((P, 0), np.float32)
    if pad_rest:                                                     # combine_splats: pad or truncate (:507-512)
        if rest.shape[1] < expected:
            rest = np.hstack((rest, np.zeros((P, expected - rest.shape[1]), np.float32)))
        rest = rest[:, :expected]
    elif rest.shape[1] != expected:                                  # load_ply asserts (:437)
        raise AssertionError(f"{path}: {rest.shape[1]} f_rest properties, SH degree {sh_degree} needs {expected}")
    f_rest = f32(rest.reshape(P, 3, (sh_degree + 1) ** 2 - 1))      # file is [P,3,15] flattened
    scales = f32(np.stack([v[n] for n in _numbered(v, "scale_")], axis=1))
    rots = f32(np.stack([v[n] for n in _numbered(v, "rot")], axis=1))
    objects = torch.zeros(P, NUM_OBJECTS, 1)
    if not zero_objects and all(f"obj_dc_{i}" in v for i in range(NUM_OBJECTS)):
        objects = f32(np.stack([v[f"obj_dc_{i}"] for i in range(NUM_OBJECTS)], axis=1))[:, :, None]
    return dict(xyz=xyz, features_dc=f_dc.transpose(1, 2).contiguous(), features_rest=f_rest.transpose(1, 2).contiguous(),
                scaling=scales, rotation=rots, opacity=opacity, objects_dc=objects.transpose(1, 2).contiguous())


def load_gaussians(path: str, sh_degree: int = 3, device=None) -> GaussianModel:
    """scene/gaussian_model.py:418-467; a file without obj_dc_* properties loads with zero object features."""
    return GaussianModel.from_tensors(**_load_attrs(path, sh_degree, False, False), sh_degree=sh_degree, device=device)


def combine_splats(paths: Sequence[str], sh_degree: int = 3, device=None):
    """scene/gaussian_model.py:469-556 -> (model, masks): concatenation of several files, object features zeroed,
    f_rest padded / truncated to the model's SH degree, one boolean mask per file over the combined Gaussians.
    (The reference's masks are each only as long as their own file; here every mask spans the combined model, which
    is what its callers index with.)"""
    if not paths:
        raise ValueError("No valid .ply files were loaded.")
    parts = [_load_attrs(p, sh_degree, True, True) for p in paths]
    cat = {k: torch.cat([d[k] for d in parts]) for k in parts[0]}
    model = GaussianModel.from_tensors(**cat, sh_degree=sh_degree, device=device)
    total = cat["xyz"].shape[0]
    masks, pos = [], 0
    for d in parts:
        m = torch.zeros(total, dtype=torch.bool, device=device)
        m[pos:pos + d["xyz"].shape[0]] = True
        masks.append(m)
        pos += d["xyz"].shape[0]
    model.masks = masks
    return model, masks
