"""Attribute container with the names/shapes/activations the render boundary and
the PGD step functions read.

Counterpart of the getters of the reference's ``scene/gaussian_model.py:24-39,97-124``
(seven parameter tensors + exp / sigmoid / normalize / cat activations).  Training-time
machinery (densify / prune / optimiser state, :264-711) is out of scope: the attack never
calls it.
"""
from __future__ import annotations

import torch
from torch import nn

NUM_OBJECTS = 16  # scene/gaussian_model.py:52


class GaussianModel:
    def __init__(self, sh_degree: int = 3):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree      # load_ply sets active = max (scene/gaussian_model.py:467)
        self.num_objects = NUM_OBJECTS
        e = torch.empty(0)
        self._xyz = e
        self._features_dc = e
        self._features_rest = e
        self._scaling = e
        self._rotation = e
        self._opacity = e
        self._objects_dc = e

    # ---- construction -------------------------------------------------
    @classmethod
    def from_tensors(cls, xyz, features_dc, features_rest, scaling, rotation, opacity, objects_dc=None,
                     sh_degree: int = 3, device=None, requires_grad: bool = True) -> "GaussianModel":
        """Raw (pre-activation) tensors with the load_ply layout (scene/gaussian_model.py:459-467):
        xyz [P,3], features_dc [P,1,3], features_rest [P,K-1,3], scaling (log) [P,3],
        rotation (w,x,y,z un-normalised) [P,4], opacity (logit) [P,1], objects_dc [P,1,16]."""
        m = cls(sh_degree)
        P = xyz.shape[0]
        if objects_dc is None:
            objects_dc = torch.zeros(P, 1, NUM_OBJECTS)

        def par(t):
            t = t.detach().to(torch.float32)
            if device is not None:
                t = t.to(device)
            return nn.Parameter(t.contiguous().clone(), requires_grad=requires_grad)
        m._xyz = par(xyz)
        m._features_dc = par(features_dc)
        m._features_rest = par(features_rest)
        m._scaling = par(scaling)
        m._rotation = par(rotation)
        m._opacity = par(opacity)
        m._objects_dc = par(objects_dc)
        return m

    @classmethod
    def create_from_pcd(cls, points, colors, sh_degree: int = 3, device="cpu", generator=None) -> "GaussianModel":
        """Initial model from a sparse point cloud (reference scene/gaussian_model.py:130-158): DC colour = RGB2SH(rgb),
        higher bands 0, isotropic log-scale = log sqrt(mean squared distance to the 3 nearest neighbours)
        (``simple_knn._C.distCUDA2`` -- the HIP kernel on a device, clamped at 1e-7), identity rotations, opacity
        logit(0.1), random object features RGB2SH(U[0,1)).  points [P,3], colors [P,3] in [0,1] (array-likes)."""
        from simple_knn._C import distCUDA2
        from .sh import RGB2SH
        xyz = torch.as_tensor(points, dtype=torch.float32).to(device).contiguous()
        rgb = torch.as_tensor(colors, dtype=torch.float32).to(device)
        P = xyz.shape[0]
        K = (sh_degree + 1) ** 2
        dist2 = torch.clamp_min(distCUDA2(xyz), 1e-7)
        scales = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3)
        rots = torch.zeros(P, 4, device=xyz.device)
        rots[:, 0] = 1.0
        opac = torch.full((P, 1), 0.1, device=xyz.device)
        opac = torch.log(opac / (1.0 - opac))                            # inverse_sigmoid, utils/general_utils.py:18-19
        objs = RGB2SH(torch.rand(P, NUM_OBJECTS, generator=generator).to(xyz.device))[:, None, :]
        return cls.from_tensors(xyz, RGB2SH(rgb)[:, None, :], torch.zeros(P, K - 1, 3, device=xyz.device), scales, rots,
                                opac, objs, sh_degree=sh_degree, device=xyz.device)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self._objects_dc]

    def named_parameters(self):
        return dict(xyz=self._xyz, f_dc=self._features_dc, f_rest=self._features_rest, scaling=self._scaling,
                    rotation=self._rotation, opacity=self._opacity, objects_dc=self._objects_dc)

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None

    _PARAM_ATTRS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_objects_dc")

    # ---- scene surgery used by the attack set-up (scene/gaussian_model.py:216-262) ------------------------------
    def removal_setup(self, mask3d: torch.Tensor) -> None:
        """Keep the Gaussians NOT selected by mask3d (reference :216-241).  As in the reference the survivors are
        re-wrapped in nn.Parameter, whose default requires_grad=True wins."""
        keep = ~mask3d.bool().reshape(-1)
        for n in self._PARAM_ATTRS:
            setattr(self, n, nn.Parameter(getattr(self, n)[keep].detach().clone()))

    def concat_setup(self, feature_name: str, tensor_to_concat: torch.Tensor, requires_grad: bool) -> None:
        """Append rows to one attribute and re-wrap it (reference :243-262)."""
        cur = getattr(self, f"_{feature_name}")
        cat = torch.cat((cur, tensor_to_concat.to(cur.device)), dim=0).detach().clone().requires_grad_(requires_grad)
        setattr(self, f"_{feature_name}", nn.Parameter(cat, requires_grad=requires_grad))

    def clone(self) -> "GaussianModel":
        m = GaussianModel(self.max_sh_degree)
        m.active_sh_degree = self.active_sh_degree
        for n in self._PARAM_ATTRS:
            p = getattr(self, n)
            setattr(m, n, nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad))
        return m

    def save_ply(self, path: str) -> None:
        from .ply import save_gaussians
        save_gaussians(self, path)

    @classmethod
    def load_ply(cls, path: str, sh_degree: int = 3, device=None) -> "GaussianModel":
        from .ply import load_gaussians
        return load_gaussians(path, sh_degree, device)

    # ---- getters (scene/gaussian_model.py:97-124) ---------------------
    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_objects(self):
        return self._objects_dc

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier: float = 1.0):
        """scene/gaussian_model.py:25-29 with utils/general_utils.py:64-110 (which re-normalises
        the raw quaternion inside build_rotation): packed (xx,xy,xz,yy,yz,zz)."""
        q = self._rotation / self._rotation.norm(dim=1, keepdim=True)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)
