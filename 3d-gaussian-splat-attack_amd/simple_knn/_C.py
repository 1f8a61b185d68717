"""simple_knn._C -- the reference's second native import (``from simple_knn._C import distCUDA2``, reference
scene/gaussian_model.py:17; called at :144 in create_from_pcd to seed the initial scales).

On a HIP device it calls ``gsr_knn_dist2`` of libgsraster.so (uniform-grid exact 3-NN, csrc/gsr_knn.hip.h).  CPU tensors
-- which the reference never passes, its name says so -- get a chunked torch.cdist evaluation of the same definition,
so that a GaussianModel-style container can be exercised in CPU-only unit tests; that branch is not on the raster path.
"""
import ctypes

import torch


def _dist2_torch(pts: torch.Tensor) -> torch.Tensor:
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    k = min(4, P)
    chunk = max(1, min(P, (64 << 20) // max(4 * P, 1)))
    for s in range(0, P, chunk):
        d2 = torch.cdist(pts[s:s + chunk], pts).square_()
        near = torch.topk(d2, k, dim=1, largest=False).values[:, 1:]
        out[s:s + chunk] = near.mean(dim=1) if near.shape[1] else 0.0
    return out


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest neighbours, [P] float32."""
    pts = points.detach().float().contiguous()
    if not pts.is_cuda:
        return _dist2_torch(pts)
    import diff_gaussian_rasterization as D
    lib = D._load()
    if not hasattr(lib, "_knn_ready"):
        lib.gsr_knn_dist2.restype = ctypes.c_int
        lib.gsr_knn_dist2.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        lib._knn_ready = True
    P = int(pts.shape[0])
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    with torch.cuda.device(pts.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
        rc = lib.gsr_knn_dist2(pts.data_ptr(), P, out.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError(D._err(lib))
    return out
