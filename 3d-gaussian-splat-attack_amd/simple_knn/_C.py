"""simple_knn._C -- import-time stand-in for the reference's second native dependency.

The reference imports ``distCUDA2`` at ``scene/gaussian_model.py:17`` but only calls it from
``create_from_pcd`` (:144), which the attack never reaches (a trained .ply is always loaded).  The
function is provided so a GaussianModel-style container imports; it is NOT on the raster hot path.
A HIP kNN kernel is a later-round item.
"""
import torch


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest neighbours, [P] float32 (chunked cdist)."""
    pts = points.float()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    k = min(4, P)
    chunk = max(1, min(P, (64 << 20) // max(4 * P, 1)))
    for s in range(0, P, chunk):
        d2 = torch.cdist(pts[s:s + chunk], pts).square_()
        near = torch.topk(d2, k, dim=1, largest=False).values[:, 1:]
        out[s:s + chunk] = near.mean(dim=1) if near.shape[1] else 0.0
    return out
