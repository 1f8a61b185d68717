"""distCUDA2 on the HIP path against a brute-force evaluation of its definition."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _brute(pts):
    d2 = torch.cdist(pts.double(), pts.double()).square()
    d2.fill_diagonal_(float("inf"))
    return torch.topk(d2, 3, dim=1, largest=False).values.mean(dim=1)


@pytest.mark.parametrize("kind", ["uniform", "city", "flat", "duplicates", "outliers", "tiny"])
def test_matches_brute_force(kind):
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(sum(map(ord, kind)))   # str hashes change per process
    if kind == "uniform":
        pts = torch.rand(6000, 3, generator=g) * 10
    elif kind == "city":
        from gsplat_attack.scenes import city
        pts = city(8000, 3, 20.0, 60)["xyz"]
    elif kind == "flat":                      # a degenerate bounding box (all z equal)
        pts = torch.rand(3000, 3, generator=g)
        pts[:, 2] = 0.25
    elif kind == "duplicates":                # coincident points: zero distances count
        base = torch.rand(500, 3, generator=g)
        pts = torch.cat([base, base[:200], base[:50]])
    elif kind == "outliers":                  # isolated points far from a dense cluster
        pts = torch.cat([torch.randn(4000, 3, generator=g) * 0.1, torch.randn(12, 3, generator=g) * 500])
    else:
        pts = torch.rand(7, 3, generator=g)
    got = distCUDA2(pts.cuda()).cpu().double()
    ref = _brute(pts)
    assert torch.allclose(got, ref, rtol=2e-5, atol=1e-9), (kind, (got - ref).abs().max().item())


def test_one_million_points_runs_and_is_plausible():
    from simple_knn._C import distCUDA2
    from gsplat_attack.scenes import city
    pts = city(1_000_000, 3, 20.0, 60)["xyz"].cuda()
    d = distCUDA2(pts)
    assert d.shape == (1_000_000,) and torch.isfinite(d).all() and float(d.min()) >= 0
    # spot check 200 random points against brute force
    idx = torch.randperm(1_000_000, generator=torch.Generator().manual_seed(0))[:200].cuda()
    d2 = torch.cdist(pts[idx].double(), pts.double()).square()
    d2[torch.arange(200), idx] = float("inf")
    ref = torch.topk(d2, 3, dim=1, largest=False).values.mean(dim=1)
    assert torch.allclose(d[idx].double(), ref, rtol=2e-5, atol=1e-9)


def test_model_from_colmap_points_renders():
    """COLMAP sparse points -> create_from_pcd (distCUDA2 on the HIP path) -> render(): the data-format side of the
    path end to end on the device."""
    import os
    from gsplat_attack.colmap import cameras_from_colmap, read_points3D_binary
    from gsplat_attack.gaussian_model import GaussianModel
    from gsplat_attack.renderer import PipelineParams, render
    here = os.path.dirname(os.path.abspath(__file__))
    xyz, rgb, _ = read_points3D_binary(os.path.join(here, "golden", "colmap_sample_bin", "sparse", "0", "points3D.bin"))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    pts = torch.cat([torch.from_numpy(xyz).float(), torch.randn(2000, 3, generator=g) * 0.5])
    cols = torch.cat([torch.from_numpy(rgb).float() / 255.0, torch.rand(2000, 3, generator=g)])
    model = GaussianModel.create_from_pcd(pts, cols, device=dev)
    cpu = GaussianModel.create_from_pcd(pts, cols, device="cpu")
    assert torch.allclose(model._scaling.detach().cpu(), cpu._scaling.detach(), atol=1e-4)
    cam = cameras_from_colmap(os.path.join(here, "golden", "colmap_sample_bin"), device=dev)[0]
    out = render(cam, model, PipelineParams(), torch.zeros(3, device=dev))
    out["render"].sum().backward()
    torch.cuda.synchronize()
    assert out["render"].shape == (3, cam.image_height, cam.image_width) and torch.isfinite(out["render"]).all()
    assert torch.isfinite(model._xyz.grad).all()
