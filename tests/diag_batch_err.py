"""Diagnostic: where a batch's attribute gradients differ most from the single-view loop's (per tensor: the element, its
per-view values, both sums)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splat-attack_amd")); sys.path.insert(0, ROOT)
import torch
import diff_gaussian_rasterization as D
from gsplat_attack.renderer import PipelineParams, render, render_batch
from gsplat_attack.scenes import make_scene

P, W, H, B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (60000, 640, 360, 4)))
dev = torch.device("cuda:0")
model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=W, height=H, n_views=B)
g = torch.Generator().manual_seed(P)
gcs = [torch.randn(3, H, W, generator=g).to(dev) for _ in cams]
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
per = []
for v, c in enumerate(cams):
    b = D.GradBucket(P, dev)
    render(c, model, PipelineParams(skip_objects=True, grad_bucket=b), bg)["render"].backward(gcs[v])
    per.append(b.flat.clone())
loop = D.GradBucket(P, dev)
for v, c in enumerate(cams):
    render(c, model, PipelineParams(skip_objects=True, grad_bucket=loop), bg)["render"].backward(gcs[v])
bat = D.GradBucket(P, dev)
render_batch(cams, model, PipelineParams(skip_objects=True, grad_bucket=bat), bg)["render"].backward(torch.stack(gcs))
torch.cuda.synchronize()
exact = sum(p.double() for p in per)
Wd = (3, 3, 45, 1, 3, 4)
for name, c0, c1, w in zip(bat.NAMES, bat.CUTS[:-1], bat.CUTS[1:], Wd):
    sl = slice(c0 * P, c1 * P)
    e = (bat.flat[sl].double() - exact[sl]).abs()
    i = int(e.argmax())
    gi = i // w
    print(f"{name}: scale {exact[sl].abs().max().item():.4e} worst err {e[i].item():.3e} at elem {i} (Gaussian {gi}, comp {i % w}); "
          f"loop err there {abs(loop.flat[sl][i].item() - exact[sl][i].item()):.3e}")
    print("   per view:", [f"{p[sl][i].item():.6e}" for p in per], " exact", f"{exact[sl][i].item():.8e}", " batch", f"{bat.flat[sl][i].item():.8e}",
          " loop", f"{loop.flat[sl][i].item():.8e}")
    if name == "_scaling":
        print("   raw scaling", model._scaling[gi].tolist(), "rot", model._rotation[gi].tolist())
        for nm, cc0, ww in (("_rotation", 55, 4), ("_xyz", 0, 3)):
            s2 = slice(cc0 * P + gi * ww, cc0 * P + gi * ww + ww)
            print("   ", nm, "batch", bat.flat[s2].tolist(), "exact", exact[s2].tolist())
