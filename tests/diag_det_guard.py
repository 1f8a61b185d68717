"""Diagnostic: what the published backward's 1 / (det^2 + 1e-7) (gsr_math.h GSR_DET_GUARD, oracle_r.DET_GUARD) changes against
the exact derivative 1 / det^2 of the 2D covariance inversion, on full-size scenes: the attribute gradients of one view
from the default library and from a build with -DGSR_DET_GUARD=0.0f (libgsraster_exactdet.so next to the default one:
`make -C 3d-gaussian-splat-attack_amd/csrc exactdet`).

    python tests/diag_det_guard.py [scene ...]          (default: nyc-1M airport-4K)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splat-attack_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def grads(scene):
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(scene, device=dev, n_views=1)
    cam = cams[0]
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(99)).to(dev)
    out = render(cam, model, PipelineParams(skip_objects=True), torch.zeros(3, device=dev))
    out["render"].backward(gc)
    torch.cuda.synchronize()
    g = {n: getattr(model, n).grad.detach().cpu() for n in NAMES}
    g["_viewspace"] = out["viewspace_points"].grad.detach().cpu()
    return g, D.library_path()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        g, path = grads(sys.argv[2])
        torch.save({"g": g, "lib": path}, sys.argv[3])
        return
    scenes = sys.argv[1:] or ["nyc-1M", "airport-4K"]
    import diff_gaussian_rasterization as D
    exact = os.path.join(os.path.dirname(D.library_path()), "libgsraster_exactdet.so")
    if not os.path.exists(exact):
        raise SystemExit(f"{exact} is missing (see the module docstring)")
    for scene in scenes:
        g0, lib0 = grads(scene)
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "g.pt")
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", scene, f], check=True,
                           env=dict(os.environ, GSR_LIBRARY=exact))
            d = torch.load(f)
        g1 = d["g"]
        print(f"{scene}: default library (det^2 + 1e-7) against {os.path.basename(d['lib'])} (det^2), one view, dL/dC ~ N(0,1)")
        for n in list(NAMES) + ["_viewspace"]:
            a, b = g0[n].double(), g1[n].double()
            scale = b.abs().max().item()
            diff = (a - b).abs()
            rel_el = (diff / b.abs().clamp_min(1e-30))[b.abs() > 1e-3 * scale]
            print(f"   {n:15s} max|d| / max|g| = {diff.max().item() / max(scale, 1e-30):.3e}   worst element-wise relative "
                  f"(elements above 1e-3 of the largest) = {rel_el.max().item() if rel_el.numel() else 0.0:.3e}")


if __name__ == "__main__":
    main()
