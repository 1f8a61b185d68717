"""The CPU legs of the benchmark line: oracle-R (the checker, a CPU port -- never the product path) timed on the host cores.

Part of bench.py (the repo-root benchmark driver), split out in round 6: bench.py keeps the command line, the timed regions
of the headline metric and the assembly of the ONE JSON line; this module holds `cpu_baseline` and the config-1 parity + CPU time."""
from __future__ import annotations

import os
import time

import torch


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_P: int, sample_W: int, sample_H: int) -> dict:
    """oracle-R (the checker, a CPU port -- never the product path) timed on the host cores of this node."""
    from gsplat_attack.scenes import make_scene
    from oracle import oracle_r as O
    import math
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    avail, machine = cores, os.cpu_count() or cores
    cores = max(1, min(cores, 16))      # a 1-GPU box shares its host: 16 worker threads is this pool's CPU share
    torch.set_num_threads(cores)
    model, cams, _ = make_scene("nyc-1M", device="cpu", P=sample_P, width=sample_W, height=sample_H, n_views=1)
    cam = cams[0]
    st = O.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                    torch.zeros(3), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                    False, False)
    inp = dict(means3D=model.get_xyz.detach(), shs=model.get_features.detach(), opacities=model.get_opacity.detach(),
               scales=model.get_scaling.detach(), rotations=model.get_rotation.detach())
    gc = torch.randn(3, sample_H, sample_W, generator=torch.Generator().manual_seed(99))
    t0 = time.perf_counter()
    out, _ = O.forward_backward(inp, st, gc, dtype=torch.float32)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "views/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cores": machine, "host_cores_available_to_this_process": avail, "cpu_model": _cpu_model(),
            "seconds": dt,
            "sample": f"oracle-R float32 fwd+bwd, ONE view of S-nyc-1M subsampled to {sample_P} Gaussians at "
                      f"{sample_W}x{sample_H} (N={out.num_rendered} pairs); not extrapolated to 1M/1080p"}


def parity_and_cfg1(dev) -> dict:
    """Second half of BASELINE.json's metric, measured in the same job: the HIP path against oracle-R (float64, the
    checker) on S-hydrant-1k @128x128 (BASELINE config 1), plus oracle-R's float32 CPU time on that config (median of
    5 after one warm-up)."""
    import math
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene
    from oracle import oracle_r as O
    model, cams, _ = make_scene("hydrant-1k", device=dev, n_views=1)
    cam = cams[0]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(99))
    out = render(cam, model, PipelineParams(skip_objects=True), bg)
    out["render"].backward(gc.to(dev))
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu()
    st = O.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), cpu(bg), 1.0,
                    cpu(cam.world_view_transform), cpu(cam.full_proj_transform), 3, cpu(cam.camera_center), False, False)
    ref, _, _ = make_scene("hydrant-1k", device="cpu", n_views=1)
    ro = O.rasterize(ref.get_xyz, None, ref.get_opacity, st, shs=ref.get_features, scales=ref.get_scaling,
                     rotations=ref.get_rotation)
    (ro.color * gc.double()).sum().backward()
    solid = ~ro.fragile_px
    rgb = (cpu(out["render"]).double() - ro.color.detach()).abs().max(dim=0).values[solid].max().item()
    rel = 0.0
    for p_hip, p_ref in zip(model.parameters(), ref.parameters()):
        if p_ref.grad is None or p_hip.grad is None or p_ref.grad.abs().max().item() == 0.0:
            continue
        rel = max(rel, ((cpu(p_hip.grad).double() - p_ref.grad).abs().max() / p_ref.grad.abs().max()).item())
    inp = dict(means3D=ref.get_xyz.detach(), shs=ref.get_features.detach(), opacities=ref.get_opacity.detach(),
               scales=ref.get_scaling.detach(), rotations=ref.get_rotation.detach())
    st32 = O.Settings(st.image_height, st.image_width, st.tanfovx, st.tanfovy, st.bg.float(), 1.0, st.viewmatrix,
                      st.projmatrix, 3, st.campos, False, False)
    times = []
    for i in range(6):
        t0 = time.perf_counter()
        O.forward_backward(inp, st32, gc, dtype=torch.float32)
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[2]
    return {"parity": {"scene": "S-hydrant-1k 128x128 (BASELINE config 1) vs oracle-R float64",
                       "rgb_max_abs_err": rgb, "grad_max_rel_err": rel,
                       "tolerance": {"rgb_abs": 1e-4, "grad_rel": 1e-3}},
            "cfg1_cpu": {"value": 1.0 / med, "unit": "views/s", "seconds_median_of_5": med,
                         "sample": "oracle-R float32 fwd+bwd, S-hydrant-1k 128x128"}}
