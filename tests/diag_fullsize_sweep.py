"""Full-size image parity over MANY tile windows -- an evidence generator, not part of the timed suite.

    python tests/diag_fullsize_sweep.py <cfg3|cfg5|cfg5-noobj|cfg2> <tiles> [seed] [chunk_tiles]

Renders the whole image once with the HIP path and composites `tiles` tiles of it (seeded random 2x2 windows among the
non-empty tiles, `chunk_tiles` tiles per oracle call) with oracle-R in float64 AND float32 on the same inputs, windows and
depth keys.  Per chunk: window pixels, fragile share, worst solid-pixel error, pixels on neither clause of the float32
yardstick (tests/util.py::pixel_yardstick).  At the end: the totals and the WORST SOLID PIXEL with what explains it -- its
tile's list length, the pixel's last contributor and final T, the float32 oracle's own error at that pixel, and the
magnitude of what is summed there.  Appends to gpurun_out/r05_fullsize_sweep.txt.
"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splat-attack_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle_r as O  # noqa: E402
from util import settings_for, pixel_yardstick  # noqa: E402

CFG = {
    "cfg3": dict(key="nyc-1M", view=2, n_views=3, bg=(0.1, 0.2, 0.3), objects=False),
    "cfg5": dict(key="airport-4K", view=0, n_views=1, bg=(0.2, 0.1, 0.0), objects=True),
    "cfg5-noobj": dict(key="airport-4K", view=0, n_views=1, bg=(0.2, 0.1, 0.0), objects=False),
    "cfg2": dict(key="hydrant-full", view=0, n_views=1, bg=(0.0, 0.0, 0.0), objects=False),
}


def note(line):
    print(line, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "r05_fullsize_sweep.txt"), "a") as f:
            f.write(line + "\n")


def main():
    name = sys.argv[1]
    n_tiles = int(sys.argv[2])
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 48
    cfg = CFG[name]
    import diff_gaussian_rasterization as D
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene
    D._load()
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(cfg["key"], device=dev, n_views=cfg["n_views"])
    cam = cams[cfg["view"]]
    H, W = cam.image_height, cam.image_width
    gx, gy = (W + 15) // 16, (H + 15) // 16
    bg = torch.tensor(cfg["bg"])
    pipe = PipelineParams(skip_objects=not cfg["objects"])
    # (grad mode on: export_state reads the arrays of the context a differentiable forward keeps)
    out = render(cam, model, pipe, bg.to(dev))
    hip = out["render"].detach().cpu().double()
    hip_obj = out["render_object"].detach().cpu().double() if cfg["objects"] else None
    ranges = D.export_state(out["render"], "ranges").view(-1, 2).long().cpu()
    ncon = D.export_state(out["render"], "n_contrib").view(H, W).long().cpu()
    fT = D.export_state(out["render"], "final_T").view(H, W).cpu()
    radii = out["radii"].cpu()
    with D.extra_flags(D.FLAG_NO_CULL):
        full = render(cam, model, PipelineParams(skip_objects=True), bg.to(dev))
    depth = D.export_state(full["render"], "G").view(-1, 12)[:, 9].cpu()
    depth = torch.where(radii > 0, depth, torch.zeros_like(depth))
    del full, out
    lens = (ranges[:, 1] - ranges[:, 0]).view(gy, gx)
    g = torch.Generator().manual_seed(1000 + seed)
    nz = torch.nonzero(lens.flatten() > 0).flatten()
    picks = nz[torch.randperm(nz.numel(), generator=g)].tolist()
    wins, seen = [], set()
    for t in picks:
        ty, tx = t // gx, t % gx
        w = (tx, ty, min(tx + 2, gx), min(ty + 2, gy))
        tiles = {(x, y) for y in range(w[1], w[3]) for x in range(w[0], w[2])}
        if tiles & seen:
            continue
        seen |= tiles
        wins.append(w)
        if len(seen) >= n_tiles:
            break
    note(f"== {name} ({cfg['key']}, {W}x{H}, {gx * gy} tiles, objects {'on' if cfg['objects'] else 'off'}), seed {seed}: "
         f"{len(seen)} tiles = {100.0 * len(seen) / (gx * gy):.2f} % of the image in {len(wins)} windows ==")

    ref, rcams, _ = make_scene(cfg["key"], device="cpu", n_views=cfg["n_views"])
    st = settings_for(rcams[cfg["view"]], bg, 3, 1.0)
    O.check_depth_keys(depth, ref.get_xyz, st, radii)

    def oracle(ws, dtype):
        with torch.no_grad():
            return O.rasterize(ref.get_xyz.detach(), None, ref.get_opacity.detach(), st, shs=ref.get_features.detach(),
                               sh_objs=ref.get_objects.detach() if cfg["objects"] else None,
                               scales=ref.get_scaling.detach(), rotations=ref.get_rotation.detach(), tile_windows=ws,
                               depth_key=depth, dtype=dtype)
    tot = dict(px=0, frag=0, neither=0, neither_solid=0, need_b=0)
    solid_hip, solid_r32, solid_x = [], [], []       # per solid pixel: |HIP - r64|, |r32 - r64|, its x coordinate
    worst = None
    worst_obj = 0.0
    per_win = 4
    step = max(1, chunk // per_win)
    t0 = time.time()
    for c0 in range(0, len(wins), step):
        ws = wins[c0:c0 + step]
        r64, r32 = oracle(ws, torch.float64), oracle(ws, torch.float32)
        m = r64.window_px
        y = pixel_yardstick(hip, r64.color, r32.color, r64.fragile_px, mask=m, tol=1e-4)
        n = y["n"]
        tot["px"] += n
        tot["frag"] += int(round(y["fragile"] * n))
        tot["neither"] += y["neither_px"]
        tot["neither_solid"] += y["neither_solid"]
        tot["need_b"] += int(round(y["need_b"] * n))
        solid = m & ~r64.fragile_px
        e64 = y["e64"]
        solid_hip.append(e64[solid])
        solid_r32.append((r32.color - r64.color).abs().amax(dim=0)[solid])
        solid_x.append(torch.nonzero(solid)[:, 1])
        if cfg["objects"]:
            eo = (hip_obj - r64.objects).abs().amax(dim=0)
            worst_obj = max(worst_obj, float(eo[solid].max()) if bool(solid.any()) else 0.0)
        if bool(solid.any()):
            es = torch.where(solid, e64, torch.zeros_like(e64))
            idx = int(es.argmax())
            py, px = idx // W, idx % W
            if worst is None or float(es.flatten()[idx]) > worst["err"]:
                d32 = float((r32.color[:, py, px] - r64.color[:, py, px]).abs().max())
                tile = (py // 16) * gx + px // 16
                worst = dict(err=float(es.flatten()[idx]), px=(px, py), tile=(px // 16, py // 16), list_len=int(lens.flatten()[tile]),
                             n_contrib=int(ncon[py, px]), final_T=float(fT[py, px]), f32_oracle_err=d32,
                             colour64=[float(v) for v in r64.color[:, py, px]], hip=[float(v) for v in hip[:, py, px]],
                             r32=[float(v) for v in r32.color[:, py, px]], oracle_n_contrib=int(r64.n_contrib[py, px]))
        note(f"[{name} chunk {c0 // step}] {len(ws)} windows, px {n}, fragile {y['fragile']:.4f}, solid err {y['worst_solid']:.2e}, "
             f"worst err {y['worst_any']:.2e}, neither: {y['neither_px']} fragile px (worst {y['worst_neither']:.2e}) + "
             f"{y['neither_solid']} solid px, clause B {y['need_b']:.5f}, oracle f32 vs f64 {y['f32_vs_f64']:.2e}  "
             f"[{time.time() - t0:.0f} s]")
    note(f"== {name} total: {tot['px']} px in {len(seen)} tiles, fragile {tot['frag'] / max(tot['px'], 1):.4f}, fragile px on neither "
         f"clause {tot['neither']} ({tot['neither'] / max(tot['frag'], 1):.5f} of the fragile), solid px on neither {tot['neither_solid']}, "
         f"clause B {tot['need_b']}, worst solid err {worst['err']:.2e}" + (f", worst solid objects err {worst_obj:.2e}" if cfg["objects"] else ""))
    note(f"== {name} worst solid pixel: {worst}")
    # The float32 yardstick as a DISTRIBUTION over the solid pixels: a float32 implementation's rounding is independent of the
    # float32 oracle's at any one pixel, so the tails are compared, not the pixels.  (At 4K the pixel centre itself is the
    # limit: one float32 ulp of a coordinate beyond 2048 is 2.4e-4 px, and d ln(alpha) / d centre of a one-pixel splat is O(1).)
    eh, er, xs = torch.cat(solid_hip), torch.cat(solid_r32), torch.cat(solid_x)
    def q(t, p):
        return float(torch.quantile(t, p)) if t.numel() else 0.0
    for label, sel in (("all solid px", torch.ones_like(xs, dtype=torch.bool)), ("solid px with x < 2048", xs < 2048), ("solid px with x >= 2048", xs >= 2048)):
        if int(sel.sum()) == 0:
            continue
        a, b = eh[sel], er[sel]
        note(f"== {name} {label} ({int(sel.sum())}): |HIP - r64| max {float(a.max()):.2e}, p99.99 {q(a, 0.9999):.2e}, p99.9 {q(a, 0.999):.2e}, "
             f"median {q(a, 0.5):.2e}, above 1e-4: {int((a > 1e-4).sum())}   |  float32 oracle |r32 - r64| max {float(b.max()):.2e}, "
             f"p99.99 {q(b, 0.9999):.2e}, p99.9 {q(b, 0.999):.2e}, median {q(b, 0.5):.2e}, above 1e-4: {int((b > 1e-4).sum())}")
    # what is summed at that pixel: the oracle's float64 weights alpha_i T_i c_i of its tile, and the error float32 makes of them
    px, py = worst["px"]
    tx, ty = worst["tile"]
    r64 = oracle([(tx, ty, tx + 1, ty + 1)], torch.float64)
    r32 = oracle([(tx, ty, tx + 1, ty + 1)], torch.float32)
    e_tile = (hip - r64.color).abs().amax(dim=0)[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    d_tile = (r32.color - r64.color).abs().amax(dim=0)[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    sol = ~r64.fragile_px[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    note(f"== {name} that pixel's tile ({tx},{ty}): list {worst['list_len']}, solid px {int(sol.sum())}; HIP err over its solid px: max "
         f"{float(e_tile[sol].max()):.2e}, median {float(e_tile[sol].median()):.2e}; float32 oracle err over them: max "
         f"{float(d_tile[sol].max()):.2e}, median {float(d_tile[sol].median()):.2e}; |colour| max {float(r64.color[:, ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].abs().max()):.2f}")


if __name__ == "__main__":
    main()
