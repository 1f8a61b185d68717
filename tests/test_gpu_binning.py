"""The binning stage as integers: tile lists against oracle_r.build_tile_lists, the depth order against a stable argsort for
every shape of depth distribution, scan chunks, empty and nearly empty scenes, the asynchronous pair count."""
import math
import pytest
import torch
from oracle import oracle_r as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _hip():
    import diff_gaussian_rasterization as D
    D._load()
    return D


def _tile_list_parity(scene, n_views=1, windows=None, **kw):
    D = _hip()
    from gsplat_attack.renderer import PipelineParams, render
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene(scene, device=dev, n_views=n_views, **kw)
    cam = cams[-1]
    H, W = cam.image_height, cam.image_width
    gx = (W + 15) // 16
    bg = torch.zeros(3, device=dev)
    with D.extra_flags(D.FLAG_NO_CULL):
        out = render(cam, model, PipelineParams(skip_objects=True), bg)
    img = out["render"]
    ranges = D.export_state(img, "ranges").view(-1, 2).long().cpu()
    pairs = D.export_state(img, "pair_rank").long().cpu() & 0xFFFFFFFF   # exported as int32
    gids = pairs & ((1 << 28) - 1)
    assert int((pairs >> 28).min()) == 15                  # no cull: every strip bit set
    N = D.last_num_rendered(img)
    depth = D.export_state(img, "G").view(-1, 12)[:, 9].cpu()
    radii = out["radii"].cpu()
    depth = torch.where(radii > 0, depth, torch.zeros_like(depth))
    cpu = lambda t: t.detach().cpu()
    st = O.Settings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), 1.0,
                    cpu(cam.world_view_transform), cpu(cam.full_proj_transform), 3, cpu(cam.camera_center), False, False)
    xyz, sc, ro = cpu(model.get_xyz).double(), cpu(model.get_scaling).double(), cpu(model.get_rotation).double()
    O.check_depth_keys(depth, xyz, st, radii)
    g = O.preprocess(xyz, sc, ro, None, st)
    frag = g.fragile
    bad_r = (radii != g.radii) & ~frag
    assert int(bad_r.sum()) == 0
    gid_o, ranges_o, _ = O.build_tile_lists(g, H, W, tile_windows=windows, depth_key=depth)
    if windows is None:
        tiles = torch.arange(ranges.shape[0])
    else:
        tiles = torch.tensor([ty * gx + tx for (x0, y0, x1, y1) in windows for ty in range(y0, y1) for tx in range(x0, x1)])
    n_frag_h = n_frag_o = 0
    for t in tiles.tolist():
        lh = gids[ranges[t, 0]:ranges[t, 1]]
        lo = gid_o[ranges_o[t, 0]:ranges_o[t, 1]]
        kh, ko = ~frag[lh], ~frag[lo]
        n_frag_h += int((~kh).sum())
        n_frag_o += int((~ko).sum())
        assert torch.equal(lh[kh], lo[ko]), f"tile {t}: list differs ({lh[kh][:8].tolist()} vs {lo[ko][:8].tolist()})"
    if windows is None:
        assert int((ranges[:, 1] - ranges[:, 0]).sum()) == N == int(pairs.numel())
        assert N - n_frag_h == int(gid_o.numel()) - n_frag_o
        if int(frag.sum()) == 0:
            lens, lens_o = ranges[:, 1] - ranges[:, 0], ranges_o[:, 1] - ranges_o[:, 0]
            nz = lens_o > 0                                # (an empty tile's span is (0, 0) here, (start, start) there)
            assert torch.equal(lens, lens_o) and torch.equal(ranges[nz], ranges_o[nz])
            assert torch.equal(gids, gid_o) and N == int(gid_o.numel())
    return int(frag.sum()), N


def _raw_render(D, dev, xyz, scale_log, opac_logit, W=160, H=96, bg=(0.2, 0.4, 0.6), flags=0, grad=True):
    from gsplat_attack.cameras import Camera
    from gsplat_attack.gaussian_model import GaussianModel
    from gsplat_attack.renderer import PipelineParams, render
    P = xyz.shape[0]
    g = torch.Generator().manual_seed(P % 1000 + 3)
    model = GaussianModel.from_tensors(xyz, torch.randn(P, 1, 3, generator=g) * 0.5, torch.randn(P, 15, 3, generator=g) * 0.1,
                                       scale_log, torch.randn(P, 4, generator=g), opac_logit, torch.zeros(P, 1, 16),
                                       sh_degree=3, device=dev, requires_grad=grad)
    cam = Camera(R=torch.eye(3).numpy(), T=torch.zeros(3).numpy(), FoVx=1.0, FoVy=2 * math.atan(math.tan(0.5) * H / W),
                 width=W, height=H, device=dev)
    with D.extra_flags(flags):
        out = render(cam, model, PipelineParams(skip_objects=True), torch.tensor(bg, device=dev))
    return model, cam, out


def _check_lists_against_stable_argsort(D, out):
    """Every tile's list == the Gaussians of that tile in the order of a STABLE argsort of the float32 depths of all visible
    Gaussians (what the reference's one 64-bit (tile | depth) sort of storage-ordered pairs gives), position by position."""
    from util import check_tile_lists_depth_order
    img = out["render"]
    ranges, g = check_tile_lists_depth_order(D, img)
    depth = D.export_state(img, "G").view(-1, 12)[:, 9]
    vis = out["radii"] > 0
    ids = torch.nonzero(vis).flatten()
    order = ids[torch.argsort(depth[ids], stable=True)]                 # global stable depth order
    rank = torch.full((vis.numel(),), -1, dtype=torch.long, device=img.device)
    rank[order] = torch.arange(order.numel(), device=img.device)
    r = rank[g]
    assert int(r.min()) >= 0
    lens = (ranges[:, 1] - ranges[:, 0])
    tile_of = torch.repeat_interleave(torch.arange(ranges.shape[0], device=img.device), lens)
    pos = torch.cat([torch.arange(int(a), int(b), device=img.device) for a, b in ranges[lens > 0].tolist()]) if int(lens.sum()) else r[:0]
    rr = r[pos]
    same = tile_of[1:] == tile_of[:-1]
    assert bool((rr[1:] > rr[:-1])[same].all())                         # strictly increasing global rank inside every tile
    assert int(lens.sum()) > 0


def _small_scene(n_views=3, P=20000, w=320, h=192):
    from gsplat_attack.scenes import make_scene
    dev = torch.device("cuda:0")
    model, cams, _ = make_scene("nyc-1M", device=dev, P=P, width=w, height=h, n_views=n_views)
    return dev, model, cams


def test_tile_lists_equal_the_oracles_hydrant_1k():
    nfrag, N = _tile_list_parity("hydrant-1k")
    print(f"hydrant-1k: N={N}, fragile Gaussians {nfrag}")
    assert N > 1000


def test_tile_lists_equal_the_oracles_60k_at_640x360():
    nfrag, N = _tile_list_parity("nyc-1M", P=60000, width=640, height=360)
    print(f"nyc 60k @640x360: N={N}, fragile Gaussians {nfrag}")
    assert N > 50000


def test_tile_lists_equal_the_oracles_nyc_1m_on_windows():
    # full size (1 M Gaussians, 1080p): the oracle builds the lists of three 6x4-tile windows
    windows = [(10, 10, 16, 14), (57, 30, 63, 34), (114, 64, 120, 68)]
    nfrag, N = _tile_list_parity("nyc-1M", n_views=3, windows=windows)
    print(f"nyc-1M @1080p: N={N}, fragile Gaussians {nfrag}")
    assert N > 3_000_000


def test_no_gaussian_in_front_of_the_camera_renders_the_background():
    D = _hip()
    dev = torch.device("cuda:0")
    P = 5000
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(P, 3, generator=g)
    xyz[:, 2] = -xyz[:, 2].abs() - 0.5                      # all behind the camera
    model, cam, out = _raw_render(D, dev, xyz, torch.full((P, 3), -3.0), torch.zeros(P, 1))
    assert int(out["radii"].max()) == 0 and D.last_num_rendered(out["render"]) == 0
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    assert torch.equal(out["render"], bg[:, None, None].expand_as(out["render"]))
    out["render"].sum().backward()
    for p in model.parameters():
        assert p.grad is None or float(p.grad.abs().max()) == 0.0


def test_one_visible_gaussian_among_culled_ones():
    D = _hip()
    dev = torch.device("cuda:0")
    P = 3000
    xyz = torch.zeros(P, 3)
    xyz[:, 2] = -1.0
    xyz[1234] = torch.tensor([0.0, 0.0, 3.0])
    model, cam, out = _raw_render(D, dev, xyz, torch.full((P, 3), -2.0), torch.full((P, 1), 2.0))
    radii = out["radii"]
    assert int((radii > 0).sum()) == 1 and int(radii[1234]) > 0
    dv = D.export_state(out["render"], "dv")
    assert int(dv[1]) == 1 and set((D.export_state(out["render"], "pair_rank") & ((1 << 28) - 1)).tolist()) == {1234}
    n = D.last_num_rendered(out["render"])
    assert n >= 1 and float((out["render"].detach() - torch.tensor([0.2, 0.4, 0.6], device=dev)[:, None, None]).abs().max()) > 0.01
    out["render"].sum().backward()
    gx = model._xyz.grad.detach().clone()
    assert float(gx[1234].abs().max()) > 0
    gx[1234] = 0
    assert float(gx.abs().max()) == 0.0


def test_depth_range_that_needs_the_widest_digits_sorts_exactly():
    """View depths from 0.25 to 3e5: max - min of the float keys spans 30+ bits, so the three depth passes use 10/11-bit
    digits (the benchmark scene: 9).  Every tile's list must be in the order of the stable argsort of the float32 depth."""
    D = _hip()
    dev = torch.device("cuda:0")
    P = 40000
    g = torch.Generator().manual_seed(7)
    z = torch.exp(torch.rand(P, generator=g) * (math.log(3e5) - math.log(0.25)) + math.log(0.25))
    z[::7] = z[3]                                           # exact ties: stability
    xy = (torch.rand(P, 2, generator=g) - 0.5) * 0.6
    xyz = torch.cat([xy * z[:, None], z[:, None]], dim=1)
    scale = torch.log(0.01 * z)[:, None].expand(P, 3).contiguous()      # ~1.7 px on screen whatever the depth
    model, cam, out = _raw_render(D, dev, xyz, scale, torch.zeros(P, 1), flags=D.FLAG_NO_CULL)
    dv = D.export_state(out["render"], "dv")
    assert int(dv[3]) >= 10, f"digit width {int(dv[3])}"
    order = D.export_state(out["render"], "order").long()
    depth = D.export_state(out["render"], "G").view(-1, 12)[:, 9]
    ids = torch.nonzero(out["radii"] > 0).flatten()
    assert torch.equal(order, ids[torch.argsort(depth[ids], stable=True)])
    _check_lists_against_stable_argsort(D, out)


@pytest.mark.parametrize("dist", ["constant", "two-values", "cluster+outliers", "shell-16-buckets", "shell-40-buckets", "narrow",
                                  "uniform", "log-uniform"])
def test_depth_order_is_the_stable_argsort_whatever_the_depth_distribution(dist):
    """The depth sort (three counting passes whose digit width follows the keys' range) on distributions that stress a
    sort by depth: every key equal, two values, 95 % of the Gaussians in a shell 1e-3 thick with outliers stretching the
    key range, shells a few thousand keys deep per 2^16 float steps, a range narrower than one digit, and spread-out
    ones.  Ties everywhere: the order must be the STABLE argsort of the float32 depths.  (These are the cases a bucketed
    variant of the sort -- EXPERIMENTS.md, round 3 -- was developed against; they hold for any implementation.)"""
    D = _hip()
    dev = torch.device("cuda:0")
    P = 60000
    g = torch.Generator().manual_seed(11)
    u = torch.rand(P, generator=g)
    if dist == "constant":
        z = torch.full((P,), 5.0)
    elif dist == "two-values":
        z = torch.where(u < 0.3, torch.tensor(2.0), torch.tensor(7.5))
    elif dist == "cluster+outliers":
        z = 10.0 + (u - 0.5) * 1e-3
        z[::20] = torch.exp(torch.rand(P // 20, generator=g) * math.log(400.0)) * 0.3      # 0.3 .. 120
    elif dist.startswith("shell"):
        # outliers 0.3 .. 120 make the key range ~27 bits; the shell is 16 (or 40) runs of 2^16 float32 steps deep
        z = 8.0 + u * (1.0 if dist == "shell-16-buckets" else 2.5)
        z[::20] = torch.exp(torch.rand(P // 20, generator=g) * math.log(400.0)) * 0.3
    elif dist == "narrow":
        z = 5.0 + torch.floor(u * 700.0) * 4.76837158203125e-07                           # 700 adjacent float32 values
    elif dist == "uniform":
        z = 1.0 + u * 30.0
    else:
        z = torch.exp(u * math.log(1000.0)) * 0.3
    z[::9] = z[4]                                           # exact ties: stability
    xy = (torch.rand(P, 2, generator=g) - 0.5) * 0.6
    xyz = torch.cat([xy * z[:, None], z[:, None]], dim=1)
    scale = torch.log(0.01 * z)[:, None].expand(P, 3).contiguous()
    model, cam, out = _raw_render(D, dev, xyz, scale, torch.zeros(P, 1), flags=D.FLAG_NO_CULL, grad=False)
    assert int((out["radii"] > 0).sum()) > P // 2
    _check_lists_against_stable_argsort(D, out)


def test_more_gaussians_than_the_small_scan_chunk_handles():
    """P > 2^21: the rank-order scan runs its 4096-element variant (and records the emission chunks' owners from it);
    the default path must give the bits of the full-rect path."""
    D = _hip()
    dev = torch.device("cuda:0")
    P = (1 << 21) + 70001
    g = torch.Generator().manual_seed(2)
    xyz = torch.randn(P, 3, generator=g) * torch.tensor([1.5, 0.9, 1.0]) + torch.tensor([0.0, 0.0, 4.0])
    scale = torch.full((P, 3), math.log(0.004)) + torch.randn(P, 3, generator=g) * 0.3
    opac = torch.randn(P, 1, generator=g)
    a = _raw_render(D, dev, xyz, scale, opac, W=256, H=160, grad=False)[2]
    b = _raw_render(D, dev, xyz, scale, opac, W=256, H=160, grad=False, flags=D.FLAG_NO_CULL)[2]
    assert int((a["radii"] > 0).sum()) > 100000
    assert torch.equal(a["radii"], b["radii"]) and torch.equal(a["render"], b["render"])


def test_asynchronous_pair_count_matches_and_reports_overflow():
    """GSR_FLAG_ASYNC_COUNT: same bits as the synchronous forward when the capacity guess holds; a scene that emits more
    pairs than the guess gets a NaN image and a PairCapacityExceeded from backward; the next forward recovers."""
    D = _hip()
    from gsplat_attack.renderer import PipelineParams, render
    dev, model, cams = _small_scene(n_views=2, P=50000, w=640, h=360)
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    gc = torch.randn(3, 360, 640, generator=torch.Generator().manual_seed(5)).to(dev)
    pipe = PipelineParams(skip_objects=True)

    def run(scale=1.0):
        model.zero_grad()
        out = render(cams[0], model, pipe, bg, scale)
        out["render"].backward(gc)
        return out["render"].detach().clone(), {n: getattr(model, n).grad.clone() for n in D.GradBucket.NAMES}, D.last_num_rendered(out["render"])
    img0, g0, n0 = run()                                   # synchronous: seeds the capacity table for this (P, H, W)
    try:
        D.set_flags(D.FLAG_ASYNC_COUNT)
        img1, g1, n1 = run()
        assert n1 == n0 and torch.equal(img0, img1)
        for n in g0:
            assert torch.equal(g0[n], g1[n]), n
        # three times larger splats: far more pairs than 1.25 x n0 + 64K
        model.zero_grad()
        out = render(cams[0], model, pipe, bg, 4.0)
        assert torch.isnan(out["render"]).all()
        with pytest.raises(D.PairCapacityExceeded):
            out["render"].backward(gc)
        img2, g2, n2 = run(4.0)                            # counted synchronously again: correct
        assert n2 > 1.25 * n0 + 65536 and torch.isfinite(img2).all()
        img3, _, n3 = run(4.0)                             # and asynchronously with the new capacity
        assert n3 == n2 and torch.equal(img2, img3)
    finally:
        D.set_flags(0)
