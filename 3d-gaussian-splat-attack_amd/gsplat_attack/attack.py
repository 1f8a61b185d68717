"""DAGGER-style PGD loop over Gaussian attributes, on top of render().

Counterpart of the hot loop of the reference's ``attack.py:463-604`` restricted to what touches the raster path:
per iteration, render the current batch of views (``attack.py:476-485``), turn the renders into a scalar loss,
``loss.backward()`` (:494), apply a projected step to the raw attributes (:496-511; the live call is the L2 colour step
with alpha 0.5 / epsilon 5.0, ``configs/config.yaml:47-48``) and clear the gradients.  The victim detector is third
party and out of scope: a fixed random convolutional "surrogate detector" supplies a
differentiable scalar per render, or any callable ``loss_fn(renders[B,3,H,W]) -> scalar`` can be passed in.

Differences from the reference that are deliberate and documented in SURVEY.md section 3.1:
  * gradients are zeroed every iteration (the reference's optimizer.zero_grad is a no-op on the live tensors, so its
    .grad accumulates over iterations); the per-step gradient is what gets all-reduced in multi-GPU runs;
  * with torch.distributed initialised, the batch's views are sharded over ranks and the attribute gradients are
    sum-all-reduced once per iteration (gsplat_attack.dist) before the identical step on every rank.
"""
from __future__ import annotations

import argparse
import contextlib
import copy
import json
import time
from typing import Callable, Iterable, List, Optional, Sequence

import torch
from torch import nn

from . import dist as gdist
from . import pgd
from .streams import StreamRing
from .renderer import PipelineParams, render

GROUPS = ("color", "position", "scaling", "rotation", "opacity")


class SurrogateDetector(nn.Module):
    """Fixed random 3-layer conv net; the 'targeted loss' is the mean logit of one channel (to be minimised)."""

    def __init__(self, seed: int = 7, width: int = 16):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.net = nn.Sequential(nn.Conv2d(3, width, 5, stride=2, padding=2), nn.ReLU(),
                                 nn.Conv2d(width, width, 3, stride=2, padding=1), nn.ReLU(),
                                 nn.Conv2d(width, 4, 3, stride=2, padding=1))
        with torch.no_grad():
            for p in self.net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.05))
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, renders: torch.Tensor) -> torch.Tensor:
        return self.net(renders.clamp(0.0, 1.0))[:, 0].mean()


def _step(model, originals, groups: Sequence[str], norm: str, alpha: float, epsilon: float) -> None:
    fn = {("color", "l2"): lambda: pgd.gaussian_color_l2_attack(model, alpha, epsilon, originals["_features_rest"],
                                                                  originals["_features_dc"]),
          ("color", "linf"): lambda: pgd.gaussian_color_linf_attack(model, alpha, epsilon, originals["_features_rest"],
                                                                      originals["_features_dc"])}
    single = {"position": ("_xyz", "position"), "scaling": ("_scaling", "scaling"),
              "rotation": ("_rotation", "rotation"), "opacity": ("_opacity", "opacity")}
    for g in groups:
        if g == "color":
            fn[(g, norm)]()
        else:
            attr, name = single[g]
            getattr(pgd, f"gaussian_{name}_{norm}_attack")(model, alpha, epsilon, originals[attr])


def pgd_attack(model, cameras: Sequence, *, iters: int = 20, alpha: float = 0.5, epsilon: float = 5.0,
               groups: Iterable[str] = ("color",), norm: str = "l2", bg: Optional[torch.Tensor] = None,
               loss_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, pipe: Optional[PipelineParams] = None,
               log: Optional[Callable[[dict], None]] = None, streams: int = 3) -> List[float]:
    """Runs `iters` PGD iterations over the batch `cameras` (sharded over ranks when torch.distributed is
    initialised).  Returns the per-iteration global loss (sum over the batch's views).  The rank's views are
    pipelined over `streams` HIP streams (gsplat_attack.streams); 1 = the reference's strictly sequential order."""
    groups = tuple(groups)
    assert all(g in GROUPS for g in groups) and norm in ("l2", "linf")
    dev = model.get_xyz.device
    pipe = pipe or PipelineParams(skip_objects=True)
    bg = torch.zeros(3, device=dev) if bg is None else bg.to(dev)
    loss_fn = loss_fn or SurrogateDetector().to(dev)
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) \
        if torch.distributed.is_available() and torch.distributed.is_initialized() else (0, 1)
    mine = [cameras[i] for i in gdist.views_of_rank(len(cameras), rank, world)]
    originals = {n: getattr(model, n).detach().clone() for n in gdist.ATTACK_PARAMS}
    # A colour-only attack needs no geometry gradients: freeze those parameters for the duration of the attack and the
    # rasteriser's backward drops the geometry sums and the projection chain rule (the reference computes and discards
    # them: all seven tensors are re-wrapped with requires_grad=True).
    frozen = []
    if groups == ("color",):
        pipe = copy.copy(pipe)
        pipe.viewspace_grad = False
        for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
            p = getattr(model, n)
            if p.requires_grad:
                p.requires_grad_(False)
                frozen.append(p)
    try:
        history = []
        ring = StreamRing(min(streams, max(len(mine), 1)), dev) if dev.type == "cuda" else None
        for it in range(iters):
            t0 = time.perf_counter()
            model.zero_grad()
            losses = []
            for cam in mine:                                   # one forward+backward per view: peak memory = one view per stream
                with (ring.next() if ring is not None else contextlib.nullcontext()):
                    img = render(cam, model, pipe, bg)["render"]
                    loss = loss_fn(img[None])
                    loss.backward()
                    losses.append(loss.detach())
            if ring is not None:
                ring.join()
            total = torch.stack(losses).sum() if losses else torch.zeros((), device=dev)
            if world > 1:
                if frozen:
                    gdist.allreduce_attribute_grads(model, names=("_features_dc", "_features_rest"))   # 192 MB instead of 236
                else:
                    gdist.allreduce_attribute_grads(model)
                torch.distributed.all_reduce(total)
            _step(model, originals, groups, norm, alpha, epsilon)
            history.append(float(total))
            if log is not None:
                if dev.type == "cuda":
                    torch.cuda.synchronize()
                log({"iter": it, "loss": history[-1], "seconds": time.perf_counter() - t0, "views": len(cameras)})
    finally:
        for p in frozen:
            p.requires_grad_(True)
    return history


def combine_with_background(attacked, background):
    """The scene the reference evaluates after every step: the attacked target Gaussians followed by the frozen
    background (reference attack.py:513-520: deepcopy + seven concat_setup calls).  Here: one concatenation per
    attribute into a fresh model, no deep copy of the attacked one."""
    from .gaussian_model import GaussianModel
    cat = {k: torch.cat((a.detach(), b.detach().to(a.device)), dim=0)
           for k, a, b in ((n, getattr(attacked, n), getattr(background, n)) for n in GaussianModel._PARAM_ATTRS)}
    return GaussianModel.from_tensors(cat["_xyz"], cat["_features_dc"], cat["_features_rest"], cat["_scaling"],
                                      cat["_rotation"], cat["_opacity"], cat["_objects_dc"],
                                      sh_degree=attacked.max_sh_degree, device=cat["_xyz"].device, requires_grad=False)


@torch.no_grad()
def render_combined(attacked, background, cameras: Sequence, bg: torch.Tensor, pipe: Optional[PipelineParams] = None):
    """Forward-only renders of target + background for the success check (reference attack.py:522-530)."""
    pipe = pipe or PipelineParams(skip_objects=True)
    scene = combine_with_background(attacked, background)
    return [render(cam, scene, pipe, bg)["render"] for cam in cameras]


def main():
    ap = argparse.ArgumentParser(description="PGD over Gaussian attributes with a surrogate detector (synthetic scenes)")
    ap.add_argument("--scene", default="nyc-1M")
    ap.add_argument("--P", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--groups", default="color", help="comma list of " + ",".join(GROUPS))
    ap.add_argument("--norm", default="l2")
    ap.add_argument("--alpha", type=float, default=0.5)
    ap.add_argument("--epsilon", type=float, default=5.0)
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the rank's views are pipelined over")
    args = ap.parse_args()
    from .scenes import make_scene
    rank, world, local = gdist.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    model, cams, spec = make_scene(args.scene, device=dev, P=args.P, width=args.width, height=args.height,
                                   n_views=max(args.views, 1))
    recs = []
    hist = pgd_attack(model, cams[:args.views], iters=args.iters, alpha=args.alpha, epsilon=args.epsilon,
                      groups=args.groups.split(","), norm=args.norm, log=recs.append, streams=args.streams)
    if rank == 0:
        secs = [r["seconds"] for r in recs[2:]] or [r["seconds"] for r in recs]
        print(json.dumps({"scene": spec.name, "P": int(model.get_xyz.shape[0]), "views": args.views, "gpus": world,
                          "iters": args.iters, "groups": args.groups, "loss_first": hist[0], "loss_last": hist[-1],
                          "s_per_pgd_iter": sum(secs) / len(secs)}))


if __name__ == "__main__":
    main()
