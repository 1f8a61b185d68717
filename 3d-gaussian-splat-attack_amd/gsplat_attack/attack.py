"""DAGGER-style PGD loop over Gaussian attributes, on top of render().

Counterpart of the hot loop of the reference's ``attack.py:463-604`` restricted to what touches the raster path:
per iteration, render the current batch of views (``attack.py:476-485``), turn the renders into a scalar loss,
``loss.backward()`` (:494), apply a projected step to the raw attributes (:496-511; the live call is the L2 colour step
with alpha 0.5 / epsilon 5.0, ``configs/config.yaml:47-48``) and clear the gradients.  The victim detector is third
party and out of scope: a fixed random convolutional "surrogate detector" supplies a
differentiable scalar per render, or any callable ``loss_fn(renders[B,3,H,W]) -> scalar`` can be passed in.

Defaults that differ from the reference, each with a switch that restores its behaviour:
  * gradients are zeroed every iteration; ``accumulate_grads=True`` reproduces the reference, whose
    optimizer.zero_grad is a no-op on the live tensors so that .grad accumulates over iterations (attack.py:602-604);
  * one loss and one backward per view; ``batch_loss=True`` stacks the B renders and calls the loss once, with all B
    rasteriser contexts alive through one backward (attack.py:476-494);
  * the success check after each step (attack.py:513-569) renders target + frozen background WITHOUT the deep copy and
    the seven concatenations (``render_pair``); ``run_attack`` is the batch schedule around it;
  * with torch.distributed initialised, the batch's views are sharded over ranks and the per-step attribute gradients
    are sum-all-reduced once per iteration (gsplat_attack.dist) before the identical step on every rank.
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
from .renderer import (PipelineParams, _has_raw_layout, can_batch, render, render_batch, render_pair, render_pair_batch,
                       takes_fused_path)

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


MULTI_STEP = True       # the stepped tensors of an iteration in one launch (gsr_pgd_step_multi); False: tensor by tensor (A/B, tests)


def _step(model, originals, groups: Sequence[str], norm: str, alpha: float, epsilon: float, norms=None) -> None:
    """norms: a diff_gaussian_rasterization.GradNorms whose sums of squares -- when still valid for this iteration's
    gradients -- spare the L2 rules their own pass over the gradient (same rule, attack.py:53-119, 138-173)."""
    attrs = {"color": ("_features_rest", "_features_dc"), "position": ("_xyz",), "scaling": ("_scaling",),
             "rotation": ("_rotation",), "opacity": ("_opacity",)}
    todo = [a for g in groups for a in attrs[g]]
    have = all(getattr(model, a).grad is not None for a in todo)
    if norm == "l2" and norms is not None and have and all(norms.sumsq_of(a) is not None for a in todo):
        # the raster backward left ||grad||^2 of every tensor on the device: one launch for all of them
        if MULTI_STEP and pgd.multi_step_([(getattr(model, a), getattr(model, a).grad, originals[a], norms.sumsq_of(a))
                                           for a in todo], alpha, epsilon, True):
            return
        for a in todo:
            t = getattr(model, a)
            pgd.l2_step_(t, t.grad, alpha, epsilon, originals[a], sumsq=norms.sumsq_of(a))
        return
    if have and len(todo) > 1 and MULTI_STEP:
        # every stepped tensor has a gradient (the reference's own condition for stepping a group, attack.py:496): the
        # same rules on all of them in one launch (an L2 norm that is not on the device yet is summed in front)
        if pgd.multi_step_([(getattr(model, a), getattr(model, a).grad, originals[a], None) for a in todo], alpha, epsilon,
                           norm == "l2"):
            return
    fn = {("color", "l2"): lambda: pgd.gaussian_color_l2_attack(model, alpha, epsilon, originals["_features_rest"],
                                                                  originals["_features_dc"]),
          ("color", "linf"): lambda: pgd.gaussian_color_linf_attack(model, alpha, epsilon, originals["_features_rest"],
                                                                      originals["_features_dc"])}
    single = {"position": ("_xyz", "position"), "scaling": ("_scaling", "scaling"),
              "rotation": ("_rotation", "rotation"), "opacity": ("_opacity", "opacity")}
    for g in groups:
        if g == "color":
            # the reference steps only when both colour gradients exist (attack.py:496)
            if model._features_rest.grad is None or model._features_dc.grad is None:
                continue
            fn[(g, norm)]()
        else:
            attr, name = single[g]
            if getattr(model, attr).grad is None:      # no gradient reached this group (e.g. a rank without views)
                continue
            getattr(pgd, f"gaussian_{name}_{norm}_attack")(model, alpha, epsilon, originals[attr])


class PhaseTimer:
    """Optional per-phase GPU timing of pgd_attack (bench.py's `pgd` block): HIP events on the current stream at the
    phase boundaries of every iteration, read back once at the end.  Meaningful with streams=1 (one stream, phases do
    not overlap).  Phases: render (rasteriser forward), loss (detector forward), backward (detector backward +
    rasteriser backward), reduce (bucket fold / all-reduce / .grad assignment), step, rerender."""
    PHASES = ("render", "loss", "backward", "reduce", "step", "rerender")

    def __init__(self):
        self.marks = []          # (phase, start event, end event)
        self._t = None

    def start(self):
        self._t = torch.cuda.Event(enable_timing=True)
        self._t.record()

    def lap(self, phase: str):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((phase, self._t, e))
        self._t = e

    def totals_ms(self) -> dict:
        torch.cuda.synchronize()
        out = {p: 0.0 for p in self.PHASES}
        for ph, a, b in self.marks:
            out[ph] += a.elapsed_time(b)
        return out


_CHECK_STREAMS = {}


def _check_stream(dev):
    """One extra stream per device for the success re-renders of pgd_attack (kept: libgsraster's workspace blocks stay
    with the stream that used them last)."""
    key = str(dev)
    if key not in _CHECK_STREAMS:
        _CHECK_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _CHECK_STREAMS[key]


def _world():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def gather_success(flags: Sequence[bool], n_views: int, rank: int, world: int, device) -> List[bool]:
    """The batch's per-view success flags on every rank (reference attack.py:556-560 counts them on one GPU): rank r
    owns views r, r+world, ...; one tiny all-reduce of a B-element vector."""
    full = torch.zeros(n_views, dtype=torch.int32, device=device)
    idx = gdist.views_of_rank(n_views, rank, world)
    if idx:
        full[torch.tensor(idx, device=device)] = torch.tensor([int(bool(f)) for f in flags], dtype=torch.int32, device=device)
    if world > 1:
        if device.type == "cuda" and torch.distributed.get_backend() == "gloo":
            full = full.cpu()
        torch.distributed.all_reduce(full)
    return [bool(v) for v in full.tolist()]


def batch_done(flags: Sequence[bool], n_views: int) -> bool:
    """The stopping rule of a batch, in ONE place for pgd_attack and run_attack: all views of the batch fooled, or all
    but one (reference attack.py:560, `num_successes >= len(batch) - 1`); a batch of a single view (batch_mode false,
    attack.py:590-598) has to succeed itself -- the literal B - 1 = 0 would retire it after one iteration whatever the
    detector says."""
    return sum(bool(f) for f in flags) >= max(n_views - 1, 1)


def pgd_attack(model, cameras: Sequence, *, iters: int = 20, alpha: float = 0.5, epsilon: float = 5.0,
               groups: Iterable[str] = ("color",), norm: str = "l2", bg: Optional[torch.Tensor] = None,
               loss_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, pipe: Optional[PipelineParams] = None,
               log: Optional[Callable[[dict], None]] = None, streams: int = 4, accumulate_grads: bool = False,
               batch_loss: bool = False, loss_reduction: str = "sum", background=None,
               success_fn: Optional[Callable[[torch.Tensor, int], bool]] = None, save_path: Optional[str] = None,
               originals: Optional[dict] = None, use_buckets: bool = True,
               timer: Optional["PhaseTimer"] = None, overlap_success: bool = True,
               cache_binning: bool = True, fused_norms: bool = True, batched: bool = True) -> List[float]:
    """Runs up to `iters` PGD iterations over the batch `cameras` (sharded over ranks when torch.distributed is
    initialised).  Returns the per-iteration global loss (sum over the batch's views).  The rank's views are
    pipelined over `streams` HIP streams (gsplat_attack.streams); 1 = the reference's strictly sequential order.

    Reference-faithful switches (all off by default):
      accumulate_grads  .grad is NOT cleared between iterations -- what the reference does in effect: its
                        optimizer.zero_grad (attack.py:602) does not hold the live tensors, so every step uses the SUM of
                        all gradients so far.  Multi-GPU: the running sum is kept on every rank from the all-reduced
                        per-step gradients, so ranks stay identical.
      batch_loss        the B renders are stacked and `loss_fn` is called ONCE on [B,3,H,W], one backward through all B
                        live rasteriser contexts (attack.py:476-494), instead of one loss per view.  A detector loss that
                        normalises over the batch needs this.  loss_reduction="mean" declares that loss_fn averages over
                        its inputs: each rank's loss is then weighted by (its views / B) so that the all-reduced gradient
                        is that of the single-GPU batch (SURVEY.md section 8e caveat).
      background, success_fn, save_path
                        after every step the target is re-rendered together with the frozen `background` model
                        (render_pair: no deep copy, no concatenation; attack.py:513-530), success_fn(image, view index)
                        says whether the detector was fooled on that view, the B flags are gathered over the ranks, and
                        when at least B-1 views succeed (attack.py:560) the loop stops and the attacked model is written
                        to save_path (attack.py:566-568).  The flags of the last iteration are in log records
                        ("successes") and in pgd_attack.last_successes.
      use_buckets       (default on) the fused backward adds each view's attribute gradients into a per-stream
                        GradBucket instead of handing autograd a fresh 59-float-per-Gaussian buffer per view; off =
                        round-2 behaviour (A/B, tests).
      overlap_success   (default on; device only) the success re-render of iteration i (attack.py:522-530) and the FORWARD of
                        iteration i + 1's first view both depend on nothing but the stepped parameters: the re-render goes
                        to a side stream, the next forward is enqueued beside it, and the flags are read when that forward is
                        in the queue.  If the batch turns out to be done, the speculative forward is dropped -- nothing has
                        been differentiated, accumulated or stepped -- so the iteration count, the history and the saved
                        model are those of the serial loop (off: strictly serial, what a PhaseTimer measures).
      cache_binning     (default on; colour-only attacks on the device) the attack steps _features_dc / _features_rest and
                        nothing else (attack.py:25-49), so every iteration renders the same cameras with the same means,
                        scales, rotations and opacities: each camera's rasteriser context -- projection, depth and tile
                        sorts, tile lists, schedule -- is kept in HBM after its first render (RenderCache, ~250 MB per camera
                        at 1 M Gaussians) and later renders of it, the attack's and the success check's, run the colour
                        kernel and the compositor only.  Losses, flags and the saved model are bit for bit those of the
                        uncached loop (tests/test_gpu_rerender.py).
      batched           (default on; device, fused path without object channels, a rank with two or more views) the rank's
                        views go through ONE launch chain per iteration (render_batch ->
                        gsr_forward_raw_batch / gsr_backward_raw_batch_into): one scan, one depth sort, one emission, one tile
                        sort and one schedule for all of them, every SH row read once, and the 59 gradient floats per Gaussian
                        written once instead of read and rewritten per view.  Images bit for bit those of the per-view
                        renders; the summed gradient within float32 rounding of the per-view accumulation
                        (tests/test_gpu_batch.py).  The per-view losses are summed and differentiated by ONE backward.
      fused_norms       (default on; L2 steps on one GPU, one view per iteration) the step's global gradient norms are the
                        sums of squares the raster backward leaves next to the gradients it writes (GradNorms /
                        gsr_ctx_request_sumsq): one launch per tensor and one read of the gradient instead of two.  With
                        more than one view per iteration, several ranks or accumulate_grads the sums do not describe the
                        gradient the step uses and the step sums it itself, as before."""
    groups = tuple(groups)
    assert all(g in GROUPS for g in groups) and norm in ("l2", "linf") and loss_reduction in ("sum", "mean")
    dev = model.get_xyz.device
    pipe = pipe or PipelineParams(skip_objects=True)
    bg = torch.zeros(3, device=dev) if bg is None else bg.to(dev)
    loss_fn = loss_fn or SurrogateDetector().to(dev)
    rank, world = _world()
    my_idx = gdist.views_of_rank(len(cameras), rank, world)
    mine = [cameras[i] for i in my_idx]
    if originals is None:
        originals = {n: getattr(model, n).detach().clone() for n in gdist.ATTACK_PARAMS}
    pgd_attack.last_successes = None
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
    if (cache_binning and groups == ("color",) and dev.type == "cuda" and getattr(pipe, "render_cache", None) is None
            and takes_fused_path(model, pipe)):
        from diff_gaussian_rasterization import RenderCache
        pipe.render_cache = RenderCache(max_entries=4 * max(len(mine), 1) + 8)      # pipe is this call's own copy here
    # the rank's views through one launch chain (see `batched`)
    # (a colour-only attack with kept binning keeps the BATCH's context: colour kernel + one compositor launch for all views)
    use_batch = bool(batched and dev.type == "cuda" and len(mine) >= 2 and can_batch(mine, model, pipe))
    reduce_names = ("_features_dc", "_features_rest") if frozen else gdist.ATTACK_PARAMS
    running = {}                                           # accumulate_grads with world > 1: the running sums
    try:
        history = []
        ring = StreamRing(min(streams, max(len(mine), 1)), dev) if dev.type == "cuda" and not batch_loss and not use_batch else None
        # Gradient buckets (fused path, all five groups differentiated): the rasteriser's backward writes a view's 59
        # attribute gradients per Gaussian straight into a caller-owned flat buffer -- the first view of the iteration
        # overwrites it, the others add -- instead of allocating 236 bytes per Gaussian per view and leaving the
        # accumulation into .grad to autograd's read-modify-write.  One bucket per stream, folded once per iteration;
        # the folded bucket IS the all-reduce buffer and becomes .grad without a copy.
        buckets = None
        # (only render()'s fused path fills a bucket: the same predicate decides here -- with convert_SHs_python /
        # compute_cov3D_python the classic surface runs and autograd accumulates into .grad as before)
        if (use_buckets and dev.type == "cuda" and not frozen and takes_fused_path(model, pipe)
                and all(getattr(model, n).requires_grad for n in gdist.ATTACK_PARAMS)):
            from diff_gaussian_rasterization import GradBucket
            P = int(model.get_xyz.shape[0])
            buckets = [GradBucket(P, dev) for _ in range(ring.n if ring is not None else 1)]
            pipe = copy.copy(pipe)
            pipe.grad_bucket = (lambda: buckets[ring.current]) if ring is not None else buckets[0]
        # L2 steps: the global norms come out of the raster backward (GradNorms) while one backward per iteration writes the
        # gradients the step uses -- one view per iteration on one GPU without the running-sum quirk (BASELINE config 3)
        norms = None
        # (or one BATCH per iteration: its one backward writes the summed gradient)
        if (fused_norms and norm == "l2" and dev.type == "cuda" and world == 1 and not accumulate_grads and takes_fused_path(model, pipe)
                and getattr(pipe, "grad_norms", None) is None):
            from diff_gaussian_rasterization import GradNorms
            norms = GradNorms(dev)
            pipe = copy.copy(pipe)
            pipe.grad_norms = norms
        run_flat = None                                    # accumulate_grads with buckets: the running sum
        overlap = (overlap_success and success_fn is not None and dev.type == "cuda" and not batch_loss and timer is None)
        check_stream = _check_stream(dev) if overlap else None
        pending = None                                     # overlap: the success renders whose flags have not been read yet
        t_mark = [time.perf_counter()]

        def finish_iteration(rec, imgs_and_event):
            """Reads the success flags of an iteration (its renders are in `imgs_and_event`), logs it; True: batch done."""
            done = False
            if imgs_and_event is not None:
                imgs, ev = imgs_and_event
                if ev is not None:
                    with torch.cuda.stream(check_stream):
                        mine_flags = [success_fn(im, i) for im, i in zip(imgs, my_idx)]
                    torch.cuda.current_stream(dev).wait_event(ev)      # the parameters are not stepped under the renders
                else:
                    mine_flags = [success_fn(im, i) for im, i in zip(imgs, my_idx)]
                flags = gather_success(mine_flags, len(cameras), rank, world, dev)
                pgd_attack.last_successes = flags
                rec["successes"] = flags
                done = batch_done(flags, len(cameras))
            if log is not None:
                if dev.type == "cuda":
                    torch.cuda.synchronize()
                now = time.perf_counter()
                rec["seconds"] = now - t_mark[0]
                t_mark[0] = now
                log(rec)
            if done and save_path is not None and rank == 0:
                model.save_ply(save_path)
            return done

        stopped = False
        for it in range(iters):
            if timer is not None:
                timer.start()
            if pending is not None and not mine:           # a rank without views has no forward to hide the wait behind
                stopped = finish_iteration(*pending)
                pending = None
                if stopped:
                    break
            def clear_gradients():
                if norms is not None:
                    norms.begin()
                if buckets is not None:
                    for b in buckets:
                        b.reset()
                elif not accumulate_grads or world > 1:
                    model.zero_grad()                      # multi-GPU: .grad holds THIS step's gradient until reduced
            # (with a success check pending, the gradients of the previous iteration stay in place until its flags are
            # read: a batch that turns out to be done returns with them, like the serial loop)
            cleared = pending is None
            if cleared:
                clear_gradients()
            losses = []
            if use_batch:
                imgs_b = render_batch(mine, model, pipe, bg)["render"]
                if pending is not None:
                    # the previous iteration's success renders ran beside this forward: now read their flags
                    stopped = finish_iteration(*pending)
                    pending = None
                    if stopped:
                        del imgs_b                         # speculative: never differentiated, nothing accumulated
                        break
                if not cleared:
                    clear_gradients()
                    cleared = True
                if timer is not None:
                    timer.lap("render")
                if batch_loss:
                    loss = loss_fn(imgs_b)
                    if loss_reduction == "mean":
                        loss = loss * (len(mine) / len(cameras))
                    losses.append(loss.detach())
                else:
                    # (unbind, not slices: a slice's backward materialises a zero [B,3,H,W] tensor per view and adds it in)
                    per_view = [loss_fn(im[None]) for im in imgs_b.unbind(0)]
                    if loss_reduction == "mean":
                        per_view = [l_ / len(cameras) for l_ in per_view]
                    losses.extend(l_.detach() for l_ in per_view)
                    loss = torch.stack(per_view).sum()
                if timer is not None:
                    timer.lap("loss")
                loss.backward()                            # ONE raster backward for the rank's views
                if timer is not None:
                    timer.lap("backward")
            elif batch_loss:
                if mine:
                    renders = torch.stack([render(cam, model, pipe, bg)["render"] for cam in mine])
                    loss = loss_fn(renders)
                    if loss_reduction == "mean":
                        loss = loss * (len(mine) / len(cameras))
                    loss.backward()
                    losses.append(loss.detach())
            else:
                for vi, cam in enumerate(mine):            # one forward+backward per view: peak memory = one view per stream
                    with (ring.next() if ring is not None else contextlib.nullcontext()):
                        img = render(cam, model, pipe, bg)["render"]
                        if pending is not None:
                            # the previous iteration's success renders ran beside this forward: now read their flags
                            stopped = finish_iteration(*pending)
                            pending = None
                            if stopped:
                                del img                    # speculative: never differentiated, nothing accumulated
                                break
                        if not cleared:
                            clear_gradients()
                            cleared = True
                        if timer is not None:
                            timer.lap("render")
                        loss = loss_fn(img[None])
                        if loss_reduction == "mean":
                            loss = loss / len(cameras)
                        if timer is not None:
                            timer.lap("loss")
                        loss.backward()
                        if timer is not None:
                            timer.lap("backward")
                        losses.append(loss.detach())
                if ring is not None:
                    ring.join()
                if stopped:
                    break
            total = torch.stack(losses).sum() if losses else torch.zeros((), device=dev)
            if buckets is not None:
                tot = buckets[0]
                for b in buckets[1:]:
                    tot.add_(b)
                if not tot.used:
                    if mine:                               # a bucket that no backward of this rank's views wrote: the
                        raise RuntimeError(                # renders did not take the path the buckets were made for
                            "pgd_attack: the gradient bucket was not written by any of this rank's views (render() did "
                            "not take the fused raw-parameter path); pass use_buckets=False")
                    tot.flat.zero_()                       # a rank without views contributes zeros
                    tot.used, tot.fresh = True, False
                if world > 1:
                    flat = tot.flat
                    if flat.is_cuda and torch.distributed.get_backend() == "gloo":
                        host = flat.cpu()                  # rehearsal on one GPU: gloo reduces host tensors
                        torch.distributed.all_reduce(host)
                        flat.copy_(host)
                    else:
                        torch.distributed.all_reduce(flat)     # ONE collective over 59 floats per Gaussian
                    torch.distributed.all_reduce(total)
                if accumulate_grads:
                    run_flat = tot.flat.clone() if run_flat is None else run_flat.add_(tot.flat)
                    keep = tot.flat
                    tot.flat = run_flat
                    tot.assign_to(model)
                    tot.flat = keep
                else:
                    tot.assign_to(model)
            elif world > 1:
                gdist.allreduce_attribute_grads(model, names=reduce_names)     # one bucket: 236 MB, or 192 MB colour-only
                torch.distributed.all_reduce(total)
                if accumulate_grads:
                    for n in reduce_names:
                        p = getattr(model, n)
                        running[n] = p.grad.clone() if n not in running else running[n].add_(p.grad)
                        p.grad = running[n]
            if timer is not None:
                timer.lap("reduce")
            _step(model, originals, groups, norm, alpha, epsilon, norms)
            if timer is not None:
                timer.lap("step")
            history.append(float(total))
            rec = {"iter": it, "loss": history[-1], "views": len(cameras)}
            if success_fn is None:
                finish_iteration(rec, None)
                continue
            if overlap:
                # success renders on the side stream, behind the step; the next iteration's first forward goes beside them
                stepped = torch.cuda.Event()
                stepped.record(torch.cuda.current_stream(dev))
                check_stream.wait_event(stepped)
                with torch.cuda.stream(check_stream):
                    imgs = render_combined(model, background, mine, bg, pipe)
                    rendered = torch.cuda.Event()
                    rendered.record(check_stream)
                pending = (rec, (imgs, rendered))
                continue
            imgs = render_combined(model, background, mine, bg, pipe)
            if timer is not None:
                timer.lap("rerender")
            if finish_iteration(rec, (imgs, None)):
                break
        if pending is not None:                            # the last iteration's flags
            finish_iteration(*pending)
    finally:
        for p in frozen:
            p.requires_grad_(True)
    return history


pgd_attack.last_successes = None


def run_attack(model, cameras: Sequence, *, background=None, batch_size: int = 5, max_iters: int = 20,
               success_fn: Callable[[torch.Tensor, int], bool], save_path: Optional[str] = None,
               truncate: bool = True, add_cams: int = 1, benign: bool = False, **kw) -> dict:
    """The batch schedule of the reference's run() (attack.py:463-475, 556-569), iteration for iteration: ONE global
    counter `it` runs over range(max_iters * num_batches); the pending views are attacked `batch_size` at a time; an
    iteration with (it + 1) % max_iters == 0 attacks nothing and DROPS the current batch (:470-475) -- so a batch that
    starts after an early success only gets what is left of the current max_iters window; a batch that reaches B-1
    successes is retired (:560-563), and the attacked model is saved when the retired batch was the last pending one
    (:564-569), whether or not earlier batches were dropped.  truncate=False keeps the views beyond a multiple of
    batch_size as a smaller last batch instead of dropping them.  Perturbations accumulate across batches: every batch
    projects onto the eps-ball around the ORIGINAL attributes (attack.py:389-394 captures them once).
    add_cams > 1: the yawed copies of the first camera join the views first (augment_cameras; attack.py:404-415).
    benign: the benign pass runs in front of the loop and its boxes are returned under "gt_bboxes" (attack.py:434-461).
    -> {"batches": [{"views", "iters", "success", "loss"}], "all_succeeded", "saved", "iterations"[, "gt_bboxes"]}."""
    cameras = augment_cameras(cameras, add_cams)
    pending = list(range(len(cameras)))
    if truncate and len(pending) % batch_size:             # the reference drops the views beyond a multiple of B (:417-423)
        pending = pending[:len(pending) - len(pending) % batch_size]
    # the benign pass sees the views that are KEPT (the reference truncates viewpoint_stack first, :417-423, and renders
    # the boxes afterwards, :434-461): gt_bboxes has one row per attacked view
    gt_bboxes = benign_bboxes(model, [cameras[i] for i in pending], kw.get("pipe")) if benign else None
    num_batches = max(1, -(-len(pending) // batch_size))   # ceil (:428); truncate=False keeps a smaller last batch
    originals = {n: getattr(model, n).detach().clone() for n in gdist.ATTACK_PARAMS}
    report, saved = [], False
    it, budget = 0, max_iters * num_batches
    rank, _ = _world()
    while pending and it < budget:
        if (it + 1) % max_iters == 0:                      # the window's last slot: the batch is dropped, nothing is attacked
            it += 1
            if report and report[-1]["views"] == pending[:batch_size] and not report[-1]["success"]:
                report[-1]["dropped"] = True
            else:
                report.append({"views": pending[:batch_size], "iters": 0, "success": False, "loss": None, "dropped": True})
            pending = pending[batch_size:]
            continue
        cur = pending[:batch_size]
        batch = [cameras[i] for i in cur]
        left = min(max_iters - 1 - (it % max_iters), budget - it)     # attack iterations before the window's last slot
        hist = pgd_attack(model, batch, iters=left, background=background,
                          success_fn=lambda im, j: success_fn(im, cur[j]), originals=originals, **kw)
        it += len(hist)
        flags = pgd_attack.last_successes or []
        ok = bool(hist) and batch_done(flags, len(cur))
        report.append({"views": cur, "iters": len(hist), "success": bool(ok), "loss": hist[-1] if hist else None})
        if ok:
            pending = pending[len(cur):]
            if not pending:                                # attack.py:564-569
                if save_path is not None and rank == 0:
                    model.save_ply(save_path)
                saved = True
        # not ok: the window is exhausted; the next loop turn is its last slot and drops the batch
    out = {"batches": report, "all_succeeded": all(b["success"] for b in report) and not pending, "saved": saved,
           "iterations": it}
    if gt_bboxes is not None:
        out["gt_bboxes"] = gt_bboxes
    return out


def augment_cameras(cameras: Sequence, add_cams: int = 1, yaw_step_deg: float = 7.0) -> list:
    """The reference's view augmentation (attack.py:404-415; configs/config.yaml `add_cams`): add_cams - 1 deep copies of
    the FIRST camera, copy i yawed by 7 i degrees (Camera.yaw: view and projection refreshed, camera_center kept -- SURVEY.md
    section 3.1 quirk 4), appended behind the given cameras.  add_cams <= 1: the list unchanged."""
    out = list(cameras)
    for i in range(1, int(add_cams)):
        cam = copy.deepcopy(out[0])
        cam.yaw(yaw_step_deg * i)
        out.append(cam)
    return out


def bbox_from_render(image: torch.Tensor, threshold: int = 20):
    """The bounding box the reference takes from a benign render (attack.py:438-449): clamp to [0, 1], scale by 255 and
    truncate to bytes, ITU-R 601-2 luma as PIL's convert('L') computes it (integer: (19595 R + 38470 G + 7471 B + 32768)
    >> 16), pixels with luma > threshold are "object", getbbox() of those -- (left, upper, right, lower) with right /
    lower exclusive, or None when nothing is above the threshold.  Tensor ops on the image's device; one small read-back."""
    rgb = (torch.clamp(image.detach(), 0.0, 1.0) * 255.0).to(torch.uint8).to(torch.int32)
    luma = (19595 * rgb[0] + 38470 * rgb[1] + 7471 * rgb[2] + 32768) >> 16
    on = luma > int(threshold)
    rows = torch.nonzero(on.any(dim=1)).flatten()
    if rows.numel() == 0:
        return None
    cols = torch.nonzero(on.any(dim=0)).flatten()
    return (int(cols[0]), int(rows[0]), int(cols[-1]) + 1, int(rows[-1]) + 1)


@torch.no_grad()
def benign_bboxes(model, cameras: Sequence, pipe: Optional[PipelineParams] = None, threshold: int = 20) -> list:
    """The benign pass in front of the attack loop (attack.py:434-461): every view rendered on a BLACK background
    (whatever the attack's background is) and turned into the ground-truth box the detector loss is given."""
    pipe = pipe or PipelineParams(skip_objects=True)
    if getattr(pipe, "render_cache", None) is not None or getattr(pipe, "grad_bucket", None) is not None:
        # one forward per view, never rendered again under this background: no kept context (~250 MB per view at 1 M
        # Gaussians / 1080p), no gradient bucket
        pipe = copy.copy(pipe)
        pipe.render_cache = None
        pipe.grad_bucket = None
    black = torch.zeros(3, device=model.get_xyz.device)
    with torch.no_grad():
        return [bbox_from_render(render(cam, model, pipe, black)["render"], threshold) for cam in cameras]


def combine_with_background(attacked, background):
    """The scene the reference evaluates after every step, as an explicit model: the attacked target Gaussians followed
    by the frozen background (reference attack.py:513-520: deepcopy + seven concat_setup calls).  Only for callers that
    need the concatenated model itself (e.g. to save it); rendering it goes through render_combined, which does not
    build it."""
    from .gaussian_model import GaussianModel
    cat = {k: torch.cat((a.detach(), b.detach().to(a.device)), dim=0)
           for k, a, b in ((n, getattr(attacked, n), getattr(background, n)) for n in GaussianModel._PARAM_ATTRS)}
    return GaussianModel.from_tensors(cat["_xyz"], cat["_features_dc"], cat["_features_rest"], cat["_scaling"],
                                      cat["_rotation"], cat["_opacity"], cat["_objects_dc"],
                                      sh_degree=attacked.max_sh_degree, device=cat["_xyz"].device, requires_grad=False)


@torch.no_grad()
def render_combined(attacked, background, cameras: Sequence, bg: torch.Tensor, pipe: Optional[PipelineParams] = None):
    """Forward-only renders of target + background for the success check (reference attack.py:522-530).  The two
    parameter sets go to the rasteriser side by side (render_pair -> gsr_forward_raw2): no deep copy of the model, no
    concatenation of its seven tensors per PGD iteration.  background None: the target alone."""
    pipe = pipe or PipelineParams(skip_objects=True)
    cams = list(cameras)
    # two or more cameras without object channels: ONE launch chain for all of them (render_batch / render_pair_batch), every
    # image bit for bit the per-camera render's; `batched_checks=False` on the pipe keeps the per-camera loop (A/B, tests)
    batch = (len(cams) >= 2 and bool(getattr(pipe, "batched_checks", True)) and attacked.get_xyz.is_cuda
             and can_batch(cams, attacked, pipe))
    if background is None:
        if getattr(pipe, "render_cache", None) is not None:
            pipe = copy.copy(pipe)
            pipe.cache_tag = "check"      # its own kept contexts: the attack's forward of the same camera may run beside it
        with torch.no_grad():
            if batch:
                return list(render_batch(cams, attacked, pipe, bg)["render"].unbind(0))
            return [render(cam, attacked, pipe, bg)["render"] for cam in cams]
    if (batch and _has_raw_layout(attacked) and _has_raw_layout(background) and attacked.get_xyz.shape[0] > 0
            and background.get_xyz.shape[0] > 0):
        return list(render_pair_batch(cams, attacked, background, pipe, bg)["render"].unbind(0))
    return [render_pair(cam, attacked, background, pipe, bg)["render"] for cam in cams]


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
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the rank's views are pipelined over")
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
