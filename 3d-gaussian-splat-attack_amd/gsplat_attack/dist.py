"""View-sharded data parallelism for the PGD loop: one process per GPU, one view per rank, gradients of the
(replicated) Gaussian attributes summed across ranks.

The reference is single-process: its "batch" is a Python loop over cameras whose gradients add up in .grad at
``loss.backward()`` (reference attack.py:476-494).  Here rank r renders views r, r+G, ... of the batch and the
per-step attribute gradients are all-reduced (RCCL over xGMI through torch.distributed's "nccl" backend; gloo
on CPU for tests) -- the only exchange the path has.  Every rank then applies the identical PGD step, so the
replicas stay bit-identical (an all-reduce returns the same bits on every rank).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

# raw parameters whose gradients the attack consumes (59 floats per Gaussian at SH degree 3)
ATTACK_PARAMS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def init_from_env(backend: Optional[str] = None):
    """-> (rank, world, local_rank).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* set by torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def views_of_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Indices of the batch's views this rank renders (round-robin, like one view per GPU at n_views == world)."""
    return list(range(rank, n_views, world))


def _flat_view_of(grads):
    """If the gradients are back-to-back slices of one buffer (the fused backward allocates them that way), return a
    1-D tensor aliasing exactly that span, else None."""
    try:
        base = grads[0].untyped_storage().data_ptr()
        if any(g.untyped_storage().data_ptr() != base or not g.is_contiguous() or g.dtype != grads[0].dtype for g in grads):
            return None
        spans = sorted((g.storage_offset(), g.numel()) for g in grads)
        pos = spans[0][0]
        for off, n in spans:
            if off != pos:
                return None
            pos += n
        flat = torch.empty(0, dtype=grads[0].dtype, device=grads[0].device)
        flat.set_(grads[0].untyped_storage(), spans[0][0], (pos - spans[0][0],))
        return flat
    except Exception:
        return None


def allreduce_attribute_grads(model, names: Iterable[str] = ATTACK_PARAMS, group=None) -> int:
    """Sum the per-step gradients over ranks, in place.  Returns the number of bytes reduced.
    A parameter that got no gradient on this rank (e.g. no view assigned) contributes zeros.  When the gradients sit
    in one flat buffer (the fused render path) this is ONE all-reduce of 59 floats per Gaussian; otherwise one
    asynchronous all-reduce per tensor."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    grads = []
    for n in names:
        p = getattr(model, n)
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        if not p.grad.is_contiguous():
            p.grad = p.grad.contiguous()
        grads.append(p.grad)
    flat = _flat_view_of(grads)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return flat.numel() * flat.element_size()
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True) for g in grads]
    for w in works:
        w.wait()
    return sum(g.numel() * g.element_size() for g in grads)


class BucketAllReduce:
    """Sum all-reduce of a GradBucket issued range by range while the backward that fills it is still running
    (SURVEY.md section 8e).  Arm it on the bucket before the LAST backward of the step:

        ar = BucketAllReduce(bucket, chunks=4)          # bucket.chunks / bucket.on_chunk are set
        loss.backward()                                 # K9 runs in 4 ranges; each range's six slices are all-reduced
        ar.wait()                                       # asynchronously (RCCL's own stream) as soon as it is enqueued

    With "nccl" the six slices of a range are six asynchronous all-reduces (the public entry point; RCCL's own stream
    orders them behind the work enqueued before them and the ranges behind overlap with them).  With "gloo" (tests,
    rehearsals on one GPU) the slices travel through the host synchronously.  The result is that of one all-reduce of
    the whole bucket (element-wise sums)."""

    def __init__(self, bucket, chunks: int = 4, group=None):
        self.bucket, self.group, self.works, self.bytes = bucket, group, [], 0
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if self.active:
            bucket.chunks = max(int(chunks), 1)
            bucket.on_chunk = self._on_chunk
            if bucket.chunks == 1:                       # one range: the hook is not called by the unchunked backward
                bucket.on_chunk = None

    def _on_chunk(self, chunk: int, g0: int, g1: int):
        if g1 <= g0:
            return
        pieces = self.bucket.range_slices(g0, g1)
        self.bytes += sum(p.numel() for p in pieces) * 4
        if dist.get_backend(self.group) == "gloo" and pieces[0].is_cuda:
            for p in pieces:
                h = p.cpu()
                dist.all_reduce(h, group=self.group)
                p.copy_(h)
        else:
            for p in pieces:
                self.works.append(dist.all_reduce(p, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> int:
        """Blocks the current stream until every issued range is reduced (and reduces the whole bucket now if the
        backward was not chunked).  Returns the bytes reduced."""
        if not self.active:
            return 0
        if self.bytes == 0:                              # unchunked: one collective over the flat buffer
            flat = self.bucket.flat
            if dist.get_backend(self.group) == "gloo" and flat.is_cuda:
                h = flat.cpu()
                dist.all_reduce(h, group=self.group)
                flat.copy_(h)
            else:
                dist.all_reduce(flat, group=self.group)
            self.bytes = flat.numel() * 4
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []
        return self.bytes
