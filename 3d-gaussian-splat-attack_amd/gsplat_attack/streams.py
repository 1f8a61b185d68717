"""View pipelining over HIP streams (an extension; the reference renders its views one after another on one stream).

One view's forward + backward is a chain of ~35 dependent kernels of which only the two compositing kernels fill
the chip; the sorts, scans and the per-Gaussian kernels in between leave most CUs idle.  The views of a PGD batch
are independent until the optimiser step, so dealing them round-robin over a few streams lets the small kernels of
one view run beside the compositing kernels of another (+27 % views/s on the benchmark scene with four streams and
GPU_MAX_HW_QUEUES=8, which gsplat_attack/__init__.py asks for: with the runtime's default of four hardware queues
three streams are the optimum and a fourth loses 6 %).

    ring = StreamRing(4, device)
    for cam in batch:
        with ring.next():                         # this view's kernels (and its loss) go to the ring's next stream
            loss_fn(render(cam, model, pipe, bg)["render"]).backward()
    ring.join()                                   # the caller's stream waits for every view; .grad is then complete

Gradient accumulation across streams is ordered by autograd itself (it runs each backward on its forward's stream
and synchronises the leaf accumulation), and libgsraster keeps per-stream workspace blocks, so views on different
streams share no scratch memory.
"""
from __future__ import annotations

import contextlib

import torch


_STREAM_POOL = {}      # device -> streams shared by every StreamRing on it


class StreamRing:
    def __init__(self, n: int, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = max(int(n), 1)
        # The streams are shared by all rings of a device (and handed out in the same order): libgsraster keeps its
        # workspace blocks with the stream that used them last, so a ring on fresh streams would make every view of its
        # first round allocate device memory again (tens of milliseconds per PGD call at 1 M Gaussians).
        pool = _STREAM_POOL.setdefault(str(self.device), [])
        while self.n > 1 and len(pool) < self.n:
            pool.append(torch.cuda.Stream(device=self.device))
        self.streams = pool[:self.n] if self.n > 1 else []
        self._i = 0
        self._forked = False
        self.current = 0            # index of the stream the innermost `with ring.next()` block runs on
        if self.streams:
            # leaf gradients are accumulated on the stream their AccumulateGrad node was first made on, whatever
            # stream a view's backward runs on: intended here, and ordered by autograd
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)

    def _fork(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)                      # everything the caller enqueued so far is visible to the views
        self._forked = True

    @contextlib.contextmanager
    def next(self):
        if not self.streams:
            self.current = 0
            yield None
            return
        if not self._forked:
            self._fork()
        self.current = self._i % self.n
        s = self.streams[self.current]
        self._i += 1
        with torch.cuda.stream(s):
            yield s

    def join(self):
        """Order the caller's current stream after every view enqueued so far."""
        if self.streams and self._forked:
            cur = torch.cuda.current_stream(self.device)
            for s in self.streams:
                cur.wait_stream(s)
        self._forked = False
