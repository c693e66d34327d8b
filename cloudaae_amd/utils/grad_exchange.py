"""Data-parallel gradient exchange over one flat gradient buffer (new; the reference is
single-GPU).  One process per GPU; backend "nccl" is RCCL over xGMI on MI355X, "gloo" in
the CPU tests.  The buffer is reduced (sum) in up to three contiguous pieces:

    [0, lo)  [lo, hi)  [hi, n)

`[lo, hi)` is the "early" range -- the fully connected stack (decoder + pose heads: 97 % of
the model at N=1024), whose gradients are the first ones backward produces; `early_ready()`
is called once per variable of the range and, at the `early_count`-th call of a step,
launches the range's all-reduce asynchronously so that it overlaps the rest of backward
(the encoder, ~1.2 ms at B=32).  `finish()`
launches the remaining pieces and waits for everything.  Averaging (1/world) is folded
into the optimiser kernel (`scale`), not applied here.
"""
import os

import torch
import torch.distributed as dist


def shard_range(global_batch, world, rank):
    """Contiguous slice of the global batch owned by `rank` (clouds are independent units)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d does not divide by %d ranks" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradExchange(object):
    def __init__(self, flat_grads, early=None, group=None, world=None, early_count=1):
        self.g = flat_grads
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.world = world
        # CLOUDAAE_FORCE_COLLECTIVES=1: issue the collectives even with one rank (exercises the
        # RCCL code path on a single-GPU box)
        self.active = world > 1 or (dist.is_initialized() and os.environ.get("CLOUDAAE_FORCE_COLLECTIVES") == "1")
        n = flat_grads.numel()
        if early is not None:
            lo, hi = early
            if not (0 <= lo < hi <= n):
                raise ValueError("bad early range")
        self.early = early
        self.early_count = max(1, int(early_count))
        self._early_seen = 0
        self._pending = []
        self._early_sent = False

    @property
    def scale(self):
        return 1.0 / self.world

    def early_ready(self):
        """Call right after the kernel writing the early range was enqueued."""
        if not self.active or self.early is None or self._early_sent:
            return
        self._early_seen += 1
        if self._early_seen < self.early_count:
            return
        lo, hi = self.early
        self._pending.append(dist.all_reduce(self.g[lo:hi], group=self.group, async_op=True))
        self._early_sent = True

    def finish(self):
        """Reduce whatever has not been sent yet and wait for all pieces."""
        if not self.active:
            return
        n = self.g.numel()
        if self.early is None:
            pieces = [(0, n)]
        else:
            lo, hi = self.early
            pieces = [(0, lo), (hi, n)]
            if not self._early_sent:
                pieces.insert(1, (lo, hi))
        for a, b in pieces:
            if b > a:
                self._pending.append(dist.all_reduce(self.g[a:b], group=self.group, async_op=True))
        for w in self._pending:
            w.wait()
        self._pending = []
        self._early_sent = False
        self._early_seen = 0

    def broadcast_params(self, flat_params, src=0):
        if self.active:
            dist.broadcast(flat_params, src=src, group=self.group)
