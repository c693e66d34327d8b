"""Data-parallel gradient exchange over one flat gradient buffer (new; the reference is
single-GPU).  One process per GPU; backend "nccl" is RCCL over xGMI on MI355X, "gloo" in
the CPU tests.  The buffer is reduced (sum) in up to three contiguous pieces:

    [0, lo)  [lo, hi)  [hi, n)

`[lo, hi)` is the "early" range -- the decoder output weights (12*N*1024 floats, 77 % of
the model at N=1024), whose gradient is the first one backward produces; `early_ready()`
launches its all-reduce asynchronously so it overlaps the rest of backward.  `finish()`
launches the remaining pieces and waits for everything.  Averaging (1/world) is folded
into the optimiser kernel (`scale`), not applied here.
"""
import os

import torch.distributed as dist


def shard_range(global_batch, world, rank):
    """Contiguous slice of the global batch owned by `rank` (clouds are independent units)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d does not divide by %d ranks" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradExchange(object):
    def __init__(self, flat_grads, early=None, group=None, world=None):
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
        self._pending = []
        self._early_sent = False

    @property
    def scale(self):
        return 1.0 / self.world

    def early_ready(self):
        """Call right after the kernel writing the early range was enqueued."""
        if not self.active or self.early is None or self._early_sent:
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

    def broadcast_params(self, flat_params, src=0):
        if self.active:
            dist.broadcast(flat_params, src=src, group=self.group)
