"""SyncBN: batch-norm moments over a batch that is sharded across ranks (new; the reference is
single-GPU, where tf.nn.moments sees the whole batch: utils/tf_util.py:492, :514-555).

The HIP library does not link a communication library.  Its `_sync` entry points
(include/cloudaae_hip.h: cloudaae_bn_forward_sync, cloudaae_bn_backward_sync,
cloudaae_edgeconv_forward_sync, cloudaae_edgeconv_backward_sync) reduce a rank's rows to 2*C fp64
sums in a caller-owned device buffer and call back into the host to add them across ranks --
`BnSync` is that host side: one `torch.distributed.all_reduce` (RCCL over xGMI with the "nccl"
backend, gloo in the CPU tests) per layer and direction, 2*C doubles each (<= 16 KB), issued on
torch's current stream between the layer's statistics pass and its finalise kernel.

What is exchanged, per batch-norm layer (11 in get_model_dgcnn_mean_6d):
    forward : [sum x, sum x^2]          -> mean / biased variance of the GLOBAL batch (and the EMA)
    backward: [sum dz, sum dz * x_hat]  -> the two means in dx = gamma*rstd*(dz - m1 - x_hat*m2)
dgamma = sum dz*x_hat, dbeta = sum dz and the bias gradient stay LOCAL sums: the gradient exchange
adds parameter gradients across ranks anyway (utils/grad_exchange.py).

`global_moments` / `global_backward_means` are the same arithmetic on host tensors -- the
definition the kernels implement, used by the CPU (gloo) tests.
"""
import ctypes

import torch
import torch.distributed as dist

from .. import _lib


# the batch norms' own communicator, one per set of ranks and process (created at the first BnSync over those ranks,
# reused by every later one, destroyed by close_communicators())
_OWN_GROUPS = {}


def close_communicators():
    """Destroy the communicators BnSync created (end of a process that builds many graphs, tests)."""
    for _, g in _OWN_GROUPS.values():
        try:
            dist.destroy_process_group(g)
        except Exception:      # noqa: BLE001 -- already gone with the default group
            pass
    _OWN_GROUPS.clear()


class BnSync(object):
    def __init__(self, group=None, world=None, own_communicator=True):
        # own_communicator: the small blocking all-reduces of the batch norms get their OWN communicator (a second
        # group over the same ranks): collectives of one communicator are serialised on its stream, so on the
        # gradient exchange's group every batch norm of the encoder's backward pass would wait behind the large
        # asynchronous all-reduce of the fully connected gradients (utils/grad_exchange.py) and undo its overlap
        self.world = int(world if world is not None else dist.get_world_size(group))
        if own_communicator and dist.is_initialized() and self.world > 1:
            # dist.new_group is a collective over the DEFAULT group: every rank of the default group must construct its
            # first BnSync over these ranks (ranks outside `group` included), in the same order relative to other
            # new_group calls.  Later BnSyncs over the same ranks reuse the communicator: no further collective, no leak.
            ranks = tuple(dist.get_process_group_ranks(group) if group is not None else range(dist.get_world_size()))
            # (a handle made under an earlier default group -- destroy_process_group + init_process_group in one process:
            # tests, notebooks, an elastic restart -- is dead: an entry is good only for the default group OBJECT it was made
            # under, which the entry keeps alive so that its identity cannot be handed to a new one)
            made = _OWN_GROUPS.get(ranks)
            if made is None or made[0] is not dist.group.WORLD:
                made = (dist.group.WORLD, dist.new_group(ranks=list(ranks)))
                _OWN_GROUPS[ranks] = made
            group = made[1]
        self.group = group
        self.calls = 0               # all-reduces issued (tests and bench read it)
        self.error = None            # exception raised inside the callback (ctypes cannot propagate it)
        self._planned = {}           # data_ptr -> buffer of a recorded step (lives as long as its plan)
        self._pool = {}              # (slot of the step, C) -> buffer reused step after step when not recording
        self._live = {}              # data_ptr -> pooled buffer
        self._slot = 0
        self._cb = _lib.ALLREDUCE_FN(self._allreduce)     # must outlive every struct that points at it

    # -- called by the library, between a layer's statistics pass and its finalise kernel -----------
    def _allreduce(self, ctx, buf, count, stream):
        try:
            t = self._planned.get(buf)
            if t is None:
                t = self._live[buf]
            dist.all_reduce(t[:count], group=self.group)       # sum; ordered on torch's current stream
            self.calls += 1
            return 0
        except BaseException as e:                             # noqa: BLE001 -- reported through the return code
            self.error = e
            return 1

    def begin_step(self):
        self._slot = 0

    def forget(self, plan):
        """Drop the buffers of a recorded step that is being discarded (they live in its arena)."""
        self._planned = {p: t for p, t in self._planned.items() if getattr(t, "_cloudaae_plan", None) is not plan}

    def arg(self, C, device):
        """A `cloudaae_bn_sync *` for one layer and direction with C channels."""
        n = 2 * int(C)
        if _lib.recording() is not None:
            buf = _lib.empty(n, dtype=torch.float64, device=device)
            buf._cloudaae_plan = _lib.recording()
            self._planned[buf.data_ptr()] = buf
        else:
            key = (self._slot, n)
            self._slot += 1
            buf = self._pool.get(key)
            if buf is None:
                buf = torch.empty(n, dtype=torch.float64, device=device)
                self._pool[key] = buf
                self._live[buf.data_ptr()] = buf
        st = _lib.BnSyncStruct(self._cb, None, self.world, buf.data_ptr())
        return ctypes.pointer(st)

    def check(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e


# ---- the definition, on host tensors (CPU tests) ------------------------------------------------
def global_moments(x_local, group=None):
    """mean and biased variance over the rows of EVERY rank's x [M_local, C] (fp64 sums, as the kernels)."""
    x = x_local.double()
    sums = torch.stack([x.sum(0), (x * x).sum(0)])
    count = torch.tensor([float(x.shape[0])], dtype=torch.float64)
    dist.all_reduce(sums, group=group)
    dist.all_reduce(count, group=group)
    mean = sums[0] / count
    var = (sums[1] / count - mean * mean).clamp_min(0.0)
    return mean.float(), var.float()


def global_backward_means(dz_local, xhat_local, group=None):
    """m1 = mean dz, m2 = mean dz*x_hat over every rank's rows; also the LOCAL sums (dbeta, dgamma)."""
    dz, xh = dz_local.double(), xhat_local.double()
    local = torch.stack([dz.sum(0), (dz * xh).sum(0)])
    sums = local.clone()
    count = torch.tensor([float(dz.shape[0])], dtype=torch.float64)
    dist.all_reduce(sums, group=group)
    dist.all_reduce(count, group=group)
    return (sums[0] / count).float(), (sums[1] / count).float(), local[0].float(), local[1].float()
