"""CPU: SyncBN host logic with world_size 2 over gloo -- the decomposition the `_sync` kernels implement
(utils/sync_bn.py: local fp64 sums -> all-reduce -> global moments / global backward means, LOCAL dgamma,
dbeta) equals batch_norm_template (utils/tf_util.py:473-511, restated in oracle/model_oracle.py) on the
unsharded batch, forward and backward; and the ctypes all-reduce callback the library calls
(struct cloudaae_bn_sync) adds a buffer across ranks.  No HIP compute is involved."""
import ctypes
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

EPS = 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cloudaae_amd.utils import sync_bn as S
        from oracle import model_oracle as MO
        g = torch.Generator().manual_seed(5)
        M, C = 24, 7                                  # global rows, channels
        x = torch.randn(M, C, generator=g) * 2 + 1
        gamma = torch.rand(C, generator=g) + 0.5
        beta = torch.randn(C, generator=g)
        up = torch.randn(M, C, generator=g)            # upstream gradient of the GLOBAL loss per row
        lo, hi = rank * M // world, (rank + 1) * M // world

        # the reference computation on the whole batch (one process would do this)
        V = MO.Vars(seed=0)
        V.p["bn/gamma"] = gamma.clone().requires_grad_(True)
        V.p["bn/beta"] = beta.clone().requires_grad_(True)
        xg = x.clone().requires_grad_(True)
        y = torch.relu(MO.batch_norm(xg, "bn", V, True, 0.9))
        (y * up).sum().backward()
        ema_mean = V.s["bn/moments/Squeeze/ExponentialMovingAverage"]
        ema_var = V.s["bn/moments/Squeeze_1/ExponentialMovingAverage"]

        # the sharded computation: this rank's rows only + two all-reduces of 2*C sums
        xl, ul = x[lo:hi], up[lo:hi]
        mean, var = S.global_moments(xl)
        rstd = 1.0 / torch.sqrt(var + EPS)
        xhat = (xl - mean) * rstd
        z = torch.relu(xhat * gamma + beta)
        dz = ul * (z > 0).float()
        m1, m2, dbeta_local, dgamma_local = S.global_backward_means(dz, xhat)
        dx = gamma * rstd * ((dz - m1) - xhat * m2)
        ok_fwd = torch.allclose(z, y.detach()[lo:hi], rtol=1e-5, atol=1e-6)
        ok_dx = torch.allclose(dx, xg.grad[lo:hi], rtol=1e-4, atol=1e-6)
        # parameter gradients are LOCAL sums; the gradient exchange (sum over ranks) completes them
        pg = torch.stack([dgamma_local, dbeta_local])
        dist.all_reduce(pg)
        ok_pg = torch.allclose(pg[0], V.p["bn/gamma"].grad, rtol=1e-4, atol=1e-5) and \
            torch.allclose(pg[1], V.p["bn/beta"].grad, rtol=1e-4, atol=1e-5)
        # every rank derives the same EMA update from the global moments
        ok_ema = torch.allclose(mean * 0.1, ema_mean, rtol=1e-5, atol=1e-7) and \
            torch.allclose(var * 0.1, ema_var, rtol=1e-5, atol=1e-7)
        # local moments would NOT have matched (the test has teeth)
        ok_teeth = not torch.allclose(xl.mean(0), mean, rtol=1e-3, atol=1e-4)

        # the callback the library calls: struct cloudaae_bn_sync { allreduce, ctx, world, buf }
        sync = S.BnSync(None, world)
        sync.begin_step()
        arg = sync.arg(3, "cpu")
        st = arg.contents
        buf = sync._live[st.buf]
        buf.copy_(torch.arange(6, dtype=torch.float64) * (rank + 1))
        rc = st.allreduce(None, ctypes.c_void_p(st.buf), 6, None)
        want = torch.arange(6, dtype=torch.float64) * sum(r + 1 for r in range(world))
        ok_cb = rc == 0 and torch.equal(buf, want) and st.world == world and sync.calls == 1
        sync.begin_step()
        ok_pool = sync.arg(3, "cpu").contents.buf == st.buf          # the same slot reuses its buffer next step
        rc_bad = st.allreduce(None, ctypes.c_void_p(12345), 6, None)  # unknown buffer: reported, not raised
        ok_err = rc_bad == 1 and isinstance(sync.error, KeyError)
        out[rank] = dict(fwd=ok_fwd, dx=ok_dx, pg=ok_pg, ema=ok_ema, teeth=ok_teeth, cb=ok_cb, pool=ok_pool, err=ok_err)
    finally:
        dist.destroy_process_group()


def test_sync_bn_host_logic_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = dict(out)
    assert set(res) == {0, 1}
    for rank, flags in res.items():
        assert all(flags.values()), (rank, flags)
