"""GPU: SyncBN end to end -- two ranks (sharing the one GPU of the test box, gloo backend: RCCL refuses two
ranks on one device; the code path is identical) with TrainGraph(sync_bn=True), each on its half of a batch,
against ONE rank running the whole batch: the reference is single-GPU, so its batch-norm moments are over
the whole batch (utils/tf_util.py:492, :514-555), and an N-rank run must reproduce exactly that.  Compared:
the three losses (mean over ranks = the global mean), every gradient after the exchange, the BN moving
averages, and the weights after the optimiser -- eagerly and through the recorded step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, B, N, replay, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cloudaae_amd import train_cloudAAE_ycbv as T
        dev = torch.device("cuda:0")
        whole = T.synthetic_element(B, N, dev, seed=21)              # the GLOBAL batch, identical on every rank
        whole["noise"] = torch.randn((B, N, 3), generator=torch.Generator(device=dev).manual_seed(3), device=dev) * 0.001
        lo, hi = rank * B // world, (rank + 1) * B // world
        mine = {k: v[lo:hi].contiguous() for k, v in whole.items()}
        keys = ("xyz_loss", "trans_loss", "axag_loss", "total_loss")

        # one rank, the whole batch: what the reference computes
        solo = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, process_group=False, seed=9)
        # two ranks, SyncBN
        dp = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, seed=9, sync_bn=True, replay=replay)
        ok_setup = dp.sync_bn and dp.world == world and torch.equal(dp.store.flat_params, solo.store.flat_params)
        # and, for contrast, two ranks with per-rank statistics
        local = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, seed=9)

        res = {}
        for step in range(steps):
            if step:
                # every compared step starts from ONE state: rank 0's single-rank graph (two runs of the same
                # step differ by fp32-atomics round-off, which Adam turns into +-lr moves -- also between the
                # single-rank graphs of the two processes)
                with torch.no_grad():
                    for src, dst in ((solo.store.flat_params, dp.store.flat_params),
                                     (solo.store.flat_state, dp.store.flat_state), (solo.adam_m, dp.adam_m),
                                     (solo.adam_v, dp.adam_v)):
                        dist.broadcast(src, 0)
                        dst.copy_(src)
            o1 = solo.train_step(whole)
            o2 = dp.train_step(mine)
            dp.bn_sync.check()
            losses = torch.tensor([float(o2[k]) for k in keys], dtype=torch.float64)
            dist.all_reduce(losses)
            losses /= world
            want = torch.tensor([float(o1[k].detach()) for k in keys], dtype=torch.float64)
            e_loss = float(((losses - want).abs() / want.abs().clamp_min(1.0)).max())
            g1, g2 = solo.store.flat_grads, dp.store.flat_grads / world
            e_grad = float((g1 - g2).abs().max() / g1.abs().max())
            e_grad_l2 = float((g1 - g2).norm() / g1.norm())
            e_state = float((solo.store.flat_state - dp.store.flat_state).abs().max() /
                            solo.store.flat_state.abs().max())
            # (Adam turns a round-off-sized difference of a near-zero gradient into a +-lr move)
            diff = (solo.store.flat_params - dp.store.flat_params).abs()
            res["step%d" % step] = dict(e_loss=e_loss, e_grad=e_grad, e_grad_l2=e_grad_l2, e_state=e_state,
                                        moved=float((diff > 1e-5).float().mean()))
        o3 = local.train_step(mine)
        e_local = float((solo.store.flat_state - local.store.flat_state).abs().max() /
                        solo.store.flat_state.abs().max())
        params = [torch.empty_like(dp.store.flat_params) for _ in range(world)]
        dist.all_gather(params, dp.store.flat_params)
        states = [torch.empty_like(dp.store.flat_state) for _ in range(world)]
        dist.all_gather(states, dp.store.flat_state)
        out[rank] = dict(setup=ok_setup, res=res, e_local_bn=e_local, calls=dp.bn_sync.calls,
                         same_params=all(torch.equal(params[0], p) for p in params),
                         same_state=all(torch.equal(states[0], s) for s in states),
                         replayed=(not replay) or (dp.replay and dp._plan is not None and not dp._plan.foreign_ops))
    finally:
        dist.destroy_process_group()


def _compare_once(B, N, replay, steps):
    """One two-process run; returns None when every bound holds, else the first violated (rank, what, numbers)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, B, N, replay, steps), nprocs=2, join=True)
    res = dict(out)
    assert set(res) == {0, 1}
    for rank, r in res.items():
        print(rank, r)
        assert r["setup"] and r["replayed"], (rank, r)
        # 11 batch-norm layers x (forward + backward) all-reduces per training step
        assert len(r["res"]) == steps
        assert r["calls"] == 22 * steps, (rank, r["calls"])
        assert r["same_params"] and r["same_state"], (rank, r)    # replicas stay bit-identical
        assert r["e_local_bn"] > 1e-3, (rank, r)                  # per-rank statistics do NOT reproduce it
        for step, e in r["res"].items():
            # north-star loss tolerance, N-rank vs 1-rank; moving averages: the same global moments.
            # Gradients: fp32 round-off through BN-coupled layers; a Chamfer near-tie that flips between the two runs
            # moves one point's whole gradient.  Measured over 300 steps (profiles/notes_two_processes_one_gpu.md):
            # largest entry difference median 2.7e-6 of the largest gradient entry, L2 distance 3e-6, a near-tie step
            # (one in five) up to 2e-3 in both; parameters moved by more than 1e-5: median 0.05 %, 1.8 % at a near-tie.
            ok = (e["e_loss"] < 1e-5 and e["e_state"] < 1e-5 and e["e_grad"] < 5e-3 and e["e_grad_l2"] < 5e-3 and
                  e["moved"] < 0.03)
            if not ok:
                return (rank, step, e)
    return None


@pytest.mark.parametrize("B,N,replay,steps", [(8, 128, False, 1), (16, 256, False, 3), (16, 256, True, 3),
                                               (64, 128, True, 3)])
def test_sync_bn_two_ranks_equal_one_rank_of_the_global_batch(hip, B, N, replay, steps):
    """Two ranks with SyncBN against one rank on the global batch.  BOTH RANKS SHARE THE ONE GPU of the test box, which is
    not the deployment (one rank per GPU).  Rounds 4 and 5 repeated a failed first attempt here: about one step in a hundred of
    two processes on one GPU came out with a wrong first kNN.  Round 6 found the instruction (a packed-fp32 add with op_sel
    in the candidate norms, profiles/notes_two_processes_one_gpu.md) and the library no longer contains it
    (tests/test_isa_rules.py): ONE attempt, no retry, no xfail canary."""
    bad = _compare_once(B, N, replay, steps)
    assert bad is None, bad
