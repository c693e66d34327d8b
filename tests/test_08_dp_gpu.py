"""GPU: the data-parallel train step end to end with two ranks (sharing the one GPU of the
test box, gloo backend -- RCCL refuses two ranks on one device; the code path is identical:
early all-reduce of the decoder-output gradient from inside backward, remainder at the end,
1/world folded into Adam)."""
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


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cloudaae_amd import train_cloudAAE_ycbv as T
        B, N = 4, 128                       # global batch 4 -> 2 clouds per rank
        el = T.synthetic_element(B // world, N, torch.device("cuda:0"), seed=11, rank=rank)
        el["noise"] = torch.zeros((B // world, N, 3), device="cuda")
        # (a) every rank alone: its local gradient
        solo = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B // world}, process_group=False, seed=5)
        solo.store.begin_step()
        solo.forward(el)["total_loss"].backward()
        g_local = solo.store.flat_grads.clone()
        p_init = solo.store.flat_params.clone()
        gathered = [torch.empty_like(g_local) for _ in range(world)]
        dist.all_gather(gathered, g_local)
        g_sum = sum(gathered)
        # (b) the data-parallel graph (same seed -> same initial weights, then broadcast)
        dp = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, seed=5)
        ok_world = dp.world == world and dp.local_batch == B // world and dp.exchange.early is not None
        ok_init = torch.equal(dp.store.flat_params, p_init)
        dp.train_step(el)
        g_dp = dp.store.flat_grads
        scale = float(g_sum.abs().max())
        e_grad = float((g_dp - g_sum).abs().max()) / scale
        ok_grad = e_grad <= 1e-3          # fp32 atomics order; two clouds per rank: BN over 2 samples amplifies it
        lo, hi = dp.exchange.early
        ok_early = float((g_dp[lo:hi] - g_sum[lo:hi]).abs().max()) <= 1e-3 * scale
        params = [torch.empty_like(dp.store.flat_params) for _ in range(world)]
        dist.all_gather(params, dp.store.flat_params)
        ok_same = all(torch.equal(params[0], p) for p in params)            # replicas stay bit-identical
        ok_moved = not torch.equal(dp.store.flat_params, p_init)
        # (c) the same with the recorded step: step 1 records, steps 2-3 are replays whose
        # all-reduces are host callbacks of the plan; replicas must stay bit-identical and the
        # replayed gradient must be the all-reduced sum of what each rank computes alone
        rp = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, seed=5, replay=True)
        for _ in range(3):
            rp.train_step(el)
        ok_replayed = rp.replay and rp._plan is not None
        params = [torch.empty_like(rp.store.flat_params) for _ in range(world)]
        dist.all_gather(params, rp.store.flat_params)
        ok_rsame = all(torch.equal(params[0], p) for p in params)
        solo2 = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B // world}, process_group=False, seed=5)
        solo2.store.load_state_dict(rp.store.state_dict())
        snap = rp.store.flat_params.clone()
        rp.train_step(el)                                   # 4th step (replay): its gradients ...
        with torch.no_grad():
            solo2.store.flat_params.copy_(snap)             # ... vs one rank alone on the same weights
        solo2.bn_decay.copy_(rp.bn_decay)
        solo2.store.begin_step()
        solo2.forward(el)["total_loss"].backward()
        g2 = [torch.empty_like(g_local) for _ in range(world)]
        dist.all_gather(g2, solo2.store.flat_grads.clone())
        g2sum = sum(g2)
        e_rgrad = float((rp.store.flat_grads - g2sum).abs().max()) / float(g2sum.abs().max())
        ok_rgrad = e_rgrad <= 2e-3
        out[rank] = dict(world=ok_world, init=ok_init, grad=ok_grad, early=ok_early, same=ok_same, moved=ok_moved,
                         replayed=ok_replayed, rsame=ok_rsame, rgrad=ok_rgrad, e_grad=e_grad, e_rgrad=e_rgrad)
    finally:
        dist.destroy_process_group()


def _run_once():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = dict(out)
    assert set(res) == {0, 1}
    for rank, flags in res.items():
        bad = [k for k, v in flags.items() if v is False]
        if bad:
            return (rank, bad, flags)
    return None


def test_data_parallel_two_ranks(hip):
    """(Both ranks share the test box's one GPU -- not the deployment.  Rounds 4 and 5 repeated a failed first attempt: about one
    step in a hundred came out with a wrong first kNN under two processes.  Round 6 found the instruction -- a packed-fp32 add
    with op_sel, profiles/notes_two_processes_one_gpu.md -- and the library no longer contains it (tests/test_isa_rules.py):
    one attempt, no retry.)"""
    bad = _run_once()
    assert bad is None, bad
