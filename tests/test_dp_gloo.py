"""CPU: the N>1 path's host logic with world_size 2 and 8 over gloo -- flat-buffer gradient
exchange (early bucket + remainder), averaging scale, parameter broadcast, batch sharding,
and the VariableStore's flat layout.  No HIP compute is involved (there is no GPU here)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
        from cloudaae_amd.utils.grad_exchange import GradExchange, shard_range
        n = 1000
        # (1) early bucket in the middle, sent before the rest
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        ex = GradExchange(g, early=(100, 700))
        ex.early_ready()
        ex.early_ready()          # idempotent within a step
        ex.finish()
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok1 = torch.equal(g, want) and ex.scale == 1.0 / world
        # (2) a second step on the same object, early hook never fired -> finish sends everything
        g.copy_(torch.full((n,), float(rank + 1)))
        ex.finish()
        ok2 = torch.equal(g, torch.full((n,), float(sum(r + 1 for r in range(world)))))
        # (3) no early range at all
        h = torch.full((17,), float(rank))
        ex2 = GradExchange(h)
        ex2.finish()
        ok3 = torch.equal(h, torch.full((17,), float(sum(range(world)))))
        # (4) identical weights after broadcast
        p = torch.full((5,), float(rank + 7))
        ex2.broadcast_params(p)
        ok4 = torch.equal(p, torch.full((5,), 7.0))
        # (5) sharding + DP semantics: averaged shard gradients == gradient of the global mean
        lo, hi = shard_range(8, world, rank)
        x = torch.arange(8, dtype=torch.float32)
        w = torch.tensor([0.5], requires_grad=True)
        loss = ((w * x[lo:hi] - 1.0) ** 2).mean()
        loss.backward()
        gw = w.grad.clone()
        GradExchange(gw).finish()
        w2 = torch.tensor([0.5], requires_grad=True)
        ((w2 * x - 1.0) ** 2).mean().backward()
        ok5 = torch.allclose(gw / world, w2.grad, rtol=1e-6) and (hi - lo) == 8 // world
        # (6) an early range fed by several variables: the piece leaves at the LAST of their hooks, once,
        # and the count starts over with the next step
        g6 = torch.arange(n, dtype=torch.float32) * (rank + 1)
        ex6 = GradExchange(g6, early=(400, n), early_count=3)
        sent = []
        for step in range(2):
            for call in range(3):
                ex6.early_ready()
                sent.append(ex6._early_sent)
            ex6.finish()
            if step == 0:
                ok6 = torch.equal(g6, want)
                g6.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        ok6 = ok6 and torch.equal(g6, want) and sent == [False, False, True] * 2
        out[rank] = all([ok1, ok2, ok3, ok4, ok5, ok6])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_grad_exchange(world):
    """world 8 = the deployment's rank count (BASELINE configs[3]: 8 x 128 clouds): shard ranges, the early bucket, the
    remainder and the averaging scale at eight ranks -- the cheapest rehearsal of the run only the driver can make."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}


def _replica_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        import types
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        g = torch.Generator().manual_seed(3)
        params, grads = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
        graph = types.SimpleNamespace(device=torch.device("cpu"),
                                      store=types.SimpleNamespace(flat_params=params.clone(), flat_grads=grads.clone()))
        same = bench.replica_check(graph, 1.0 + 0.25 * rank, world)
        # one ulp in one weight of ONE rank must be seen (and a permutation of equal values must not matter)
        if rank == 1:
            graph.store.flat_params[17] = torch.nextafter(graph.store.flat_params[17], torch.tensor(10.0))
        diff = bench.replica_check(graph, 1.0, world)
        graph.store.flat_params.copy_(params.flip(0) if rank == 1 else params)
        perm = bench.replica_check(graph, 1.0, world)
        out[rank] = (same["identical"], same["loss_rank_spread"], diff["identical"], perm["identical"],
                     len(same["checksums"]["flat_params"]), bench.bit_checksum(torch.tensor([1.0, -1.0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_replica_check(world):
    """bench.py's N > 1 self-check: identical replicas pass, a one-ulp difference on one rank fails, the per-rank loss
    spread is reported (world 2 and 8 over gloo)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_replica_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    want = (True, 0.25 * (world - 1), False, True, world, 0x3f800000 + 0xbf800000 - (1 << 32))
    assert dict(out) == {r: want for r in range(world)}


def test_shard_range_errors():
    from cloudaae_amd.utils.grad_exchange import shard_range
    assert shard_range(1024, 8, 3) == (384, 512)
    with pytest.raises(ValueError):
        shard_range(10, 4, 0)


def test_variable_store_flat_layout_cpu():
    """Host logic of the store: reference names, creation order, 16-byte aligned offsets,
    gradient views, state_dict round trip (runs on CPU tensors)."""
    from cloudaae_amd.utils.variables import VariableStore
    st = VariableStore(device="cpu", seed=1)
    with st.variable_scope("dgcnn1"):
        w = st.get_variable("weights", [1, 1, 48, 64], VariableStore.xavier_uniform(48, 64))
        b = st.get_variable("biases", [64], VariableStore.constant(0.0))
        with st.variable_scope("bn"):
            st.get_variable("beta", [3], VariableStore.constant(0.0))
            st.get_variable("moments/Squeeze/ExponentialMovingAverage", [3], VariableStore.constant(0.0),
                            trainable=False)
    assert list(st.vars) == ["dgcnn1/weights", "dgcnn1/biases", "dgcnn1/bn/beta",
                             "dgcnn1/bn/moments/Squeeze/ExponentialMovingAverage"]
    lim = (6.0 / (48 + 64)) ** 0.5
    assert float(w.data.abs().max()) <= lim and float(w.data.abs().max()) > 0.8 * lim
    with st.variable_scope("dgcnn1"):
        assert st.get_variable("weights", [1, 1, 48, 64], None) is w          # AUTO_REUSE
        with pytest.raises(ValueError):
            st.get_variable("weights", [1, 1, 48, 32], None)
    before = st.state_dict()
    st.flatten()
    assert st.num_params == 48 * 64 + 64 + 3
    assert all(o % 4 == 0 for o in st.offsets.values())                        # 16-byte aligned
    assert st.offsets["dgcnn1/bn/beta"] == 48 * 64 + 64
    for n, t in st.state_dict().items():
        assert torch.equal(t, before[n])
    assert w.data.data_ptr() == st.flat_params.data_ptr() and w.grad.data_ptr() == st.flat_grads.data_ptr()
    assert w.data.requires_grad and w.data.grad is w.grad
    st.flat_params.add_(1.0)                                                    # an optimiser update is visible
    assert torch.equal(st.vars["dgcnn1/biases"].data.detach(), torch.ones(64))
    with pytest.raises(RuntimeError):
        st.get_variable("late", [1], VariableStore.constant(0.0))
    sd = st.state_dict()
    sd["dgcnn1/biases"] = torch.full((64,), 3.0)
    st.load_state_dict(sd)
    assert float(st.flat_params[48 * 64]) == 3.0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) must start N ranks itself, as children of a
    parent that never touches the GPU, and exit with their code.  Checked on CPU with CLOUDAAE_BENCH_DRYRUN=1 (every
    rank reports who it is and stops before any GPU call)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CLOUDAAE_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1] and all(l["world"] == 2 for l in lines), r.stdout[-2000:]
    assert all(l["per_gpu_batch"] == 128 for l in lines)            # the per-GPU shape of BASELINE configs[3]
    # one GPU: no children, BASELINE configs[1]'s batch
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")], env=env, capture_output=True, text=True,
                        timeout=300)
    l1 = [json.loads(l) for l in r1.stdout.splitlines() if l.startswith("{")]
    assert r1.returncode == 0 and len(l1) == 1 and l1[0]["world"] == 1 and l1[0]["per_gpu_batch"] == 32
    # under a launcher it is a rank itself: no second level of children
    env2 = dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2,
                        capture_output=True, text=True, timeout=300)
    l2 = [json.loads(l) for l in r2.stdout.splitlines() if l.startswith("{")]
    assert r2.returncode == 0 and len(l2) == 1 and l2[0]["rank"] == 1
