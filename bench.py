"""bench.py -- throughput of the CloudAAE training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL -- either launched by torch.distributed.run, which sets
    WORLD_SIZE/RANK/LOCAL_RANK, or by bench.py itself: with WORLD_SIZE unset it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before
    anything here touches the GPU, relays rank 0's JSON line and exits with the child's code)

A "step" is one pass of the hot path over one batch of synthetic input, i.e. one
iteration of the reference's session loop (train_cloudAAE_ycbv.py:350-368): BN-decay
schedule, input assembly, DGCNN encoder + decoder + pose heads, Chamfer / translation /
SO(3) losses, backward, TF-Adam (+ gradient all-reduce for N > 1).  Workload at N=1 is
BASELINE.json configs[1]: all 21 classes, batch 32, 1024 points, fp32.  For N > 1 the workload is
BASELINE.json configs[3]'s per-GPU shape: 128 clouds per GPU (global batch 1024 on 8 GPUs), fixed as
N grows ("weak"); `--per-gpu-batch` overrides either.  Because the N=1 line is a different (smaller)
batch than the N>1 lines, every N>1 line also carries `one_rank_same_shape`: rank 0 alone, no
collectives, the same 128-cloud batch -- the honest denominator for a scaling efficiency.
Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the dominant kernel = the top row of the step's kernel trace: the kNN over 64 feature
                 channels (knn64_wide_kernel, three launches per step), algorithmic flops 2 B N^2 64 against the
                 fp32 matrix pipe; timed live with HIP events on the launch stream over the timed region
  roofline_agg_fwd : the dgcnn_agg forward product, algorithmic flops (frac, frac_algorithmic) kept apart
                 from the bf16 pipe's utilisation by the six issued piece products (frac_issued)
  roofline_edgeconv: the four edge-convolution blocks, forward + backward, against HBM
  cpu_baseline : the CPU oracle's train step timed on this box's host cores on a bounded
                 sample (rank 0, N=1 only)
  comm (N > 1) : ranks_seen (an RCCL all-reduce of ones), allreduce_exposed_ms (HIP events on the
                 compute stream around the end-of-backward exchange: the time the step waits for
                 gradients, the early fully-connected piece having been reduced behind backward)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
HBM_PEAK_GBS = 8000.0
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense (the headline figure with 2:1 sparsity is never used)


def cpu_baseline(num_point, sample_batch, steps=1):
    """CPU oracle (oracle/model_oracle.py: torch-CPU restatement + C kNN/Chamfer, all host
    cores) on a bounded sample of the same workload.  Checker code used as the reported
    baseline only (kind "port": TensorFlow 1.12 cannot run in this image)."""
    from oracle import model_oracle as MO
    # torch's intra-op pool stops scaling (and then regresses) well before the 256 hardware
    # threads of the GPU box on these layer sizes (measured there with tools/cpu_threads.py:
    # 8: 8.8, 16: 9.7, 32: 9.2, 64: 5.6, 128: 1.8, 256: 0.15 clouds/s).
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    V = MO.Vars(seed=0)
    batch = MO.synthetic_batch(sample_batch, num_point, seed=123456789)
    opt = MO.AdamTF()
    warm = MO.synthetic_batch(2, num_point, seed=1)                # warm-up: allocations, thread pools
    MO.train_step(warm, V, opt, 0, num_point, 2)
    t0 = time.time()
    for i in range(steps):
        MO.train_step(batch, V, opt, i + 1, num_point, sample_batch)
    dt = time.time() - t0
    return {"value": round(sample_batch * steps / dt, 3), "unit": "clouds/s", "cores": cores,
            "kind": "port",
            "sample": "%d train step(s) of batch %d, N=%d (same graph), %.1f s"
                      % (steps, sample_batch, num_point, dt)}


def chamfer_cpu_rate(n, m, clouds=4):
    """The reference's CPU Chamfer next to the kernel: its own nnsearch loops (tf_nndistance.cpp:21-43,
    compiled from the reference's lines into oracle/_ref, kind "reference") when that library
    travelled with the snapshot, else the oracle's restatement (kind "port"); single-threaded, as
    tf_nndistance.cpp:79-80 runs them, on a few clouds of the same shape and seed."""
    import numpy as np
    from oracle import native as O
    rng = np.random.default_rng(100)
    a = rng.standard_normal((clouds, n, 3)).astype(np.float32)
    c = rng.standard_normal((clouds, m, 3)).astype(np.float32)
    fn, kind = (O.ref_nn_distance, "reference") if O.have_ref() else (lambda x, y: O.nn_distance(x, y, threads=1), "port")
    fn(a[:1], c[:1])
    t0 = time.time()
    fn(a, c)
    dt = time.time() - t0
    return {"clouds/s": round(clouds / dt, 2), "Gpairs/s": round(2.0 * clouds * n * m / dt / 1e9, 3), "cores": 1,
            "kind": kind, "sample": "%d clouds of %dx%d, %.2f s" % (clouds, n, m, dt)}


def cpu_table(num_point):
    """BASELINE.md section 2's CPU legs that are not a train step (rank 0, N=1, bounded samples of ~1 s each):
      C1  Chamfer forward AND backward on the reference's own loops (tf_nndistance.cpp:21-43 / :126-163, compiled from
          the reference's lines into oracle/_ref: kind "reference"; the oracle's restatement, kind "port", when that
          library did not travel), single-threaded as the reference op runs them (:79-80), and on all host cores
          with the clouds of the batch spread over threads (clouds are independent: one call per cloud; ctypes
          releases the GIL) -- at the train shape [B,4N]x[B,4N] and the reference's benchmark shape
          [32,16384]x[32,1024] (tf_nndistance.py:48-49), randn, seed 100;
      C3  farthest point sampling 4N -> N, the restatement of tf_sampling_g.cu:105-170 (the reference has no CPU
          kernel for it: kind "port"), 1 thread and all cores (OpenMP over the batch)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import native as O
    cores = min(os.cpu_count() or 1, 64)
    have_ref = O.have_ref()
    fwd = O.ref_nn_distance if have_ref else (lambda x, y: O.nn_distance(x, y, threads=1))
    bwd = O.ref_nn_distance_grad if have_ref else (lambda *a: O.nn_distance_grad(*a, threads=1))
    kind = "reference" if have_ref else "port"
    rng = np.random.default_rng(100)
    rows = []

    def timed(fn, reps=1):
        fn()
        t0 = time.time()
        for _ in range(reps):
            fn()
        return (time.time() - t0) / reps

    for n, m, c1 in ((4 * num_point, 4 * num_point, 2), (16384, 1024, 2)):
        clouds = max(cores, c1)
        a = rng.standard_normal((clouds, n, 3)).astype(np.float32)
        c = rng.standard_normal((clouds, m, 3)).astype(np.float32)
        d1, i1, d2, i2 = fwd(a[:c1], c[:c1])
        g1, g2 = np.ones_like(d1), np.ones_like(d2)
        t_f = timed(lambda: fwd(a[:c1], c[:c1]))
        t_b = timed(lambda: bwd(a[:c1], c[:c1], g1, i1, g2, i2), reps=20)
        with ThreadPoolExecutor(cores) as pool:
            t_fa = timed(lambda: list(pool.map(lambda j: fwd(a[j:j + 1], c[j:j + 1]), range(clouds))))
        shape = "[B,%d,3]x[B,%d,3]" % (n, m)
        rows.append({"leg": "C1 chamfer forward", "shape": shape, "clouds/s": round(c1 / t_f, 2), "cores": 1, "kind": kind,
                     "Gpairs/s": round(2.0 * c1 * n * m / t_f / 1e9, 3), "sample": "%d clouds" % c1})
        rows.append({"leg": "C1 chamfer forward, all cores", "shape": shape, "clouds/s": round(clouds / t_fa, 2),
                     "cores": cores, "kind": kind, "Gpairs/s": round(2.0 * clouds * n * m / t_fa / 1e9, 3),
                     "sample": "%d clouds, one per thread" % clouds})
        rows.append({"leg": "C1b chamfer backward (tf_nndistance.cpp:126-163)", "shape": shape,
                     "clouds/s": round(c1 / t_b, 1), "cores": 1, "kind": kind, "sample": "%d clouds x 20" % c1})
    n = 4 * num_point
    x = rng.standard_normal((cores, n, 3)).astype(np.float32)
    t1 = timed(lambda: O.farthest_point_sample(num_point, x[:2], threads=1))
    ta = timed(lambda: O.farthest_point_sample(num_point, x, threads=cores))
    rows.append({"leg": "C3 farthest point sampling", "shape": "[B,%d,3] -> %d" % (n, num_point),
                 "clouds/s": round(2 / t1, 2), "cores": 1, "kind": "port", "sample": "2 clouds"})
    rows.append({"leg": "C3 farthest point sampling, all cores", "shape": "[B,%d,3] -> %d" % (n, num_point),
                 "clouds/s": round(cores / ta, 2), "cores": cores, "kind": "port", "sample": "%d clouds" % cores})
    return rows


def chamfer_kernel_rate(batch, n, m, iters=20, distinct=None, hint=False):
    """The second half of BASELINE's metric: Chamfer nn_distance forward kernel rate.
    Algorithmic bytes = B*(n+m)*20 (12 B read + 4 B dist + 4 B idx per point): arithmetic-bound
    by three orders of magnitude.  Large clouds take nn_distance_filter_kernel<SPLIT>: the nearest candidate is
    searched with scores on the bf16 matrix cores -- error-free three-piece splits, two v_mfma_f32_32x32x16_bf16
    (32 cycles each) per 32 x 32 pairs -- and the reference's un-fused arithmetic decides among the candidates of
    the two best units (bit parity with the CPU reference).  `frac_of_bf16_matrix_pipe_bound` prices the pipe the
    scores run on: 1024 pairs per 64 cycles per SIMD = 39.3 T pairs/s (27 us at [32, 4096]^2); the vector digest of
    the scores (8 v_min3 per 512 pairs, ~25 us, overlapping only partly) is what the kernel adds to it:
    `frac_of_pipe_plus_digest_floor` prices against both."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    g = torch.Generator(device="cuda").manual_seed(100)       # tf_nndistance.py:45-46 seeds
    a = torch.randn((batch, n, 3), generator=g, device="cuda")
    c = torch.randn((batch, m, 3), generator=g, device="cuda")
    if distinct is not None:
        # the reference's training targets: `distinct` visible points, then random re-draws of them
        # (utils/hidden_point_removal.py:38-40) -- every target point exists m / distinct times
        pick = torch.randint(0, distinct, (batch, m - distinct), generator=g, device="cuda")
        c[:, distinct:] = torch.gather(c[:, :distinct], 1, pick[:, :, None].expand(-1, -1, 3))
    d2 = None
    if hint:
        # what the on-line synthesis knows about its targets (cloudaae_hidden_point_removal_rows: num_vis, row_src) and the
        # train step passes on: the search visits the distinct points only (cloudaae_nn_distance_prefix), same results
        rows = torch.arange(distinct, device="cuda", dtype=torch.int32)[None].expand(batch, -1)
        d2 = (torch.full((batch,), distinct, dtype=torch.int64, device="cuda"),
              torch.cat([rows, pick.to(torch.int32)], 1).contiguous())
    # the C-ABI call itself, outputs allocated once: through the autograd wrapper a 50 us kernel would be timed at the
    # host's pace (the wrapper is checked against this call once, below)
    from cloudaae_amd import _lib
    L = _lib.lib()._cdll
    d1 = torch.empty((batch, n), device="cuda"); i1 = torch.empty((batch, n), dtype=torch.int32, device="cuda")
    dd2 = torch.empty((batch, m), device="cuda"); i2 = torch.empty((batch, m), dtype=torch.int32, device="cuda")
    P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731

    def go():
        rc = L.cloudaae_nn_distance_prefix(batch, n, P(a), m, P(c), P(d2[0]) if d2 else None, P(d2[1]) if d2 else None,
                                           P(d1), P(i1), P(dd2), P(i2), _lib.stream())
        assert rc == 0, L.cloudaae_last_error()
    ref = tf_nndistance.nn_distance(a, c, distinct2=d2)
    go()
    assert all(torch.equal(x, y) for x, y in zip(ref, (d1, i1, dd2, i2)))
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    pairs = 2.0 * batch * n * m
    bound = 256 * 4 * 2.4e9 * 1024 / 64             # two 32-cycle bf16 MFMAs per 1024 pairs per SIMD
    digest = 256 * 4 * 2.4e9 * 512 / (8 * 4)        # the digest: 8 v_min3 (4 cycles each) per 512 pairs per SIMD
    floor = 1.0 / (1.0 / bound + 1.0 / digest)
    return {"shape": "[%d,%d,3]x[%d,%d,3]%s" % (batch, n, batch, m, "" if distinct is None else
                                                 " (target = %d points + re-draws, as the reference pads%s)"
                                                 % (distinct, "; the search is told which rows are re-draws" if hint else "")),
            "us_per_launch": round(sec * 1e6, 2),
            "GB/s": round(batch * (n + m) * 20 / sec / 1e9, 3), "Tpairs/s": round(pairs / sec / 1e12, 3),
            "clouds/s": round(batch / sec, 1), "frac_of_bf16_matrix_pipe_bound": round(pairs / sec / bound, 4),
            "frac_of_pipe_plus_digest_floor": round(pairs / sec / floor, 4)}


def fps_kernel_rate(batch, n, m, iters=5):
    """Farthest point sampling n -> m (the op the inference path runs on the 4N reconstructed points,
    evaluate_cloudAAE_ycbv.py:450): m - 1 dependent rounds, one workgroup per cloud -- a latency chain, priced per round."""
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    g = torch.Generator(device="cuda").manual_seed(100)
    x = torch.randn((batch, n, 3), generator=g, device="cuda")
    for _ in range(2):
        tf_sampling.farthest_point_sample(m, x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        tf_sampling.farthest_point_sample(m, x)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    return {"shape": "[%d,%d,3] -> %d" % (batch, n, m), "us_per_launch": round(sec * 1e6, 1),
            "us_per_round": round(sec * 1e6 / max(m - 1, 1), 3), "clouds/s": round(batch / sec, 1)}


def chamfer_train_rate(batch=32, n=16384, m=1024, iters=20):
    """The reference's own micro-benchmark, as it runs it (tf_ops/nn_distance/tf_nndistance.py:45-66):
    an SGD step on loss = sum(dist1) + sum(dist2) with xyz1 the variable, i.e. per iteration the forward
    search both ways, the two reductions, NnDistanceGrad with unit upstream gradients and
    xyz1 -= 0.05 * grad.  Same shapes and seed; everything through the C ABI."""
    from cloudaae_amd import _lib
    L, P, S = _lib.lib(), _lib.ptr, _lib.stream
    g = torch.Generator(device="cuda").manual_seed(100)
    a = torch.randn((batch, n, 3), generator=g, device="cuda")
    c = torch.randn((batch, m, 3), generator=g, device="cuda")
    d1, d2 = torch.empty((batch, n), device="cuda"), torch.empty((batch, m), device="cuda")
    i1 = torch.empty((batch, n), dtype=torch.int32, device="cuda")
    i2 = torch.empty((batch, m), dtype=torch.int32, device="cuda")
    one1, one2 = torch.ones_like(d1), torch.ones_like(d2)
    ga, gc = torch.empty_like(a), torch.empty_like(c)
    s1, s2 = torch.empty((), device="cuda"), torch.empty((), device="cuda")
    ws = torch.empty(int(L.cloudaae_mean_workspace_bytes()) // 8 + 1, dtype=torch.float64, device="cuda")

    def step():
        _lib.check(L.cloudaae_nn_distance(batch, n, P(a), m, P(c), P(d1), P(i1), P(d2), P(i2), S()), "nn_distance")
        _lib.check(L.cloudaae_mean_f32(batch * n, P(d1), P(s1), P(ws), S()), "mean")        # tf.reduce_sum(reta)
        _lib.check(L.cloudaae_mean_f32(batch * m, P(d2), P(s2), P(ws), S()), "mean")        # tf.reduce_sum(retc)
        _lib.check(L.cloudaae_nn_distance_grad(batch, n, P(a), m, P(c), P(one1), P(i1), P(one2), P(i2), P(ga), P(gc),
                                               S()), "nn_distance_grad")
        _lib.check(L.cloudaae_sgd(batch * n * 3, P(a), P(ga), 0.05, 1.0, S()), "sgd")     # GradientDescentOptimizer(0.05)
    for _ in range(3):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    return {"what": "tf_nndistance.py:45-66: forward + sums + NnDistanceGrad + SGD on xyz1",
            "shape": "[%d,%d,3]x[%d,%d,3]" % (batch, n, batch, m), "us_per_iteration": round(sec * 1e6, 2),
            "clouds/s": round(batch / sec, 1)}


def git_blob_sha(path):
    """`git hash-object` of a file, without git (the GPU box has no .git)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def measured_traffic(B, N, kernel_key):
    """HBM bytes per launch of a named kernel from the rocprofv3 PMC passes recorded in
    profiles/roofline_traffic.json -- used ONLY if that file was collected on the very sources that are
    running (git blob hashes of the kernel's source files match) and on this workload; else None."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if not os.path.exists(tpath):
        return None
    rec = json.load(open(tpath)).get(kernel_key)
    if not rec or rec.get("workload") != "B=%d,N=%d" % (B, N):
        return None
    for rel, sha in rec.get("source_blobs", {}).items():
        f = os.path.join(ROOT, rel)
        if not os.path.exists(f) or git_blob_sha(f) != sha:
            return None
    return rec.get("traffic_bytes_per_launch")


def bit_checksum(t):
    """64-bit sum of the 32-bit patterns of a float32 tensor (order independent, exact)."""
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())


def replica_check(graph, loss, world):
    """After the timed steps: gather (checksum of flat_params, checksum of the post-exchange flat_grads, loss) from every
    rank.  Data-parallel replicas start from broadcast weights and apply the same all-reduced gradient, so the first two
    must agree bit for bit on every rank; the per-rank losses differ (different shards) and their spread is reported."""
    mine = torch.tensor([bit_checksum(graph.store.flat_params), bit_checksum(graph.store.flat_grads)], dtype=torch.int64,
                        device=graph.device)
    lmine = torch.tensor([loss], dtype=torch.float64, device=graph.device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        allc = [torch.zeros_like(mine) for _ in range(world)]
        alll = [torch.zeros_like(lmine) for _ in range(world)]
        dist.all_gather(allc, mine)
        dist.all_gather(alll, lmine)
    else:
        allc, alll = [mine], [lmine]
    sums = [[int(v) for v in c.tolist()] for c in allc]
    losses = [float(v) for v in alll]
    return {"identical": all(c == sums[0] for c in sums), "checksums": {"flat_params": [c[0] for c in sums],
                                                                          "flat_grads": [c[1] for c in sums]},
            "loss_rank_spread": round(max(losses) - min(losses), 6)}


def launch_children(args):
    """--gpus N without a launcher: run N ranks as children of THIS process (which has not touched the
    GPU and will not), relay their output, return the exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_steps(graph, el, steps, warmup, world, sites):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    from cloudaae_amd.utils import _functions as F
    for _ in range(warmup):
        graph.train_step(el)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    for k in sites:
        F.TIMED_SITES[k].clear()
    # one HIP event per step on the launch stream: the spread of the single steps (min / median / max) goes into the
    # line next to the mean, so that a box effect on a 30 ms timed region can be told from a regression
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        F.TIMED_ON = i % 4 == 0          # the live kernel timings sample one step in four of the timed region
        out = graph.train_step(el)
        marks[i + 1].record()
    F.TIMED_ON = True
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=graph.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    spread = {"step_ms_min": round(per_step[0], 4), "step_ms_median": round(per_step[len(per_step) // 2], 4),
              "step_ms_max": round(per_step[-1], 4)} if per_step else {}
    return elapsed, out, spread


def site_ms(events):
    ms = [events[i].elapsed_time(events[i + 1]) for i in range(0, len(events) - 1, 2)]
    return (sum(ms) / len(ms) if ms else 0.0), len(ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--per-gpu-batch", type=int, default=0, help="clouds per GPU (0 = 32 on one GPU: BASELINE "
                    "configs[1]; 128 on several: the per-GPU shape of configs[3])")
    ap.add_argument("--num-point", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10, help="neighbours of the edge convolution (BASELINE configs[4]: 20)")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="train steps of the CPU-baseline sample")
    ap.add_argument("--gemm-dtype", default="bf16x3", choices=["f32", "bf16", "bf16x3"], help="bf16x3 (default): an fp32 step "
                    "whose three dgcnn_agg products run as error-free 3 x bf16 split products on the bf16 matrix cores (fp32 "
                    "accuracy: six exact piece products, fp32 accumulate); f32: those products on the fp32 matrix cores; bf16: "
                    "BASELINE configs[2]'s arithmetic (dense-layer operands rounded to bf16, fp32 accumulate; everything else fp32)")
    ap.add_argument("--sync-bn", action="store_true", help="batch-norm moments over the GLOBAL batch (all ranks)")
    ap.add_argument("--step-only", action="store_true", help="skip the Chamfer kernel micro-benchmarks and the CPU "
                    "baseline (profiling runs: only the train step's kernels in the trace)")
    ap.add_argument("--config5", action="store_true", help="BASELINE configs[4]: N=4096, k=20 and the on-line synthesis "
                    "(pose -> transform -> occluder -> spherical flip -> hidden point removal x2, train...:96-117) of "
                    "every batch INSIDE the timed loop, from synthetic 8192-point object models")
    ap.add_argument("--eager", action="store_true", help="step through Python/autograd every time instead of "
                    "replaying the recorded step (TrainGraph(replay=False))")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing in this process has initialised the GPU (importing torch does not).
        sys.exit(launch_children(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CLOUDAAE_BENCH_DRYRUN") == "1":      # (CPU test of the launch path: no GPU is touched)
        print(json.dumps({"dryrun": True, "rank": rank, "world": world, "local_rank": local,
                          "per_gpu_batch": args.per_gpu_batch or (32 if world == 1 else 128)}), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    force = os.environ.get("CLOUDAAE_FORCE_COLLECTIVES") == "1" and "MASTER_ADDR" in os.environ
    stdout_fd = None
    if world > 1 or force:
        # RCCL prints a version banner on STDOUT when its first communicator comes up: this process's stdout is
        # pointed at stderr until the first collective has run, so that rank 0's ONE JSON line stays alone there
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        # a rank that never shows up (or a communicator that cannot form) is a non-zero exit after two minutes with
        # the rendezvous error on stderr, not a hang for the length of the driver's own limit
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local),
                                timeout=datetime.timedelta(seconds=int(os.environ.get("CLOUDAAE_PG_TIMEOUT_S", "120"))))

    from cloudaae_amd import train_cloudAAE_ycbv as T
    from cloudaae_amd.utils import _functions as F

    if args.config5:
        args.num_point, args.k = 4096, 20
    B = args.per_gpu_batch or (32 if world == 1 else 128)
    N = args.num_point
    # live HIP events on the launch stream around: the dgcnn_agg forward GEMM, the three kNN launches over 64
    # feature channels, and (N > 1) the end-of-backward gradient exchange -- host callbacks of the recorded step
    sites = ["agg_fwd", "knn64"] + (["exchange"] if (world > 1 or force) else [])
    for k in sites:
        F.TIMED_SITES[k] = []
    F.TIMED_SITES["edgeconv"] = []        # recorded with the step, switched off until the timed region is over
    F.SITES_OFF.add("edgeconv")
    graph = T.TrainGraph({"num_point": N, "gpu": local}, {"optimizer": "adam"},
                         {"batch_size": B * world, "learning_rate": 0.0008}, replay=not args.eager,
                         gemm_dtype=args.gemm_dtype, k_neighbor=args.k, sync_bn=args.sync_bn)
    el = T.synthetic_element(B, N, graph.device, seed=123456789, rank=rank)
    graph.reuse_staged_inputs = True     # one fixed batch, resident in HBM: do not re-copy it every step
    synth = None
    if args.config5:
        # every step draws its batch from the synthesis pipeline: poses of the synthetic element, 8192-point models
        # (their + the occluder's 400 points + the viewpoint fit the LDS-resident hull test), the input cloud = the first
        # N of the visible points of model + occluder, the Chamfer target = 4N rows of the model's visible points
        # (visible points, then random re-draws: the reference's padding rule, hidden_point_removal.py:38-40)
        models = T.synthetic_object_models(T.NUM_CLASS, 8192, device=graph.device)
        poses = {k: el[k] for k in ("translation", "axisangle", "class_id")}
        graph.reuse_staged_inputs = False
        counter = [0]

        class _Synth(object):
            device = graph.device

            def train_step(self, _):
                counter[0] += 1
                return graph.train_step(T.get_small_data(poses, models, seed=counter[0], rows_org=4 * N))
        synth = _Synth()

    ranks_seen = 1
    if world > 1 or force:
        ones = torch.ones(1, device=graph.device)
        dist.all_reduce(ones)                                   # RCCL: every rank contributes 1
        ranks_seen = int(round(float(ones)))
    if stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)

    elapsed, out, spread = timed_steps(synth if synth is not None else graph, el, args.steps, args.warmup, world, sites)
    events = {k: F.TIMED_SITES.pop(k) for k in sites}
    loss = float(out["total_loss"])
    # the edge-convolution blocks are timed AFTER the timed region (sixteen event pairs per step would cost the
    # headline ~1 %): eight more steps of the same replayed plan with only that site switched on
    F.TIMED_SITES["edgeconv"].clear()
    F.SITES_OFF.discard("edgeconv")
    F.TIMED_ON = True
    runner = synth if synth is not None else graph
    for _ in range(8):
        runner.train_step(el)
    torch.cuda.synchronize()
    events["edgeconv"] = F.TIMED_SITES.pop("edgeconv")

    # N > 1 (and the forced one-rank RCCL run): the replicas must be IDENTICAL after the timed steps -- same weights, same
    # post-exchange gradient buffer on every rank (a 64-bit sum of the bit patterns of each, gathered over the ranks);
    # a stream-ordering slip between the asynchronous early all-reduce and the kernels around it shows up here
    replicas = None
    if world > 1 or force:
        replicas = replica_check(graph, float(out["total_loss"]), world)

    one_rank = None
    if world > 1:
        # the same per-GPU batch on rank 0 ALONE (no collectives, the others wait at the barrier)
        if rank == 0:
            solo = T.TrainGraph({"num_point": N, "gpu": local}, {"optimizer": "adam"},
                                {"batch_size": B, "learning_rate": 0.0008}, replay=not args.eager,
                                gemm_dtype=args.gemm_dtype, k_neighbor=args.k, process_group=False)
            solo.reuse_staged_inputs = True
            t1, _, _ = timed_steps(solo, el, args.steps, args.warmup, 1, [])
            one_rank = {"per_gpu_batch": B, "clouds/s": round(B * args.steps / t1, 2),
                        "ms_per_step": round(t1 / args.steps * 1e3, 4)}
        dist.barrier()

    if rank == 0:
        k_ms, k_n = site_ms(events["agg_fwd"])
        M, Nn, K = B * N, 1024, 320
        flops = 2.0 * M * Nn * K                      # algorithmic flops of one dgcnn_agg forward launch
        achieved = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        line = {
            "metric": "point-clouds/sec (train step, N=%d)" % N,
            "value": round(B * world * args.steps / elapsed, 2),
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "step_ms_min": spread.get("step_ms_min"), "step_ms_median": spread.get("step_ms_median"),
            "step_ms_max": spread.get("step_ms_max"),       # per-step HIP events of this rank, inside the timed region
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32" if args.gemm_dtype == "f32" else
                      "f32, the dgcnn_agg products as error-free 3 x bf16 splits on the bf16 matrix cores (fp32 accumulate)"
                      if args.gemm_dtype == "bf16x3" else "bf16 dense-layer operands, f32 accumulate and everything else"),
            "data": "synthetic",
            "config": {"workload": "CloudAAE train step: get_model_dgcnn_mean_6d, all 21 YCB classes, "
                                   "batch %d/GPU, N=%d points, k=%d, 4N-point Chamfer target, TF-Adam%s%s"
                                   % (B, N, args.k, ", SyncBN" if args.sync_bn else "",
                                      ", every batch synthesised on the GPU inside the loop (8192-point models, occluder, "
                                      "spherical flip, hidden point removal x2)" if args.config5 else ""),
                       "baseline_config": ("configs[4] (one GPU; on-line synthesis inside the timed loop)" if args.config5 else
                                           "configs[1]" if (world == 1 and B == 32 and args.gemm_dtype == "bf16x3") else
                                           "configs[1], dgcnn_agg products on the fp32 matrix cores" if (world == 1 and B == 32 and args.gemm_dtype == "f32") else
                                           "configs[2]" if (world == 1 and B == 256 and args.gemm_dtype == "bf16") else
                                           "configs[3] (128 clouds per GPU)" if (B == 128 and world > 1) else "custom"),
                       "global_batch": B * world, "per_gpu_batch": B, "num_point": N, "parallelism": "dp%d" % world,
                       "step_issue": "recorded step replay" if graph.replay else "eager",
                       "final_total_loss": round(loss, 4)},
        }
        # `roofline`: the DOMINANT kernel = the top row of the step's kernel trace at every batch size
        # (profiles/r0N_trainstep_*_kernel_stats.csv): the kNN over 64 feature channels, three launches per step
        # (layers 2-4, tf_util.py:597-632).  Algorithmic flops = the N x N x 64 inner products of every cloud
        # (2 B N^2 C) against the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s); the selection
        # that follows the products is what keeps the kernel from that bound.
        q_ms, q_n = site_ms(events["knn64"])
        qf = 2.0 * B * N * N * 64 / (q_ms * 1e-3) / 1e12 if q_ms > 0 else 0.0
        line["roofline"] = {"bound": "mfma", "kernel": "knn64_wide_kernel (cloudaae_knn, C=64, k=%d) [%d x %d x %d], "
                                                        "3 launches per step" % (args.k, B, N, N),
                            "achieved": round(qf, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(qf / FP32_MFMA_PEAK_TFLOPS, 4),
                            "traffic": measured_traffic(B, N, "knn64"),
                            "algorithmic_flops_per_launch": int(2.0 * B * N * N * 64),
                            "launch_ms": round(q_ms, 4), "launches_timed": q_n}
        # second entry: the dgcnn_agg forward product [B N x 320] x [320 x 1024] (tf_util.py:161-166).  ALGORITHMIC flops
        # 2 M N K whatever the arithmetic.  f32: on the fp32 matrix pipe.  bf16x3: six bf16 piece products are ISSUED per
        # algorithmic product, so the line carries frac_issued (6 x flops / 2.5 PFLOP/s: how busy the bf16 pipe is -- a
        # utilisation figure, NOT a roofline fraction) next to frac_algorithmic against both pipes (vs the bf16 pipe
        # it runs on; vs the fp32 pipe an fp32 kernel would be bound by: > 1 means it beats every fp32-MFMA kernel).
        agg = {"bound": "mfma", "algorithmic_tflops": round(achieved, 3), "launch_ms": round(k_ms, 4), "launches_timed": k_n,
               "algorithmic_flops_per_launch": int(flops)}
        if args.gemm_dtype == "f32":
            agg.update({"kernel": "gemm_f32_kernel dgcnn_agg forward [%d x 320] x [320 x 1024]" % M,
                        "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": measured_traffic(B, N, "agg_fwd")})
        elif args.gemm_dtype == "bf16x3":
            agg.update({"kernel": "gemm_x3s_kernel dgcnn_agg forward [%d x 320] x [320 x 1024] as 3 x bf16 splits "
                                  "(6 piece products issued per product)" % M,
                        "achieved": round(achieved, 3), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                        "frac_algorithmic": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                        "frac_algorithmic_vs_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                        "frac_issued": round(6.0 * achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                        "issued_tflops": round(6.0 * achieved, 2), "traffic": measured_traffic(B, N, "agg_fwd_x3")})
        else:
            # with bf16 operands the same product leaves the matrix pipe (2.5 PFLOP/s dense) and is bound by
            # HBM.  Activations kept as bfloat16 (F.ACT_BF16, the default): x (M x 320) and W in as bf16, y
            # (M x 1024) out as bf16; else fp32 tensors in and out (rounded on the way into LDS)
            act16 = bool(F.ACT_BF16) and M % 128 == 0 and not args.sync_bn
            esz = 2.0 if act16 else 4.0
            nbytes = esz * (M * K + K * Nn + M * Nn) + 4.0 * Nn
            gbs = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            kname = ("gemm_b16 (bf16 x, W in; bf16 y out)" if act16 else "gemm_bf16_kernel<128,128,2,2> (fp32 in / out)")
            agg.update({"bound": "hbm", "kernel": kname + " dgcnn_agg forward [%d x 320] x [320 x 1024]" % M,
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(nbytes),
                        "traffic": measured_traffic(B, N, "agg_fwd_b16" if act16 else "agg_fwd_bf16")})
        line["roofline_agg_fwd"] = agg
        # third entry: the four edge-convolution blocks, forward and backward (the calls cloudaae_edgeconv_forward /
        # _backward: their products, statistics, finalise and apply / gather kernels; the grouped weight-gradient
        # launch that follows backward is not inside).  HBM-bound by construction (the k-fold edge tensor never
        # exists); algorithmic bytes = every array of a layer read or written once per pass that needs it.
        e_ms, e_n = site_ms(events["edgeconv"])
        if e_n:
            P = B * N
            tot = 0.0
            for cin, cout in ((24, 64), (64, 64), (64, 64), (64, 128)):
                idx, pq, out_, est = 4.0 * P * args.k, 4.0 * P * 2 * cout, 4.0 * P * cout, 4.0 * P * 3 * cout
                fwd = 4.0 * P * cin + pq + (pq + idx) + (pq + idx + out_ + est)          # product | statistics | apply
                bwd = (est + out_) + (est + out_ + pq + 2 * idx + pq) + (pq + 4.0 * P * cin)   # stats | apply (dpq) | dX
                tot += fwd + bwd
            per_call = tot / 8.0                       # eight timed calls per step (4 forward + 4 backward)
            gbs = per_call / (e_ms * 1e-3) / 1e9
            line["roofline_edgeconv"] = {"bound": "hbm", "kernel": "cloudaae_edgeconv_forward + _backward, 4 layers "
                                                                   "(ec_stats / ec_apply / ec_bwd_* + their products)",
                                         "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                         "step_ms_in_edgeconv": round(e_ms * 8, 4), "calls_timed": e_n,
                                         "algorithmic_bytes_per_step": int(tot)}
        if world > 1 or force:
            x_ms, x_n = site_ms(events["exchange"])
            line["comm"] = {"backend": "nccl (RCCL)", "ranks_seen": ranks_seen,
                            "allreduce_exposed_ms": round(x_ms, 4), "exchanges_timed": x_n,
                            "bytes_per_step": int(graph.store.flat_grads.numel()) * 4,
                            "sync_bn": bool(args.sync_bn)}
        if replicas is not None:
            line["replicas_identical"] = replicas["identical"]
            line["loss_rank_spread"] = replicas["loss_rank_spread"]
            line["comm"]["replica_checksums"] = replicas["checksums"]
        if one_rank is not None:
            line["one_rank_same_shape"] = one_rank
        if world == 1 and not args.step_only:
            # "Chamfer kernel GB/s": the train shape (n = m = 4N) and the reference's own
            # micro-benchmark shape (tf_nndistance.py:48-49), forward alone and the whole iteration it times
            line["chamfer_kernel"] = [chamfer_kernel_rate(B, 4 * N, 4 * N), chamfer_kernel_rate(32, 16384, 1024),
                                      chamfer_kernel_rate(B, 4 * N, 4 * N, distinct=N),
                                      chamfer_kernel_rate(B, 4 * N, 4 * N, distinct=N, hint=True)]
            line["chamfer_reference_microbench"] = chamfer_train_rate()
            line["fps_kernel"] = [fps_kernel_rate(B, 4 * N, N), fps_kernel_rate(1, 4 * N, N)]
            if args.cpu_batch > 0:
                line["chamfer_kernel"][0]["cpu"] = chamfer_cpu_rate(4 * N, 4 * N)
        if world == 1 and args.cpu_batch > 0 and not args.step_only:
            line["cpu_table"] = cpu_table(N)
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_batch, args.cpu_steps)
        print(json.dumps(line))
    diverged = replicas is not None and not replicas["identical"]
    if dist.is_initialized():
        dist.destroy_process_group()
    if diverged:
        sys.stderr.write("bench.py: the data-parallel replicas DIVERGED (see replica_checksums)\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
