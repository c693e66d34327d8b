"""bench.py -- throughput of the CloudAAE training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" is one pass of the hot path over one batch of synthetic input, i.e. one
iteration of the reference's session loop (train_cloudAAE_ycbv.py:350-368): BN-decay
schedule, input assembly, DGCNN encoder + decoder + pose heads, Chamfer / translation /
SO(3) losses, backward, TF-Adam (+ gradient all-reduce for N > 1).  Workload at N=1 is
BASELINE.json configs[1]: all 21 classes, batch 32, 1024 points, fp32.  For N > 1 the
per-GPU batch stays 32 (weak scaling).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the dominant kernel (the dgcnn_agg forward GEMM on the matrix cores),
                 timed live with HIP events on the launch stream over the timed region
  cpu_baseline : the CPU oracle's train step timed on this box's host cores on a bounded
                 sample (rank 0, N=1 only)
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


def chamfer_kernel_rate(batch, n, m, iters=20):
    """The second half of BASELINE's metric: Chamfer nn_distance forward kernel rate.
    Algorithmic bytes = B*(n+m)*20 (12 B read + 4 B dist + 4 B idx per point): arithmetic-bound
    by three orders of magnitude.  Large clouds take nn_distance_filter_kernel: the nearest candidate is
    searched with scores on the matrix cores (two v_mfma_f32_32x32x2_f32 per 32 x 32 pairs) and the
    reference's un-fused arithmetic decides among the candidates of the two best units (bit parity with
    the CPU reference).  Its bound is the matrix pipe: 1024 pairs per 128 cycles per SIMD = 19.7 T pairs/s
    (the first-generation kernel: 8.6 fp32 lane-operations per pair, 9.1 T pairs/s at best)."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    g = torch.Generator(device="cuda").manual_seed(100)       # tf_nndistance.py:45-46 seeds
    a = torch.randn((batch, n, 3), generator=g, device="cuda")
    c = torch.randn((batch, m, 3), generator=g, device="cuda")
    for _ in range(3):
        tf_nndistance.nn_distance(a, c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        tf_nndistance.nn_distance(a, c)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    pairs = 2.0 * batch * n * m
    bound = 256 * 4 * 2.4e9 * 1024 / 128
    return {"shape": "[%d,%d,3]x[%d,%d,3]" % (batch, n, batch, m), "us_per_launch": round(sec * 1e6, 2),
            "GB/s": round(batch * (n + m) * 20 / sec / 1e9, 3), "Tpairs/s": round(pairs / sec / 1e12, 3),
            "clouds/s": round(batch / sec, 1), "frac_of_matrix_pipe_bound": round(pairs / sec / bound, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--per-gpu-batch", type=int, default=32)
    ap.add_argument("--num-point", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10, help="neighbours of the edge convolution (BASELINE config 4: 20)")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="train steps of the CPU-baseline sample")
    ap.add_argument("--gemm-dtype", default="f32", choices=["f32", "bf16"], help="bf16: BASELINE config 3's "
                    "arithmetic (dense-layer operands rounded to bf16, fp32 accumulate; everything else fp32)")
    ap.add_argument("--eager", action="store_true", help="step through Python/autograd every time instead of "
                    "replaying the recorded step (TrainGraph(replay=False))")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (WORLD_SIZE is unset)" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    force = os.environ.get("CLOUDAAE_FORCE_COLLECTIVES") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from cloudaae_amd import train_cloudAAE_ycbv as T
    from cloudaae_amd.utils import _functions as F

    B = args.per_gpu_batch
    N = args.num_point
    # the dgcnn_agg forward GEMM records a HIP event before and after itself on its launch stream,
    # live in every step of the timed region (host callbacks of the recorded step)
    F.TIMED_SITES["agg_fwd"] = []
    graph = T.TrainGraph({"num_point": N, "gpu": local}, {"optimizer": "adam"},
                         {"batch_size": B * world, "learning_rate": 0.0008}, replay=not args.eager,
                         gemm_dtype=args.gemm_dtype, k_neighbor=args.k)
    el = T.synthetic_element(B, N, graph.device, seed=123456789, rank=rank)
    graph.reuse_staged_inputs = True     # one fixed batch, resident in HBM: do not re-copy it every step

    for _ in range(args.warmup):
        graph.train_step(el)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    F.TIMED_SITES["agg_fwd"].clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = graph.train_step(el)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    events = F.TIMED_SITES.pop("agg_fwd")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=graph.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(out["total_loss"])

    if rank == 0:
        ms = [events[i].elapsed_time(events[i + 1]) for i in range(0, len(events) - 1, 2)]
        k_ms = sum(ms) / max(1, len(ms))
        M, Nn, K = B * N, 1024, 320
        flops = 2.0 * M * Nn * K                      # algorithmic flops of one dgcnn_agg forward launch
        achieved = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_roofline_traffic.json")
        if os.path.exists(tpath) and B == 32 and N == 1024:
            # HBM bytes per launch of the same kernel from rocprofv3 PMC passes (FETCH_SIZE x2
            # corrected + WRITE_SIZE; collected separately, see profiles/ and DESIGN.md section 5)
            traffic = json.load(open(tpath)).get("traffic_bytes_per_launch")
        line = {
            "metric": "point-clouds/sec (train step, N=%d)" % N,
            "value": round(B * world * args.steps / elapsed, 2),
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.gemm_dtype == "f32" else "bf16 dense-layer operands, f32 accumulate and everything else",
            "data": "synthetic",
            "config": {"workload": "CloudAAE train step: get_model_dgcnn_mean_6d, all 21 YCB classes, "
                                   "batch %d/GPU, N=%d points, k=%d, 4N-point Chamfer target, TF-Adam" % (B, N, args.k),
                       "global_batch": B * world, "num_point": N, "parallelism": "dp%d" % world,
                       "step_issue": "recorded step replay" if graph.replay else "eager",
                       "final_total_loss": round(loss, 4)},
            "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel<128,128,2,2> dgcnn_agg forward "
                                                     "[%d x 320] x [320 x 1024]" % M,
                         "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "launch_ms": round(k_ms, 4), "launches_timed": len(ms)},
        }
        if args.gemm_dtype == "bf16":
            # with bf16 operands the same product leaves the matrix pipe (2.5 PFLOP/s dense) and is bound by
            # HBM: fp32 activations in (M x 320), weights, fp32 output out (M x 1024)
            nbytes = 4.0 * (M * K + K * Nn + Nn + M * Nn)
            gbs = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            line["roofline"] = {"bound": "hbm", "kernel": "gemm_bf16_kernel<128,128,2,2> dgcnn_agg forward "
                                                          "[%d x 320] x [320 x 1024]" % M,
                                "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                "frac": round(gbs / 8000.0, 4), "traffic": None, "launch_ms": round(k_ms, 4),
                                "launches_timed": len(ms)}
        if world == 1:
            # "Chamfer kernel GB/s": the train shape (n = m = 4N) and the reference's own
            # micro-benchmark shape (tf_nndistance.py:48-49)
            line["chamfer_kernel"] = [chamfer_kernel_rate(B, 4 * N, 4 * N), chamfer_kernel_rate(32, 16384, 1024)]
            if args.cpu_batch > 0:
                line["chamfer_kernel"][0]["cpu"] = chamfer_cpu_rate(4 * N, 4 * N)
        if world == 1 and args.cpu_batch > 0:
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_batch, args.cpu_steps)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
