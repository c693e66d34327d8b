"""dev: is the forward pass bit-reproducible under load?  python tools/dev/fwd_repro_stress.py STEPS [PROCS]
Each process: one single-rank TrainGraph (B=16, N=256), the same batch and the same starting state every step; prints how many
steps gave a different total loss / reconstruction than the first one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.multiprocessing as mp


def work(rank, steps, B, N):
    from cloudaae_amd import train_cloudAAE_ycbv as T
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if os.environ.get("RCCL") == "1":          # one rank, the collectives really issued (CLOUDAAE_FORCE_COLLECTIVES=1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29700 + rank))
        dist.init_process_group("nccl", rank=0, world_size=1)
        g = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, seed=9, sync_bn=os.environ.get("SYNCBN") == "1")
    else:
        g = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, process_group=False, seed=9)
    el = T.synthetic_element(B, N, dev, seed=21)
    el["noise"] = torch.randn((B, N, 3), generator=torch.Generator(device=dev).manual_seed(3), device=dev) * 0.001
    snap = [t.clone() for t in (g.store.flat_params, g.store.flat_state, g.adam_m, g.adam_v, g.batch, g.beta1_power,
                                g.beta2_power, g.bn_decay)]
    first = None
    bad = []
    # tap: the step's FIRST kNN call (layer 1, on the assembled cloud) is issued twice, the second into our own buffer; a
    # third result comes from re-running the kernel on the same pointer after the step
    from cloudaae_amd import _lib
    L = _lib.lib()
    orig = L.cloudaae_knn
    tap = {}
    twin = torch.empty((B, N, 10), dtype=torch.int32, device=dev)

    def knn_tap(*a):
        rc = orig(*a)
        if "args" not in tap:
            tap["args"] = a
            orig(*(a[:6] + (twin.data_ptr(),) + a[7:]))
        return rc
    L.cloudaae_knn = knn_tap
    if os.environ.get("SYNC_AFTER_ASSEMBLE") == "1":
        orig_as = L.cloudaae_input_assemble

        def as_sync(*a):
            rc = orig_as(*a)
            torch.cuda.synchronize()
            return rc
        L.cloudaae_input_assemble = as_sync
    if os.environ.get("SYNC_BEFORE_ASSEMBLE") == "1":
        orig_as2 = L.cloudaae_input_assemble

        def as_sync2(*a):
            torch.cuda.synchronize()
            return orig_as2(*a)
        L.cloudaae_input_assemble = as_sync2
    twin_bad = 0
    for s in range(steps):
        tap.clear()
        with torch.no_grad():
            for dst, src in zip((g.store.flat_params, g.store.flat_state, g.adam_m, g.adam_v, g.batch, g.beta1_power,
                                 g.beta2_power, g.bn_decay), snap):
                dst.copy_(src)
        mode = os.environ.get("MODE", "train")
        if mode == "eval":
            o = g.eval_step(el)
        elif mode == "fwd":                       # training-mode forward only (batch statistics), no backward, no Adam
            with torch.no_grad():
                o = g.forward(el, is_training=True)
        else:
            o = g.train_step(el)
        torch.cuda.synchronize()
        ep = o["end_points"]
        if not torch.equal(twin, ep["nn_idx1"].to(torch.int32).reshape(B, N, 10)):
            twin_bad += 1
            if twin_bad <= 4:
                a1 = ep["nn_idx1"].to(torch.int32).reshape(B, N, 10)
                d = (twin != a1).any(-1).nonzero()
                clouds = sorted(set(d[:, 0].tolist()))
                pts = d[:, 1].tolist()
                print("proc", rank, "step", s, "points with differing neighbours:", d.shape[0], "clouds", clouds,
                      "point range", min(pts), max(pts), "sample", a1[d[0, 0], d[0, 1]].tolist(), twin[d[0, 0], d[0, 1]].tolist(),
                      flush=True)
        cur = tuple(int(ep["nn_idx%d" % i].long().sum().item()) for i in (1, 2, 3, 4)) + tuple(
            ep[k].double().sum().item() for k in sorted(ep) if torch.is_tensor(ep[k]) and ep[k].is_floating_point()) + (
            float(o["total_loss"]), o["xyz_recon"].double().sum().item())
        if first is None:
            names = ["nn1", "nn2", "nn3", "nn4"] + [k for k in sorted(ep) if torch.is_tensor(ep[k]) and ep[k].is_floating_point()] + ["total_loss", "recon"]
        if first is None:
            first = cur
        elif cur != first:
            bad.append((s, [n for n, a, b in zip(names, cur, first) if a != b]))
    print("proc", rank, "steps", steps, "differing from the first:", len(bad), bad[:6], "| twin of the first kNN differs from it:", twin_bad, flush=True)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    if procs == 1:
        work(0, steps, B, N)
    else:
        mp.spawn(work, args=(steps, B, N), nprocs=procs, join=True)
