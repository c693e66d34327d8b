"""dev: the two-processes-on-one-GPU kNN effect, with the scan kernel checking itself (make -C cloudaae_amd/csrc dbg).
    CLOUDAAE_HIP_LIB=cloudaae_amd/libcloudaae_hip_dbg.so CLOUDAAE_KNN3_WIDE=0 python tools/dev/knn3_debug_stress.py STEPS PROCS [B N]
Each process steps one single-rank TrainGraph on the same batch from the same state.  The step's first kNN (layer 1) is issued
twice (the twin); both results are compared with the first step's.  The debug build's knn3_scan_kernel records
  kind 2 / 3: a row staged in LDS (x, y, z) != memory, right after the staging barrier / at the end of the workgroup's life,
  kind 4 / 5: the staged norm (w) != the norm of the staged x, y, z recomputed by the checker, at the same two places.
DETERMINISTIC=1 builds the graph with deterministic=True: then the sums of the state Adam wrote must repeat too.
A wrong list is reported with the candidate rows it lacks / has in excess (the stretch of the cloud that was read wrong)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.multiprocessing as mp


def hw(h):   # HW_ID of gfx9: wave 0-3, simd 4-5, pipe 6-7, cu 8-11, sh 12, se 13-15, tg 16-19, vm 20-23, queue 24-26
    return "se%d sh%d cu%d simd%d wave%d vmid%d queue%d" % ((h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15, (h >> 4) & 3, h & 15,
                                                           (h >> 20) & 15, (h >> 24) & 7)


def work(rank, steps, B, N):
    from cloudaae_amd import train_cloudAAE_ycbv as T
    from cloudaae_amd import _lib
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    g = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, process_group=False, seed=9,
                     deterministic=os.environ.get("DETERMINISTIC") == "1")
    el = T.synthetic_element(B, N, dev, seed=21)
    el["noise"] = torch.randn((B, N, 3), generator=torch.Generator(device=dev).manual_seed(3), device=dev) * 0.001
    state = (g.store.flat_params, g.store.flat_state, g.adam_m, g.adam_v, g.batch, g.beta1_power, g.beta2_power, g.bn_decay)
    snap = [t.clone() for t in state]
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    have_dbg = hasattr(raw, "cloudaae_debug_knn3_records")
    recbuf = (ctypes.c_uint * (2048 * 16))()
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
    ref = None
    ref_sums = None
    wrong = [0, 0]
    sums_wrong = []
    recs_total = 0
    for s in range(steps):
        tap.clear()
        with torch.no_grad():
            for dst, src in zip(state, snap):
                dst.copy_(src)
        o = g.eval_step(el) if os.environ.get("MODE") == "eval" else g.train_step(el)
        torch.cuda.synchronize()
        # everything else the step produced, as exact sums: the other neighbour lists, the losses, the state Adam wrote
        ep = o["end_points"]
        sums = tuple(int(ep["nn_idx%d" % i].long().sum().item()) for i in (2, 3, 4)) + (
            float(o["total_loss"].detach()), o["xyz_recon"].detach().double().sum().item()) + tuple(t_.double().sum().item() for t_ in state[:4])
        a1 = o["end_points"]["nn_idx1"].to(torch.int32).reshape(B, N, 10).cpu().numpy()
        tw = twin.cpu().numpy()
        nrec = raw.cloudaae_debug_knn3_records(recbuf, 2048) if have_dbg else 0
        if ref is None:
            ref = a1.copy()
            if not np.array_equal(a1, tw):
                print("proc", rank, "the FIRST step's twin differs: reference unsure", flush=True)
        if ref_sums is None:
            ref_sums = sums
        elif sums != ref_sums and np.array_equal(a1, ref):
            names = ("nn2", "nn3", "nn4", "total_loss", "recon", "params", "bn_state", "adam_m", "adam_v")
            sums_wrong.append((s, [n_ for n_, x_, y_ in zip(names, sums, ref_sums) if x_ != y_]))
        for which, got in ((0, a1), (1, tw)):
            if np.array_equal(got, ref):
                continue
            wrong[which] += 1
            if wrong[0] + wrong[1] > 12:
                continue
            bad = np.argwhere((got != ref).any(-1))
            for c in sorted(set(bad[:, 0].tolist())):
                pts = bad[bad[:, 0] == c][:, 1]
                lack, extra = set(), set()
                for p in pts:
                    lack |= set(ref[c, p].tolist()) - set(got[c, p].tolist())
                    extra |= set(got[c, p].tolist()) - set(ref[c, p].tolist())
                print("proc %d step %d %s: cloud %d, %d queries wrong (range %d..%d); rows lacking %s; rows in excess %s" % (
                    rank, s, ("launch", "twin")[which], c, len(pts), pts.min(), pts.max(), sorted(lack), sorted(extra)), flush=True)
        if nrec:
            recs_total += nrec
            r = np.frombuffer(recbuf, dtype=np.uint32).reshape(2048, 16)[:min(nrec, 2048)].copy()
            out_launch = tap["args"][6] >> 8 if "args" in tap else -1
            print("proc %d step %d: %d debug records" % (rank, s, nrec), flush=True)
            for kind in (2, 3, 4, 5, 6):
                rk = r[r[:, 0] == kind]
                if not len(rk):
                    continue
                f = rk[:, 8:14].view(np.float32)
                groups = {}
                for q in range(len(rk)):
                    groups.setdefault((int(rk[q, 1]), int(rk[q, 2]), int(rk[q, 4])), []).append(q)
                for (c, qg, outp), qs in sorted(groups.items())[:6]:
                    rows = sorted(int(rk[q, 3]) for q in qs)
                    q0 = qs[0]
                    what = ("staged xyz %s, memory %s" if kind < 4 else "staged xyz %s, (staged w, its norm, -) %s" if kind < 6 else
                            "(w, norm, z*z) %s, registers after the sequence (z*z, x*x+y*y, z*z+y*y; variants 6, 7: z*z, w read again, w of a second pass) %s")
                    print(("   kind %d cloud %d query group %d (%s): rows %s | %s xcc%d t=%d | first: " + what) % (
                        kind, c, qg, "launch" if (outp & 0xffffffff) == (out_launch & 0xffffffff) else "twin/other", rows, hw(int(rk[q0, 5])), int(rk[q0, 6]) & 15,
                        int(rk[q0, 7]), f[q0, :3].tolist(), f[q0, 3:].tolist()), flush=True)
    print("proc", rank, "steps", steps, "| launch wrong:", wrong[0], "| twin wrong:", wrong[1], "| debug records:", recs_total,
          "| steps with a right first kNN and anything else different:", len(sums_wrong), sums_wrong[:5], flush=True)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    if procs == 1:
        work(0, steps, B, N)
    else:
        mp.spawn(work, args=(steps, B, N), nprocs=procs, join=True)
