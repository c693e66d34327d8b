"""dev: cloudaae_knn on a fixed input, many launches, PROCS processes sharing the GPU: how many results differ from the first?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.multiprocessing as mp


def work(rank, iters, b, n, c, ld, k):
    from cloudaae_amd import _lib
    L = _lib.lib()
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((b, n, ld), device="cuda", generator=g)
    out = torch.empty((b, n, k), dtype=torch.int32, device="cuda")
    first = None
    bad = 0
    DEPTH = int(os.environ.get("DEPTH", "1"))          # launches queued back to back before a synchronisation
    outs = [torch.empty((b, n, k), dtype=torch.int32, device="cuda") for _ in range(DEPTH)]
    for i in range(iters // DEPTH):
        for o in outs:
            o.fill_(-1)
        for o in outs:
            _lib.check(L.cloudaae_knn(b, n, c, ld, k, x.data_ptr(), o.data_ptr(), _lib.stream()), "knn")
        torch.cuda.synchronize()
        for o in outs:
            if first is None:
                first = o.clone()
            elif not torch.equal(o, first):
                bad += 1
                if bad <= 3:
                    d = (o != first).nonzero()
                    print("proc", rank, "iter", i, "differing entries", d.shape[0], "first at", d[0].tolist(), o[tuple(d[0].tolist())].item(), first[tuple(d[0].tolist())].item(), flush=True)
    print("proc", rank, (b, n, c, ld, k), "iters", iters, "differing:", bad, flush=True)


if __name__ == "__main__":
    iters, procs = int(sys.argv[1]), int(sys.argv[2])
    shape = tuple(int(a) for a in sys.argv[3:8])
    if procs == 1:
        work(0, iters, *shape)
    else:
        mp.spawn(work, args=(iters,) + shape, nprocs=procs, join=True)
