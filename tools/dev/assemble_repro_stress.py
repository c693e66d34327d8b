"""dev: input assembly + layer-1 kNN as the step issues them, fixed inputs, PROCS processes sharing the GPU"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.multiprocessing as mp


def work(rank, iters, B, N):
    from cloudaae_amd import _lib
    L = _lib.lib()
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(5)
    P, NC, k = N, 21, 10
    vis = torch.randn((B, P, 3), device="cuda", generator=g)
    noise = torch.randn((B, N, 3), device="cuda", generator=g) * 0.001
    cls = torch.randint(0, NC, (B,), device="cuda", generator=g)
    first = None
    bad = [0, 0, 0, 0]
    for i in range(iters):
        # the blocks the next allocations get held something else a moment ago (as inside a step)
        if os.environ.get("JUNK", "1") == "1":
            j1 = torch.empty((B, N, 3 + NC), device="cuda").normal_(); j2 = torch.empty((B, N, k), dtype=torch.int32, device="cuda").fill_(7)
            js = j1.sum()
            del j1, j2
        # fresh buffers like the eager step (the caching allocator hands the same blocks back)
        pc = torch.empty((B, N, 3 + NC), device="cuda"); mean = torch.empty((B, 3), device="cuda")
        noisy = torch.empty((B, N, 3), device="cuda"); nn = torch.empty((B, N, k), dtype=torch.int32, device="cuda")
        _lib.check(L.cloudaae_input_assemble(B, P, N, NC, vis.data_ptr(), noise.data_ptr(), cls.data_ptr(), pc.data_ptr(),
                                             mean.data_ptr(), noisy.data_ptr(), _lib.stream()), "assemble")
        _lib.check(L.cloudaae_knn(B, N, 3, 3 + NC, k, pc.data_ptr(), nn.data_ptr(), _lib.stream()), "knn")
        torch.cuda.synchronize()
        cur = (pc.clone(), mean.clone(), noisy.clone(), nn.clone())
        if first is None:
            first = cur
        else:
            for j in range(4):
                if not torch.equal(cur[j], first[j]):
                    bad[j] += 1
    print("proc", rank, "iters", iters, "differing [pc, mean, noisy, nn]:", bad, flush=True)


if __name__ == "__main__":
    iters, procs = int(sys.argv[1]), int(sys.argv[2])
    B, N = int(sys.argv[3]), int(sys.argv[4])
    if procs == 1:
        work(0, iters, B, N)
    else:
        mp.spawn(work, args=(iters, B, N), nprocs=procs, join=True)
