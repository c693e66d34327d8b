"""GPU: the kernels that once returned wrong lists next to another process repeat bit for bit next to one.

profiles/notes_two_processes_one_gpu.md, round 6: a packed-fp32 instruction that takes its low half from the high register of
a pair (`v_pk_add_f32 ... op_sel:[0,1]`, the squared norms of the kNN / Chamfer / sampling kernels) now and then returned
`src0 + 0` in lanes 48-63 -- only while a wave of ANOTHER process shared the SIMD, about 62 launches of 9600 at these shapes.
tests/test_isa_rules.py checks the built code for the instruction; this is the behaviour: next to a process that runs training
steps on the same GPU, a process issues the same launches over and over and every result must equal the first (the reference op, utils/tf_util.py:621-632, is
deterministic).  Measured with a build that has the instruction (the five files of csrc/Makefile's NOPK_OBJS compiled without the
flag): 803 of 66 640 launches of the [32,128] kNN wrong in its five seconds; the shipped build: 0 of 314 000 over the six cases."""
import time

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SECONDS = 5.0       # per case and process: both processes walk the cases by the clock, so they stay side by side


def _worker(rank, out, running, done):
    torch.cuda.set_device(0)
    if rank == 1:
        # the load: a process that steps the train graph (the reproducer of tools/dev/pk_opsel_repro_run.sh found its wrong
        # lanes next to exactly this; the same kernels from a second stream of ONE process never showed any)
        from cloudaae_amd import train_cloudAAE_ycbv as T
        g = T.TrainGraph({"num_point": 1024, "gpu": 0}, {}, {"batch_size": 32}, process_group=False, seed=3, replay=True)
        el = T.synthetic_element(32, 1024, torch.device("cuda:0"), seed=4)
        steps = 0
        while not done.is_set():
            for _ in range(10):
                g.train_step(el)
            torch.cuda.synchronize()
            steps += 10
            running.set()
        out[rank] = {"steps of the load": (0, steps)}
        return
    from cloudaae_amd.utils import tf_util
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    g = torch.Generator(device="cuda").manual_seed(100)
    wrong = {}
    # the layer-1 kNN over xyz at the shapes the two-rank tests step (clouds under 256 points: the scan kernel) and at the
    # headline's (the wide kernel), the 64-channel kNN, the Chamfer search, farthest point sampling
    cases = []
    for b, n in ((32, 128), (16, 256), (32, 1024)):
        x = torch.randn((b, n, 3), generator=g, device="cuda")
        cases.append(("knn3 [%d,%d]" % (b, n), lambda x=x: tf_util.knn(tf_util.pairwise_xyz_distance(x), k=10)))
    f = torch.randn((32, 1024, 64), generator=g, device="cuda").relu_()
    cases.append(("knn64 [32,1024]", lambda f=f: tf_util.knn(tf_util.pairwise_xyz_distance(f[:, :, None, :]), k=10)))
    a, c = torch.randn((32, 1024, 3), generator=g, device="cuda"), torch.randn((32, 1024, 3), generator=g, device="cuda")
    cases.append(("chamfer [32,1024]", lambda a=a, c=c: torch.cat([t.reshape(-1).view(torch.int32) for t in tf_nndistance.nn_distance(a, c)])))
    p = torch.randn((32, 1024, 3), generator=g, device="cuda")
    cases.append(("fps [32,1024]->256", lambda p=p: tf_sampling.farthest_point_sample(256, p)))
    firsts = [fn().clone() for _, fn in cases]          # (before the load starts)
    torch.cuda.synchronize()
    try:
        assert running.wait(240), "the load did not start"
        for (name, fn), first in zip(cases, firsts):
            bad = launches = 0
            t_end = time.time() + SECONDS
            while time.time() < t_end:
                got = [fn() for _ in range(16)]                 # (a queue of launches between two host reads)
                bad += sum(int(not torch.equal(t, first)) for t in got)
                launches += len(got)
            wrong[name] = (bad, launches)
    finally:
        done.set()
    out[rank] = wrong


def test_results_repeat_next_to_another_process(hip):
    mgr = mp.Manager()
    out, running, done = mgr.dict(), mgr.Event(), mgr.Event()
    mp.spawn(_worker, args=(out, running, done), nprocs=2, join=True)
    res = dict(out)
    assert set(res) == {0, 1}
    print(res)
    assert res[1]["steps of the load"][1] >= 100, "the load hardly ran: %r" % (res[1],)
    bad = {k: v for k, v in res[0].items() if v[0]}
    assert not bad, "(launches that differ from the first, launches): %r" % (bad,)
