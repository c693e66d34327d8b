"""dev: stage times of fps_kernel's round (a build with -DFPS_STAMPS writes per-stage average cycles into the tail of the output)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd.tf_ops.sampling import tf_sampling
for n, m in ((1024, 256), (4096, 1024)):
    x = torch.randn((2, n, 3), device="cuda")
    o = tf_sampling.farthest_point_sample(m, x)[0].cpu().tolist()
    names = ["distances", "wave max (DPP, ballot)", "point + xyz (readlanes)", "store, barrier", "table reduce"]
    for w, off in ((0, 0), (7, 5)):
        vals = [o[m - 1 - i - off] for i in range(5)]
        print("n=%d wave %d: " % (n, w) + ", ".join("%s %d" % (a, b) for a, b in zip(names, vals)), "| sum", sum(vals), "(100 MHz ticks)")
