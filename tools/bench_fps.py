"""Dev: farthest point sampling kernel time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd.tf_ops.sampling import tf_sampling
for b, n, m in ((1, 1024, 256), (32, 1024, 256), (1, 4096, 1024), (32, 4096, 1024), (256, 4096, 1024), (32, 2048, 512)):
    x = torch.randn((b, n, 3), device="cuda")
    for _ in range(2): tf_sampling.farthest_point_sample(m, x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): tf_sampling.farthest_point_sample(m, x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print("b=%d n=%d m=%d: %.1f us (%.2f us per round)" % (b, n, m, us, us / (m - 1)))
