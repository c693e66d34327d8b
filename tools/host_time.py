"""Dev: host enqueue time per train step vs GPU time (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloudaae_amd import train_cloudAAE_ycbv as T
for B, rp in ((32, False), (32, True), (8, False), (8, True), (128, True)):
    graph = T.TrainGraph({"num_point": 1024, "gpu": 0}, {}, {"batch_size": B}, replay=rp)
    el = T.synthetic_element(B, 1024, graph.device)
    for _ in range(10):
        graph.train_step(el)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(50):
        graph.train_step(el)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print("replay" if rp else "eager ", "B=%d host enqueue %.3f ms/step, total %.3f ms/step" % (B, (t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3), flush=True)
