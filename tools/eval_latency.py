"""Dev: latency of the inference step (evaluate_batch) at small batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloudaae_amd import train_cloudAAE_ycbv as T, evaluate_cloudAAE_ycbv as E
for B, N in ((1, 256), (1, 1024), (8, 1024), (32, 1024)):
    graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": max(B, 2)})
    el = T.synthetic_element(B, N, graph.device)
    el["xyz_inlier"] = el["visiblePoints"]
    for rp in (False, True):
        for _ in range(5):
            E.evaluate_batch(graph, el, replay=rp)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(30):
            E.evaluate_batch(graph, el, replay=rp)
        torch.cuda.synchronize()
        print("B=%d N=%d %s: %.3f ms per batch" % (B, N, "replay" if rp else "eager ", (time.time() - t0) / 30 * 1e3), flush=True)
