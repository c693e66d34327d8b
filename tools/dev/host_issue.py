"""Dev: host time to issue one replayed step (no GPU wait) against the step time, and the split of the host time between
the Python loop + ctypes and the library call itself (CLOUDAAE-side launches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloudaae_amd import train_cloudAAE_ycbv as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
graph = T.TrainGraph({"num_point": 1024, "gpu": 0}, {"optimizer": "adam"}, {"batch_size": B, "learning_rate": 0.0008}, replay=True)
el = T.synthetic_element(B, 1024, graph.device, seed=1)
graph.reuse_staged_inputs = True
for _ in range(30):
    graph.train_step(el)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        graph.train_step(el)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d host issue %.3f ms/step, total %.3f ms/step" % (B, (t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3), flush=True)
plan = graph._plan
print("entries per step:", len(plan.entries), "of which host callbacks:", sum(1 for e in plan.entries if e[2] is None))
# per-entry host cost with the GPU idle in between (sync after each): which calls are slow to issue
import collections
cost = collections.defaultdict(lambda: [0, 0.0])
for _ in range(20):
    for fn, args, name in plan.entries:
        a = time.perf_counter()
        fn(*args)
        b = time.perf_counter()
        c = cost[name or "host:" + getattr(fn, "__name__", "?")]
        c[0] += 1
        c[1] += b - a
    torch.cuda.synchronize()
for k, (n, t) in sorted(cost.items(), key=lambda kv: -kv[1][1]):
    print("%-40s %4d calls/step %7.2f us each %8.1f us/step" % (k, n // 20, t / n * 1e6, t / 20 * 1e6))
print("sum %.1f us/step" % (sum(t for _, t in cost.values()) / 20 * 1e6))
