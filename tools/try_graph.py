"""Dev experiment: capture one train step in a HIP graph (torch.cuda.CUDAGraph) and replay it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloudaae_amd import train_cloudAAE_ycbv as T
from cloudaae_amd.utils import _functions as F

B, N = 32, 1024
graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B})
el = T.synthetic_element(B, N, graph.device)
for _ in range(5):
    out = graph.train_step(el)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(50):
    out = graph.train_step(el)
torch.cuda.synchronize()
print("eager ms/step", (time.time() - t0) / 50 * 1e3, float(out["total_loss"].detach()))

# events inside a capture?
ext_ok = False
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True); e1 = torch.cuda.Event(enable_timing=True, external=True)
    ext_ok = True
except TypeError as ex:
    print("no external events:", ex)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        out = graph.train_step(el)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
if ext_ok:
    F.TIMED_SITES["agg_fwd_external"] = (e0, e1)
with torch.cuda.graph(g):
    sout = graph.train_step(el)
torch.cuda.synchronize()
print("captured")
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.time() - t0) / 50 * 1e3, float(sout["total_loss"].detach()), float(graph.batch))
