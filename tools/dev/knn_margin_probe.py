"""Dev probe: how often would an APPROXIMATE kNN search (distances known to +-eps) leave the order of the first k
neighbours undecided on the features the network really produces?  eps = c * |x_i| * max_j |x_j| (bf16 hi/lo split
products accumulated in fp32: c ~ 6e-5).  Prints, per encoder layer, the fraction of queries with any gap <= 2 eps among
their first k+1 sorted distances, and the fraction whose ambiguity zone is not closed within k+6 candidates."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import train_cloudAAE_ycbv as T
from cloudaae_amd.utils import _functions as F

slots = []
orig = F.ConcatSlot.__init__
def init(self, buf):
    orig(self, buf); slots.append(self)
F.ConcatSlot.__init__ = init
B, N = 8, 1024
for k, steps in ((10, 0), (10, 30)):
    g = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, k_neighbor=k)
    el = T.synthetic_element(B, N, g.device, seed=1)
    for _ in range(steps):
        g.train_step(el)
    del slots[:]
    g.train_step(el)
    buf = slots[-1].buf.detach()                  # [B, N, 320]: net1 | net2 | net3 | net4
    for layer, off in ((2, 0), (3, 64), (4, 128)):
        x = buf[:, :, off:off + 64].double()
        sq = (x * x).sum(-1)
        D = sq[:, :, None] + sq[:, None, :] - 2 * x @ x.transpose(1, 2)
        d, _ = D.sort(-1)
        d = d[:, :, :k + 7]
        nrm = sq.sqrt()
        for c in (6e-5, 2e-5):
            eps = c * nrm * nrm.max(1, keepdim=True).values + 1e-6 * (sq + sq.max(1, keepdim=True).values)
            gaps = d[:, :, 1:k + 1] - d[:, :, :k]
            dirty = (gaps <= 2 * eps[:, :, None]).any(-1).float().mean()
            open_ = ((d[:, :, k + 5] - d[:, :, k - 1]) <= 2 * eps).float().mean()
            print("steps %d layer %d c=%.0e: |x|^2 mean %.2f  d_k mean %.3f  eps mean %.2e | dirty %.3f  not closed in k+6 %.4f"
                  % (steps, layer, c, float(sq.mean()), float(d[:, :, k - 1].mean()), float(eps.mean()), float(dirty), float(open_)))
