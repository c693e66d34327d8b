"""dev: do a query tile's neighbours sit in few candidate tiles?  For the C = 64 kNN inputs of a training step: per tile of
32 consecutive queries, the number of 32-row candidate tiles that hold at least one of their k nearest neighbours.
   python tools/dev/knn_tile_coherence.py [B] [N] [k] [--config5]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import train_cloudAAE_ycbv as T
from cloudaae_amd.utils import tf_util
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if len(args) > 0 else 32
N = int(args[1]) if len(args) > 1 else 1024
K = int(args[2]) if len(args) > 2 else 10
graph = T.TrainGraph({"num_point": N, "gpu": 0}, {"optimizer": "adam"}, {"batch_size": B, "learning_rate": 0.0008}, replay=False,
                     k_neighbor=K)
el = T.synthetic_element(B, N, graph.device, seed=123456789)
if "--config5" in sys.argv:
    models = T.synthetic_object_models(T.NUM_CLASS, 8192, device=graph.device)
    el = T.get_small_data({k: el[k] for k in ("translation", "axisangle", "class_id")}, models, seed=1, rows_org=4 * N)
for s in range(3):
    tf_util.KNN_TAP = [] if s == 2 else None
    graph.train_step(el)
torch.cuda.synchronize()
for li, x in enumerate(tf_util.KNN_TAP):
    x = x[:8, :, :64].double()
    d = torch.cdist(x, x) ** 2
    idx = d.topk(K, largest=False).indices                      # [b, n, k]
    tiles = idx // 32
    b, n, _ = tiles.shape
    hit = torch.zeros((b, n // 32, n // 32), dtype=torch.bool, device=x.device)
    qt = (torch.arange(n, device=x.device) // 32)[None, :, None].expand_as(tiles)
    bb = torch.arange(b, device=x.device)[:, None, None].expand_as(tiles)
    hit[bb.reshape(-1), qt.reshape(-1), tiles.reshape(-1)] = True
    per = hit.sum(-1).double()
    print("layer %d: candidate tiles with a neighbour, per query tile: mean %.1f  median %.0f  max %.0f  of %d" %
          (li + 2, per.mean(), per.median(), per.max(), n // 32))
