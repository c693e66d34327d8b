import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloudaae_amd import train_cloudAAE_ycbv as T
B, N = 32, 1024
graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B})
el = T.synthetic_element(B, N, graph.device)
for i in range(400):
    out = graph.train_step(el)
    if i % 20 == 0 or i > 100 and i < 125:
        print(i, [round(float(out[k].detach()), 6) for k in ("total_loss", "xyz_loss", "trans_loss", "axag_loss")], flush=True)
