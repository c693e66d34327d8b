"""Run the hidden point removal of one synthetic batch a few times (for counter passes): [B] [points per model] [reps]."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import train_cloudAAE_ycbv as T
from cloudaae_amd.utils import generate_occluder, hidden_point_removal as hpr
import math
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda")
models = T.synthetic_object_models(T.NUM_CLASS, NP, device=dev)
el = T.synthetic_element(B, 1024, dev, seed=1)
x = {k: el[k] for k in ("translation", "axisangle", "class_id")}
x = T.get_object_model(x, models); x = T.get_rotation_matrix(x); x = T.transform_object_model(x)
x = generate_occluder.get_random_spherical_occluder(x, 'ycbv', seed=1)
x = hpr.sphericalFlip(x, None, 0.8 * math.pi)
x = hpr.sphericalFlip_org(x, None, 0.8 * math.pi)
for _ in range(reps):
    v, n = hpr.convexHull(x['flippedPoints'], x['orgPoints'], 1)
    v2, n2 = hpr.convexHull(x['flippedPoints_org'], x['orgPoints_org'], 2)
torch.cuda.synchronize()
print("B=%d, %d / %d points per hull; visible %.0f / %.0f" % (B, x['flippedPoints'].shape[1], x['flippedPoints_org'].shape[1],
                                                             float(n.float().mean()), float(n2.float().mean())))
