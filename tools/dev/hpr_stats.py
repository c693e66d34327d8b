"""dev: counters of the culled hull-vertex LP (libcloudaae_hip_prof.so, make -C cloudaae_amd/csrc prof):
   CLOUDAAE_HIP_LIB=cloudaae_amd/libcloudaae_hip_prof.so python tools/dev/hpr_stats.py [points per model]"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib, train_cloudAAE_ycbv as T
from cloudaae_amd.utils import generate_occluder, hidden_point_removal as hpr
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = 32
dev = torch.device("cuda")
models = T.synthetic_object_models(T.NUM_CLASS, NP, device=dev)
el = T.synthetic_element(B, 1024, dev, seed=1)
x = {k: el[k] for k in ("translation", "axisangle", "class_id")}
x = T.get_object_model(x, models); x = T.get_rotation_matrix(x); x = T.transform_object_model(x)
x = generate_occluder.get_random_spherical_occluder(x, 'ycbv', seed=1)
x = hpr.sphericalFlip(x, None, 0.8 * math.pi)
cd = _lib.lib()._cdll
out = (ctypes.c_ulonglong * 16)()
cd.cloudaae_hpr_stats_read(out, 1)
v, n = hpr.convexHull(x['flippedPoints'], x['orgPoints'], 1)
cd.cloudaae_hpr_stats_read(out, 1)
names = ["points", "local re-solves", "passes", "box iterations", "groups scanned", "joined", "fallbacks", "vertices",
         "local re-solve iterations", "re-solves at [0,128)", "at [128,256)", "at [256,384)", "at [384,512)", "rejected locally",
         "sum of reject positions", "-"]
pts = max(out[0], 1)
for k, c in zip(names, out):
    print("%-16s %10d  %.3f per point" % (k, c, c / pts))
print("per vertex-candidate pass: groups scanned / pass = %.2f of %d" % (out[4] / max(out[2], 1), (x['flippedPoints'].shape[1] + 63) // 64))
