"""Dev: hidden point removal vs scipy/qhull on many random poses of the fixture object model (+ occluder blobs)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import tfrecord_io as TR
from cloudaae_amd.utils import hidden_point_removal as hpr
from oracle import synth_oracle as SO
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
models, _ = TR.read_and_decode_obj_model(os.path.join(g, "obj_model_first1.tfrecords"))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
B = 48
clouds = []
for i in range(B):
    ax = rng.standard_normal(3); ax = ax / np.linalg.norm(ax) * rng.uniform(0, np.pi)
    t = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(0.5, 1.5)], np.float32)
    pts = SO.transform_object_model(models[0][:, :3], ax.astype(np.float32), t)
    occ = (rng.standard_normal((400, 3)) * rng.uniform(0.005, 0.03) + [t[0] + rng.uniform(-.05, .05), t[1] + rng.uniform(-.05, .05), t[2] * rng.uniform(0.4, 0.9)]).astype(np.float32)
    clouds.append(np.concatenate([pts, occ], 0))
bad = 0
for sl, name in ((slice(None), "with occluder"), (slice(0, 2048), "object only")):
    fl, org = zip(*[SO.spherical_flip(c[sl]) for c in clouds])
    vis, num, ids = hpr.convexHull(torch.from_numpy(np.stack(fl)).cuda(), torch.from_numpy(np.stack(org)).cuda(), return_ids=True)
    for i in range(B):
        want, _ = SO.convex_hull_visible(fl[i])
        got = ids[i, :int(num[i])].cpu().numpy()
        if not np.array_equal(got, want):
            bad += 1
            print(name, i, "mismatch: qhull", len(want), "gpu", len(got), "sym diff", len(set(want) ^ set(got)))
print("clouds checked", 2 * B, "mismatches", bad)
