"""dev: kernels at sizes below their internal thresholds, against the CPU oracle (a sweep, not a test: prints every mismatch)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
from oracle import native as O
L = _lib.lib()
rng = np.random.default_rng(0)
bad = 0
# kNN: tiny clouds, k up to n, both channel counts, odd leading dimensions
for c, ld in ((3, 24), (3, 3), (64, 64), (16, 16)):
    for n in (1, 2, 3, 5, 17, 31, 32, 33, 63, 65, 100, 129, 255, 256, 257):
        for k in sorted(set([1, min(n, 5), min(n, 10), min(n, 20), min(n, 32)])):
            x = (rng.standard_normal((3, n, ld)) * 0.3).astype(np.float32)
            if n > 8:
                x[:, n // 2] = x[:, 0]
            want = O.knn(x, k, channels=c)
            xd = torch.from_numpy(x).cuda()
            got = torch.full((3, n, k), -7, dtype=torch.int32, device="cuda")
            rc = L.cloudaae_knn(3, n, c, ld, k, xd.data_ptr(), got.data_ptr(), _lib.stream())
            if rc != 0:
                print("knn rc", rc, (c, ld, n, k)); bad += 1; continue
            if not np.array_equal(want, got.cpu().numpy()):
                print("knn MISMATCH", (c, ld, n, k), int((want != got.cpu().numpy()).sum())); bad += 1
# Chamfer: tiny and ragged
from cloudaae_amd.tf_ops.nn_distance import tf_nndistance as NND
for n, m in ((1, 1), (1, 2), (2, 1), (3, 31), (31, 3), (33, 32), (32, 33), (1, 300), (300, 1), (65, 4097)):
    a = rng.standard_normal((2, n, 3)).astype(np.float32); b = rng.standard_normal((2, m, 3)).astype(np.float32)
    d1, i1, d2, i2 = O.nn_distance(a, b)
    g = NND.nn_distance(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    for w, t, nm in ((d1, g[0], "d1"), (i1, g[1], "i1"), (d2, g[2], "d2"), (i2, g[3], "i2")):
        if not np.array_equal(np.asarray(w), t.cpu().numpy()):
            print("nnd MISMATCH", (n, m), nm); bad += 1
print("mismatches:", bad)
