"""Chamfer forward (cloudaae_nn_distance) at the train step's size: both kernels.
    python tools/bench_chamfer.py [B] [N]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib  # noqa: E402
from tools.bench_fc import timeit  # noqa: E402

L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for scale, off in ((0.05, 0.0), (0.05, 1.0)):
    a = torch.randn(B, N, 3, device="cuda") * scale + off
    c = torch.randn(B, N, 3, device="cuda") * scale + off
    d1, d2 = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
    i1, i2 = torch.empty(B, N, dtype=torch.int32, device="cuda"), torch.empty(B, N, dtype=torch.int32, device="cuda")
    out = {}
    for k in ("0", "1"):             # first generation | matrix-core filter (scores as bf16 splits)
        _lib.set_knob("CLOUDAAE_NN_FILTER", int(k))
        f = lambda: L.cloudaae_nn_distance(B, N, a.data_ptr(), N, c.data_ptr(), d1.data_ptr(), i1.data_ptr(),
                                           d2.data_ptr(), i2.data_ptr(), _lib.stream())
        us = timeit(f, 30)
        out[k] = (us, d1.clone(), i1.clone(), d2.clone(), i2.clone())
        print("offset %.1f kernel %s: %.1f us  (%.2f T pairs/s)" % (off, k, us, 2.0 * B * N * N / us / 1e6))
    same = all(torch.equal(x, y) for x, y in zip(out["0"][1:], out["1"][1:]))
    print("  identical results:", same)
