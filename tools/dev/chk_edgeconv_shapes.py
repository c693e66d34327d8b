"""Dev: the edge-convolution parity case of tests/test_01_layers_gpu.py over a list of shapes, every error printed
(python tools/dev/chk_edgeconv_shapes.py B,N,cin,cout,k,pool ...)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib  # noqa: E402
from cloudaae_amd.utils import _functions as F  # noqa: E402
from oracle import model_oracle as MO, native  # noqa: E402


def rel(got, want):
    got, want = got.detach().cpu().double().numpy(), want.detach().cpu().double().numpy()
    d = np.abs(got - want)
    return d.max() / (np.abs(want).max() + 1e-30), np.unravel_index(d.argmax(), d.shape)


def run(B, N, cin, cout, k, pool):
    g = torch.Generator().manual_seed(B * N + cin + cout + k)
    x = (torch.randn(B, N, cin, generator=g) * 0.5).requires_grad_(True)
    nn_idx = torch.from_numpy(native.knn(x.detach().numpy(), k, channels=min(cin, 64))).long()
    V = MO.Vars(seed=3)
    edge = MO.get_edge_feature(x, nn_idx, k)
    y = MO.conv2d_1x1(edge, cout, "ec", V, True, True, 0.5)
    V.p["ec/biases"].data.normal_(0, 0.1, generator=g)
    V.p["ec/bn/gamma"].data.uniform_(0.5, 1.5, generator=g)
    V.p["ec/bn/beta"].data.normal_(0, 0.1, generator=g)
    y = MO.conv2d_1x1(edge, cout, "ec", V, True, True, 0.5)
    want = y.mean(2) if pool == "mean" else y.amax(2)
    w = torch.randn(B, N, cout, generator=g)
    (want * w).sum().backward()
    xd = x.detach().cuda().requires_grad_(True)
    P = {n: p.detach().cuda().requires_grad_(True) for n, p in V.p.items()}
    sm, sv = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")
    decay = torch.full((1,), 0.5, device="cuda")
    out = F.EdgeConvFn.apply(xd, nn_idx.int().cuda(), P["ec/weights"].reshape(2 * cin, cout), P["ec/biases"],
                             P["ec/bn/gamma"], P["ec/bn/beta"], sm, sv, decay, True, 1 if pool == "mean" else 2, None)
    (out * w.cuda()).sum().backward()
    print((B, N, cin, cout, k, pool), "out %.2e" % rel(out, want)[0], "dx %.2e at %s" % rel(xd.grad, x.grad),
          " ".join("%s %.2e" % (n.split("/")[-1], rel(P[n].grad, V.p[n].grad)[0]) for n in ("ec/weights", "ec/bn/gamma", "ec/bn/beta")),
          flush=True)
    d = (xd.grad.cpu() - x.grad).abs().amax(-1) / x.grad.abs().max()
    bad = (d > 2e-5).nonzero()
    print("   rows of dx off by more than 2e-5: %d of %d" % (len(bad), B * N), bad[:24].tolist(), flush=True)
    # an edge on the ReLU corner: |bn(y)| within round-off of zero flips between the two sides
    C = cout
    yy = y.detach().reshape(-1, C)
    print("   edge values with |z| < 2e-6 in the oracle: %d" % int(((yy.abs() < 2e-6) & (yy != 0)).sum()), flush=True)


if __name__ == "__main__":
    _lib.lib()
    for a in sys.argv[1:]:
        B, N, cin, cout, k, pool = a.split(",")
        run(int(B), int(N), int(cin), int(cout), int(k), pool)
