"""Experiments on csrc/fc.hip: K=.. N=.. BN=0/1 python tools/try_fc.py -- sweeps the launch shapes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib  # noqa: E402
from tools.bench_fc import timeit  # noqa: E402

L = _lib.lib()
s = _lib.stream()
M, K, N = 32, int(os.environ.get("K", 1024)), int(os.environ.get("N", 12288))
bn = os.environ.get("BN", "0") == "1"
x = torch.randn(M, K, device="cuda")
W = torch.randn(K, N, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda")
y, out = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
sm, sv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
mean, var = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
decay = torch.full((1,), 0.9, device="cuda")
tk = torch.zeros(L.cloudaae_fc_forward_tickets(N), dtype=torch.int32, device="cuda")
dout = torch.randn(M, N, device="cuda")
dx, dw = torch.zeros(M, K, device="cuda"), torch.empty(K, N, device="cuda")
dbias, dg, db = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
G = (lambda t: P(t)) if bn else (lambda t: None)


def fwd():
    L.cloudaae_fc_forward(M, K, N, P(x), K, P(W), P(b), G(gamma), G(beta), 1, G(decay), G(sm), G(sv), G(mean), G(var),
                          1, P(y), G(out), 1, G(tk), s)


def bwd(dxb, dwb):
    def f():
        L.cloudaae_fc_backward(M, K, N, P(x), K, P(W), G(y), G(gamma), G(beta), G(mean), G(var), 1, 1, P(dout), N,
                               P(dxb), K, P(dwb), 0, G(dg), G(db), P(dbias), 0, s)
    return f


fwd()
for sp in (1, 2, 4, 8, 16):
    os.environ["CLOUDAAE_FC_FWD_SPLITS"] = str(sp)
    print("fwd splits %2d: %.1f us" % (sp, timeit(fwd, 200)))
for blocks in (32, 64, 128, 256, 512, 768):
    os.environ["CLOUDAAE_FC_BWD_BLOCKS"] = str(blocks)
    print("bwd blocks %4d: both %.1f us, dw only %.1f, dx only %.1f, neither %.1f" % (
        blocks, timeit(bwd(dx, dw), 200), timeit(bwd(None, dw), 200), timeit(bwd(dx, None), 200),
        timeit(bwd(None, None), 200)))
