"""dev: the grouped backward launch of one depth against float64 (dx, dw of every member)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloudaae_amd import _lib
from tools.bench_fc import Layer, P
L = _lib.lib(); s = _lib.stream()
for kv in sys.argv[2:]:
    k, v = kv.split("="); _lib.set_knob(k, int(v))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for shapes in ([(1024, 12288, False), (256, 3, False), (256, 3, False)], [(1024, 1024, True), (512, 256, True), (512, 256, True)]):
    layers = [Layer(L, M, K, N, bn) for K, N, bn in shapes]
    arr = (_lib.FcLayer * len(layers))()
    for l, rec in zip(layers, arr):
        l.fill(rec)
    decay = torch.full((1,), 0.9, device="cuda")
    assert L.cloudaae_fc_forward_group(M, len(layers), arr, 1, P(decay), s) == 0
    for l in layers:
        l.dx.zero_()
    assert L.cloudaae_fc_backward_group(M, len(layers), arr, 1, s) == 0, L.cloudaae_last_error()
    torch.cuda.synchronize()
    for i, l in enumerate(layers):
        x, W, d = l.x.double(), l.W.double(), l.dout.double()
        if l.bn:
            y = x @ W + l.b.double()
            mu, var = y.mean(0), y.var(0, unbiased=False)
            xh = (y - mu) / torch.sqrt(var + 1e-3)
            z = xh * l.gamma.double() + l.beta.double()
            d = d * (z > 0)
            dxh = d * l.gamma.double()
            d = (dxh - dxh.mean(0) - xh * (dxh * xh).mean(0)) / torch.sqrt(var + 1e-3)
        dx, dw = d @ W.T, x.T @ d
        e1 = float((l.dx.double() - dx).abs().max() / dx.abs().max())
        e2 = float((l.dw.double() - dw).abs().max() / dw.abs().max())
        print("M=%d member %d %s: dx err %.2e  dw err %.2e" % (M, i, shapes[i], e1, e2))
