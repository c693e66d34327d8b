"""Time the three dgcnn_agg products with bf16 operands (forward, dX, dW) at B clouds of 1024 points."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
for B in (32, 128, 256):
    M, K, N = B * 1024, 320, 1024
    X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") / 18; Y = torch.empty(M, N, device="cuda")
    dY = torch.randn(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda")
    b = torch.randn(N, device="cuda")
    P = lambda v: v.data_ptr()
    fwd = lambda: L.cloudaae_gemm_bf16(0, 0, M, N, K, P(X), K, P(W), N, P(Y), N, P(b), 0, s)
    dx = lambda: L.cloudaae_gemm_bf16(0, 1, M, K, N, P(dY), N, P(W), N, P(dX), K, None, 0, s)
    dw = lambda: L.cloudaae_gemm_bf16(1, 0, K, N, M, P(X), K, P(dY), N, P(dW), N, None, 0, s)
    fwd(); dx(); dw()
    xb, wb = X.bfloat16().float(), W.bfloat16().float()
    e1 = ((xb[:4096] @ wb + b) - Y[:4096]).abs().max().item()
    e2 = ((dY[:4096].bfloat16().float() @ wb.t()) - dX[:4096]).abs().max().item()
    ref = xb.t().double() @ dY.bfloat16().double()
    e3 = ((ref - dW.double()).abs().max() / ref.abs().max()).item()
    _lib.set_knob("CLOUDAAE_BF16_LOADSONLY", 1)
    print("  loads only: fwd %7.1f us  dX %7.1f us  dW %7.1f us" % (t(fwd), t(dx), t(dw)))
    for kb in (56, 130):
        _lib.set_knob("CLOUDAAE_BF16_DYNLDS", kb * 1024)
        print("  loads only +%d KB LDS: fwd %7.1f us" % (kb, t(fwd)))
    _lib.set_knob("CLOUDAAE_BF16_DYNLDS", None)
    _lib.set_knob("CLOUDAAE_BF16_LOADSONLY", None)
    print("B=%3d  fwd %7.1f us  dX %7.1f us  dW %7.1f us   err %.2e %.2e %.2e" % (B, t(fwd), t(dx), t(dw), e1, e2, e3))
