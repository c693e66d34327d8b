"""Time the dgcnn_agg block with bf16 operands: the three products from fp32 tensors (gemm_bf16.hip) and from
bfloat16 tensors (gemm_b16.hip), the batch-norm passes on an fp32 / bfloat16 y, the conversions."""
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
P = lambda v: v.data_ptr() if v is not None else None
for B in (32, 128, 256):
    N_ = 1024
    M, K, N = B * N_, 320, 1024
    X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") / 18; Y = torch.empty(M, N, device="cuda")
    dY = torch.randn(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda")
    b = torch.randn(N, device="cuda")
    X16, W16, dY16 = X.bfloat16(), W.bfloat16(), dY.bfloat16()
    Y16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    parts = L.cloudaae_gemm_b16_colstats_parts(M, N, K)
    cs = torch.empty(parts * 2 * N, dtype=torch.float64, device="cuda")
    fwd = lambda: L.cloudaae_gemm_bf16_colstats(0, 0, M, N, K, P(X), K, P(W), N, P(Y), N, P(b), P(cs), s)
    dx = lambda: L.cloudaae_gemm_bf16(0, 1, M, K, N, P(dY), N, P(W), N, P(dX), K, None, 0, s)
    dw = lambda: L.cloudaae_gemm_bf16(1, 0, K, N, M, P(X), K, P(dY), N, P(dW), N, None, 0, s)
    fwd16 = lambda: L.cloudaae_gemm_b16(0, 0, M, N, K, P(X16), K, P(W16), N, P(Y16), N, 1, P(b), 0, P(cs), s)
    dx16 = lambda: L.cloudaae_gemm_b16(0, 1, M, K, N, P(dY16), N, P(W16), N, P(dX), K, 0, None, 0, None, s)
    dw16 = lambda: L.cloudaae_gemm_b16(1, 0, K, N, M, P(X16), K, P(dY16), N, P(dW), N, 0, None, 0, None, s)
    cvx = lambda: L.cloudaae_to_bf16(M * K, P(X), P(X16), s)
    # batch norm passes
    gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    decay = torch.full((1,), 0.9, device="cuda")
    em, ev, sm, sv = (torch.zeros(N, device="cuda") for _ in range(4))
    pooled = torch.empty(B, N, device="cuda"); dpooled = torch.randn(B, N, device="cuda")
    ps = torch.empty(B * 3 * N, dtype=torch.float64, device="cuda")
    dg, db = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ws = torch.empty(int(L.cloudaae_bn_workspace_bytes(N)) // 8 + 1, dtype=torch.float64, device="cuda")
    fwd(); fwd16()
    bnf = lambda: L.cloudaae_bn_forward_colstats(M, N, P(Y), N, P(gamma), P(beta), 1, P(decay), P(em), P(ev), P(sm), P(sv), 1, None, N, N_, 1, P(pooled), None, P(ps), P(ws), P(cs), parts, s)
    bnb = lambda: L.cloudaae_bn_backward(M, N, P(Y), N, P(gamma), P(beta), P(sm), P(sv), 1, 1, None, N, N_, 1, P(dpooled), P(pooled), None, P(dY), N, P(dg), P(db), None, 0, P(ps), P(ws), s)
    bnf16 = lambda: L.cloudaae_bn_meanpool_forward16(M, N, P(Y16), N, P(gamma), P(beta), P(decay), P(em), P(ev), P(sm), P(sv), N_, P(pooled), P(ps), P(ws), P(cs), parts, s)
    bnb16 = lambda: L.cloudaae_bn_meanpool_backward16(M, N, P(Y16), N, P(gamma), P(beta), P(sm), P(sv), N_, P(dpooled), P(dY16), N, P(dg), P(db), None, 0, P(ps), P(ws), s)
    for f in (fwd, dx, dw, fwd16, dx16, dw16, cvx, bnf, bnb, bnf16, bnb16):
        assert f() == 0, L.cloudaae_last_error()
    print("B=%3d fp32 storage: fwd %7.1f  dX %7.1f  dW %7.1f  bn fwd %6.1f  bn bwd %6.1f   (us)" % (B, t(fwd), t(dx), t(dw), t(bnf), t(bnb)))
    print("      bf16 storage: fwd %7.1f  dX %7.1f  dW %7.1f  bn fwd %6.1f  bn bwd %6.1f   x -> bf16 %6.1f" % (t(fwd16), t(dx16), t(dw16), t(bnf16), t(bnb16), t(cvx)))
