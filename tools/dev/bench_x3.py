"""The three dgcnn_agg products as split (3 x bf16) products against the fp32 MFMA kernels: time and error vs float64."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream()
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
P = lambda v: v.data_ptr() if v is not None else None
for B in (32, 128):
    M, K, N = B * 1024, 320, 1024
    X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") / 18
    dY = torch.randn(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    Y = torch.empty(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda")
    Y3 = torch.empty(M, N, device="cuda"); dX3 = torch.empty(M, K, device="cuda"); dW3 = torch.empty(K, N, device="cuda")
    parts = L.cloudaae_gemm_f32_colstats_parts(M, N, K); cs = torch.empty(parts * 2 * N, dtype=torch.float64, device="cuda")
    parts3 = L.cloudaae_gemm_bf16x3_colstats_parts(M, N, K); cs3 = torch.empty(parts3 * 2 * N, dtype=torch.float64, device="cuda")
    fwd = lambda: L.cloudaae_gemm_f32_colstats(0, 0, M, N, K, P(X), K, P(W), N, P(Y), N, P(b), P(cs), s)
    dx = lambda: L.cloudaae_gemm_f32(0, 1, M, K, N, P(dY), N, P(W), N, P(dX), K, None, 0, s)
    dw = lambda: L.cloudaae_gemm_f32(1, 0, K, N, M, P(X), K, P(dY), N, P(dW), N, None, 0, s)
    fwd3 = lambda: L.cloudaae_gemm_bf16x3(0, 0, M, N, K, P(X), K, P(W), N, P(Y3), N, P(b), 0, P(cs3), s)
    dx3 = lambda: L.cloudaae_gemm_bf16x3(0, 1, M, K, N, P(dY), N, P(W), N, P(dX3), K, None, 0, None, s)
    dw3 = lambda: L.cloudaae_gemm_bf16x3(1, 0, K, N, M, P(X), K, P(dY), N, P(dW3), N, None, 0, None, s)
    for f in (fwd, dx, dw, fwd3, dx3, dw3):
        assert f() == 0, L.cloudaae_last_error()
    torch.cuda.synchronize()
    rows = slice(0, 8192)
    ref_y = X[rows].double() @ W.double() + b.double()
    ref_dx = dY[rows].double() @ W.double().t()
    ref_dw = X.double().t() @ dY.double()
    e = lambda got, ref: float((got.double() - ref).abs().max() / ref.abs().max())
    print("B=%d  max |err| / max |ref| vs float64:  fp32 MFMA  y %.2e dX %.2e dW %.2e | split  y %.2e dX %.2e dW %.2e"
          % (B, e(Y[rows], ref_y), e(dX[rows], ref_dx), e(dW, ref_dw), e(Y3[rows], ref_y), e(dX3[rows], ref_dx), e(dW3, ref_dw)))
    st = cs.reshape(parts, 2, N).sum(0); st3 = cs3.reshape(parts3, 2, N).sum(0)
    print("      column sums: split vs fp32 kernel rel %.2e" % float((st - st3).abs().max() / st.abs().max()))
    print("      time us: fp32 MFMA fwd %6.1f dX %6.1f dW %6.1f | split fwd %6.1f dX %6.1f dW %6.1f"
          % (t(fwd), t(dx), t(dw), t(fwd3), t(dx3), t(dw3)))
