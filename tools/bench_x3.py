"""The three dgcnn_agg products (utils/tf_util.py:161-166 and its two gradient products) as split (3 x bf16) products:
time per launch of the first-generation kernel (CLOUDAAE_X3_GEN1=1), of the streamed kernels with the weight planes
split inside the call, and with planes split once (cloudaae_x3_split + cloudaae_gemm_bf16x3p), against the fp32 MFMA
kernels; error vs float64; and whether the generations agree bit for bit (they must: same piece products, same order)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream()
C = L._cdll


def t(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it


P = lambda v: v.data_ptr() if v is not None else None  # noqa: E731
for B in [int(a) for a in sys.argv[1:]] or (32, 128):
    M, K, N = B * 1024, 320, 1024
    X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") / 18
    dY = torch.randn(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    Y = torch.empty(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda")
    Y3 = torch.empty(M, N, device="cuda"); dX3 = torch.empty(M, K, device="cuda"); dW3 = torch.empty(K, N, device="cuda")
    Y1 = torch.empty(M, N, device="cuda"); dX1 = torch.empty(M, K, device="cuda")
    parts = C.cloudaae_gemm_f32_colstats_parts(M, N, K); cs = torch.empty(parts * 2 * N, dtype=torch.float64, device="cuda")
    fwd = lambda: C.cloudaae_gemm_f32_colstats(0, 0, M, N, K, P(X), K, P(W), N, P(Y), N, P(b), P(cs), s)
    dx = lambda: C.cloudaae_gemm_f32(0, 1, M, K, N, P(dY), N, P(W), N, P(dX), K, None, 0, s)
    dw = lambda: C.cloudaae_gemm_f32(1, 0, K, N, M, P(X), K, P(dY), N, P(dW), N, None, 0, s)

    def x3(Yo, dXo, cso):
        return (lambda: C.cloudaae_gemm_bf16x3(0, 0, M, N, K, P(X), K, P(W), N, P(Yo), N, P(b), 0, P(cso), s),
                lambda: C.cloudaae_gemm_bf16x3(0, 1, M, K, N, P(dY), N, P(W), N, P(dXo), K, None, 0, None, s),
                lambda: C.cloudaae_gemm_bf16x3(1, 0, K, N, M, P(X), K, P(dY), N, P(dW3), N, None, 0, None, s))
    C.cloudaae_set_knob(b"CLOUDAAE_X3_GEN1", 1)
    parts1 = C.cloudaae_gemm_bf16x3_colstats_parts(M, N, K); cs1 = torch.empty(parts1 * 2 * N, dtype=torch.float64, device="cuda")
    f1 = x3(Y1, dX1, cs1)
    for f in f1:
        assert f() == 0, C.cloudaae_last_error()
    t1 = [t(f) for f in f1]
    C.cloudaae_unset_knob(b"CLOUDAAE_X3_GEN1")
    parts3 = C.cloudaae_gemm_bf16x3_colstats_parts(M, N, K); cs3 = torch.empty(parts3 * 2 * N, dtype=torch.float64, device="cuda")
    f3 = x3(Y3, dX3, cs3)
    for f in (fwd, dx, dw) + f3:
        assert f() == 0, C.cloudaae_last_error()
    torch.cuda.synchronize()
    print("B=%d  generations agree bit for bit: y %s  dX %s   column sums rel %.1e" % (
        B, torch.equal(Y1, Y3), torch.equal(dX1, dX3),
        float((cs1.reshape(parts1, 2, N).sum(0) - cs3.reshape(parts3, 2, N).sum(0)).abs().max() / cs1.abs().max())))
    rows = slice(0, 8192)
    ref_y = X[rows].double() @ W.double() + b.double()
    ref_dx = dY[rows].double() @ W.double().t()
    ref_dw = X.double().t() @ dY.double()
    e = lambda got, ref: float((got.double() - ref).abs().max() / ref.abs().max())  # noqa: E731
    print("      max |err| / max |ref| vs float64:  fp32 MFMA  y %.2e dX %.2e dW %.2e | split  y %.2e dX %.2e dW %.2e"
          % (e(Y[rows], ref_y), e(dX[rows], ref_dx), e(dW, ref_dw), e(Y3[rows], ref_y), e(dX3[rows], ref_dx), e(dW3, ref_dw)))
    # planes split once
    pf = torch.empty(C.cloudaae_x3_planes_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
    pb = torch.empty(C.cloudaae_x3_planes_bytes(K, N) // 2, dtype=torch.bfloat16, device="cuda")
    split = lambda: (C.cloudaae_x3_split(N, K, P(W), N, 1, P(pf), s), C.cloudaae_x3_split(K, N, P(W), N, 0, P(pb), s))
    assert split() == (0, 0), C.cloudaae_last_error()
    Yp = torch.empty(M, N, device="cuda"); dXp = torch.empty(M, K, device="cuda")
    fwdp = lambda: C.cloudaae_gemm_bf16x3p(M, N, K, P(X), K, P(pf), P(Yp), N, P(b), 0, P(cs3), s)
    dxp = lambda: C.cloudaae_gemm_bf16x3p(M, K, N, P(dY), N, P(pb), P(dXp), K, None, 0, None, s)
    assert fwdp() == 0 and dxp() == 0, C.cloudaae_last_error()
    torch.cuda.synchronize()
    print("      planes split once: y %s dX %s" % (torch.equal(Yp, Y3), torch.equal(dXp, dX3)))
    print("      time us: fp32 MFMA fwd %6.1f dX %6.1f dW %6.1f | gen1 fwd %6.1f dX %6.1f dW %6.1f | streamed (split in call) fwd %6.1f "
          "dX %6.1f dW %6.1f | planes given fwd %6.1f dX %6.1f, weight splits %5.1f"
          % (t(fwd), t(dx), t(dw), t1[0], t1[1], t1[2], t(f3[0]), t(f3[1]), t(f3[2]), t(fwdp), t(dxp), t(split)))
