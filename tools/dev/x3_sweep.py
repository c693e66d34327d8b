"""Streamed split products at B = 32 / 128 under the tile / staging knobs (CLOUDAAE_X3_TM, CLOUDAAE_X3_AS); checks against gen1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream(); C = L._cdll
P = lambda v: v.data_ptr() if v is not None else None  # noqa: E731
def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
for B in (32, 128):
    M, K, N = B * 1024, 320, 1024
    X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") / 18; dY = torch.randn(M, N, device="cuda")
    Y = torch.empty(M, N, device="cuda"); dX = torch.empty(M, K, device="cuda")
    Y1 = torch.empty(M, N, device="cuda"); dX1 = torch.empty(M, K, device="cuda")
    C.cloudaae_set_knob(b"CLOUDAAE_X3_GEN1", 1)
    C.cloudaae_gemm_bf16x3(0, 0, M, N, K, P(X), K, P(W), N, P(Y1), N, None, 0, None, s)
    C.cloudaae_gemm_bf16x3(0, 1, M, K, N, P(dY), N, P(W), N, P(dX1), K, None, 0, None, s)
    C.cloudaae_unset_knob(b"CLOUDAAE_X3_GEN1")
    pf = torch.empty(C.cloudaae_x3_planes_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
    pb = torch.empty(C.cloudaae_x3_planes_bytes(K, N) // 2, dtype=torch.bfloat16, device="cuda")
    C.cloudaae_x3_split(N, K, P(W), N, 1, P(pf), s); C.cloudaae_x3_split(K, N, P(W), N, 0, P(pb), s)
    fwd = lambda: C.cloudaae_gemm_bf16x3p(M, N, K, P(X), K, P(pf), P(Y), N, None, 0, None, s)
    dx = lambda: C.cloudaae_gemm_bf16x3p(M, K, N, P(dY), N, P(pb), P(dX), K, None, 0, None, s)
    for tm in (2, 1):
        for as_ in (1, 2):
            C.cloudaae_set_knob(b"CLOUDAAE_X3_TM", tm); C.cloudaae_set_knob(b"CLOUDAAE_X3_AS", as_)
            Y.fill_(float("nan")); dX.fill_(float("nan"))
            assert fwd() == 0 and dx() == 0, C.cloudaae_last_error()
            torch.cuda.synchronize()
            print("B=%d TM=%d AS=%d  fwd %7.1f us  dX %7.1f us   equal to gen1: %s %s" % (B, tm, as_, t(fwd), t(dx), torch.equal(Y, Y1), torch.equal(dX, dX1)), flush=True)
