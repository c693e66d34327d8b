"""dev: time of the dgcnn_agg weight-gradient product (first-generation split-product kernel), warm clocks"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream(); C = L._cdll
P = lambda v: v.data_ptr()  # noqa: E731
for B in (32, 128):
    M, K, N = B * 1024, 320, 1024
    X = torch.randn(M, K, device="cuda"); Y = torch.randn(M, N, device="cuda"); dW = torch.empty(K, N, device="cuda")
    go = lambda: C.cloudaae_gemm_bf16x3(1, 0, K, N, M, P(X), K, P(Y), N, P(dW), N, None, 0, None, s)
    t_end = time.time() + 1.0
    while time.time() < t_end:
        for _ in range(20): go()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): go()
    b.record(); torch.cuda.synchronize()
    ref = X.double().t() @ Y.double()
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    print("dW B=%d: %.1f us  rel.err vs fp64 %.2e  checksum %.9e" % (B, a.elapsed_time(b) * 10, err, float(dW.double().sum())))
