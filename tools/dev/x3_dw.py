"""The dgcnn_agg weight-gradient product (x^T dy) as a split product: knob sweep, variants interleaved."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream(); C = L._cdll
P = lambda v: v.data_ptr() if v is not None else None  # noqa: E731
KNOB = sys.argv[1].encode()
VALUES = [int(v) for v in sys.argv[2].split(",")]
def group(fn, it=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
for B in (32, 128):
    M, K, N = B * 1024, 320, 1024
    X = torch.randn(M, K, device="cuda"); dY = torch.randn(M, N, device="cuda")
    dW = torch.empty(K, N, device="cuda")
    ref = X.double().t() @ dY.double()
    fn = lambda: C.cloudaae_gemm_bf16x3(1, 0, K, N, M, P(X), K, P(dY), N, P(dW), N, None, 0, None, s)
    times = {v: [] for v in VALUES}
    err = {}
    for rep in range(10):
        for v in VALUES:
            C.cloudaae_set_knob(KNOB, v)
            assert fn() == 0, C.cloudaae_last_error()
            torch.cuda.synchronize()
            err[v] = float((dW.double() - ref).abs().max() / ref.abs().max())
            times[v].append(group(fn))
    for v in VALUES:
        t = sorted(times[v][2:])
        print("B=%d dW %s=%d  median %7.1f us  min %7.1f   err vs f64 %.2e" % (B, KNOB.decode(), v, t[len(t) // 2], t[0], err[v]), flush=True)
