"""Streamed split products under a knob sweep: median of event-timed groups, variants interleaved (the boxes' clocks drift)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream(); C = L._cdll
P = lambda v: v.data_ptr() if v is not None else None  # noqa: E731
KNOB = sys.argv[1].encode()
VALUES = [int(v) for v in sys.argv[2].split(",")]
WHICH = sys.argv[3] if len(sys.argv) > 3 else "fwd"
def group(fn, it=10):
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
    pf = torch.empty(C.cloudaae_x3_planes_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
    pb = torch.empty(C.cloudaae_x3_planes_bytes(K, N) // 2, dtype=torch.bfloat16, device="cuda")
    C.cloudaae_x3_split(N, K, P(W), N, 1, P(pf), s); C.cloudaae_x3_split(K, N, P(W), N, 0, P(pb), s)
    fns = {"fwd": lambda: C.cloudaae_gemm_bf16x3p(M, N, K, P(X), K, P(pf), P(Y), N, None, 0, None, s),
           "dx": lambda: C.cloudaae_gemm_bf16x3p(M, K, N, P(dY), N, P(pb), P(dX), K, None, 0, None, s)}
    fn = fns[WHICH]
    times = {v: [] for v in VALUES}
    for rep in range(12):
        for v in VALUES:
            C.cloudaae_set_knob(KNOB, v)
            assert fn() == 0, C.cloudaae_last_error()
            times[v].append(group(fn))
    for v in VALUES:
        t = sorted(times[v][2:])
        print("B=%d %s %s=%d  median %7.1f us  min %7.1f" % (B, WHICH, KNOB.decode(), v, t[len(t) // 2], t[0]), flush=True)
