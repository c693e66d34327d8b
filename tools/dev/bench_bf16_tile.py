"""Time the bf16-operand GEMM at the dgcnn_agg forward shape with and without the 128 x 256 tile knob."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream()
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
for M in (32768, 131072, 262144):
    K, N = 320, 1024
    A = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda"); C = torch.empty(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    f = lambda: L.cloudaae_gemm_bf16(0, 0, M, N, K, A.data_ptr(), K, W.data_ptr(), N, C.data_ptr(), N, b.data_ptr(), 0, s)
    _lib.set_knob("CLOUDAAE_BF16_TILE256", None); f(); ref = C.clone(); t0 = t(f)
    _lib.set_knob("CLOUDAAE_BF16_TILE256", 1); f(); same = torch.equal(ref, C); t1 = t(f)
    _lib.set_knob("CLOUDAAE_BF16_TILE256", None)
    print("M=%d  128x128: %.1f us   128x256: %.1f us  same bits: %s" % (M, t0, t1, same))
