"""Times the GEMM shapes of the B=32, N=1024 train step (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib
L = _lib.lib()
FN = "cloudaae_gemm_bf16" if os.environ.get("BF16") else "cloudaae_gemm_f32"
def run(ta, tb, M, N, K, iters=20):
    A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
    C = torch.empty((M, N), device="cuda")
    def go():
        _lib.check(getattr(L, FN)(ta, tb, M, N, K, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], C.data_ptr(), N, None, 0, _lib.stream()), "g")
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6
if __name__ == "__main__":
  for name, args in [("agg fwd", (0, 0, 32768, 1024, 320)), ("agg dX", (0, 1, 32768, 320, 1024)), ("agg dW", (1, 0, 320, 1024, 32768)),
                     ("edge fwd", (0, 0, 32768, 64, 64)), ("ec fwd2", (0, 0, 32768, 128, 64)), ("ec fwd4", (0, 0, 32768, 256, 64)), ("ec dX2", (0, 1, 32768, 64, 128)), ("ec dX4", (0, 1, 32768, 64, 256)), ("edge dX", (0, 1, 32768, 64, 64)), ("edge dW", (1, 0, 64, 64, 32768)),
                     ("edge dW2", (1, 0, 64, 128, 32768)), ("edge dW1", (1, 0, 24, 64, 32768)), ("fc fwd", (0, 0, 32, 1024, 1024)), ("fc dW", (1, 0, 1024, 1024, 32)), ("fc dX", (0, 1, 32, 1024, 1024)), ("out fwd", (0, 0, 32, 12288, 1024)), ("out dW", (1, 0, 1024, 12288, 32)),
                     ("big sq", (0, 0, 8192, 8192, 8192))]:
      us, tf = run(*args)
      print("%-9s %-28s %9.1f us %7.1f TF" % (name, args, us, tf))
