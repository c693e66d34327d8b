"""Where does the streamed split product differ from the first generation? (debugging aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream(); C = L._cdll
P = lambda v: v.data_ptr() if v is not None else None  # noqa: E731
for (tb, M, N, K) in [(0, 32768, 1024, 320), (1, 32768, 320, 1024)]:
    torch.manual_seed(1)
    A = torch.randn(M, K, device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
    C1 = torch.empty(M, N, device="cuda"); C2 = torch.empty(M, N, device="cuda")
    C.cloudaae_set_knob(b"CLOUDAAE_X3_GEN1", 1)
    assert C.cloudaae_gemm_bf16x3(0, tb, M, N, K, P(A), K, P(B), B.shape[1], P(C1), N, None, 0, None, s) == 0
    C.cloudaae_unset_knob(b"CLOUDAAE_X3_GEN1")
    for rep in range(3):
        assert C.cloudaae_gemm_bf16x3(0, tb, M, N, K, P(A), K, P(B), B.shape[1], P(C2), N, None, 0, None, s) == 0
        torch.cuda.synchronize()
        bad = (C1 != C2)
        ref = A.double() @ (B.double().t() if tb else B.double())
        print(tb, M, N, K, "rep", rep, "mismatches", int(bad.sum()), "max err gen1 %.2e gen2 %.2e" % (float((C1 - ref).abs().max()), float((C2 - ref).abs().max())))
        if bad.any():
            idx = bad.nonzero()
            rows = idx[:, 0].unique(); cols = idx[:, 1].unique()
            print("   rows", rows[:20].tolist(), "n", len(rows), " cols", cols[:20].tolist(), "n", len(cols))
            print("   rows%32", sorted(set((rows % 32).tolist()))[:40], " rows//32", sorted(set((rows // 32).tolist()))[:40])
            print("   cols%32", sorted(set((cols % 32).tolist()))[:40], " cols//32", sorted(set((cols // 32).tolist()))[:40])
