"""Run one dgcnn_agg product as a split (3 x bf16) product a few times (for counter passes): fwd | dx | dw [B]."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream(); C = L._cdll
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
M, K, N = B * 1024, 320, 1024
X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") / 18; Y = torch.randn(M, N, device="cuda")
dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda")
P = lambda v: v.data_ptr()  # noqa: E731
pf = torch.empty(C.cloudaae_x3_planes_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
pb = torch.empty(C.cloudaae_x3_planes_bytes(K, N) // 2, dtype=torch.bfloat16, device="cuda")
C.cloudaae_x3_split(N, K, P(W), N, 1, P(pf), s); C.cloudaae_x3_split(K, N, P(W), N, 0, P(pb), s)
for _ in range(6):
    if which == "fwd":
        C.cloudaae_gemm_bf16x3p(M, N, K, P(X), K, P(pf), P(Y), N, None, 0, None, s)
    elif which == "dx":
        C.cloudaae_gemm_bf16x3p(M, K, N, P(Y), N, P(pb), P(dX), K, None, 0, None, s)
    else:
        C.cloudaae_gemm_bf16x3(1, 0, K, N, M, P(X), K, P(Y), N, P(dW), N, None, 0, None, s)
torch.cuda.synchronize()
