"""Run one dgcnn_agg product as a split (3 x bf16) product a few times (for counter passes): fwd | dx | dw."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib(); s = _lib.stream()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
M, K, N = B * 1024, 320, 1024
X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") / 18; Y = torch.randn(M, N, device="cuda")
dX = torch.empty(M, K, device="cuda"); dW = torch.empty(K, N, device="cuda")
P = lambda v: v.data_ptr()
for _ in range(6):
    if which == "fwd":
        L.cloudaae_gemm_bf16x3(0, 0, M, N, K, P(X), K, P(W), N, P(Y), N, None, 0, None, s)
    elif which == "dx":
        L.cloudaae_gemm_bf16x3(0, 1, M, K, N, P(Y), N, P(W), N, P(dX), K, None, 0, None, s)
    else:
        L.cloudaae_gemm_bf16x3(1, 0, K, N, M, P(X), K, P(Y), N, P(dW), N, None, 0, None, s)
torch.cuda.synchronize()
