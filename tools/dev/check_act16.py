"""dgcnn_agg block with bf16 activations against a torch emulation (dev check)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd.utils import _functions as F
from cloudaae_amd.utils import tf_util
B, N, K, C = 4, 1024, 320, 1024
M = B * N
g = torch.Generator().manual_seed(0)
X = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
W = (torch.randn(K, C, generator=g) / 18).cuda().requires_grad_(True)
b = torch.randn(C, generator=g).cuda()
gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True), torch.randn(C, generator=g).cuda().requires_grad_(True)
em, ev = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
decay = torch.full((1,), 0.9, device="cuda")
dpooled = torch.randn(B, C, generator=g).cuda()
F.GEMM_DTYPE = "bf16"
rows = [X[:, :64], X[:, 64:128], X[:, 128:192], X[:, 192:]]
y = F.ConcatLinearFn.apply(None, W, b, 7, *rows)
print("y dtype", y.dtype)
pooled, sm, sv = F.BatchNormFn.apply(y, gamma, beta, em, ev, decay, True, True, N, 1, False, b)
pooled.backward(dpooled)
gX, gW, gg, gb = X.grad.clone(), W.grad.clone(), gamma.grad.clone(), beta.grad.clone()
# emulation
def bf(t): return t.bfloat16().float()
class Mm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w); return bf(x) @ bf(w)
    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors; d = bf(dy); return d @ bf(w).t(), bf(x).t() @ d
class Ste(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return bf(x)
    @staticmethod
    def backward(ctx, g): return g
X2, W2 = X.detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True)
g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
y2 = Mm.apply(X2, W2) + b
ys = Ste.apply(y2)
print("stored y: fraction of elements that differ from the emulation: %.2e" % float((y.float() != ys).float().mean()))
mean = y2.mean(0); var = ((y2 - mean.detach()) ** 2).mean(0)
inv = g2 * torch.rsqrt(var + 1e-3)
z = torch.relu(ys * inv + (b2 - mean * inv))
p2 = z.reshape(B, N, C).mean(1)
p2.backward(dpooled)
rel = lambda a, c: float((a - c).norm() / c.norm())
print("pooled %.2e  mean %.2e var %.2e" % (rel(pooled, p2), rel(sm, mean), rel(sv, var)))
print("dX %.2e  dW %.2e  dgamma %.2e  dbeta %.2e" % (rel(gX, X2.grad), rel(gW, W2.grad), rel(gg, g2.grad), rel(gb, b2.grad)))
