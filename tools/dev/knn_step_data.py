"""dev: what do the C = 64 kNN inputs of a training step look like?  (norms, norms around the centre, neighbour distances
and the gaps between consecutive neighbours, all relative to |x|^2)   python tools/dev/knn_step_data.py [B] [N] [k] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import train_cloudAAE_ycbv as T
from cloudaae_amd.utils import tf_util
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
graph = T.TrainGraph({"num_point": N, "gpu": 0}, {"optimizer": "adam"}, {"batch_size": B, "learning_rate": 0.0008}, replay=False,
                     k_neighbor=K)
el = T.synthetic_element(B, N, graph.device, seed=123456789)
for s in range(steps):
    tf_util.KNN_TAP = [] if s == steps - 1 else None
    graph.train_step(el)
torch.cuda.synchronize()
for li, x in enumerate(tf_util.KNN_TAP):
    x = x[:4, :, :64].double()
    sq = (x * x).sum(-1)
    c = x.mean(1, keepdim=True)
    sc = ((x - c) ** 2).sum(-1)
    d = torch.cdist(x, x) ** 2
    dk, _ = d.topk(K + 5, largest=False)
    gaps = (dk[..., 1:] - dk[..., :-1])
    rel = sq.mean()
    print("layer %d: |x|^2 mean %.3g max %.3g | centred mean %.3g max %.3g (ratio %.3g) | d_k / |x|^2 %.3g | gap between neighbours / |x|^2: "
          "median %.2g  10%% %.2g  1%% %.2g | zero gaps %.2g%%"
          % (li + 2, sq.mean(), sq.max(), sc.mean(), sc.max(), sc.mean() / sq.mean(), dk[..., K - 1].mean() / rel,
             gaps.median() / rel, gaps.flatten().kthvalue(int(gaps.numel() * 0.1)).values / rel,
             gaps.flatten().kthvalue(int(gaps.numel() * 0.01)).values / rel, 100.0 * (gaps == 0).double().mean()))

# the kNN launch itself on these inputs, against the same shape of post-ReLU Gaussian features
from cloudaae_amd import _lib
L = _lib.lib()


def time_knn(x, tag):
    b, n, _ = x.shape
    ld = x.stride(1)
    out = torch.empty((b, n, K), dtype=torch.int32, device="cuda")
    go = lambda: _lib.check(L.cloudaae_knn(b, n, 64, ld, K, x.data_ptr(), out.data_ptr(), _lib.stream()), "knn")
    for _ in range(20):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        go()
    e1.record(); torch.cuda.synchronize()
    print("  %-28s %.1f us" % (tag, e0.elapsed_time(e1) * 1e3 / 20))


for li, x in enumerate(tf_util.KNN_TAP):
    x = x.contiguous()
    time_knn(x, "layer %d features" % (li + 2))
    perm = torch.stack([torch.randperm(x.shape[1], device="cuda") for _ in range(x.shape[0])])
    time_knn(torch.gather(x, 1, perm[..., None].expand_as(x)).contiguous(), "  ... rows shuffled")
time_knn(torch.relu(torch.randn_like(tf_util.KNN_TAP[0])).contiguous(), "relu(randn)")
