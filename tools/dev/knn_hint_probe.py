"""Would the PREVIOUS layer's neighbours give the kNN over 64 channels a usable bound?  For layers 2-4 of a B = 32 step:
tau_hint = max over the previous layer's k neighbours of the distance in THIS layer's features; how many candidates lie at or
below it (the queue of knn64_wide_kernel holds 144), against the count below the true k-th distance (= k)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import train_cloudAAE_ycbv as T
B, N, k = 32, 1024, 10
g = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, replay=False)
el = T.synthetic_element(B, N, g.device, seed=123456789)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    out = g.train_step(el)
out = g.forward(el, is_training=True)
ep = out["end_points"]
idx = [ep["nn_idx%d" % i].long() for i in range(1, 5)]
# the concat buffer: net1..net3 are the kNN inputs of layers 2..4
from cloudaae_amd.utils import _functions as F
cat = None
for obj in (ep.get("concat"), getattr(g, "_last_concat", None)):
    if obj is not None:
        cat = obj
if cat is None:
    # rebuild the features from the graph: run the encoder pieces eagerly
    import cloudaae_amd.utils.tf_util as tu
    store = []
    orig = tu.knn
    def spy(adj, k=9):
        store.append(adj.points)
        return orig(adj, k=k)
    tu.knn = spy
    import cloudaae_amd.models.pointnet_ycb_23_decoder_4 as M
    M.tf_util.knn = spy
    g.forward(el, is_training=True)
    tu.knn = orig
    feats = [s.reshape(B, N, -1)[..., :64].contiguous() if s.shape[-1] >= 64 else None for s in store[-4:]]
else:
    feats = [None, cat[..., 0:64], cat[..., 64:128], cat[..., 128:192]]
for layer in (2, 3, 4):
    x = feats[layer - 1]
    if x is None:
        continue
    x = x.double()
    D = torch.cdist(x, x) ** 2                                   # [B,N,N]
    prev = idx[layer - 2]                                        # previous layer's neighbours
    dprev = torch.gather(D, 2, prev)
    tau = dprev.max(2, keepdim=True).values * (1 + 1e-5)
    cnt = (D <= tau).sum(2).float()
    kth = D.topk(k, dim=2, largest=False).values[..., -1:]
    same = (torch.gather(D, 2, idx[layer - 1]) <= tau).float().mean()
    print("layer %d: candidates <= tau_hint per query: mean %.1f  p99 %.0f  max %.0f   (k = %d; fraction of true neighbours inside: %.4f; "
          "tau_hint / d_k: median %.2f  max %.1f)" % (layer, float(cnt.mean()), float(cnt.flatten().kthvalue(int(0.99 * cnt.numel())).values),
                                                      float(cnt.max()), k, float(same), float((tau / kth).median()), float((tau / kth).max())))
