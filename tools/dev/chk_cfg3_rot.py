"""dev: configs[3]'s per-GPU step (B = 128) against the oracle: the pose heads' gradients in detail"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cloudaae_amd import train_cloudAAE_ycbv as T, _lib
from oracle import model_oracle as MO
for kv in sys.argv[2:]:
    k, v = kv.split("="); _lib.set_knob(k, int(v))
B, N, kn = 128, 1024, 10
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1000 + B
graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, replay=False, gemm_dtype="f32", k_neighbor=kn)
V = MO.Vars(seed=31)
with torch.no_grad():
    MO.forward_losses(MO.synthetic_batch(2, N, seed=1), V, N, is_training=False)
batch = MO.synthetic_batch(B, N, seed=seed)
graph.store.load_state_dict(V.state_dict())
opt = MO.AdamTF()
dev = {k: v.cuda() for k, v in batch.items()}
out = graph.train_step(dev)
torch.cuda.synchronize()
idx = [out["end_points"]["nn_idx%d" % i].cpu().clone() for i in (1, 2, 3, 4)]
ref, grads = MO.train_step(batch, V, opt, 0, N, B, k=kn, nn_override=idx)
print("losses gpu", {k: float(out[k]) for k in ("xyz_loss", "trans_loss", "axag_loss")}, "oracle", {k: float(ref[k]) for k in ("xyz_loss", "trans_loss", "axag_loss")})
print("rot_pred max abs diff", float((out["rot_pred"].cpu().double() - ref["rot_pred"].double()).abs().max()) if "rot_pred" in out else None)
offs = graph.store.offsets
for nme, g in grads.items():
    if "rot" in nme or "trans" in nme:
        got = graph.store.flat_grads[offs[nme]:offs[nme] + g.numel()].view(g.shape).cpu().double()
        e = float((got - g.double()).abs().max() / (g.double().abs().max() + 1e-30))
        print("  %-36s max-rel err %.2e   |g|max %.3e" % (nme, e, float(g.abs().max())))
th = ref["axag_loss_perSample"]
print("largest angles", th.sort().values[-4:].tolist())
for nme in ("dgcnn_rot_fc2/weights", "dgcnn_rot_fc2/bn/beta", "dgcnn_rot_fc1/weights", "dgcnn_rot_fc1/bn/beta"):
    g = grads[nme].double()
    got = graph.store.flat_grads[offs[nme]:offs[nme] + g.numel()].view(g.shape).cpu().double()
    err = (got - g).abs()
    col = err.reshape(-1, g.shape[-1]).max(0).values / g.abs().max()
    top = col.sort(descending=True)
    print(nme, "columns with the largest error:", top.indices[:4].tolist(), ["%.1e" % v for v in top.values[:4].tolist()],
          "| columns above 1e-4:", int((col > 1e-4).sum()), "of", col.numel())
