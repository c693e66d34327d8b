import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cloudaae_amd import train_cloudAAE_ycbv as T
from oracle import model_oracle as MO
for (B, N, k, seed) in [(4, 256, 10, 17), (4, 256, 20, 17), (4, 256, 20, 18), (8, 256, 20, 17), (4, 1024, 20, 17)]:
    graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, k_neighbor=k)
    V = MO.Vars(seed=13)
    batch = MO.synthetic_batch(B, N, seed=seed)
    with torch.no_grad():
        MO.forward_losses(batch, V, N, is_training=False, k=k)
    graph.store.load_state_dict(V.state_dict())
    out = graph.train_step({kk: v.cuda() for kk, v in batch.items()})
    ref, grads = MO.train_step(batch, V, MO.AdamTF(), 0, N, B, k=k)
    print(B, N, k, seed, [(kk, abs(float(out[kk].detach()) - float(ref[kk]))) for kk in ("xyz_loss", "trans_loss", "axag_loss")])
    for i in range(1, 5):
        a = out["end_points"]["nn_idx%d" % i].cpu().long(); b = ref["end_points"]["nn_idx%d" % i].long()
        print("   layer", i, "idx mismatches", int((a != b).sum()), "of", a.numel())
    print("   emb rel", float((out["end_points"]["embedding"].cpu() - ref["end_points"]["embedding"]).abs().max() / ref["end_points"]["embedding"].abs().max()))
