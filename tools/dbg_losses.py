import sys; sys.path.insert(0,'.')
import torch
from cloudaae_amd import train_cloudAAE_ycbv as T
from oracle import model_oracle as MO
for (B,N,seed) in [(2,256,5),(2,256,6),(4,128,5),(8,256,5)]:
    graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B})
    V = MO.Vars(seed=11)
    batch = MO.synthetic_batch(B, N, seed=seed, single_class=0 if B == 2 else None)
    with torch.no_grad():
        MO.forward_losses(batch, V, N, is_training=False)
    graph.store.load_state_dict(V.state_dict())
    out = graph.train_step({k: v.cuda() for k, v in batch.items()})
    ref, grads = MO.train_step(batch, V, MO.AdamTF(), 0, N, B)
    print(B,N,seed,[(k, float(out[k]), float(ref[k]), abs(float(out[k])-float(ref[k]))) for k in ("xyz_loss","trans_loss","axag_loss")])
    print("  recon rel", float((out["xyz_recon"].cpu()-ref["xyz_recon"]).abs().max()/ref["xyz_recon"].abs().max()),
          "emb rel", float((out["end_points"]["embedding"].cpu()-ref["end_points"]["embedding"]).abs().max()/ref["end_points"]["embedding"].abs().max()))
