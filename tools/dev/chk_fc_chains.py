"""dev: decoder + pose heads as the model runs them (tf_util.fully_connected_chains) against float64 autograd; real widths"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloudaae_amd import _lib
from cloudaae_amd.utils import tf_util
from cloudaae_amd.utils.variables import VariableStore, set_default_store
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for kv in sys.argv[2:]:
    k, v = kv.split("="); _lib.set_knob(k, int(v))
E, P = 1024, 12288
chains = [[('d_fc1', 1024, True), ('d_fc2', 1024, True), ('d_out', P, False)],
          [('r_fc1', 512, True), ('r_fc2', 256, True), ('r_out', 3, False)],
          [('t_fc1', 512, True), ('t_fc2', 256, True), ('t_out', 3, False)]]
g = torch.Generator().manual_seed(3)
emb = torch.randn(B, E, generator=g)
ups = [torch.randn(B, c[-1][1], generator=g) for c in chains]
store = VariableStore(device="cuda", seed=1)
set_default_store(store)
xd = emb.cuda().requires_grad_(True)
outs = tf_util.fully_connected_chains(xd, chains, bn_decay=0.9, is_training=True)
sum((o * u.cuda()).sum() for o, u in zip(outs, ups)).backward()
torch.cuda.synchronize()
# float64 reference
params = {n: v.data.detach().double().cpu().requires_grad_(True) for n, v in store.vars.items() if v.trainable}
x = emb.double().requires_grad_(True)
want = []
for chain in chains:
    net = x
    for scope, n, bn in chain:
        net = net @ params[scope + "/weights"] + params[scope + "/biases"]
        if bn:
            mu, var = net.mean(0), net.var(0, unbiased=False)
            net = torch.relu((net - mu) / torch.sqrt(var + 1e-3) * params[scope + "/bn/gamma"] + params[scope + "/bn/beta"])
    want.append(net)
sum((o * u.double()).sum() for o, u in zip(want, ups)).backward()
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
print("outputs", [("%.1e" % rel(o, w.detach())) for o, w in zip(outs, want)], "dx %.1e" % rel(xd.grad, x.grad))
for n, p in params.items():
    e = rel(store.vars[n].data.grad, p.grad)
    if e > 1e-4 and p.grad.abs().max() > 1e-5:
        print("  %-22s err %.2e" % (n, e))
