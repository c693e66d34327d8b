import sys, os
sys.path.insert(0, "/root/repo")
import torch
from cloudaae_amd import _lib
L = _lib.lib()
B, N, M = 4, 8192, 8192
g = torch.Generator(device="cuda").manual_seed(100)
a = torch.randn((B, N, 3), generator=g, device="cuda")
c = torch.randn((B, M, 3), generator=g, device="cuda")
def run(k):
    _lib.set_knob("CLOUDAAE_NN_SPLIT_SCORES", k)
    d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
    i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
    assert L.cloudaae_nn_distance(B, N, a.data_ptr(), M, c.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), _lib.stream()) == 0
    torch.cuda.synchronize()
    return i1, i2
ref = run(0)
for rep in range(6):
    o = run(1)
    print("rep", rep, "mismatches vs fp32 scores:", int((o[0] != ref[0]).sum()), int((o[1] != ref[1]).sum()),
          "first bad queries", (o[0] != ref[0]).nonzero()[:4].tolist())
_lib.set_knob("CLOUDAAE_NN_FILTER", 0)
gen1 = run(0)
_lib.set_knob("CLOUDAAE_NN_FILTER", 1)
for rep in range(3):
    o = run(0)
    print("fp32-score filter vs first-generation kernel:", int((o[0] != gen1[0]).sum()), int((o[1] != gen1[1]).sum()))
for rep in range(3):
    o = run(1)
    print("split-score filter vs first-generation kernel:", int((o[0] != gen1[0]).sum()), int((o[1] != gen1[1]).sum()))
