"""dev: a few launches of cloudaae_nn_distance at the train shape (for tools/pmc_kernel.sh):  python run_chamfer.py [B] [N] [split 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cloudaae_amd import _lib
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
_lib.set_knob("CLOUDAAE_NN_SPLIT_SCORES", int(sys.argv[3]) if len(sys.argv) > 3 else 1)
g = torch.Generator(device="cuda").manual_seed(100)
a = torch.randn(B, N, 3, device="cuda", generator=g) * 0.05 + 1.0
c = torch.randn(B, N, 3, device="cuda", generator=g) * 0.05 + 1.0
d1, d2 = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
i1, i2 = torch.empty(B, N, dtype=torch.int32, device="cuda"), torch.empty(B, N, dtype=torch.int32, device="cuda")
for _ in range(6):
    assert L.cloudaae_nn_distance(B, N, a.data_ptr(), N, c.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), _lib.stream()) == 0
torch.cuda.synchronize()
