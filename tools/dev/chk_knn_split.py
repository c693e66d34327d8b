"""dev: the bf16-pipe kNN without its repair launch: how many query groups were flagged (rows left at -1), and do the
rows it did write agree with the fp32-pipe kernel?   python tools/dev/chk_knn_split.py B N K"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib
L = _lib.lib()
b, n, k = (int(a) for a in sys.argv[1:4])
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.relu(torch.randn((b, n, 64), device="cuda", generator=g))


def run(split, fixup):
    _lib.set_knob("CLOUDAAE_KNN_SPLIT", split)
    _lib.set_knob("CLOUDAAE_KNN_SPLIT_FIXUP", fixup)
    out = torch.full((b, n, k), -1, dtype=torch.int32, device="cuda")
    _lib.check(L.cloudaae_knn(b, n, 64, 64, k, x.data_ptr(), out.data_ptr(), _lib.stream()), "knn")
    torch.cuda.synchronize()
    return out


ref = run(0, 1)
got = run(2, 0)
unwritten = (got[:, :, 0] < 0)
groups = unwritten.reshape(b, -1, 128).any(-1) if n % 128 == 0 else None
print("rows unwritten: %d of %d" % (int(unwritten.sum()), b * n), "" if groups is None else "groups flagged: %d of %d" % (int(groups.sum()), groups.numel()))
w = ~unwritten
bad = ((got != ref).any(-1) & w)
print("written rows that differ from the fp32-pipe kernel: %d" % int(bad.sum()))
if int(bad.sum()):
    i = bad.nonzero()[0]
    print(i.tolist(), got[i[0], i[1]].tolist(), ref[i[0], i[1]].tolist())
