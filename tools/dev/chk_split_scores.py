import sys, os
sys.path.insert(0, "/root/repo")
import torch
from cloudaae_amd import _lib
L = _lib.lib()
for (B, N, M, scale) in [(2, 4096, 4096, 1.0), (2, 4096, 4096, 0.05), (1, 2048, 2048, 1.0), (1, 1024, 3000, 1.0), (4, 8192, 8192, 1.0)]:
    g = torch.Generator(device="cuda").manual_seed(100)
    a = torch.randn((B, N, 3), generator=g, device="cuda") * scale
    c = torch.randn((B, M, 3), generator=g, device="cuda") * scale
    outs = {}
    for k in (0, 1):
        _lib.set_knob("CLOUDAAE_NN_SPLIT_SCORES", k)
        d1 = torch.empty(B, N, device="cuda"); d2 = torch.empty(B, M, device="cuda")
        i1 = torch.empty(B, N, dtype=torch.int32, device="cuda"); i2 = torch.empty(B, M, dtype=torch.int32, device="cuda")
        rc = L.cloudaae_nn_distance(B, N, a.data_ptr(), M, c.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), _lib.stream())
        assert rc == 0
        torch.cuda.synchronize()
        outs[k] = (d1, i1, d2, i2)
    bad1 = (outs[0][1] != outs[1][1]).nonzero()
    bad2 = (outs[0][3] != outs[1][3]).nonzero()
    print((B, N, M, scale), "dir1 mismatches", bad1.shape[0], "dir2", bad2.shape[0])
    if bad1.shape[0]:
        b, j = bad1[0].tolist()
        print("  first: cloud", b, "query", j, "fp32 idx", outs[0][1][b, j].item(), outs[0][0][b, j].item(), "split idx", outs[1][1][b, j].item(), outs[1][0][b, j].item())
        print("  queries affected (first cloud):", sorted(set((bad1[bad1[:,0]==b][:,1]).tolist()))[:20])
        print("  right answers' tiles:", sorted(set((outs[0][1][b][bad1[bad1[:,0]==b][:,1]] // 32).tolist()))[:20])

# ---- a failing query under the microscope: scores of its true nearest candidate's unit and of the chosen one
import ctypes, numpy as np
L._cdll.cloudaae_dev_nn_split_scores.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
B, N, M, scale = 4, 8192, 8192, 1.0
g = torch.Generator(device="cuda").manual_seed(100)
a = torch.randn((B, N, 3), generator=g, device="cuda") * scale
c = torch.randn((B, M, 3), generator=g, device="cuda") * scale
b, j = 0, 1417
q = a[b, j].cpu().numpy().astype(np.float64)
C = c[b].cpu().numpy()
c0 = C[0].astype(np.float32)
aa = (a[b, j].cpu().numpy() - c0).astype(np.float64)
bb = (C - c0).astype(np.float64)
score = (bb * bb).sum(1) - 2.0 * bb @ aa
d2 = ((C.astype(np.float64) - q) ** 2).sum(1)
order = np.argsort(score)
print("true nn", int(d2.argmin()), "d2", d2.min(), "best by exact score", order[:4], score[order[:4]])
R = (np.sqrt((aa * aa).sum()) + np.sqrt((bb * bb).sum(1)).max()) ** 2
print("R", R, "margin", 160 * 2.0 ** -24 * R)
units = np.minimum.reduceat(score.reshape(-1, 64)[:, [r + 4 * h + 32 * t for h in (0, 1) for t in (0, 1) for r in (0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19, 24, 25, 26, 27)]].reshape(-1, 2, 32).min(2).reshape(-1), np.arange(0, 2 * (M // 64), 1))
uo = np.argsort(units)
print("best units", uo[:4], units[uo[:4]], "unit of the true nn:", 2 * (int(d2.argmin()) // 64) + ((int(d2.argmin()) % 32) // 4) % 2)
