"""Dev: one kNN shape, many launches (for rocprofv3 PMC passes): python tools/bench_knn1.py B N C LD K [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib
L = _lib.lib()
b, n, c, ld, k = (int(a) for a in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((b, n, ld), device="cuda", generator=g)
if c != 3:
    x = torch.relu(x)                                                    # post-ReLU features, like net1..net3 (xyz: as drawn)
out = torch.empty((b, n, k), dtype=torch.int32, device="cuda")
go = lambda: _lib.check(L.cloudaae_knn(b, n, c, ld, k, x.data_ptr(), out.data_ptr(), _lib.stream()), "knn")
if os.environ.get("KNN_HINT"):          # "exact": the answer itself as the hint; "noisy": the lists of slightly different features
    go()
    hint = out.clone()
    if os.environ["KNN_HINT"] == "noisy":
        y = (x + 0.05 * torch.randn_like(x)).contiguous()
        _lib.check(L.cloudaae_knn(b, n, c, ld, k, y.data_ptr(), hint.data_ptr(), _lib.stream()), "knn")
    tau = torch.empty((b, n), device="cuda")
    want = out.clone()
    go = lambda: _lib.check(L.cloudaae_knn_hinted(b, n, c, ld, k, x.data_ptr(), hint.data_ptr(), tau.data_ptr(), out.data_ptr(),
                                                  _lib.stream()), "knn_hinted")
    go()
    assert torch.equal(out, want), "hinted result differs"
import time
t_end = time.time() + float(os.environ.get('WARM_S', '1.5'))      # clocks ramp from idle: warm up by time
while time.time() < t_end:
    for _ in range(50): go()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): go()
e1.record(); torch.cuda.synchronize()
print((b, n, c, ld, k), "mode", os.environ.get("CLOUDAAE_KNN_SCAN"), "hint", os.environ.get("KNN_HINT"), "%.1f us" % (e0.elapsed_time(e1) * 1e3 / iters))
