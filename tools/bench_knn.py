"""Dev: time the kNN kernels at the train-step shapes (CLOUDAAE_KNN_OLD=1 selects the previous C=64 kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib
L = _lib.lib()
def run(b, n, c, ld, k, iters=20):
    x = torch.randn((b, n, ld), device="cuda")
    out = torch.empty((b, n, k), dtype=torch.int32, device="cuda")
    go = lambda: _lib.check(L.cloudaae_knn(b, n, c, ld, k, x.data_ptr(), out.data_ptr(), _lib.stream()), "knn")
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, out
for shape in [(32, 1024, 64, 320, 10), (32, 1024, 3, 24, 10), (8, 1024, 64, 320, 10), (256, 1024, 64, 320, 10), (8, 4096, 64, 320, 20), (2, 333, 64, 64, 10)]:
    us, out = run(*shape)
    print(shape, "%.1f us" % us, int(out.long().sum()))
