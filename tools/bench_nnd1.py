"""Dev: one Chamfer forward shape, many launches (for rocprofv3 PMC passes): python tools/bench_nnd1.py B N M [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib
L = _lib.lib()
b, n, m = (int(a) for a in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
g = torch.Generator(device="cuda").manual_seed(1)
x1 = torch.randn((b, n, 3), device="cuda", generator=g) * 0.05
x2 = torch.randn((b, m, 3), device="cuda", generator=g) * 0.05
d1 = torch.empty((b, n), device="cuda"); i1 = torch.empty((b, n), dtype=torch.int32, device="cuda")
d2 = torch.empty((b, m), device="cuda"); i2 = torch.empty((b, m), dtype=torch.int32, device="cuda")
go = lambda: _lib.check(L.cloudaae_nn_distance(b, n, x1.data_ptr(), m, x2.data_ptr(), d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), _lib.stream()), "nnd")
for _ in range(3): go()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): go()
e1.record(); torch.cuda.synchronize()
print((b, n, m), "%.1f us" % (e0.elapsed_time(e1) * 1e3 / iters))
