"""Dev: the four edge-conv weight-gradient products (B=32, N=1024) one after the other vs concurrently on four
streams -- what a grouped launch could gain."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib
L = _lib.lib()
P = 32768
shapes = [(24, 128), (64, 128), (64, 128), (64, 256)]          # (cin, 2*cout) of dgcnn1..4
X = [torch.randn(P, 320, device="cuda") for _ in shapes]
D = [torch.randn(P, n, device="cuda") for _, n in shapes]
W = [torch.zeros(m, n, device="cuda") for m, n in shapes]
streams = [torch.cuda.Stream() for _ in shapes]
def go(i, s):
    m, n = shapes[i]
    _lib.check(L.cloudaae_gemm_f32(1, 0, m, n, P, X[i].data_ptr(), 320, D[i].data_ptr(), n, W[i].data_ptr(), n, None, 2, s), "g")
def seq():
    s = _lib.stream()
    for i in range(4): go(i, s)
def par():
    cur = torch.cuda.current_stream()
    for i, st in enumerate(streams):
        st.wait_stream(cur)
        go(i, st.cuda_stream)
    for st in streams: cur.wait_stream(st)
for name, fn in (("sequential", seq), ("four streams", par), ("sequential", seq), ("four streams", par)):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us per set of four" % (e0.elapsed_time(e1) * 1e3 / 50))
