"""Dev: time given GEMM shapes: python tools/bench_gemm_shapes.py ta,tb,M,N,K [...]  (BF16=1 for the bf16 kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import run
for a in sys.argv[1:]:
    ta, tb, M, N, K = (int(x) for x in a.split(","))
    us, tf = run(ta, tb, M, N, K, iters=10)
    print("%-34s %9.1f us %7.1f TF" % (a, us, tf))
