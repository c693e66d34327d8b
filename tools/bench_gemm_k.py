import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_gemm import run
for M, N in [(32768, 1024), (32768, 768), (24576, 1024), (16384, 1024)]:
    for K in (64, 320, 640, 1280, 2560):
        us, tf = run(0, 0, M, N, K, iters=10)
        print("M=%d N=%d K=%d: %8.1f us %6.1f TF  tiles=%d" % (M, N, K, us, tf, (M // 128) * (N // 128)))
