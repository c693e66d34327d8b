import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for shape in [(32, 4096, 4096), (256, 4096, 4096), (32, 16384, 1024)]:
    print(shape, {k: v for k, v in bench.chamfer_kernel_rate(*shape).items() if k in ("us_per_launch", "Tpairs/s", "GB/s")})
