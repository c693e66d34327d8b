"""Dev helper: CPU-oracle train-step rate vs torch thread count (run on the GPU box's host)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model_oracle as MO
V = MO.Vars(seed=0); opt = MO.AdamTF()
MO.train_step(MO.synthetic_batch(2, 1024, seed=1), V, opt, 0, 1024, 2)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    b = MO.synthetic_batch(8, 1024, seed=2)
    t0 = time.time(); MO.train_step(b, V, opt, 1, 1024, 8); dt = time.time() - t0
    print("threads", th, "clouds/s %.3f" % (8 / dt), flush=True)
