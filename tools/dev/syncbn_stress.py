"""dev: the SyncBN two-rank comparison of tests/test_09_sync_bn_gpu.py over many steps: python tools/dev/syncbn_stress.py STEPS REPLAY"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.multiprocessing as mp
from tests import test_09_sync_bn_gpu as t

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    replay = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(t._worker, args=(2, t._free_port(), out, 16, 256, replay, steps), nprocs=2, join=True)
    for rank, r in dict(out).items():
        bad = {k: v for k, v in r["res"].items() if v["e_loss"] > 1e-6 or v["e_state"] > 1e-5 or v["e_grad_l2"] > 1e-4}
        print("rank", rank, "steps", len(r["res"]), "off:", {k: (round(v["e_loss"], 8), round(v["e_state"], 6), round(v["e_grad_l2"], 5)) for k, v in bad.items()})
