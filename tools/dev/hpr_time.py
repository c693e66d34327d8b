"""dev: time of the two hull launches of one config-5 batch (32 clouds, 8192-point models), events around convexHull x2"""
import os, sys, math, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import train_cloudAAE_ycbv as T
from cloudaae_amd.utils import generate_occluder, hidden_point_removal as hpr
for NP in (8192, 2048):
    B = 32
    dev = torch.device("cuda")
    models = T.synthetic_object_models(T.NUM_CLASS, NP, device=dev)
    el = T.synthetic_element(B, 1024, dev, seed=1)
    x = {k: el[k] for k in ("translation", "axisangle", "class_id")}
    x = T.get_object_model(x, models); x = T.get_rotation_matrix(x); x = T.transform_object_model(x)
    x = generate_occluder.get_random_spherical_occluder(x, 'ycbv', seed=1)
    x = hpr.sphericalFlip(x, None, 0.8 * math.pi)
    x = hpr.sphericalFlip_org(x, None, 0.8 * math.pi)
    def go():
        v, n = hpr.convexHull(x['flippedPoints'], x['orgPoints'], 1)
        v2, n2 = hpr.convexHull(x['flippedPoints_org'], x['orgPoints_org'], 2)
        return n, n2
    t_end = time.time() + 1.0
    while time.time() < t_end:
        go(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): n, n2 = go()
    b.record(); torch.cuda.synchronize()
    print("models of %d points: %.2f ms per batch of %d (two hulls each); visible %.1f / %.1f" % (NP, a.elapsed_time(b) / 5, B, float(n.float().mean()), float(n2.float().mean())))
