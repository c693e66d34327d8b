"""dev: how compact are the 64-point groups of the hull kernel's sorted order?  (numpy replica of the 3-D Morton grid sort)"""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import train_cloudAAE_ycbv as T
from cloudaae_amd.utils import generate_occluder, hidden_point_removal as hpr
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
models = T.synthetic_object_models(T.NUM_CLASS, NP, device=dev)
el = T.synthetic_element(4, 1024, dev, seed=1)
x = {k: el[k] for k in ("translation", "axisangle", "class_id")}
x = T.get_object_model(x, models); x = T.get_rotation_matrix(x); x = T.transform_object_model(x)
x = generate_occluder.get_random_spherical_occluder(x, 'ycbv', seed=1)
x = hpr.sphericalFlip(x, None, 0.8 * math.pi)
P = x['flippedPoints'][0].cpu().numpy().astype(np.float64)
n1 = len(P)
V = P[-1]
print("n1", n1, "bbox extents", P.max(0) - P.min(0), "without the viewpoint", P[:-1].max(0) - P[:-1].min(0))


def spread(v):
    out = np.zeros_like(v)
    for b in range(10):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def report(order, tag):
    S = P[order]
    G = (n1 + 63) // 64
    rad, thick = [], []
    for g in range(G):
        q = S[g * 64:(g + 1) * 64]
        m = q.mean(0)
        n = (m - V) / np.linalg.norm(m - V)
        e = q - m
        h = e @ n
        t = np.sqrt(np.maximum((e * e).sum(1) - h * h, 0))
        rad.append(t.max()); thick.append(h.max() - h.min())
    rad, thick = np.array(rad), np.array(thick)
    # the patch: area from the lateral extent of the cloud (without the viewpoint)
    L = P[:-1]
    c = L.mean(0); nn = (c - V) / np.linalg.norm(c - V)
    e = L - c; lat = e - np.outer(e @ nn, nn)
    R = np.sqrt((lat * lat).sum(1)).max()
    ideal = R * math.sqrt(64.0 / n1)
    print("%-28s groups %d: radius median %.1f  mean %.1f  max %.1f | ideal disc %.1f | patch radius %.1f | thickness median %.2f max %.2f"
          % (tag, G, np.median(rad), rad.mean(), rad.max(), ideal, R, np.median(thick), thick.max()))


lo, hi = P.min(0), P.max(0)
for bits in (BITS, 6, 7):
    idx = np.minimum((1 << bits) - 1, ((P - lo) / np.maximum(hi - lo, 1e-30) * (1 << bits)).astype(np.int64))
    code = spread(idx[:, 0]) | (spread(idx[:, 1]) << 1) | (spread(idx[:, 2]) << 2)
    report(np.argsort(code, kind="stable"), "3-D Morton, %d bits" % bits)
# 2-D: angles seen from the viewpoint
D = P[:-1] - V
D /= np.linalg.norm(D, axis=1, keepdims=True)
c = D.mean(0); c /= np.linalg.norm(c)
a = np.cross(c, [1.0, 0, 0]); a /= np.linalg.norm(a); b2 = np.cross(c, a)
uv = np.stack([D @ a, D @ b2], 1)
for bits in (5, 6, 7):
    l2, h2 = uv.min(0), uv.max(0)
    idx = np.minimum((1 << bits) - 1, ((uv - l2) / (h2 - l2) * (1 << bits)).astype(np.int64))
    code = spread(idx[:, 0]) | (spread(idx[:, 1]) << 1)
    order = np.concatenate([np.argsort(code, kind="stable"), [n1 - 1]])
    report(order, "2-D Morton of directions, %d" % bits)
