"""Generates tests/golden/*.npz -- TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference for the `ref_*` columns):
    python -m oracle.make_golden
Every fixture is data: seeded inputs and the expected outputs.
  chamfer_*  expected outputs come from the reference's own C++ lines
             (oracle/_ref, tf_nndistance.cpp:21-43,126-163) -- they pin the oracle.
  chamfer_seed0_1x5x6 additionally carries the reference's only seeded known-answer
             input (tf_nndistance_cpu.py:28-46) and its float64 brute-force matrix.
  fps_* / knn_*  expected outputs come from OUR restatement (the reference has no
             CPU kernel or vector for them: "parity unpinned"); they guard against
             drift and travel to the GPU box, where /root/reference does not exist.
"""
import os

import numpy as np

from . import native as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def chamfer_case(name, xyz1, xyz2, seed):
    assert O.have_ref(), "build oracle/_ref first (needs /root/reference)"
    rng = np.random.default_rng(seed)
    d1, i1, d2, i2 = O.ref_nn_distance(xyz1, xyz2)
    g1 = rng.standard_normal(d1.shape).astype(np.float32)
    g2 = rng.standard_normal(d2.shape).astype(np.float32)
    gx1, gx2 = O.ref_nn_distance_grad(xyz1, xyz2, g1, i1, g2, i2)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), xyz1=xyz1, xyz2=xyz2, dist1=d1, idx1=i1,
                        dist2=d2, idx2=i2, grad_dist1=g1, grad_dist2=g2, grad_xyz1=gx1,
                        grad_xyz2=gx2)


def prob_sample_case():
    """ProbSample (tf_sampling_g.cu:7-104): weights with zeros, a ragged quad tail, more than one
    8192-value chunk; own generator, so adding it does not move the other fixtures."""
    r = np.random.default_rng(7)
    p = r.random((2, 8192 + 35)).astype(np.float32)
    p[:, ::5] = 0.0
    draws = r.random((2, 400)).astype(np.float32)
    out, cum = O.prob_sample(p, draws, return_cumsum=True)
    np.savez_compressed(os.path.join(OUT, "probsample_2x8227_400.npz"), inp=p, inpr=draws, out=out, cumsum=cum)


def main():
    os.makedirs(OUT, exist_ok=True)
    # (1) the reference's seeded known-answer input, tf_nndistance_cpu.py:28-46
    np.random.seed(0)
    pc1 = np.random.random((1, 5, 3))
    pc2 = np.random.random((1, 6, 3))
    brute = np.zeros((5, 6))
    for i in range(5):
        for j in range(6):
            brute[i, j] = np.sum((pc1[0, i, :] - pc2[0, j, :]) ** 2)
    x1, x2 = pc1.astype(np.float32), pc2.astype(np.float32)
    d1, i1, d2, i2 = O.ref_nn_distance(x1, x2)
    g1 = np.ones_like(d1)
    g2 = np.ones_like(d2)
    gx1, gx2 = O.ref_nn_distance_grad(x1, x2, g1, i1, g2, i2)
    np.savez_compressed(os.path.join(OUT, "chamfer_seed0_1x5x6.npz"), pc1_f64=pc1, pc2_f64=pc2,
                        brute_f64=brute, xyz1=x1, xyz2=x2, dist1=d1, idx1=i1, dist2=d2, idx2=i2,
                        grad_dist1=g1, grad_dist2=g2, grad_xyz1=gx1, grad_xyz2=gx2)

    rng = np.random.default_rng(20200908)
    # (2) random clouds, object scale + translation (SURVEY 8d input law)
    a = (rng.standard_normal((2, 256, 3)) * 0.05 + np.array([0.1, -0.2, 0.9])).astype(np.float32)
    b = (rng.standard_normal((2, 256, 3)) * 0.05 + np.array([0.1, -0.2, 0.9])).astype(np.float32)
    chamfer_case("chamfer_rand_2x256x256", a, b, 1)
    # (3) duplicated points: ties must resolve to the first minimum
    a = rng.standard_normal((2, 64, 3)).astype(np.float32)
    b = rng.standard_normal((2, 64, 3)).astype(np.float32)
    b[:, 32:] = b[:, :32]          # every candidate exists twice
    a[:, 10] = b[:, 5]             # exact zero distance, two candidates
    a[:, 11] = a[:, 10]
    chamfer_case("chamfer_dup_ties_2x64x64", a, b, 2)
    # (4) ragged sizes, not multiples of any tile
    a = rng.standard_normal((3, 77, 3)).astype(np.float32)
    b = rng.standard_normal((3, 1031, 3)).astype(np.float32)
    chamfer_case("chamfer_ragged_3x77x1031", a, b, 3)

    # FPS: random cloud, cloud with duplicates (ties), n not a multiple of 512
    def fps_case(name, pts, m):
        np.savez_compressed(os.path.join(OUT, name + ".npz"), inp=pts, npoint=m,
                            out=O.farthest_point_sample(m, pts))

    p = rng.standard_normal((2, 1024, 3)).astype(np.float32)
    fps_case("fps_rand_2x1024_to_256", p, 256)
    p = rng.standard_normal((2, 700, 3)).astype(np.float32)
    p[:, 350:] = p[:, :350]        # every point twice -> max ties each round
    fps_case("fps_dup_2x700_to_128", p, 128)
    p = np.round(rng.standard_normal((1, 1500, 3)) * 2).astype(np.float32) / 2  # lattice: many equal distances
    fps_case("fps_lattice_1x1500_to_300", p, 300)

    # kNN: xyz slice of a [*,24] row; 64-channel features; ties from duplicates
    def knn_case(name, x, k, c):
        idx, dist = O.knn(x, k, channels=c, return_dist=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x, k=k, channels=c, nn_idx=idx,
                            nn_dist=dist)

    x = np.zeros((2, 300, 24), np.float32)
    x[:, :, :3] = rng.standard_normal((2, 300, 3)) * 0.05
    x[:, :, 3 + 7] = 1.0
    x[:, 150:, :3] = x[:, :150, :3]   # duplicates
    knn_case("knn_xyz_dup_2x300_k10", x, 10, 3)
    f = np.maximum(rng.standard_normal((2, 257, 64)), 0).astype(np.float32)  # post-ReLU-like
    f[:, 200] = f[:, 3]
    knn_case("knn_feat64_2x257_k10", f, 10, 64)
    knn_case("knn_feat64_2x257_k20", f, 20, 64)
    prob_sample_case()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
