"""GPU: tf_ops custom operators and kNN grouping through the C-ABI vs the oracle."""
import os

import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


CHAMFER = ["chamfer_seed0_1x5x6", "chamfer_rand_2x256x256", "chamfer_dup_ties_2x64x64",
           "chamfer_ragged_3x77x1031"]


@pytest.mark.parametrize("name", CHAMFER)
def test_nn_distance_golden_bit_exact(hip, golden_dir, name):
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    g = _load(golden_dir, name)
    d1, i1, d2, i2 = tf_nndistance.nn_distance(_dev(g["xyz1"]), _dev(g["xyz2"]))
    assert i1.dtype == torch.int32 and i2.dtype == torch.int32
    assert np.array_equal(i1.cpu().numpy(), g["idx1"])
    assert np.array_equal(i2.cpu().numpy(), g["idx2"])
    assert np.array_equal(d1.cpu().numpy(), g["dist1"])  # bit-exact fp32
    assert np.array_equal(d2.cpu().numpy(), g["dist2"])
    gx1, gx2 = tf_nndistance.nn_distance_grad(_dev(g["xyz1"]), _dev(g["xyz2"]), _dev(g["grad_dist1"]),
                                              i1, _dev(g["grad_dist2"]), i2)
    # atomics: summation order differs from the sequential CPU sweep -> fp32 tolerance
    np.testing.assert_allclose(gx1.cpu().numpy(), g["grad_xyz1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gx2.cpu().numpy(), g["grad_xyz2"], rtol=1e-5, atol=1e-6)
    # the ordered kernel follows the sweep of tf_nndistance.cpp:126-163 term by term: the reference lines' own
    # gradients (the fixtures were written by oracle/_ref), bit for bit
    ox1, ox2 = tf_nndistance.nn_distance_grad(_dev(g["xyz1"]), _dev(g["xyz2"]), _dev(g["grad_dist1"]),
                                              i1, _dev(g["grad_dist2"]), i2, ordered=True)
    assert np.array_equal(ox1.cpu().numpy(), g["grad_xyz1"])
    assert np.array_equal(ox2.cpu().numpy(), g["grad_xyz2"])


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 3, 1000), (3, 257, 255), (2, 4096, 4096), (2, 5000, 17), (5, 2100, 2049)])
def test_nn_distance_grad_ordered_bit_exact(hip, oracle, b, n, m):
    """cloudaae_nn_distance_grad_ordered = the reference's sequential gradient loops (tf_nndistance.cpp:126-163), bit for
    bit: many queries share a nearest neighbour (duplicated targets, few candidates), so the ORDER of the additions
    matters; per-point and uniform upstream gradients; a NULL output is skipped."""
    from cloudaae_amd import _lib
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    rng = np.random.default_rng(b * 1000 + n + m)
    a = (rng.standard_normal((b, n, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    c = (rng.standard_normal((b, m, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    if m > 40:
        c[:, m // 2:] = c[:, :m - m // 2]      # every target point twice (the reference's padded targets)
    _, i1, _, i2 = oracle.nn_distance(a, c, threads=8)
    w1, w2 = rng.standard_normal((b, n)).astype(np.float32), rng.standard_normal((b, m)).astype(np.float32)
    want1, want2 = oracle.nn_distance_grad(a, c, w1, i1, w2, i2)
    got1, got2 = tf_nndistance.nn_distance_grad(_dev(a), _dev(c), _dev(w1), _dev(i1), _dev(w2), _dev(i2), ordered=True)
    assert np.array_equal(got1.cpu().numpy(), want1) and np.array_equal(got2.cpu().numpy(), want2)
    # uniform upstream gradient u * scale for every distance, second output not wanted
    u, scale = torch.full((), 0.37, device="cuda"), 1.0 / (b * n)
    ones1 = np.full((b, n), np.float32(np.float32(0.37) * np.float32(scale)), np.float32)
    ones2 = np.full((b, m), np.float32(np.float32(0.37) * np.float32(scale)), np.float32)
    want1, _ = oracle.nn_distance_grad(a, c, ones1, i1, ones2, i2)
    ad, cd, i1d, i2d = _dev(a), _dev(c), _dev(i1), _dev(i2)
    out1 = torch.full((b, n, 3), float("nan"), device="cuda")
    _lib.check(_lib.lib().cloudaae_nn_distance_grad_ordered(b, n, _lib.ptr(ad), m, _lib.ptr(cd), None, _lib.ptr(i1d), None,
                                                            _lib.ptr(i2d), _lib.ptr(u), scale, _lib.ptr(out1), None,
                                                            _lib.stream()), "grad_ordered")
    assert np.array_equal(out1.cpu().numpy(), want1)


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 3, 1000), (3, 257, 255), (4, 1024, 4096),
                                   (2, 4096, 4096), (2, 5000, 17), (40, 512, 300)])
@pytest.mark.parametrize("kernel", ["0", "1"])
def test_nn_distance_vs_oracle(hip, oracle, b, n, m, kernel, knobs):
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    knobs("CLOUDAAE_NN_FILTER", int(kernel))       # both kernels on every shape
    rng = np.random.default_rng(b * 1000 + n + m)
    a = (rng.standard_normal((b, n, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    c = (rng.standard_normal((b, m, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    if m > 40:
        c[:, m // 2:m // 2 + 20] = c[:, :20]   # duplicated candidates: first must win
    want = oracle.nn_distance(a, c, threads=8)
    got = tf_nndistance.nn_distance(_dev(a), _dev(c))
    for w, g_ in zip(want, got):
        assert np.array_equal(w, g_.cpu().numpy())


@pytest.mark.parametrize("lds", [1, 0])
@pytest.mark.parametrize("case", ["random-2x2048x2048", "collapsed-2x4096x4096", "ragged-3x700x5461", "long-2x6000x100", "one-1x1x1", "config5-2x4096x16384"])
def test_nn_distance_grad_forms_vs_oracle(hip, oracle, knobs, case, lds):
    """The Chamfer gradient with its additions in LDS (one workgroup per cloud and 2048 points of an output array, the
    default) and with global atomics (CLOUDAAE_NND_GRAD_LDS=0) against the oracle's sequential sweep
    (tf_nndistance.cpp:126-163): unordered fp32 additions either way -> fp32 tolerance.  'collapsed': nearly every target
    point names one of a few predicted points (a decoder at initialisation) -- thousands of additions per address."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    knobs("CLOUDAAE_NND_GRAD_LDS", lds)
    b, n, m = (int(x) for x in case.split("-")[-1].split("x"))
    rng = np.random.default_rng(n + m)
    a = rng.standard_normal((b, n, 3)).astype(np.float32)
    c = rng.standard_normal((b, m, 3)).astype(np.float32)
    if case.startswith("collapsed"):
        a *= np.float32(1e-3)
        a[:, :8] *= np.float32(1e3)
    d1, i1, d2, i2 = tf_nndistance.nn_distance(_dev(a), _dev(c))
    w1 = rng.standard_normal((b, n)).astype(np.float32)
    w2 = rng.standard_normal((b, m)).astype(np.float32)
    g1, g2 = tf_nndistance.nn_distance_grad(_dev(a), _dev(c), _dev(w1), i1, _dev(w2), i2)
    want1, want2 = oracle.nn_distance_grad(a, c, w1, i1.cpu().numpy(), w2, i2.cpu().numpy())
    # (sums of up to thousands of terms in another order: the tolerance scales with the largest sum)
    tol = 5e-5 * max(1.0, float(np.abs(want1).max()), float(np.abs(want2).max()))
    np.testing.assert_allclose(g1.cpu().numpy(), want1, rtol=1e-4, atol=tol)
    np.testing.assert_allclose(g2.cpu().numpy(), want2, rtol=1e-4, atol=tol)


def test_nn_distance_autograd_and_empty(hip, oracle):
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    rng = np.random.default_rng(11)
    a = rng.standard_normal((2, 100, 3)).astype(np.float32)
    c = rng.standard_normal((2, 90, 3)).astype(np.float32)
    ta = _dev(a).requires_grad_(True)
    tc = _dev(c).requires_grad_(True)
    d1, i1, d2, i2 = tf_nndistance.nn_distance(ta, tc)
    w1 = _dev(rng.standard_normal((2, 100)).astype(np.float32))
    w2 = _dev(rng.standard_normal((2, 90)).astype(np.float32))
    ((d1 * w1).sum() + (d2 * w2).sum()).backward()
    o = oracle.nn_distance(a, c)
    g1, g2 = oracle.nn_distance_grad(a, c, w1.cpu().numpy(), o[1], w2.cpu().numpy(), o[3])
    np.testing.assert_allclose(ta.grad.cpu().numpy(), g1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), g2, rtol=1e-5, atol=1e-6)
    # m == 0: dist 0 / idx 0 (tf_nndistance.cpp:28-29 survive an empty loop)
    e = torch.zeros((2, 0, 3), device="cuda")
    d1, i1, d2, i2 = tf_nndistance.nn_distance(_dev(a), e)
    assert (d1 == 0).all() and (i1 == 0).all() and d2.shape == (2, 0)
    # shape errors mirror the OP_REQUIREs
    with pytest.raises(ValueError):
        tf_nndistance.nn_distance(_dev(a)[:, :, :2], _dev(c))
    with pytest.raises(ValueError):
        tf_nndistance.nn_distance(_dev(a), _dev(c)[:1])


def test_nn_distance_full_size_properties(hip):
    """BASELINE config sizes (B=32, n=m=4096): size-independent properties."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    gen = torch.Generator(device="cuda").manual_seed(100)
    a = torch.randn((32, 4096, 3), generator=gen, device="cuda")
    c = torch.randn((32, 4096, 3), generator=gen, device="cuda")
    d1, i1, d2, i2 = tf_nndistance.nn_distance(a, c)
    # (1) the reported distance is the distance to the reported index (un-fused formula)
    nb = torch.gather(c, 1, i1.long()[:, :, None].expand(-1, -1, 3))
    diff = nb - a
    dd = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert torch.equal(dd, d1)
    # (2) swapping the clouds swaps the outputs
    e1, j1, e2, j2 = tf_nndistance.nn_distance(c, a)
    assert torch.equal(e1, d2) and torch.equal(j1, i2) and torch.equal(e2, d1) and torch.equal(j2, i1)
    # (3) a cloud against itself: zero distance; index = first duplicate = itself for distinct points
    z1, k1, _, _ = tf_nndistance.nn_distance(a, a)
    assert (z1 == 0).all() and torch.equal(k1, torch.arange(4096, device="cuda", dtype=torch.int32).expand(32, -1))
    # (4) brute force on a sample of rows (tf_nndistance.py:77-85), in the reference's arithmetic -- the un-fused expression of
    # tf_nndistance.cpp:30-33, one rounding per operation (elementwise torch kernels do not contract) -- and its tie rule
    # (strict <: the FIRST minimum): distance and index must be EQUAL, no tolerance
    rows = torch.randint(0, 4096, (64,), device="cuda")
    df = a[:, rows, None, :] - c[:, None, :, :]
    D = (df[..., 0] * df[..., 0] + df[..., 1] * df[..., 1]) + df[..., 2] * df[..., 2]
    dmin = D.min(-1, keepdim=True).values
    first = torch.where(D == dmin, torch.arange(4096, device="cuda").expand_as(D), torch.full_like(D, 4096, dtype=torch.int64)).min(-1).values
    assert torch.equal(dmin[..., 0], d1[:, rows])
    assert torch.equal(first.int(), i1[:, rows])


@pytest.mark.parametrize("name", ["fps_rand_2x1024_to_256", "fps_dup_2x700_to_128",
                                  "fps_lattice_1x1500_to_300"])
def test_fps_golden_bit_exact(hip, golden_dir, name):
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    g = _load(golden_dir, name)
    out = tf_sampling.farthest_point_sample(int(g["npoint"]), _dev(g["inp"]))
    assert out.dtype == torch.int32
    assert np.array_equal(out.cpu().numpy(), g["out"])


@pytest.mark.parametrize("b,n,m", [(1, 1, 4), (3, 100, 100), (2, 513, 64), (4, 4096, 1024),
                                   (1, 9000, 50), (33, 2048, 128), (1, 20000, 40)])
def test_fps_vs_oracle(hip, oracle, b, n, m):
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    rng = np.random.default_rng(n + m)
    p = rng.standard_normal((b, n, 3)).astype(np.float32)
    if n > 200:
        p[:, n // 2:n // 2 + 100] = p[:, :100]
    want = oracle.farthest_point_sample(m, p, threads=8)
    got = tf_sampling.farthest_point_sample(m, _dev(p)).cpu().numpy()
    assert np.array_equal(want, got)


def test_gather_point_and_grad(hip, oracle):
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    rng = np.random.default_rng(4)
    inp = rng.standard_normal((3, 300, 3)).astype(np.float32)
    idx = rng.integers(0, 300, (3, 70)).astype(np.int32)
    t = _dev(inp).requires_grad_(True)
    out = tf_sampling.gather_point(t, _dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.gather_point(inp, idx))
    og = rng.standard_normal((3, 70, 3)).astype(np.float32)
    out.backward(_dev(og))
    np.testing.assert_allclose(t.grad.cpu().numpy(), oracle.gather_point_grad(inp.shape, idx, og),
                               rtol=1e-5, atol=1e-6)
    # unique indices (what FPS produces): exact
    perm = np.stack([rng.permutation(300)[:70] for _ in range(3)]).astype(np.int32)
    g = tf_sampling.gather_point_grad(_dev(inp), _dev(perm), _dev(og))
    assert np.array_equal(g.cpu().numpy(), oracle.gather_point_grad(inp.shape, perm, og))
    # eval-path composition (evaluate_cloudAAE_ycbv.py:450)
    sub = tf_sampling.gather_point(_dev(inp), tf_sampling.farthest_point_sample(32, _dev(inp)))
    assert sub.shape == (3, 32, 3)


@pytest.mark.parametrize("name", ["knn_xyz_dup_2x300_k10", "knn_feat64_2x257_k10",
                                  "knn_feat64_2x257_k20"])
def test_knn_golden_bit_exact(hip, golden_dir, name):
    from cloudaae_amd.utils import tf_util
    g = _load(golden_dir, name)
    x = _dev(g["x"])
    c = int(g["channels"])
    adj = tf_util.pairwise_xyz_distance(x if c == 3 else x[:, :, None, :])
    nn_idx = tf_util.knn(adj, k=int(g["k"]))
    assert nn_idx.dtype == torch.int32
    assert np.array_equal(nn_idx.cpu().numpy(), g["nn_idx"])


@pytest.mark.parametrize("b,n,c,ld,k", [(2, 256, 3, 24, 10), (3, 1024, 3, 24, 10), (2, 1000, 3, 3, 20),
                                        (2, 256, 64, 64, 10), (4, 1024, 64, 64, 10), (2, 777, 64, 64, 20),
                                        (1, 4096, 64, 64, 20), (2, 130, 64, 64, 5), (2, 100, 16, 16, 7),
                                        (1, 64, 3, 24, 32), (70, 1000, 3, 24, 10), (130, 1024, 3, 3, 10), (40, 4096, 3, 24, 20)])
def test_knn_vs_oracle(hip, oracle, b, n, c, ld, k):
    from cloudaae_amd import _lib
    rng = np.random.default_rng(n * 7 + c + k)
    x = np.maximum(rng.standard_normal((b, n, ld)), -0.5).astype(np.float32) * 0.1
    x[:, n // 2:n // 2 + 30] = x[:, :30]      # duplicates -> exact ties
    want = oracle.knn(x, k, channels=c, threads=8)
    xd = _dev(x)
    got = torch.empty((b, n, k), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().cloudaae_knn(b, n, c, ld, k, _lib.ptr(xd), _lib.ptr(got), _lib.stream()), "knn")
    assert np.array_equal(want, got.cpu().numpy())


@pytest.mark.parametrize("b,n,k,mode", [(2, 333, 10, 1), (40, 1024, 10, 1), (3, 1500, 20, 2), (33, 1024, 10, 2),
                                          (2, 700, 20, None), (140, 1024, 10, None), (2, 257, 20, 1), (5, 1000, 5, 2),
                                          (32, 1024, 10, None), (2, 4096, 20, 2), (2, 4096, 20, None), (2, 260, 10, 2),
                                          (40, 1024, 10, 5), (3, 1500, 10, 5), (2, 260, 5, 5), (2, 2048, 10, 5), (9, 3000, 10, None),
                                          (1, 3300, 7, 5),
                                          # k = 20 and clouds of 4096+ points in the 16-wave kernel (BASELINE configs[4])
                                          (2, 4096, 20, 5), (3, 1500, 20, 5), (2, 300, 15, 5), (1, 4500, 20, 5), (1, 6000, 10, 5),
                                          (5, 4096, 20, None), (33, 1024, 20, None), (1, 3400, 10, 5),
                                          # sampled tiles reused, pass B over the rest: tile counts 17 / 25 / 29 / 9 / 32
                                          (3, 530, 10, 5), (2, 800, 10, 5), (2, 900, 7, 5), (3, 270, 10, 5), (2, 1024, 1, 5),
                                          # clouds below 256 points (the scan kernel since round 6 retired knn64_mfma_kernel),
                                          # k up to the cloud's size, and k above 20 (the generic kernel)
                                          (3, 5, 5, None), (2, 31, 10, None), (4, 64, 20, None), (2, 130, 10, 1), (2, 255, 20, None),
                                          (2, 40, 32, None), (1, 1030, 25, None)])
def test_knn_c64_kernel_choices_vs_oracle(hip, oracle, knobs, b, n, k, mode):
    """The C = 64 kernels (knob CLOUDAAE_KNN_SCAN = 1 / 2: whole-cloud scan with one / two waves per query tile, 5: bound
    pass + filtered scan in 16-wave workgroups; None: the launcher's own choice) keep the same bit-exact contract."""
    from cloudaae_amd import _lib
    knobs("CLOUDAAE_KNN_SCAN", mode)
    rng = np.random.default_rng(n + k)
    x = np.maximum(rng.standard_normal((b, n, 64)), -0.5).astype(np.float32) * 0.1
    dup = min(30, n - n // 2)
    x[:, n // 2:n // 2 + dup] = x[:, :dup]
    want = oracle.knn(x, k, channels=64, threads=8)
    xd = _dev(x)
    got = torch.empty((b, n, k), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().cloudaae_knn(b, n, 64, 64, k, _lib.ptr(xd), _lib.ptr(got), _lib.stream()), "knn")
    assert np.array_equal(want, got.cpu().numpy())


@pytest.mark.parametrize("mode,k", [(1, 10), (2, 20), (5, 10), (5, 20)])
@pytest.mark.parametrize("case", ["all_equal", "few_distinct", "lattice", "large_finite", "far_cluster", "near_ties"])
def test_knn_c64_bound_kernel_adversarial(hip, oracle, knobs, mode, k, case):
    """The bound kernel's correctness must not depend on its bound being tight: clouds where (nearly) every
    candidate ties with the k-th distance (the queue overflows and is drained over and over), where the sampled
    tiles are unrepresentative, and where distances are huge."""
    from cloudaae_amd import _lib
    knobs("CLOUDAAE_KNN_SCAN", mode)
    rng = np.random.default_rng(7)
    b, n = 3, 1024
    if case == "near_ties":
        # every point with 7 copies that differ in the last bits of a few channels: distances equal to ~1e-7 relative, the
        # order decided by the oracle's rounding alone
        base = rng.standard_normal((b, n // 8, 64)).astype(np.float32)
        x = np.repeat(base, 8, axis=1)
        jig = rng.integers(-2, 3, x.shape).astype(np.int32) * (rng.random(x.shape) < 0.1)
        x = (x.view(np.int32) + jig).view(np.float32).astype(np.float64)
    elif case == "all_equal":
        x = np.tile(rng.standard_normal((b, 1, 64)), (1, n, 1))
    elif case == "few_distinct":
        x = rng.standard_normal((b, 5, 64))[:, rng.integers(0, 5, n)]
    elif case == "lattice":
        x = rng.integers(0, 2, (b, n, 64)).astype(np.float64)        # exact small-integer distances: ties everywhere
    elif case == "large_finite":
        x = rng.standard_normal((b, n, 64)) * 1e15                   # |x|^2 ~ 6e31: finite, heavy cancellation
        # (non-finite distances are outside the contract: tf.nn.top_k over NaN is unspecified)
    else:
        x = rng.standard_normal((b, n, 64)) * 0.01
        x[:, ::128] += 50.0                                          # every sampled tile starts with an outlier
        x[:, 32:64] += 100.0                                         # and a whole unsampled tile sits far away
    x = x.astype(np.float32)
    want = oracle.knn(x, k, channels=64, threads=8)
    xd = _dev(x)
    got = torch.full((b, n, k), -1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().cloudaae_knn(b, n, 64, 64, k, _lib.ptr(xd), _lib.ptr(got), _lib.stream()), "knn")
    assert np.array_equal(want, got.cpu().numpy())


@pytest.mark.parametrize("b,n,k", [(3, 256, 10), (2, 1024, 10), (2, 1500, 20), (2, 4096, 20), (1, 4500, 20), (5, 700, 7), (2, 3400, 10)])
@pytest.mark.parametrize("hint_kind", ["previous_layer", "exact", "random", "one_point", "far"])
def test_knn_hinted_vs_oracle(hip, oracle, b, n, k, hint_kind):
    """cloudaae_knn_hinted (round 6): the largest distance to k hinted points bounds the k-th distance, the filtered scan takes
    it instead of a bound pass of its own -- and returns cloudaae_knn's answer WHATEVER the hint holds: the lists of slightly
    different features (what the encoder passes: the layer before), the exact answer (the bound IS the k-th distance: the
    filter's `<=`), random points (a loose bound: the queues overflow, the flagged rescan answers), k copies of one index
    (not distinct: the bound may be too tight -- the caller's contract is k DISTINCT indices, so this case only has to
    stay in bounds and return k valid indices) and the k farthest points."""
    from cloudaae_amd import _lib
    rng = np.random.default_rng(n + k)
    x = np.maximum(rng.standard_normal((b, n, 64)), -0.5).astype(np.float32) * 0.1
    x[:, n // 2:n // 2 + 30] = x[:, :30]
    want = oracle.knn(x, k, channels=64, threads=8)
    if hint_kind == "previous_layer":
        hint = oracle.knn((x + rng.standard_normal(x.shape).astype(np.float32) * 0.01).astype(np.float32), k, channels=64, threads=8)
    elif hint_kind == "exact":
        hint = want.copy()
    elif hint_kind == "random":
        hint = np.stack([np.stack([rng.permutation(n)[:k] for _ in range(n)]) for _ in range(b)])
    elif hint_kind == "one_point":
        hint = np.repeat(rng.integers(0, n, (b, n, 1)), k, axis=2)
    else:
        far = oracle.knn(-x, k, channels=64, threads=8)       # (some other k distinct points)
        hint = far
    hint = np.ascontiguousarray(hint.astype(np.int32))
    xd, hd = _dev(x), _dev(hint)
    tau = torch.empty((b, n), device="cuda")
    got = torch.full((b, n, k), -1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().cloudaae_knn_hinted(b, n, 64, 64, k, _lib.ptr(xd), _lib.ptr(hd), _lib.ptr(tau), _lib.ptr(got),
                                              _lib.stream()), "knn_hinted")
    g = got.cpu().numpy()
    if hint_kind == "one_point":
        assert g.min() >= 0 and g.max() < n
    else:
        assert np.array_equal(want, g)
        # the bound really is one: at least k candidates at or below it
        d = oracle.knn(x, k, channels=64, threads=8, return_dist=True)[1]
        t = tau.cpu().numpy()
        assert (t >= d[:, :, k - 1]).all()
        if hint_kind == "exact":
            # the hinted points ARE the neighbours: the bound is their largest distance plus the rounding allowance of
            # knn64_hint_bound_kernel (1.25 x 2^-17 of the two squared norms), no looser
            norms = (x.astype(np.float64) ** 2).sum(-1)
            slack = 2.0 ** -16 * (norms + norms.max(axis=1, keepdims=True)) + 1e-30
            assert (t <= d[:, :, k - 1] * (1 + 1e-5) + slack).all()


@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("b,n,ld,k,case", [(2, 300, 24, 10, "dup"), (3, 1024, 24, 10, "dup"), (2, 1000, 3, 20, "dup"),
                                            (2, 4096, 24, 20, "dup"), (1, 2500, 3, 7, "dup"), (5, 257, 24, 1, "dup"),
                                            (2, 1024, 24, 10, "all_equal"), (2, 1024, 24, 20, "lattice"),
                                            (2, 1024, 24, 10, "far_cluster"), (2, 1024, 3, 10, "large_finite"),
                                            # clouds below 256 points (always the scan kernel), k above 20 and clouds beyond
                                            # the LDS-resident forms (the generic kernel since round 6 retired knn3_kernel)
                                            (3, 5, 24, 5, "dup"), (2, 64, 3, 20, "lattice"), (4, 200, 24, 10, "dup"),
                                            (2, 300, 24, 27, "dup"), (1, 6500, 3, 10, "dup")])
def test_knn_c3_kernel_choices_vs_oracle(hip, oracle, knobs, wide, b, n, ld, k, case):
    """C = 3 (layer 1): knn3_wide_kernel (three MFMAs per 32 x 32 tile, bound pass + queues; CLOUDAAE_KNN3_WIDE=1) and
    knn3_scan_kernel (=0) keep the same bit-exact contract, also where (nearly) every candidate ties with the k-th
    distance (the queues overflow: flagged rescan), where the sampled tiles are unrepresentative and where distances are huge."""
    from cloudaae_amd import _lib
    knobs("CLOUDAAE_KNN3_WIDE", wide)
    rng = np.random.default_rng(n + k + ld)
    if case == "dup":
        x = np.maximum(rng.standard_normal((b, n, ld)), -0.5) * 0.1
        dup = min(30, n - n // 2)
        x[:, n // 2:n // 2 + dup] = x[:, :dup]    # duplicates -> exact ties
    elif case == "all_equal":
        x = np.tile(rng.standard_normal((b, 1, ld)), (1, n, 1))
    elif case == "lattice":
        x = rng.integers(0, 3, (b, n, ld)).astype(np.float64)         # exact small-integer distances: ties everywhere
    elif case == "large_finite":
        x = rng.standard_normal((b, n, ld)) * 1e15
    else:
        x = rng.standard_normal((b, n, ld)) * 0.01
        x[:, ::128] += 50.0                                          # every sampled tile starts with an outlier
        x[:, 32:64] += 100.0                                         # and a whole unsampled tile sits far away
    x = x.astype(np.float32)
    want = oracle.knn(x, k, channels=3, threads=8)
    xd = _dev(x)
    got = torch.full((b, n, k), -1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().cloudaae_knn(b, n, 3, ld, k, _lib.ptr(xd), _lib.ptr(got), _lib.stream()), "knn")
    assert np.array_equal(want, got.cpu().numpy())


# ---- ProbSample (tf_sampling_g.cu:7-104) -----------------------------------------------------
@pytest.mark.parametrize("b,n,m", [(1, 1, 5), (2, 7, 50), (3, 21, 1000), (2, 4096, 300), (1, 8192 + 37, 500),
                                   (2, 20000, 700)])
def test_prob_sample_vs_oracle(hip, oracle, b, n, m):
    """Indices AND the prefix sums (the reference's association order) bit-exact, with zero-weight
    categories (ties in the prefix sums), ragged quads and rows longer than one 8192-value chunk."""
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    from cloudaae_amd import _lib
    rng = np.random.default_rng(n + m)
    p = rng.random((b, n)).astype(np.float32)
    p[:, ::5] = 0.0
    r = rng.random((b, m)).astype(np.float32)
    want, cum = oracle.prob_sample(p, r, return_cumsum=True)
    got = tf_sampling.prob_sample(_dev(p), _dev(r))
    assert got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), want)
    temp = torch.empty((b, n), device="cuda")
    out = torch.empty((b, m), dtype=torch.int32, device="cuda")
    pd, rd = _dev(p), _dev(r)
    _lib.check(_lib.lib().cloudaae_prob_sample(b, n, m, _lib.ptr(pd), _lib.ptr(rd), _lib.ptr(temp), _lib.ptr(out),
                                               _lib.stream()), "prob_sample")
    assert np.array_equal(temp.cpu().numpy(), cum)
    # (a zero-weight category CAN be drawn at a quad boundary: the tree-scanned total and the running
    # prefix differ by an ulp there -- a property of the reference's summation order, kept as is)
    # the draws follow the weights
    if n == 21:
        freq = np.bincount(want.reshape(-1), minlength=n) / want.size
        assert np.abs(freq - p.sum(0) / p.sum()).max() < 0.05


@pytest.mark.parametrize("b,n,m,split", [(2, 16384, 1024, None), (3, 700, 9000, None), (2, 5000, 4100, "3"),
                                         (1, 4096, 4096, "2"), (2, 1000, 20000, "7"), (33, 600, 4096, None)])
def test_nn_distance_split_candidates_vs_oracle(hip, oracle, b, n, m, split, knobs):
    """Clouds of unequal size (the reference's own benchmark is 16384 x 1024 points, tf_nndistance.py:48-49): the
    direction with few queries and many candidates is cut over its candidates, the ranges meet in a 64-bit
    (distance, index) minimum.  Bit-exact incl. the first-index rule ACROSS ranges (duplicated candidates sit in
    different ranges).  split: CLOUDAAE_NN_SPLIT forces a range count on both directions."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    knobs("CLOUDAAE_NN_FILTER", 1)
    if split is not None:
        knobs("CLOUDAAE_NN_SPLIT", int(split))
    rng = np.random.default_rng(n + m)
    a = (rng.standard_normal((b, n, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    c = (rng.standard_normal((b, m, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    c[:, m - 50:m - 10] = c[:, :40]            # exact duplicates far apart: the lower index must win
    a[:, n - 50:n - 10] = a[:, 5:45]
    a[:, :8] = c[:, 100:108]                   # zero distances
    want = oracle.nn_distance(a, c, threads=8)
    got = tf_nndistance.nn_distance(_dev(a), _dev(c))
    for w, g_ in zip(want, got):
        assert np.array_equal(w, g_.cpu().numpy())


@pytest.mark.parametrize("b,n,distinct,split", [(2, 4096, 1024, None), (3, 2049, 1000, None), (2, 16384, 4000, None),
                                                (2, 6000, 300, 2)])
def test_nn_distance_padded_targets_vs_oracle(hip, oracle, knobs, b, n, distinct, split):
    """The reference's training targets: the visible points followed by random RE-DRAWS of visible points
    (utils/hidden_point_removal.py:38-40), i.e. every target point exists two to four times.  For the matrix-core
    search every such query has several units with bitwise equal best scores: settled by the second pass over the
    scores (candidates at or below best + margin evaluated exactly), bit-exact incl. the first-index rule -- the
    index returned for a duplicated point is always its FIRST copy."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    knobs("CLOUDAAE_NN_FILTER", 1)
    if split is not None:
        knobs("CLOUDAAE_NN_SPLIT", split)
    rng = np.random.default_rng(n + distinct)
    base = (rng.standard_normal((b, distinct, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    pick = rng.integers(0, distinct, (b, n - distinct))
    target = np.concatenate([base, np.take_along_axis(base, pick[:, :, None].repeat(3, 2), 1)], 1)
    pred = (rng.standard_normal((b, n, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    pred[:, :50] = target[:, 100:150]                 # exact hits
    want = oracle.nn_distance(pred, target, threads=8)
    got = tf_nndistance.nn_distance(_dev(pred), _dev(target))
    for w, g_ in zip(want, got):
        assert np.array_equal(w, g_.cpu().numpy())
    assert int(got[1].max()) < distinct                # nearest targets: always the first copy


@pytest.mark.parametrize("b,n,m,filt", [(4, 4096, 4096, 1), (3, 2049, 2049, 1), (2, 16384, 16384, 1), (4, 700, 900, 0),
                                         (3, 300, 5000, 0), (2, 6000, 6000, 1)])
def test_nn_distance_prefix_equals_full_search(hip, oracle, knobs, b, n, m, filt):
    """cloudaae_nn_distance_prefix (tf_nndistance.nn_distance(..., distinct2=(count, row_src))): the caller says that
    cloud c's targets are count[c] distinct points followed by copies of them -- the reference's Chamfer targets
    (utils/hidden_point_removal.py:38-43) -- and the search visits the distinct points only.  Bit for bit the results of
    the full search (= the oracle's: first index wins, so every answer is an original; a copy's answer is its
    original's), for both kernels, counts that differ per cloud, a cloud without copies (count = m), and counts the
    entry point must ignore (0, negative, > m); gradients follow from the indices."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    knobs("CLOUDAAE_NN_FILTER", filt)
    rng = np.random.default_rng(n + m)
    counts = [max(1, m // 4), m, max(2, m // 3), 0][:b]
    if b > 2:
        counts[2] = m + 5 if m % 2 else 37
    target = np.empty((b, m, 3), dtype=np.float32)
    src = np.empty((b, m), dtype=np.int32)
    for c in range(b):
        k = counts[c] if 0 < counts[c] <= m else m
        base = (rng.standard_normal((k, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
        pick = rng.integers(0, k, m - k)
        target[c, :k] = base
        target[c, k:] = base[pick]
        src[c, :k] = np.arange(k)
        src[c, k:] = pick
    pred = (rng.standard_normal((b, n, 3)) * 0.05 + [0.1, -0.2, 0.9]).astype(np.float32)
    pred[:, :40] = target[:, 10:50]                  # exact hits
    count = torch.tensor(counts, dtype=torch.int64, device="cuda")
    p, t = _dev(pred), _dev(target)
    full = tf_nndistance.nn_distance(p, t)
    pre = tf_nndistance.nn_distance(p, t, distinct2=(count, torch.from_numpy(src).cuda()))
    for f, g_ in zip(full, pre):
        assert torch.equal(f, g_)
    if n * m <= 4096 * 4096:
        want = oracle.nn_distance(pred, target, threads=8)
        for w, g_ in zip(want, pre):
            assert np.array_equal(w, g_.cpu().numpy())
    for c in range(b):                                   # nearest targets: always an original
        k = counts[c] if 0 < counts[c] <= m else m
        assert int(pre[1][c].max()) < k
    # through autograd: same gradients (they are a function of the indices)
    p1, t1 = p.clone().requires_grad_(True), t.clone().requires_grad_(True)
    p2, t2 = p.clone().requires_grad_(True), t.clone().requires_grad_(True)
    knobs("CLOUDAAE_DETERMINISTIC", 1)
    from cloudaae_amd.utils import _functions as F
    det = F.DETERMINISTIC
    F.DETERMINISTIC = True
    try:
        d1, _, d2, _ = tf_nndistance.nn_distance(p1, t1)
        (d1.sum() + 2 * d2.sum()).backward()
        e1, _, e2, _ = tf_nndistance.nn_distance(p2, t2, distinct2=(count, torch.from_numpy(src).cuda()))
        (e1.sum() + 2 * e2.sum()).backward()
    finally:
        F.DETERMINISTIC = det
    assert torch.equal(p1.grad, p2.grad) and torch.equal(t1.grad, t2.grad)
    # malformed hints are refused
    with pytest.raises(Exception):
        tf_nndistance.nn_distance(p, t, distinct2=(count.int(), torch.from_numpy(src).cuda()))
    with pytest.raises(Exception):
        tf_nndistance.nn_distance(p, t, distinct2=(count, torch.from_numpy(src[:, :-1].copy()).cuda()))


@pytest.mark.parametrize("case", ["lattice", "far_from_origin", "huge", "tiny_scale", "one_candidate", "ragged"])
def test_nn_distance_filter_kernel_adversarial(hip, oracle, case, knobs):
    """The matrix-core search + exact verification (nn_distance_filter_kernel) on inputs built to defeat a
    filter: exact ties everywhere (lattice), coordinates far from the origin (the filter's scores lose all
    their digits and every query falls back to the full scan), overflow, denormal-sized clouds.
    Bit-exact against the oracle, first index wins."""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    knobs("CLOUDAAE_NN_FILTER", 1)
    rng = np.random.default_rng(7)
    if case == "lattice":
        g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
        a = np.stack([rng.permutation(g) for _ in range(2)]).astype(np.float32) * 0.01
        c = np.stack([rng.permutation(g) for _ in range(2)]).astype(np.float32) * 0.01 + np.float32(0.005)
    elif case == "far_from_origin":
        a = (rng.standard_normal((2, 700, 3)) * 0.05 + [300.0, -200.0, 900.0]).astype(np.float32)
        c = (rng.standard_normal((2, 900, 3)) * 0.05 + [300.0, -200.0, 900.0]).astype(np.float32)
        c[:, 0] += 5000.0      # the centre the scores are taken around is an outlier
    elif case == "huge":
        a = (rng.standard_normal((1, 600, 3)) * 1e19).astype(np.float32)
        c = (rng.standard_normal((1, 640, 3)) * 1e19).astype(np.float32)
    elif case == "tiny_scale":
        a = (rng.standard_normal((1, 600, 3)) * 1e-21).astype(np.float32)
        c = (rng.standard_normal((1, 640, 3)) * 1e-21).astype(np.float32)
    elif case == "one_candidate":
        a = rng.standard_normal((3, 70, 3)).astype(np.float32)
        c = rng.standard_normal((3, 1, 3)).astype(np.float32)
    else:
        a = (rng.standard_normal((5, 1031, 3)) * 0.05).astype(np.float32)
        c = (rng.standard_normal((5, 2077, 3)) * 0.05).astype(np.float32)
        c[:, 1000:1040] = c[:, :40]
    want = oracle.nn_distance(a, c, threads=8)
    got = tf_nndistance.nn_distance(_dev(a), _dev(c))
    for w, g_ in zip(want, got):
        assert np.array_equal(w, g_.cpu().numpy())


def test_nn_distance_prefix_broken_hint_is_loud(hip, knobs):
    """A caller that breaks the contract of cloudaae_nn_distance_prefix: a copy row whose row_src is not a
    distinct row gets dist2 = NaN, idx2 = 0 -- never the output buffer's old contents, never an index the gradient
    kernels could follow out of bounds; with CLOUDAAE_NN_PREFIX_VERIFY = 1 a copy that is not bitwise its original
    (a target shuffled after the synthesis) is flagged the same way."""
    from cloudaae_amd import _lib
    L = _lib.lib()
    b, n, m, k = 2, 512, 640, 200
    rng = np.random.default_rng(9)
    base = rng.standard_normal((b, k, 3)).astype(np.float32)
    pick = rng.integers(0, k, (b, m - k))
    target = np.concatenate([base, np.take_along_axis(base, pick[:, :, None].repeat(3, 2), 1)], 1)
    src = np.concatenate([np.tile(np.arange(k), (b, 1)), pick], 1).astype(np.int32)
    src[0, k + 3], src[1, k + 7], src[1, m - 1] = -1, k, m + 100          # not distinct rows
    pred = rng.standard_normal((b, n, 3)).astype(np.float32)
    p, t, sd = _dev(pred), _dev(target), torch.from_numpy(src).cuda()
    count = torch.full((b,), k, dtype=torch.int64, device="cuda")
    d1 = torch.empty((b, n), device="cuda"); i1 = torch.empty((b, n), dtype=torch.int32, device="cuda")

    def run():
        d2 = torch.full((b, m), 123.0, device="cuda"); i2 = torch.full((b, m), 1 << 30, dtype=torch.int32, device="cuda")
        _lib.check(L.cloudaae_nn_distance_prefix(b, n, p.data_ptr(), m, t.data_ptr(), count.data_ptr(), sd.data_ptr(),
                                                 d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), _lib.stream()),
                   "nn_distance_prefix")
        return d2.cpu().numpy(), i2.cpu().numpy()
    d2, i2 = run()
    bad = np.zeros((b, m), dtype=bool)
    bad[0, k + 3] = bad[1, k + 7] = bad[1, m - 1] = True
    assert np.isnan(d2[bad]).all() and (i2[bad] == 0).all()
    assert not np.isnan(d2[~bad]).any() and (i2 >= 0).all() and (i2 < n).all()
    # a copy that is no longer its original
    t[0, k + 11] += 1.0
    d2, _ = run()
    assert not np.isnan(d2[0, k + 11])                    # (trusted without the knob)
    knobs("CLOUDAAE_NN_PREFIX_VERIFY", 1)
    d2, i2 = run()
    bad[0, k + 11] = True
    assert np.isnan(d2[bad]).all() and (i2[bad] == 0).all() and not np.isnan(d2[~bad]).any()


@pytest.mark.parametrize("scale,offset", [(0.05, (0.1, -0.2, 0.9)), (1.0, (0.0, 0.0, 0.0)), (1e-3, (5.0, -3.0, 40.0)),
                                          (300.0, (1e3, 2e3, -5e2))])
def test_nn_distance_split_score_error(hip, scale, offset):
    """The large-cloud Chamfer kernel searches with scores |b'|^2 - 2 a'.b' computed as error-free three-piece bfloat16
    split products on the bf16 matrix pipe (csrc/nn_distance.hip, round 5) and decides with a margin of 160 units of
    2^-24 R, R = (|a'| + max |b'|)^2, of which 64 + 1 + 3 per score are charged to the arithmetic of the score.  Here the
    scores of the same operand construction and instructions (cloudaae_dev_nn_split_scores) are compared with float64 on
    the centred fp32 coordinates: the hardware must stay inside that charge -- it stays far inside (a few units)."""
    from cloudaae_amd import _lib
    L = _lib.lib()
    L._cdll.cloudaae_dev_nn_split_scores.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(int(scale * 1000) % 997)
    worst = 0.0
    for rep in range(40):
        q = (rng.standard_normal((32, 3)) * scale + offset).astype(np.float32)
        c = (rng.standard_normal((32, 3)) * scale + offset).astype(np.float32)
        if rep % 4 == 1:
            c[5:9] = q[5:9]                     # exact hits
        if rep % 4 == 2:
            c[:, 2] = c[0, 2]                   # a flat cloud
        qd, cd = _dev(q), _dev(c)
        sc = torch.full((32, 32), float("nan"), device="cuda")
        R = torch.empty(32, device="cuda")
        _lib.check(L._cdll.cloudaae_dev_nn_split_scores(32, 32, qd.data_ptr(), cd.data_ptr(), sc.data_ptr(), R.data_ptr(),
                                                         _lib.stream()), "dev_nn_split_scores")
        torch.cuda.synchronize()
        a = (q - c[0]).astype(np.float64)       # the centred fp32 values, exactly (fp32 subtraction, then widened)
        b = (c - c[0]).astype(np.float64)
        want = (b * b).sum(1)[None, :] - 2.0 * a @ b.T
        Rw = (np.sqrt((a * a).sum(1)) + np.sqrt((b * b).sum(1)).max()) ** 2
        got, Rg = sc.cpu().numpy().astype(np.float64), R.cpu().numpy().astype(np.float64)
        assert np.allclose(Rg, Rw, rtol=1e-5)
        units = np.abs(got - want) / (2.0 ** -24 * Rw[:, None])
        worst = max(worst, float(units.max()))
    print("largest score error: %.2f units of 2^-24 R" % worst)
    assert worst <= 8.0, worst                   # (charged in the margin: 68)

