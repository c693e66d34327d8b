"""GPU: the product kernels AND the CPU oracle against the reference's OWN GPU kernels, run here.

oracle/_ref/libref_gpu.so = tf_ops/sampling/tf_sampling_g.cu (whole) and tf_ops/nn_distance/tf_nndistance_g.cu:5-151, compiled
for gfx950 by hipcc from where they lie under /root/reference (oracle/build_ref.sh, oracle/ref_gpu_shim.hip; -ffp-contract=off:
the un-fused arithmetic the oracle defines).  hipcc is not the reference's toolchain, so this is corroboration, not
a formal pin -- but the tie-break of farthest point sampling (the 512-thread strided scan and the `dists[i1] < dists[i2]` tree,
tf_sampling_g.cu:130-165), the association order of ProbSample's prefix sums (cumsumKernel, :7-82) and the first-wins argmin of
the Chamfer kernel across its tiles (tf_nndistance_g.cu:5-127) are decided by those lines, and the oracle only RESTATES them.
Three-way, bit for bit: reference kernel == oracle == cloudaae_* through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def refgpu(oracle):
    if not oracle.have_ref_gpu():
        pytest.skip("oracle/_ref/libref_gpu.so was not built (needs /root/reference at build time)")
    return oracle


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _clouds(kind, b, n, rng):
    if kind == "randn":
        p = rng.standard_normal((b, n, 3)).astype(np.float32)
    elif kind == "dup":              # copies of points: exact ties in every round
        p = rng.standard_normal((b, n, 3)).astype(np.float32)
        if n > 8:
            p[:, n // 2:n // 2 + n // 4] = p[:, :n // 4]
    elif kind == "lattice":          # integer grid: many equal distances, exact in fp32
        p = rng.integers(-4, 5, (b, n, 3)).astype(np.float32)
    elif kind == "same":             # one point n times: every round is an n-way tie
        p = np.repeat(rng.standard_normal((b, 1, 3)).astype(np.float32), n, axis=1)
    else:
        raise ValueError(kind)
    return p


@pytest.mark.parametrize("b,n,m,kind", [(1, 1, 4, "randn"), (3, 100, 100, "randn"), (2, 513, 64, "dup"), (4, 4096, 1024, "randn"),
                                        (1, 9000, 50, "dup"), (33, 2048, 128, "randn"), (1, 20000, 40, "randn"),
                                        (2, 1500, 300, "lattice"), (2, 700, 20, "same"), (5, 511, 511, "lattice"),
                                        (40, 1024, 256, "dup"), (2, 3073, 100, "lattice")])
def test_farthest_point_sampling_three_way(hip, refgpu, b, n, m, kind):
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    p = _clouds(kind, b, n, np.random.default_rng(n * 7 + m))
    d = _dev(p)
    ref = refgpu.ref_gpu_farthest_point_sample(m, d).cpu().numpy()
    ours = tf_sampling.farthest_point_sample(m, d).cpu().numpy()
    cpu = refgpu.farthest_point_sample(m, p, threads=8)
    assert np.array_equal(ref, cpu), "the oracle's restatement differs from tf_sampling_g.cu:105-170"
    assert np.array_equal(ref, ours), "cloudaae_farthest_point_sample differs from tf_sampling_g.cu:105-170"


def test_gather_and_its_gradient_against_the_reference_kernels(hip, refgpu):
    from cloudaae_amd.tf_ops.sampling import tf_sampling
    rng = np.random.default_rng(4)
    inp = rng.standard_normal((3, 300, 3)).astype(np.float32)
    idx = rng.integers(0, 300, (3, 70)).astype(np.int32)
    di, dx = _dev(inp), _dev(idx)
    assert torch.equal(tf_sampling.gather_point(di, dx), refgpu.ref_gpu_gather_point(di, dx))
    og = _dev(rng.standard_normal((3, 70, 3)).astype(np.float32))
    # unique indices (what farthest point sampling produces): one term per sum, exact
    perm = _dev(np.stack([rng.permutation(300)[:70] for _ in range(3)]).astype(np.int32))
    assert torch.equal(tf_sampling.gather_point_grad(di, perm, og), refgpu.ref_gpu_gather_point_grad(300, perm, og))
    # repeated indices: the reference adds with atomics in arrival order -- equal to round-off
    torch.testing.assert_close(tf_sampling.gather_point_grad(di, dx, og), refgpu.ref_gpu_gather_point_grad(300, dx, og),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("b,n,m", [(1, 1, 5), (2, 7, 50), (3, 21, 1000), (2, 4096, 300), (1, 8192 + 37, 500), (2, 20000, 64)])
def test_prob_sample_three_way(hip, refgpu, b, n, m):
    """indices AND prefix sums: cumsumKernel's blocked scan (tf_sampling_g.cu:7-82) == the oracle's restatement == ours"""
    from cloudaae_amd import _lib
    rng = np.random.default_rng(n + m)
    p = rng.random((b, n)).astype(np.float32)
    p[:, ::5] = 0.0
    r = rng.random((b, m)).astype(np.float32)
    pd, rd = _dev(p), _dev(r)
    ref_idx, ref_cum = refgpu.ref_gpu_prob_sample(pd, rd)
    want, cum = refgpu.prob_sample(p, r, return_cumsum=True)
    assert np.array_equal(ref_cum.cpu().numpy(), cum), "the oracle's prefix sums differ from cumsumKernel's"
    assert np.array_equal(ref_idx.cpu().numpy(), want), "the oracle's draws differ from binarysearchKernel's"
    temp = torch.empty((b, n), device="cuda")
    out = torch.empty((b, m), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().cloudaae_prob_sample(b, n, m, _lib.ptr(pd), _lib.ptr(rd), _lib.ptr(temp), _lib.ptr(out),
                                               _lib.stream()), "prob_sample")
    assert torch.equal(temp, ref_cum) and torch.equal(out, ref_idx)


@pytest.mark.parametrize("b,n,m,kind", [(1, 1, 1, "randn"), (2, 3, 1000, "randn"), (3, 257, 255, "dup"), (2, 4096, 4096, "randn"),
                                        (2, 5000, 17, "lattice"), (5, 2100, 2049, "dup"), (32, 1024, 4096, "randn"),
                                        (2, 600, 600, "same")])
def test_chamfer_search_three_way(hip, refgpu, b, n, m, kind):
    """NmDistanceKernel (tf_nndistance_g.cu:5-127, un-fused) == the reference's CPU lines (tf_nndistance.cpp:21-43) == ours:
    squared distances and first-wins indices, ties included"""
    from cloudaae_amd.tf_ops.nn_distance import tf_nndistance
    rng = np.random.default_rng(n * 3 + m)
    a, c = _clouds(kind, b, n, rng), _clouds(kind, b, m, rng)
    da, dc = _dev(a), _dev(c)
    ref = refgpu.ref_gpu_nn_distance(da, dc)
    ours = tf_nndistance.nn_distance(da, dc)
    for r, o in zip(ref, ours):
        assert torch.equal(r, o)
    cpu = (refgpu.ref_nn_distance if refgpu.have_ref() else refgpu.nn_distance)(a, c)
    for r, w in zip(ref, cpu):
        assert np.array_equal(r.cpu().numpy(), w)


def test_chamfer_gradient_against_the_reference_kernel(hip, refgpu):
    from cloudaae_amd import _lib
    rng = np.random.default_rng(9)
    b, n, m = 3, 700, 900
    a, c = _dev(rng.standard_normal((b, n, 3)).astype(np.float32)), _dev(rng.standard_normal((b, m, 3)).astype(np.float32))
    d1, i1, d2, i2 = refgpu.ref_gpu_nn_distance(a, c)
    g1, g2 = _dev(rng.standard_normal((b, n)).astype(np.float32)), _dev(rng.standard_normal((b, m)).astype(np.float32))
    ra, rc = refgpu.ref_gpu_nn_distance_grad(a, c, g1, i1, g2, i2)
    oa, oc = torch.empty_like(a), torch.empty_like(c)
    P = _lib.ptr
    _lib.check(_lib.lib().cloudaae_nn_distance_grad(b, n, P(a), m, P(c), P(g1), P(i1), P(g2), P(i2), P(oa), P(oc),
                                                    _lib.stream()), "nn_distance_grad")
    # (both add a point's terms in an order that is not fixed: equal to fp32 round-off)
    torch.testing.assert_close(oa, ra, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(oc, rc, rtol=1e-4, atol=1e-5)
