"""GPU: on-line synthesis (rigid transform, occluders, spherical flip, hidden point removal) vs the
CPU restatement (oracle/synth_oracle.py: numpy + scipy/qhull) on the reference's own object model and
pose records (tests/golden fixtures)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data(golden_dir):
    from cloudaae_amd import tfrecord_io as T
    models, labels = T.read_and_decode_obj_model(os.path.join(golden_dir, "obj_model_first1.tfrecords"))
    recs = T.PoseRecords([os.path.join(golden_dir, "pose_records_cls0_first4.tfrecords")])
    return models, recs


def _records(recs, device="cuda"):
    return {"translation": torch.from_numpy(recs.translation).to(device),
            "axisangle": torch.from_numpy(recs.axisangle).to(device),
            "class_id": torch.from_numpy(recs.class_id).to(device)}


def test_transform_and_flip_bit_exact(hip, data):
    from cloudaae_amd import train_cloudAAE_ycbv as T
    from cloudaae_amd.utils import hidden_point_removal as hpr
    from oracle import synth_oracle as SO
    models, recs = data
    x = _records(recs)
    x = T.get_object_model(x, torch.from_numpy(models).cuda())
    x = T.transform_object_model(T.get_rotation_matrix(x))
    got = x["model_xyz_rot_trans"].cpu().numpy()
    for i in range(4):
        want = SO.transform_object_model(models[0][:, :3], recs.axisangle[i], recs.translation[i])
        assert np.array_equal(got[i], want)
    rng = np.random.default_rng(0)
    x["occluder"] = torch.from_numpy((rng.standard_normal((4, 400, 3)) * 0.01 + [0.0, 0.0, 0.6]).astype(np.float32)).cuda()
    x = hpr.sphericalFlip(x, None, 0.8 * math.pi)
    x = hpr.sphericalFlip_org(x, None, 0.8 * math.pi)
    for i in range(4):
        pts = np.concatenate([got[i], x["occluder"][i].cpu().numpy()], 0)
        f, o = SO.spherical_flip(pts)
        assert np.array_equal(x["orgPoints"][i].cpu().numpy(), o)
        np.testing.assert_allclose(x["flippedPoints"][i].cpu().numpy(), f, rtol=2e-7, atol=0)
        f2, o2 = SO.spherical_flip(got[i])
        assert x["flippedPoints_org"].shape == (4, 2049, 3)
        np.testing.assert_allclose(x["flippedPoints_org"][i].cpu().numpy(), f2, rtol=2e-7, atol=0)


def test_hidden_point_removal_equals_qhull(hip, data):
    """The visible-id sets must be IDENTICAL to scipy.spatial.ConvexHull's on the same flipped points
    (real object model, real poses, with and without occluders)."""
    from cloudaae_amd.utils import hidden_point_removal as hpr
    from oracle import synth_oracle as SO
    models, recs = data
    rng = np.random.default_rng(1)
    clouds = []
    for i in range(4):
        pts = SO.transform_object_model(models[0][:, :3], recs.axisangle[i], recs.translation[i])
        z = recs.translation[i][2]
        occ = (rng.standard_normal((400, 3)) * 0.01 + [0.01 * i, -0.01, (0.5 + z) / 2]).astype(np.float32)
        clouds.append(np.concatenate([pts, occ], 0))
    flipped, org = zip(*[SO.spherical_flip(c) for c in clouds])
    F, O = torch.from_numpy(np.stack(flipped)).cuda(), torch.from_numpy(np.stack(org)).cuda()
    vis, num, ids = hpr.convexHull(F, O, seed=3, return_ids=True)
    assert vis.shape == (4, 2449, 3) and num.dtype == torch.int64
    for i in range(4):
        want, vertices = SO.convex_hull_visible(flipped[i])
        assert vertices[-1] == 2448                       # the viewpoint is a hull vertex (what [:-1] assumes)
        n = int(num[i])
        assert n == len(want)
        assert np.array_equal(ids[i, :n].cpu().numpy(), want)
        assert (ids[i, n:] == -1).all()
        v = vis[i].cpu().numpy()
        assert np.array_equal(v[:n], org[i][want])
        # padded rows are re-draws of visible points (np.random.choice(visibleId, ...))
        visible_set = {tuple(r) for r in org[i][want]}
        assert all(tuple(r) in visible_set for r in v[n:])
        assert len({tuple(r) for r in v[n:]}) > 50        # and they are spread, not one repeated row
    # without the occluder (the Chamfer target, 2048+1 points)
    f2, o2 = zip(*[SO.spherical_flip(c[:2048]) for c in clouds])
    vis2, num2, ids2 = hpr.convexHull(torch.from_numpy(np.stack(f2)).cuda(), torch.from_numpy(np.stack(o2)).cuda(),
                                      return_ids=True)
    for i in range(4):
        want, _ = SO.convex_hull_visible(f2[i])
        assert np.array_equal(ids2[i, :int(num2[i])].cpu().numpy(), want)


def test_hidden_point_removal_random_poses(hip, data):
    """24 random poses of the object model, each with a random occluder blob and without: the visible-id
    sets equal qhull's (the per-point LPs see their constraints in a data-dependent order -- spatially
    sorted neighbours first -- so this exercises many different orders)."""
    from cloudaae_amd.utils import hidden_point_removal as hpr
    from oracle import synth_oracle as SO
    models, _ = data
    rng = np.random.default_rng(2024)
    clouds = []
    for i in range(24):
        ax = rng.standard_normal(3)
        ax = (ax / np.linalg.norm(ax) * rng.uniform(0, np.pi)).astype(np.float32)
        t = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(0.5, 1.5)], np.float32)
        pts = SO.transform_object_model(models[0][:, :3], ax, t)
        occ = (rng.standard_normal((400, 3)) * rng.uniform(0.005, 0.03) +
               [t[0] + rng.uniform(-.05, .05), t[1] + rng.uniform(-.05, .05), t[2] * rng.uniform(0.4, 0.9)])
        clouds.append(np.concatenate([pts, occ.astype(np.float32)], 0))
    for sl in (slice(None), slice(0, 2048)):
        fl, org = zip(*[SO.spherical_flip(c[sl]) for c in clouds])
        _, num, ids = hpr.convexHull(torch.from_numpy(np.stack(fl)).cuda(), torch.from_numpy(np.stack(org)).cuda(),
                                     return_ids=True)
        for i in range(len(clouds)):
            want, _ = SO.convex_hull_visible(fl[i])
            assert np.array_equal(ids[i, :int(num[i])].cpu().numpy(), want), i


def test_hull_vertices_random_clouds(hip):
    """Generic clouds (not HPR-shaped): Gaussian blob, points on a sphere (all vertices), a cube
    lattice (many coplanar/interior points)."""
    from cloudaae_amd.utils import hidden_point_removal as hpr
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(5)
    blob = rng.standard_normal((500, 3))
    sph = rng.standard_normal((300, 3)); sph /= np.linalg.norm(sph, axis=1, keepdims=True)
    for pts in (blob, sph * 3 + [0, 0, 10]):
        P = np.concatenate([pts, np.zeros((1, 3))], 0).astype(np.float32)
        hull = ConvexHull(P.astype(np.float64))
        vis, num, ids = hpr.convexHull(torch.from_numpy(P[None]).cuda(), torch.from_numpy(P[None]).cuda(),
                                       return_ids=True)
        want = np.sort(hull.vertices)[:-2]
        assert np.array_equal(ids[0, :int(num[0])].cpu().numpy(), want)


@pytest.mark.parametrize("case", ["hpr_8192", "hpr_2048", "blob_small", "sphere", "tiny", "two_scales"])
def test_hull_vertex_paths_agree(hip, data, knobs, case):
    """Round 5 rebuilt the hull-vertex test (3-D Morton sort, culled verification passes over bounding slabs, a point queue per
    cloud); the round-4 paths are still there behind knobs.  Whatever the combination, the visible ids are the same -- on
    HPR-shaped clouds of both kernel sizes (8-wave / 16-wave workgroups), generic clouds, clouds smaller than a point's
    neighbourhood (round 5 found the round-4 kernel wrong there: the neighbour offsets came round to a point twice and a
    binding constraint was re-tested) and a tight cluster next to a few far points.  (Rows that repeat EXACTLY are outside
    the contract, as for the per-point LPs of round 2: a copy of a binding constraint tests as violated by round-off.)"""
    from cloudaae_amd.utils import hidden_point_removal as hpr
    from oracle import synth_oracle as SO
    models, _ = data
    rng = np.random.default_rng(77)
    if case.startswith("hpr"):
        n = int(case.split("_")[1])
        base = models[0][:, :3]
        pick = rng.integers(0, len(base), n)
        clouds = []
        for i in range(4):
            ax = rng.standard_normal(3)
            ax = (ax / np.linalg.norm(ax) * rng.uniform(0, np.pi)).astype(np.float32)
            t = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0.6, 1.4)], np.float32)
            pts = SO.transform_object_model(base[pick] + rng.standard_normal((n, 3)).astype(np.float32) * 1e-3, ax, t)
            occ = (rng.standard_normal((400, 3)) * 0.02 + [t[0], t[1], t[2] * 0.7]).astype(np.float32)
            clouds.append(np.concatenate([pts, occ], 0))
        fl, org = zip(*[SO.spherical_flip(c) for c in clouds])
        F, O = np.stack(fl), np.stack(org)
    else:
        if case == "blob_small":
            pts = rng.standard_normal((3, 150, 3))
        elif case == "sphere":
            pts = rng.standard_normal((3, 700, 3))
            pts /= np.linalg.norm(pts, axis=-1, keepdims=True)
            pts = pts * 2 + [0, 0, 9]
        elif case == "tiny":
            pts = rng.standard_normal((5, 9, 3))
        else:
            pts = np.concatenate([rng.standard_normal((3, 900, 3)) * 1e-3, rng.standard_normal((3, 40, 3)) * 5.0], 1)
        F = np.concatenate([pts, np.zeros((pts.shape[0], 1, 3))], 1).astype(np.float32)
        O = F
    Fd, Od = torch.from_numpy(F).cuda(), torch.from_numpy(O).cuda()

    def run(cull):
        knobs("CLOUDAAE_HPR_CULL", cull)
        _, num, ids = hpr.convexHull(Fd, Od, return_ids=True)
        return [ids[i, :int(num[i])].cpu().numpy() for i in range(F.shape[0])]

    ref = run(0)                            # the full strided scan behind the local problem (round 4's; the fallback of today's)
    if case.startswith("hpr"):              # qhull on the flipped cloud, at BASELINE configs[4]'s hull size too (8593 points)
        for i in range(F.shape[0]):
            want, _ = SO.convex_hull_visible(F[i])
            assert np.array_equal(ref[i], want), i
    else:
        from scipy.spatial import ConvexHull
        for i in range(F.shape[0]):
            assert np.array_equal(ref[i], np.sort(ConvexHull(F[i].astype(np.float64)).vertices)[:-2]), i
    # 1 = the default: culled verification passes; 2 / 3: no / one point may join a working set -- every point whose first pass
    # finds a violation goes through the strided scan after all: the fallback's answers are the same
    for cull in (1, 2, 3):
        got = run(cull)
        for i in range(F.shape[0]):
            assert np.array_equal(got[i], ref[i]), (cull, i)


def test_occluder_statistics_and_layout(hip):
    from cloudaae_amd.utils import generate_occluder
    from oracle import synth_oracle as SO
    B = 512
    t = torch.zeros((B, 3), device="cuda"); t[:, 2] = torch.linspace(0.6, 1.6, B, device="cuda")
    x = generate_occluder.get_random_spherical_occluder({"translation": t}, "ycbv", seed=11)
    occ = x["occluder"].cpu().numpy()
    assert occ.shape == (B, 400, 3)
    hnear, wnear = SO.get_frustum_near()
    assert abs(hnear - 0.55785) < 1e-4 and abs(wnear - 0.71901) < 1e-4      # tan(22.5 rad) quirk
    b1, b2 = occ[:, 0::2], occ[:, 1::2]                      # the two blobs interleave row by row
    for blob in (b1, b2):
        c = blob.mean(1)
        assert abs(np.std(blob - c[:, None], axis=1).mean() - 0.01) < 5e-4          # sigma 0.01
        assert abs(c[:, 0].std() - wnear / 10) < 0.012 and abs(c[:, 1].std() - hnear / 10) < 0.01
        z = t[:, 2].cpu().numpy()
        assert abs((c[:, 2] - (0.5 + z) / 2).mean()) < 0.02
        assert abs((c[:, 2] - (0.5 + z) / 2).std() - ((z - 0.5) / 6).mean()) < 0.03
    assert np.abs(b1.mean(1) - b2.mean(1)).mean() > 0.01                  # distinct centres
    y = generate_occluder.get_random_spherical_occluder({"translation": t}, "ycbv", seed=11)
    assert torch.equal(x["occluder"], y["occluder"])                      # counter-based: reproducible
    z = generate_occluder.get_random_spherical_occluder({"translation": t}, "ycbv", seed=12)
    assert not torch.equal(x["occluder"], z["occluder"])


def test_pipeline_feeds_a_train_step(hip, data):
    """tfrecord records -> get_small_data -> TrainGraph.train_step at the reference's default N=256."""
    from cloudaae_amd import train_cloudAAE_ycbv as T
    models, recs = data
    obj = torch.from_numpy(np.repeat(models, 21, axis=0)).cuda()          # 21 classes (fixture has class 0)
    el = T.get_small_data(_records(recs), obj, seed=5)
    assert el["visiblePoints"].shape == (4, 2449, 3) and el["visiblePoints_org"].shape == (4, 2049, 3)
    assert (el["num_vis_point"] > 300).all() and (el["num_vis_point_org"] >= el["num_vis_point"] - 400).all()
    graph = T.TrainGraph({"num_point": 256, "gpu": 0}, {}, {"batch_size": 4})
    out = graph.train_step(el)
    assert math.isfinite(float(out["total_loss"])) and out["xyz_recon"].shape == (4, 1024, 3)


def test_config5_synthesis_rows_and_models(hip, data):
    """BASELINE configs[4] (N = 4096 input points): synthetic 8192-point object models and a visiblePoints_org of 4N =
    16384 rows -- more rows than a model has points, filled by the reference's own rule (hidden_point_removal.py:38-40:
    the visible points in ascending index, then random re-draws of visible points).  The first num_vis rows equal the
    reference-shaped call's, every padded row is one of the visible points; the visible set of an 8193-point cloud is
    qhull's."""
    from cloudaae_amd import train_cloudAAE_ycbv as T
    from oracle import synth_oracle as SO
    _, recs = data
    models = T.synthetic_object_models(3, 8192, seed=5, device="cuda")
    assert tuple(models.shape) == (3, 8192, 6) and float(models[:, :, :3].abs().max()) < 0.12
    x = _records(recs)
    x["class_id"] = torch.tensor([0, 1, 2, 1], device="cuda")
    ref = T.get_small_data(dict(x), models, seed=3)
    big = T.get_small_data(dict(x), models, seed=3, rows_org=16384)
    assert tuple(ref["visiblePoints"].shape) == (4, 8192 + 400 + 1, 3) and tuple(ref["visiblePoints_org"].shape) == (4, 8193, 3)
    assert tuple(big["visiblePoints_org"].shape) == (4, 16384, 3) and torch.equal(big["visiblePoints"], ref["visiblePoints"])
    assert torch.equal(big["num_vis_point_org"], ref["num_vis_point_org"])
    for i in range(4):
        nv = int(ref["num_vis_point_org"][i])
        assert 500 < nv < 8192
        assert torch.equal(big["visiblePoints_org"][i, :nv], ref["visiblePoints_org"][i, :nv])
        vis = {tuple(r) for r in ref["visiblePoints_org"][i, :nv].cpu().numpy().tolist()}
        pad = big["visiblePoints_org"][i, nv:].cpu().numpy()
        assert all(tuple(r) in vis for r in pad[::37].tolist())
        assert len({tuple(r) for r in pad.tolist()}) > min(nv, 2000) // 2          # re-draws, not one repeated point
    # the visible set of one 8193-point cloud against qhull
    from cloudaae_amd.utils import hidden_point_removal as hpr
    want, _ = SO.convex_hull_visible(big["flippedPoints_org"][0].cpu().numpy())
    vis9, num, ids, rsrc = hpr.convexHull(big["flippedPoints_org"][:1].contiguous(), big["orgPoints_org"][:1].contiguous(),
                                          seed=0, return_ids=True, rows=9000, return_src=True)
    assert int(num[0]) == len(want) and np.array_equal(ids[0, :len(want)].cpu().numpy(), want)
    assert (ids[0, len(want):] == -1).all() and ids.shape == (1, 9000)
    # row_src: every output row names the row < num_vis it equals (itself for a visible point, the drawn one for a re-draw)
    nv = len(want)
    rs = rsrc[0].cpu().numpy()
    assert rsrc.shape == (1, 9000) and np.array_equal(rs[:nv], np.arange(nv)) and rs[nv:].min() >= 0 and rs[nv:].max() < nv
    assert torch.equal(vis9[0], vis9[0][rsrc[0].long()])
    assert len(np.unique(rs[nv:])) > 0.5 * min(nv, 9000 - nv)          # (re-draws are spread over the visible points)
    # ... and it feeds a k = 20 train step at N = 4096
    g = T.TrainGraph({"num_point": 4096, "gpu": 0}, {}, {"batch_size": 4}, k_neighbor=20)
    out = g.train_step(big)
    assert tuple(out["xyz_recon"].shape) == (4, 16384, 3) and np.isfinite(float(out["total_loss"]))
