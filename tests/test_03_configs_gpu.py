"""GPU: BASELINE.json's configurations AT THEIR SIZE, through the path bench.py and the CLI use (the
recorded step, replayed), against the CPU oracle (oracle/model_oracle.py):

    configs[1]  all 21 classes, batch 32,  N = 1024, fp32
    configs[3]  per-GPU shape of the 8-GPU run: batch 128, N = 1024, fp32
    configs[2]  batch 256, N = 1024, bf16 dense-layer operands, fp32 everything else
    configs[4]  its network shape: N = 4096, k = 20 (batch 8; the on-line synthesis in front of it: test_04_synth_gpu.py)

One full iteration of train_cloudAAE_ycbv.py:344-368 from a mid-training state (step counter 2, Adam
slots non-zero): the three losses, the reconstruction, the gradient of every variable, the BN moving
averages, and what the optimiser leaves behind -- weights, both Adam slots, beta powers, step counter
and the next step's bn_decay.  The compared pass is the REPLAYED one (the recording pass is compared
with it too).  The oracle groups on the GPU's neighbour indices (the kNN op itself is bit-exact on
identical inputs, test_00_ops_gpu.py); how often free-running grouping differs is measured per layer,
printed, written to gpurun_out/ and bounded.

Size-dependent kernel choices these shapes reach and the small cases do not: knn64_scan by grid
size, the column-sum epilogue of the dgcnn_agg product with many tile rows, split-K slice counts,
the fully connected stack through the GEMM + small-M batch norm path (batch > 32).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# largest |normalised value| of a fully connected unit whose ReLU the bf16 step and the fp32-activation oracle may decide
# differently (their inputs agree to ~1e-3): measured in round 6 at B = 256: ten units, the largest at 1.5e-4 (round 5 allowed 3e-2)
RELU_FLIP_MAX_BF16 = 1e-3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP0 = 2            # the step counter (`batch`, train...:192) the compared iteration starts from

CONFIGS = [
    pytest.param("configs[1]", 32, 1024, "f32", 10, id="cfg1-B32-f32"),
    pytest.param("configs[3]/gpu", 128, 1024, "f32", 10, id="cfg3-B128-f32"),
    pytest.param("configs[2]", 256, 1024, "bf16", 10, id="cfg2-B256-bf16"),
    # configs[1] with the three dgcnn_agg products as error-free 3 x bf16 splits on the bf16 matrix cores (opt-in,
    # csrc/gemm_x3.hip): held to the fp32 tolerances against the fp32 oracle -- it is an fp32-accurate product
    pytest.param("configs[1]/split products", 32, 1024, "bf16x3", 10, id="cfg1-B32-bf16x3"),
    # configs[4]'s network shape (N = 4096 points, k = 20 neighbours, a 16384-point Chamfer target) at a batch the CPU
    # oracle finishes in under a minute; 8 clouds = 1024 query tiles: the 16-wave kNN kernel with k = 20 and 128-slot
    # queues, the kernel bench.py --config5 runs
    pytest.param("configs[4]/net", 8, 4096, "f32", 20, id="cfg5-B8-N4096-k20"),
    # the same at configs[4]'s own batch of 32 (the oracle step takes ~1 minute on the GPU boxes' 256 host cores;
    # CLOUDAAE_SKIP_FULL_CFG5=1 leaves it out on a small host)
    pytest.param("configs[4]/net, full batch", 32, 4096, "f32", 20, id="cfg5-B32-N4096-k20",
                 marks=pytest.mark.skipif(os.environ.get("CLOUDAAE_SKIP_FULL_CFG5") == "1",
                                          reason="full-size configs[4] parity skipped (CLOUDAAE_SKIP_FULL_CFG5=1)")),
]


def _rel(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return float((got - want).abs().max() / (want.abs().max() + 1e-30))


def _nrm(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    return float((got - want).norm() / (want.norm() + 1e-30))


def _snapshot(g):
    return [t.clone() for t in (g.store.flat_params, g.store.flat_state, g.adam_m, g.adam_v, g.batch,
                                g.beta1_power, g.beta2_power, g.bn_decay)]


def _restore(g, snap):
    with torch.no_grad():
        for dst, src in zip((g.store.flat_params, g.store.flat_state, g.adam_m, g.adam_v, g.batch,
                             g.beta1_power, g.beta2_power, g.bn_decay), snap):
            dst.copy_(src)


def _adam_reference(p, g, m, v, lr, b1, b2, eps, b1p, b2p):
    """TF-1.x ApplyAdam in numpy fp32 (the closed form MO.AdamTF restates, train...:263-273)."""
    f = np.float32
    lr_t = f(lr) * np.sqrt(f(1) - f(b2p)) / (f(1) - f(b1p))
    m = m + (g - m) * (f(1) - f(b1))
    v = v + (g * g - v) * (f(1) - f(b2))
    p = p - (m * lr_t) / (np.sqrt(v) + f(eps))
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32)


@pytest.mark.parametrize("name,B,N,dtype,kn", CONFIGS)
def test_config_replayed_step_vs_oracle(hip, name, B, N, dtype, kn):
    from cloudaae_amd import train_cloudAAE_ycbv as T
    from oracle import model_oracle as MO
    bf16 = dtype == "bf16"
    graph = T.TrainGraph({"num_point": N, "gpu": 0}, {}, {"batch_size": B}, replay=True, gemm_dtype=dtype, k_neighbor=kn)
    V = MO.Vars(seed=31)
    with torch.no_grad():       # creates the oracle's variables (their shapes depend on N only)
        MO.forward_losses(MO.synthetic_batch(2, N, seed=1), V, N, is_training=False)
    batch = MO.synthetic_batch(B, N, seed=1000 + B)
    graph.store.load_state_dict(V.state_dict())
    assert graph.store.num_params == sum(p.numel() for p in V.p.values())

    # a mid-training state: step counter STEP0, beta powers advanced STEP0 times, Adam slots non-zero
    opt = MO.AdamTF()
    for _ in range(STEP0):
        opt.b1p = np.float32(opt.b1p * opt.b1)
        opt.b2p = np.float32(opt.b2p * opt.b2)
    gen = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for nme, p in V.p.items():
            opt.m[nme] = torch.randn(p.shape, generator=gen) * 1e-3
            opt.v[nme] = torch.rand(p.shape, generator=gen) * 1e-6
            o, n = graph.store.offsets[nme], p.numel()
            graph.adam_m[o:o + n].copy_(opt.m[nme].reshape(-1))
            graph.adam_v[o:o + n].copy_(opt.v[nme].reshape(-1))
        graph.batch.fill_(float(STEP0))
        graph.beta1_power.fill_(float(opt.b1p))
        graph.beta2_power.fill_(float(opt.b2p))
    graph.refresh_bn_decay()
    decay0 = MO.bn_decay_schedule(STEP0, B)
    assert abs(float(graph.bn_decay) - decay0) < 1e-7
    start = _snapshot(graph)
    b1p0, b2p0 = opt.b1p, opt.b2p

    dev = {k: v.cuda() for k, v in batch.items()}
    keys = ("xyz_loss", "trans_loss", "axag_loss", "total_loss")

    def run():
        out = graph.train_step(dev)
        torch.cuda.synchronize()
        return dict(losses={k: float(out[k]) for k in keys}, recon=out["xyz_recon"].clone(),
                    grads=graph.store.flat_grads.clone(), params=graph.store.flat_params.clone(),
                    state=graph.store.flat_state.clone(), m=graph.adam_m.clone(), v=graph.adam_v.clone(),
                    idx=[out["end_points"]["nn_idx%d" % i].cpu().clone() for i in (1, 2, 3, 4)],
                    scalars=[float(graph.batch), float(graph.beta1_power), float(graph.beta2_power),
                             float(graph.bn_decay)])

    from cloudaae_amd.utils import tf_util
    tf_util.FC_TAP = {}
    try:
        rec = run()                               # the recording pass (eager issue, arena buffers)
        fc_on = {scope: (t.detach() > 0).cpu() for scope, t in tf_util.FC_TAP.items()}
    finally:
        tf_util.FC_TAP = None
    assert graph.replay and graph._plan is not None and not graph._plan.foreign_ops, "step not replayable"
    _restore(graph, start)
    rep = run()                                   # the replayed pass: what bench.py times
    # the two passes ran the same kernels on the same data: the forward pass is bit-reproducible (fixed-order
    # sums everywhere), the backward pass reorders fp32 atomics (round-off only)
    for k in keys:
        assert rec["losses"][k] == rep["losses"][k], (k, rec["losses"][k], rep["losses"][k])
    assert torch.equal(rec["recon"], rep["recon"])
    assert all(torch.equal(a, b) for a, b in zip(rec["idx"], rep["idx"]))
    assert _nrm(rec["grads"], rep["grads"]) < 1e-3      # (a Chamfer near-tie may flip between the passes)

    # ---- free-running grouping: how often do the neighbour SETS differ when the oracle groups on its own? ----
    shadows = {k: v.clone() for k, v in V.s.items()}
    MO.GEMM_BF16 = MO.ACT_BF16 = bf16
    try:
        with torch.no_grad():
            free = MO.forward_losses(batch, V, N, True, decay0, kn)
    finally:
        MO.GEMM_BF16 = MO.ACT_BF16 = False
    for k, v in shadows.items():
        V.s[k].copy_(v)
    mismatch = []
    for i in range(4):
        a = rep["idx"][i].long().sort(-1).values
        b = free["end_points"]["nn_idx%d" % (i + 1)].long().sort(-1).values
        mismatch.append(float((a != b).any(-1).float().mean()))
    print("\n%s (B=%d, N=%d, %s): free-running neighbour-set mismatch per layer (fraction of points): %s"
          % (name, B, N, dtype, ", ".join("%.5f" % m for m in mismatch)))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "knn_free_running_mismatch_B%d_N%d_k%d_%s.json" % (B, N, kn, dtype)), "w") as f:
            json.dump({"config": name, "B": B, "N": N, "k": kn, "dtype": dtype,
                       "mismatch_fraction_of_points_per_layer": mismatch}, f)
    except OSError:
        pass
    # layer 1 groups on xyz (inputs differ by the round-off of the centroid only: measured 0); layers 2-4 on
    # features that went through one more batch norm each (measured, profiles/r02_knn_free_running_mismatch_*:
    # at most 0.95 % of the points at B=32, 1.14 % at B=128, 1.40 % at B=256 with bf16 operands, where a feature
    # on a rounding boundary flips).  The bounds are what was measured plus headroom for another seed, so a
    # regression of the kNN numerics cannot hide: 2 % (fp32), 3 % (bf16).
    # (k = 20 at N = 4096: twice the neighbours per point, twice the k-th / (k+1)-th near-ties: measured 0 / 0.48 / 0.93 / 2.48 %, bound 4 %)
    assert mismatch[0] < 0.001 and max(mismatch) < (0.03 if bf16 else (0.02 if kn <= 10 else 0.04)), mismatch

    # ---- the oracle's iteration, grouped on the GPU's indices ----
    p0 = {n: p.detach().clone() for n, p in V.p.items()}
    MO.GEMM_BF16 = MO.ACT_BF16 = bf16       # (bf16 mode: dgcnn_agg's y is stored as bfloat16, F.ACT_BF16)
    # ... and, for the units of the fully connected stack that sit on the ReLU's corner (|normalised value| < 1e-4: a coin
    # toss between two correct fp32 implementations, worth 1 / B of a column's gradient each -- found at B = 128: one unit
    # of dgcnn_rot_fc2, column 237 3.6 % off and every other column at 1e-7, tools/dev/chk_cfg3_rot.py), on the GPU's side
    # of the corner; everywhere else the two activation patterns must be equal (the oracle asserts it)
    MO.RELU_OVERRIDE = {s_: m for s_, m in fc_on.items() if s_ + "/bn/gamma" in V.p}
    MO.RELU_REPORT = []
    tie0 = MO.RELU_TIE
    if bf16:
        MO.RELU_TIE = RELU_FLIP_MAX_BF16   # (bf16 operands below the stack: its inputs agree to ~1e-3, not to round-off)
    try:
        ref, grads = MO.train_step(batch, V, opt, STEP0, N, B, k=kn, nn_override=rep["idx"])
        took = [(s_, a_, c_, "%.1e" % v_) for s_, a_, c_, v_ in MO.RELU_REPORT if c_]
        print("%s: fully connected units within %.0e of the ReLU corner: %d, of which the oracle took the GPU's side: %s"
              % (name, MO.RELU_TIE, sum(r_[1] for r_ in MO.RELU_REPORT), took or "none"))
        assert len(MO.RELU_REPORT) >= 6, MO.RELU_REPORT          # the six batch-normalised layers were compared
        # the override is for a HANDFUL of units on the corner, not a licence: measured (round 6) 0 or 1 unit in fp32 at every
        # size, 10 of 917 504 in bf16 mode (B = 256); more than that, or a flipped unit far from the corner, is a regression
        n_took = sum(r_[2] for r_ in MO.RELU_REPORT)
        assert n_took <= (30 if bf16 else 4), MO.RELU_REPORT
        assert max(r_[3] for r_ in MO.RELU_REPORT) <= (RELU_FLIP_MAX_BF16 if bf16 else 1e-4), MO.RELU_REPORT
    finally:
        MO.GEMM_BF16 = MO.ACT_BF16 = False
        MO.RELU_OVERRIDE = MO.RELU_REPORT = None
        MO.RELU_TIE = tie0

    ltol = 2e-3 if bf16 else 1e-5             # north star: fp32 Chamfer / pose losses within 1e-5
    for k in ("xyz_loss", "trans_loss", "axag_loss"):
        a, b = rep["losses"][k], float(ref[k])
        assert abs(a - b) <= ltol * max(1.0, abs(b)), (k, a, b)
    a, b = rep["losses"]["total_loss"], float(ref["total_loss"])
    assert abs(a - b) <= ltol * abs(b), ("total_loss", a, b)
    assert _rel(rep["recon"], ref["xyz_recon"]) < (5e-3 if bf16 else 1e-4)

    offs = graph.store.offsets
    gmax = max(float(g.abs().max()) for g in grads.values())
    report = []
    for nme, g in grads.items():
        got = rep["grads"][offs[nme]:offs[nme] + g.numel()].view(g.shape)
        if nme.endswith("/biases") and (nme.rsplit("/", 1)[0] + "/bn/beta") in grads:
            # a bias in front of a batch norm: analytically zero gradient
            report.append((float(got.abs().max()) / ((1e-2 if bf16 else 1e-3) * gmax + 1e-3), nme, "zero-bias"))
            continue
        if bf16:
            err, tol, how = _nrm(got, g), 2e-2, "L2"          # (measured <= 3.2e-3)
        elif nme.startswith("dgcnn_output/"):
            # 4096 x 4096 Chamfer pairs per cloud: a few nearest-neighbour assignments sit on 1e-7 near-ties
            # and flip between the two implementations, which moves one point's whole gradient to another
            # column of this layer -- a discrete change for those columns, invisible in the norm
            err, tol, how = _nrm(got, g), 5e-3, "L2"           # (measured < 2e-3)
        else:
            encoder = nme.startswith("dgcnn") and nme.split("/")[0] in ("dgcnn1", "dgcnn2", "dgcnn3", "dgcnn4",
                                                                         "dgcnn_agg")
            # measured at these sizes (profiles of round 3): encoder <= 6e-4, fully connected stack <= 5.6e-4
            err, tol, how = _rel(got, g), (1.5e-3 if encoder else 1e-3), "max"
        report.append((err / tol, nme, "%s %.2e (tol %.0e)" % (how, err, tol)))
    report.sort(reverse=True)
    print("%s: gradient error / tolerance, worst five: %s" % (name, "; ".join("%s %.2f [%s]" % (n, r, h)
                                                                                for r, n, h in report[:5])))
    assert report[0][0] < 1.0, report[:5]

    # BN moving averages after the step (utils/tf_util.py:493-500)
    for nme, s in V.s.items():
        assert _rel(graph.store.vars[nme].data, s) < (2e-3 if bf16 else 1e-4), nme

    # ---- the optimiser: cloudaae_adam_tf_step on ITS gradients = the TF formula, element for element ----
    sp, _, sm, sv = (t.cpu().numpy() for t in start[:4])
    gp = rep["grads"].cpu().numpy()
    want_p, want_m, want_v = _adam_reference(sp, gp, sm, sv, 0.0008, 0.9, 0.999, 1e-8, b1p0, b2p0)
    used = np.zeros(sp.shape, bool)
    for nme, p in V.p.items():
        used[offs[nme]:offs[nme] + p.numel()] = True
    assert np.abs(rep["params"].cpu().numpy() - want_p)[used].max() < 2e-7
    assert np.abs(rep["m"].cpu().numpy() - want_m)[used].max() <= 1e-6 * np.abs(want_m[used]).max()
    assert np.abs(rep["v"].cpu().numpy() - want_v)[used].max() <= 1e-6 * np.abs(want_v[used]).max()

    # ... and against the oracle's own post-step weights: the UPDATE vectors agree (where the gradient is
    # analytically zero the update is round-off amplified by 1/sqrt(v), so those biases are left out)
    num = den = 0.0
    for nme, p in V.p.items():
        if nme.endswith("/biases") and (nme.rsplit("/", 1)[0] + "/bn/beta") in grads:
            continue
        o, n = offs[nme], p.numel()
        du_ref = (p.detach() - p0[nme]).reshape(-1).double()
        du_got = (rep["params"][o:o + n].cpu() - p0[nme].reshape(-1)).double()
        num += float((du_ref - du_got).pow(2).sum())
        den += float(du_ref.pow(2).sum())
    upd_err = (num / den) ** 0.5
    print("%s: relative L2 error of the parameter update vs the oracle's: %.2e" % (name, upd_err))
    assert upd_err < (5e-2 if bf16 else 2e-3), upd_err      # (measured 1.7e-2 / 1.9e-4 ... 5.2e-4)

    # step scalars advanced by the optimiser kernel's last workgroup (step.hip: AdamTail)
    step1, b1p1, b2p1, decay1 = rep["scalars"]
    assert step1 == STEP0 + 1
    assert abs(b1p1 - float(opt.b1p)) <= 1e-7 and abs(b2p1 - float(opt.b2p)) <= 1e-7
    assert abs(decay1 - MO.bn_decay_schedule(STEP0 + 1, B)) < 1e-7
