"""GPU: a whole fully connected layer per launch (csrc/fc.hip, batches of <= 128 clouds) against the
oracle's fully_connected (oracle/model_oracle.py: matmul + bias + batch_norm + ReLU, autograd for the
gradients) on the same seeded inputs.  fp32 sums in a different order: compared to round-off."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(got, want):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    return np.abs(got - want).max() / (np.abs(want).max() + 1e-30)


def _oracle_layer(oracle, x, W, b, gamma, beta, sm, sv, training, relu, decay=0.9):
    MO = oracle
    V = MO.Vars(0)
    V.p["s/weights"], V.p["s/biases"] = W, b
    bn = gamma is not None
    if bn:
        V.p["s/bn/beta"], V.p["s/bn/gamma"] = beta, gamma
        V.s["s/bn/moments/Squeeze/ExponentialMovingAverage"] = sm
        V.s["s/bn/moments/Squeeze_1/ExponentialMovingAverage"] = sv
    return MO.fully_connected(x, W.shape[1], "s", V, bn=bn, is_training=training, bn_decay=decay, relu=relu)


@pytest.fixture(scope="module")
def model_oracle():
    from oracle import model_oracle as MO
    return MO


SHAPES = [(32, 1024, 1024, True), (32, 1024, 512, True), (32, 512, 256, True), (32, 256, 3, False),
          (32, 1024, 12288, False), (8, 1024, 1024, True), (1, 64, 40, False), (5, 100, 37, True),
          (31, 33, 130, True), (32, 1000, 516, False), (2, 8, 4, True),
          # more than one 32-row tile (33 / 64 / 128 rows; 128 = the per-GPU batch of BASELINE configs[3])
          (33, 1024, 1024, True), (64, 1024, 512, True), (128, 1024, 1024, True), (128, 1024, 512, True),
          (128, 512, 256, True), (128, 256, 3, False), (128, 1024, 12288, False), (64, 1024, 12288, False),
          (96, 520, 260, True), (100, 1000, 516, False), (65, 33, 130, True), (127, 100, 37, True)]


@pytest.mark.parametrize("M,K,N,bn", SHAPES)
@pytest.mark.parametrize("training", [True, False])
def test_fc_layer_vs_oracle(hip, model_oracle, M, K, N, bn, training):
    from cloudaae_amd.utils import _functions as F
    if M == 1 and bn and training:
        pytest.skip("variance of one row")
    g = torch.Generator().manual_seed(M * 7 + K + N)
    x = torch.randn(M, K, generator=g).requires_grad_(True)
    W = (torch.randn(K, N, generator=g) / np.sqrt(K)).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).requires_grad_(True)
    gamma = beta = sm = sv = None
    if bn:
        gamma = (torch.rand(N, generator=g) + 0.5).requires_grad_(True)
        beta = (torch.randn(N, generator=g) * 0.1).requires_grad_(True)
        sm, sv = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5
    up = torch.randn(M, N, generator=g)
    ref_sm, ref_sv = (sm.clone(), sv.clone()) if bn else (None, None)
    want = _oracle_layer(model_oracle, x, W, b, gamma, beta, ref_sm, ref_sv, training, bn)
    (want * up).sum().backward()

    def dev(t):
        return None if t is None else t.detach().cuda().requires_grad_(t.requires_grad)
    xd, Wd, bd, gd, betad = dev(x), dev(W), dev(b), dev(gamma), dev(beta)
    smd, svd = dev(sm), dev(sv)
    decay = torch.full((1,), 0.9, device="cuda") if bn else None
    out = F.FcFn.apply(xd, Wd, bd, gd, betad, smd, svd, decay, training, bn)
    (out * up.cuda()).sum().backward()
    assert _rel(out, want) < 3e-5
    assert _rel(xd.grad, x.grad) < 3e-4
    assert _rel(Wd.grad, W.grad) < 3e-4
    assert _rel(bd.grad, b.grad) < 3e-4 or (bn and training and b.grad.abs().max() < 1e-4)
    if bn:
        assert _rel(gd.grad, gamma.grad) < 3e-4 and _rel(betad.grad, beta.grad) < 3e-4
        if training:
            assert _rel(smd, ref_sm) < 1e-5 and _rel(svd, ref_sv) < 1e-5


@pytest.mark.parametrize("M,K,N", [(32, 96, 200), (128, 96, 200), (70, 64, 1300)])
def test_fc_backward_accumulates(hip, M, K, N):
    """dx is ADDED into what the buffer holds (several consumers of one input share it), dw and the
    per-column gradients add on request."""
    from cloudaae_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    x, W = torch.randn(M, K, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
    dout = torch.randn(M, N, generator=g).cuda()
    dx0, dw0, db0 = torch.randn(M, K, generator=g).cuda(), torch.randn(K, N, generator=g).cuda(), \
        torch.randn(N, generator=g).cuda()
    dx, dw, db = dx0.clone(), dw0.clone(), db0.clone()
    s = _lib.stream()
    _lib.check(L.cloudaae_fc_backward(M, K, N, x.data_ptr(), K, W.data_ptr(), None, None, None, None, None, 1, 0,
                                      dout.data_ptr(), N, dx.data_ptr(), K, dw.data_ptr(), 1, None, None,
                                      db.data_ptr(), 1, s), "fc_backward")
    torch.cuda.synchronize()
    xd, Wd, dd = x.double(), W.double(), dout.double()
    assert _rel(dx, dx0.double() + dd @ Wd.T) < 1e-5
    assert _rel(dw, dw0.double() + xd.T @ dd) < 1e-5
    assert _rel(db, db0.double() + dd.sum(0)) < 1e-5


def test_fc_rejects_large_batches(hip):
    from cloudaae_amd import _lib
    L = _lib.lib()
    assert L.cloudaae_fc_max_rows() == 128
    x, W, y = torch.zeros(129, 8).cuda(), torch.zeros(8, 8).cuda(), torch.zeros(129, 8).cuda()
    rc = L.cloudaae_fc_forward(129, 8, 8, x.data_ptr(), 8, W.data_ptr(), None, None, None, 0, None, None, None, None,
                               None, 0, y.data_ptr(), None, None, None, 0, _lib.stream())
    assert rc != 0 and "rows" in L.cloudaae_last_error().decode()
    # batch norm over several row tiles needs the counters and the scratch (the column statistics cross workgroups)
    g, b = torch.ones(8).cuda(), torch.zeros(8).cuda()
    x, y, out = torch.zeros(40, 8).cuda(), torch.zeros(40, 8).cuda(), torch.zeros(40, 8).cuda()
    rc = L.cloudaae_fc_forward(40, 8, 8, x.data_ptr(), 8, W.data_ptr(), None, g.data_ptr(), b.data_ptr(), 1, None, None,
                               None, g.data_ptr(), b.data_ptr(), 0, y.data_ptr(), out.data_ptr(), None, None, 0,
                               _lib.stream())
    assert rc != 0 and "tickets" in L.cloudaae_last_error().decode()


def test_fc_forward_without_tickets_keeps_k_whole(hip):
    """Batch norm over a product cut over K (arrival counters; slices summed in slice order through the
    partial-tile scratch) and with K whole in one workgroup (no counters) are the same layer; the counters are
    left at zero."""
    from cloudaae_amd import _lib
    L = _lib.lib()
    M, K, N = 32, 1024, 512
    g = torch.Generator().manual_seed(11)
    x, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(K, N, generator=g) / 32).cuda()
    b, gamma, beta = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda() + 0.5, \
        torch.randn(N, generator=g).cuda()
    decay = torch.full((1,), 0.9, device="cuda")
    res = []
    nparts = int(L.cloudaae_fc_forward_partials(M, K, N, 1))
    assert nparts > 0
    for use in (True, False):
        tk = torch.zeros(L.cloudaae_fc_forward_tickets(M, N), dtype=torch.int32, device="cuda") if use else None
        parts = torch.full((nparts,), float("nan"), device="cuda") if use else None
        y, out = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        sm, sv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        mean, var = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        for _ in range(3):      # repeated launches reuse the counters
            _lib.check(L.cloudaae_fc_forward(M, K, N, x.data_ptr(), K, W.data_ptr(), b.data_ptr(), gamma.data_ptr(),
                                             beta.data_ptr(), 1, decay.data_ptr(), sm.data_ptr(), sv.data_ptr(),
                                             mean.data_ptr(), var.data_ptr(), 1, y.data_ptr(), out.data_ptr(),
                                             None if tk is None else tk.data_ptr(),
                                             None if parts is None else parts.data_ptr(),
                                             0 if parts is None else parts.numel(), _lib.stream()), "fc_forward")
        torch.cuda.synchronize()
        if use:
            assert int(tk.abs().sum()) == 0
            res_tk, res_parts = tk, parts
        res.append((y, out, mean, var, sm))
    # the partial-tile scratch travels with its size: a launch whose cut needs more is refused (nothing is launched)
    rc = L.cloudaae_fc_forward(M, K, N, x.data_ptr(), K, W.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1,
                               decay.data_ptr(), sm.data_ptr(), sv.data_ptr(), mean.data_ptr(), var.data_ptr(), 1,
                               y.data_ptr(), out.data_ptr(), res_tk.data_ptr(), res_parts.data_ptr(), nparts - 1, _lib.stream())
    assert rc != 0 and "partials_floats" in L.cloudaae_last_error().decode()
    for other in res[1:]:
        for a, c in zip(res[0], other):
            assert _rel(a, c) < 1e-5
    want = x.double() @ W.double() + b.double()
    assert _rel(res[0][0], want) < 1e-5


@pytest.mark.parametrize("B", [32, 128, 48])
def test_fc_chains_vs_oracle(hip, model_oracle, B):
    """Decoder + two pose heads over one embedding (three chains, three grouped launches per direction)
    against the oracle's layer-by-layer evaluation: outputs, every parameter gradient, and the summed
    gradient of the shared input."""
    from cloudaae_amd.utils import tf_util
    from cloudaae_amd.utils.variables import VariableStore, set_default_store
    MO = model_oracle
    E, P = 1024, 3 * 4 * 64
    chains = [[('d_fc1', 1024, True), ('d_fc2', 1024, True), ('d_out', P, False)],
              [('r_fc1', 512, True), ('r_fc2', 256, True), ('r_out', 3, False)],
              [('t_fc1', 512, True), ('t_fc2', 256, True), ('t_out', 3, False)]]
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(B, E, generator=g)
    ups = [torch.randn(B, c[-1][1], generator=g) for c in chains]

    store = VariableStore(device="cuda", seed=1)
    set_default_store(store)
    xd = emb.cuda().requires_grad_(True)
    outs = tf_util.fully_connected_chains(xd, chains, bn_decay=0.9, is_training=True)
    sum((o * u.cuda()).sum() for o, u in zip(outs, ups)).backward()

    V = MO.Vars(0)
    for name, var in store.vars.items():
        if var.trainable:
            V.p[name] = var.data.detach().cpu().clone().requires_grad_(True)
    x = emb.clone().requires_grad_(True)
    want = []
    for chain in chains:
        net = x
        for scope, n, bn in chain:
            net = MO.fully_connected(net, n, scope, V, bn=bn, is_training=True, bn_decay=0.9, relu=bn)
        want.append(net)
    sum((o * u).sum() for o, u in zip(want, ups)).backward()
    for o, w in zip(outs, want):
        assert _rel(o, w) < 1e-4
    assert _rel(xd.grad, x.grad) < 1e-3
    for name, p in V.p.items():
        got = store.vars[name].data.grad
        assert got is not None, name
        assert _rel(got, p.grad) < 2e-3 or p.grad.abs().max() < 1e-5, name


@pytest.mark.parametrize("M", [32, 128])
def test_fc_forward_ticket_stress(hip, M):
    """The arrival-counter protocol of the batch-norm forward (K slices -- and row tiles -- publish their partial
    tiles at agent scope, the last one to arrive reads them back: no fence) over many launches with other work in
    between: every launch must reproduce the reference result, leave the counters at zero and give the SAME BITS."""
    from cloudaae_amd import _lib
    L = _lib.lib()
    K, N = 1024, 1024
    fixed_order = True
    g = torch.Generator().manual_seed(21)
    x, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(K, N, generator=g) / 32).cuda()
    b, gamma, beta = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda() + 0.5, \
        torch.randn(N, generator=g).cuda()
    decay = torch.full((1,), 0.9, device="cuda")
    sm, sv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    mean, var = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    tk = torch.zeros(L.cloudaae_fc_forward_tickets(M, N), dtype=torch.int32, device="cuda")
    parts = torch.full((int(L.cloudaae_fc_forward_partials(M, K, N, 1)),), float("nan"), device="cuda")

    def run(tickets, y, out):
        _lib.check(L.cloudaae_fc_forward(M, K, N, x.data_ptr(), K, W.data_ptr(), b.data_ptr(), gamma.data_ptr(),
                                         beta.data_ptr(), 1, decay.data_ptr(), sm.data_ptr(), sv.data_ptr(),
                                         mean.data_ptr(), var.data_ptr(), 1, y.data_ptr(), out.data_ptr(),
                                         tickets, parts.data_ptr() if tickets else None,
                                         parts.numel(), _lib.stream()), "fc_forward")
    # the reference result: float64 on the device
    y0 = (x.double() @ W.double() + b.double())
    mu, vr = y0.mean(0), y0.var(0, unbiased=False)
    out0 = torch.relu((y0 - mu) / torch.sqrt(vr + 1e-3) * gamma.double() + beta.double()).float()
    y0 = y0.float()
    junk = torch.randn(1 << 22, device="cuda")
    worst, first, same = 0.0, None, True
    for it in range(200):
        y, out = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        if it % 3 == 0:
            junk.mul_(1.0001)                       # dirty lines in the L2s between launches
        if it % 7 == 0:
            parts.fill_(float("nan"))               # stale partial tiles must never be read
        run(tk.data_ptr(), y, out)
        worst = max(worst, _rel(out, out0), _rel(y, y0))
        if first is None:
            first = (y, out)
        else:
            same = same and torch.equal(y, first[0]) and torch.equal(out, first[1])
    torch.cuda.synchronize()
    assert worst < 3e-5 and int(tk.abs().sum()) == 0
    assert same or not fixed_order, "fixed-order slices did not reproduce bit for bit"


@pytest.mark.parametrize("M,K,N,d", [(32, 1024, 12288, 3), (5, 256, 3, 3), (32, 512, 100, 5), (128, 1024, 12288, 3),
                                     (77, 256, 3, 3)])
def test_fc_forward_adds_a_row_vector(hip, M, K, N, d):
    """out_rowvec of cloudaae_fc_layer: y[r][c] = (x w + b)[r][c] + vec[r][c % d] -- the "+ element_mean" of
    train_cloudAAE_ycbv.py:232-233 in the output layer's epilogue; equal to a separate cloudaae_add_rowvec pass bit for
    bit (same order of additions)."""
    from cloudaae_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + N)
    x, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(K, N, generator=g) / 32).cuda()
    b, vec = torch.randn(N, generator=g).cuda(), torch.randn(M, d, generator=g).cuda()
    nparts = int(L.cloudaae_fc_forward_partials(M, K, N, 0))
    tk = torch.zeros(L.cloudaae_fc_forward_tickets(M, N), dtype=torch.int32, device="cuda")
    parts = torch.empty(max(nparts, 1), device="cuda")
    ys = []
    for with_vec in (False, True):
        layer = (_lib.FcLayer * 1)()
        l = layer[0]
        y = torch.full((M, N), float("nan"), device="cuda")
        l.K, l.N, l.x, l.ldx, l.w, l.bias, l.y = K, N, x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr()
        l.tickets, l.partials, l.partials_floats = tk.data_ptr(), parts.data_ptr() if nparts else None, nparts
        if with_vec:
            l.out_rowvec, l.out_rowvec_d = vec.data_ptr(), d
        _lib.check(L.cloudaae_fc_forward_group(M, 1, layer, 1, None, _lib.stream()), "fc_forward_group")
        ys.append(y)
    want = torch.empty(M, N // d if N % d == 0 else 1, d, device="cuda")
    if N % d == 0:
        _lib.check(L.cloudaae_add_rowvec(M, N // d, d, ys[0].data_ptr(), vec.data_ptr(), want.data_ptr(), _lib.stream()),
                   "add_rowvec")
        assert torch.equal(ys[1], want.view(M, N))
    assert torch.equal(ys[1], ys[0] + vec.repeat(1, (N + d - 1) // d)[:, :N])


@pytest.mark.parametrize("M,K,N,bn", [(32, 1024, 1024, True), (7, 1024, 512, True), (32, 256, 3, False),
                                      (32, 1024, 12288, False), (19, 520, 260, True), (32, 1024, 1000, False),
                                      (128, 1024, 1024, True), (128, 1024, 12288, False), (50, 520, 260, True)])
def test_fc_forward_is_bit_reproducible(hip, M, K, N, bn):
    """With the partial-tile scratch a forward layer gives the same bits launch after launch (north star: the
    reference's CPU path is sequential, tf_nndistance.cpp:21-43 -- and evaluate_cloudAAE_ycbv.py:421-477 returns
    the same reconstruction for the same frame), and agrees with float64 to fp32 round-off."""
    from cloudaae_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + K + N)
    x, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(K, N, generator=g) / 32).cuda()
    b = torch.randn(N, generator=g).cuda()
    gamma, beta = (torch.rand(N, generator=g).cuda() + 0.5, torch.randn(N, generator=g).cuda()) if bn else (None, None)
    decay = torch.full((1,), 0.9, device="cuda")
    P = lambda t: None if t is None else t.data_ptr()      # noqa: E731
    nparts = int(L.cloudaae_fc_forward_partials(M, K, N, int(bn)))
    tk = torch.zeros(L.cloudaae_fc_forward_tickets(M, N), dtype=torch.int32, device="cuda") if nparts else None
    parts = torch.full((max(nparts, 1),), float("nan"), device="cuda") if nparts else None
    runs = []
    for it in range(20):
        sm, sv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        mean, var = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        y = torch.full((M, N), float("nan"), device="cuda")            # y need not be cleared in fixed order
        out = torch.empty(M, N, device="cuda") if bn else None
        _lib.check(L.cloudaae_fc_forward(M, K, N, P(x), K, P(W), P(b), P(gamma), P(beta), 1, P(decay), P(sm) if bn else None,
                                         P(sv) if bn else None, P(mean) if bn else None, P(var) if bn else None, 1, P(y),
                                         P(out), P(tk), P(parts), nparts, _lib.stream()), "fc_forward")
        runs.append((y, out, mean if bn else None))
    torch.cuda.synchronize()
    for y, out, mean in runs[1:]:
        assert torch.equal(y, runs[0][0])
        if bn:
            assert torch.equal(out, runs[0][1]) and torch.equal(mean, runs[0][2])
    want = x.double() @ W.double() + b.double()
    assert _rel(runs[0][0], want) < 1e-5
    if tk is not None:
        assert int(tk.abs().sum()) == 0
