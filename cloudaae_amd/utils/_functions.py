"""torch.autograd plumbing around the C-ABI kernels (no arithmetic happens here).

Parameter gradients: a parameter tensor that belongs to a VariableStore carries its
Variable in `_cloudaae_var`; its gradient is written by the kernel straight into the
store's flat gradient buffer (first write of a backward pass stores, later ones add)
and `None` is returned to autograd.  Foreign tensors get ordinary returned gradients.
"""
import ctypes
import os

import torch

from .. import _lib
from .._lib import ptr, rows_ptr, require, stream

L = _lib.lib


def _ws(nbytes, device):
    return _lib.empty((int(nbytes) + 7) // 8, dtype=torch.float64, device=device)


def _var(t):
    return getattr(t, "_cloudaae_var", None)


class _ParamGrad(object):
    """Where the gradient of one parameter goes."""

    def __init__(self, param, needed):
        self.var = _var(param) if needed else None
        self.needed = needed
        self.own = None
        if not needed:
            self.buf, self.accumulate = None, 0
        elif self.var is not None and self.var.grad is not None:
            self.buf = self.var.grad
            self.accumulate = 0 if self.var.fresh else 1
        else:
            self.own = _lib.empty(param.shape, dtype=torch.float32, device=param.device)
            self.buf, self.accumulate = self.own, 0

    @property
    def gemm_acc(self):
        """accumulate flag for cloudaae_gemm_f32: 2 = first write of this step into a gradient buffer
        that begin_step() cleared as a whole (no per-product clear pass for split-K)."""
        if self.var is not None and self.var.grad is not None and self.var.fresh and self.var.zeroed:
            return 2
        return self.accumulate

    def done(self):
        if self.var is not None and self.var.grad is not None:
            self.var.fresh = False
            if self.var.on_ready is not None:
                _lib.host(self.var.on_ready)
            return None
        return self.own


# bench.py times individual launches: TIMED_SITES[name] = [] switches a site on; each launch of
# that site then appends a start and an end HIP event (recorded on the launch stream) to the list
# -- also on every replay of a recorded step, where the two records are host callbacks of the plan.
TIMED_SITES = {}
KNN64_SEEN = 0         # launches of the kNN over 64 channels seen while the "knn64" site is on


TIMED_ON = True        # bench.py switches the sites on for one step in four (an event pair costs the stream ~5 us)


SITES_OFF = set()      # names of sites whose (recorded) event pairs are skipped for now


def _mark(rec, site=None):
    if not TIMED_ON or (site is not None and site in SITES_OFF):
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    rec.append(e)


# Operand type of the per-point dense layers (conv2d 1x1, the edge convolution, dgcnn_agg): "f32", or
# "bf16" = operands rounded to bfloat16 on their way to the matrix cores, fp32 accumulate (BASELINE
# configs[2]).  The fully connected stack (rows = batch size) stays fp32: it is bound by reading its
# fp32 weights, not by the matrix pipe, and bf16 staging only adds work there (measured: 12.9 vs
# 7.9 us per 32 x 1024 x 1024 product).  Read when a layer's FORWARD runs; its backward follows suit.
GEMM_DTYPE = "f32"


# SyncBN (utils/sync_bn.BnSync) or None: read when a batch-norm layer's FORWARD runs in training mode; its
# backward follows suit.  With it set, moments and backward means are over the batch of ALL ranks.
BN_SYNC = None


# With bf16 dense-layer operands (GEMM_DTYPE "bf16"): keep the widest activations of the step -- the concatenated
# encoder features, dgcnn_agg's output y and its gradient -- in HBM as bfloat16 (csrc/gemm_b16.hip, bn16.hip).  The
# products see the values they would have rounded to anyway; the one new rounding point is the stored y.  Off in
# deterministic mode and under SyncBN (those take the fp32-storage kernels).
ACT_BF16 = os.environ.get("CLOUDAAE_ACT_BF16", "1") != "0"
# ... and the edge-conv layers store the bfloat16 copy of the concat themselves (0: one conversion pass in front of the product)
CONCAT_BF16 = os.environ.get("CLOUDAAE_CONCAT_BF16", "1") != "0"


def to_bf16(t):
    """bfloat16 copy (round to nearest even) of a contiguous fp32 tensor whose element count is a multiple of 8."""
    out = _lib.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    _lib.check(L().cloudaae_to_bf16(t.numel(), t.data_ptr(), out.data_ptr(), stream()), "cloudaae_to_bf16")
    return out


def gemm_is_bf16():
    require(GEMM_DTYPE in ("f32", "bf16", "bf16x3"), "GEMM_DTYPE must be 'f32', 'bf16' or 'bf16x3'")
    return GEMM_DTYPE == "bf16"


_X3_FALLBACKS = set()


def _note_x3_fallback(M, N, K):
    """gemm_dtype='bf16x3' is a request: a shape the split-product kernels do not serve (rows not a multiple of their
    tiles, unaligned operands) runs on the fp32 matrix cores instead -- same accuracy class, but not what the run's label
    says.  Said once per shape."""
    if (M, N, K) not in _X3_FALLBACKS:
        _X3_FALLBACKS.add((M, N, K))
        import warnings
        warnings.warn("cloudaae_amd: gemm_dtype='bf16x3' not available for the dgcnn_agg product [%d x %d] x [%d x %d]: "
                      "using the fp32 matrix cores (cloudaae_gemm_f32)" % (M, K, K, N), RuntimeWarning)


def gemm_is_x3():
    """GEMM_DTYPE "bf16x3" (TrainGraph's default): fp32 everywhere, but the three dgcnn_agg products run as split products on the bf16
    matrix cores (csrc/gemm_x3.hip: every operand element = three bfloat16 pieces, six piece products, fp32 accumulate --
    the accuracy of an fp32 product at 2.7 x less matrix-pipe time).  Off in deterministic mode."""
    return GEMM_DTYPE == "bf16x3" and not DETERMINISTIC


# Work nobody waits for until the optimiser runs (the weight-gradient products of the per-point layers) and
# the reverse neighbour lists go to SIDE_STREAM when it is set (a raw hipStream_t, _lib.side_stream()): they
# fill the CUs the critical path leaves idle.  Whoever sets it joins -- _lib.stream_wait(stream(), SIDE_STREAM)
# -- before consuming gradients (TrainGraph does, around backward); None = everything on the current stream.
SIDE_STREAM = None
SIDE_EDGE = True      # edge-convolution layers use it too (else only the dgcnn_agg weight gradient)
SIDE_AGG = True       # the dgcnn_agg weight gradient uses it


def gemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, accumulate=0, site=None, bf16=False, on=None, device=None):
    rec = TIMED_SITES.get(site) if site is not None else None
    if rec is not None:
        _lib.host(_mark, rec)
    if DETERMINISTIC and int(accumulate) != 1 and on is None and device is not None:
        # deterministic mode: a product cut over K keeps its slices apart and sums them in slice order
        # (cloudaae_gemm_*_ordered) instead of staying whole on a handful of CUs
        wsq = L().cloudaae_gemm_bf16_ordered_workspace if bf16 else L().cloudaae_gemm_f32_ordered_workspace
        n = int(wsq(M, N, K))
        ws = _lib.empty(n, dtype=torch.float32, device=device) if n else None
        fn = L().cloudaae_gemm_bf16_ordered if bf16 else L().cloudaae_gemm_f32_ordered
        _lib.check(fn(int(ta), int(tb), M, N, K, A, lda, B, ldb, C, ldc, bias, ptr(ws), n, stream()), "cloudaae_gemm_ordered")
    else:
        fn = L().cloudaae_gemm_bf16 if bf16 else L().cloudaae_gemm_f32
        _lib.check(fn(int(ta), int(tb), M, N, K, A, lda, B, ldb, C, ldc, bias, int(accumulate),
                      stream() if on is None else on),
                   "cloudaae_gemm_bf16" if bf16 else "cloudaae_gemm_f32")
    if rec is not None:
        _lib.host(_mark, rec)


def gemm_forward(M, N, K, A, lda, B, ldb, C, ldc, bias, device, site=None, bf16=False):
    """C = A B + bias for a FORWARD product: bit-reproducible from run to run (cloudaae_gemm_*_ordered: a product
    cut over K keeps its slices apart and sums them in slice order instead of adding them with atomics)."""
    rec = TIMED_SITES.get(site) if site is not None else None
    if rec is not None:
        _lib.host(_mark, rec)
    wsq = L().cloudaae_gemm_bf16_ordered_workspace if bf16 else L().cloudaae_gemm_f32_ordered_workspace
    n = int(wsq(M, N, K))
    ws = _lib.empty(n, dtype=torch.float32, device=device) if n else None
    fn = L().cloudaae_gemm_bf16_ordered if bf16 else L().cloudaae_gemm_f32_ordered
    _lib.check(fn(0, 0, M, N, K, A, lda, B, ldb, C, ldc, bias, ptr(ws), n, stream()),
               "cloudaae_gemm_bf16_ordered" if bf16 else "cloudaae_gemm_f32_ordered")
    if rec is not None:
        _lib.host(_mark, rec)


def _gemm_out(shape, K, device, bf16=False):
    """Output buffer of a product and the accumulate flag to pass: when the product is split over K
    while a step is being recorded, the buffer comes from the plan's zero zone (cleared by one fill
    per replay) and the product skips its own clear pass."""
    if _lib.recording() is not None and _lib.gemm_splits(shape[0], shape[1], K, bf16) > 1:
        return _lib.zeros(shape, dtype=torch.float32, device=device), 2
    return _lib.empty(shape, dtype=torch.float32, device=device), 0


class LinearFn(torch.autograd.Function):
    """y[M,N] = x[M,K] W[K,N] + b  (tf.matmul/conv2d-1x1 + bias_add)."""

    @staticmethod
    def forward(ctx, x, w, b, bias_grad_by_bn=False, allow_bf16=True):
        # bias_grad_by_bn: a BatchNormFn consumes y directly and writes d(b) itself (column sums of
        # its dy come out of its own reductions), so backward here skips the extra pass over dy
        require(x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[0], "LinearFn: shape mismatch")
        xp, ldx = rows_ptr(x)
        M, K = x.shape
        N = w.shape[1]
        ctx.bf16 = gemm_is_bf16() and bool(allow_bf16)
        y = _lib.empty((M, N), dtype=torch.float32, device=x.device)
        gemm_forward(M, N, K, xp, ldx, ptr(w), N, ptr(y), N, ptr(b) if b is not None else None, x.device, bf16=ctx.bf16)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None and not bias_grad_by_bn
        ctx.bvar = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[1]
        dy = dy.contiguous() if dy.stride(-1) != 1 else dy
        dyp, lddy = rows_ptr(dy)
        xp, ldx = rows_ptr(x)
        dx = None
        if ctx.needs_input_grad[0]:
            dx, acc = _gemm_out((M, K), N, x.device, ctx.bf16)
            gemm(0, 1, M, K, N, dyp, lddy, ptr(w), N, ptr(dx), K, None, acc, bf16=ctx.bf16, device=x.device)
        gw = _ParamGrad(w, ctx.needs_input_grad[1])
        if gw.needed:
            gemm(1, 0, K, N, M, xp, ldx, dyp, lddy, ptr(gw.buf), N, None, gw.gemm_acc, bf16=ctx.bf16, device=x.device)
        gb_ret = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _ParamGrad(ctx.bvar, True)
            ws = _ws(L().cloudaae_bn_workspace_bytes(N), x.device)
            _lib.check(L().cloudaae_colsum_f32(M, N, dyp, lddy, ptr(gb.buf), gb.accumulate, ptr(ws), stream()),
                       "cloudaae_colsum_f32")
            gb_ret = gb.done()
        return dx, gw.done(), gb_ret, None, None


# Deterministic mode (TrainGraph(deterministic=True) sets it together with the library's CLOUDAAE_DETERMINISTIC knob): the
# backward pass gives up its fp32 atomics -- the fully connected stack takes the GEMM + batch-norm route (the grouped
# kernels add their dX slices with atomics), the Chamfer gradient is accumulated in the reference's CPU order
# (cloudaae_nn_distance_grad_ordered) -- and, every product staying whole over K and the reverse neighbour lists being
# sorted on the library's side, a training step is bit-reproducible from run to run, as the reference's CPU path is.
DETERMINISTIC = False


# An offer to the next fully_connected_chains call: a row vector [B, d]; the chains the MODEL declares as point
# outputs (tf_util.fully_connected_chains(point_outputs={chain: d})) get out[b, c] += vec[b, c % d] --
# train_cloudAAE_ycbv.py:232-233's "+ element_mean" folded into the output layers' epilogue.  The call that takes the
# offer sets this back to None (TrainGraph.forward looks at it afterwards); a model that declares no point outputs, or
# whose decoder emits vectors of another width (the *_hand decoder: 5), leaves it untouched.
FC_OUT_ADD = None


def fc_fits(M):
    """Rows of a fully connected layer that the one-launch-per-direction kernels take."""
    return 0 < M <= L().cloudaae_fc_max_rows() and not DETERMINISTIC


def fc_max_group():
    return int(L().cloudaae_fc_max_group())


def _fc_scratch(M, K, N, bn, dev):
    """(tickets, partials) of one forward layer of csrc/fc.hip: arrival counters of its column / row tiles (zero before
    and after every launch) and room for the partial tiles of its K slices and row tiles, which the last to arrive sums
    in a fixed order -- the forward pass is bit-reproducible from run to run.  (None, None) when one workgroup per
    column tile finishes the layer."""
    n = int(L().cloudaae_fc_forward_partials(int(M), int(K), int(N), int(bool(bn))))
    if n == 0:
        return None, None
    return (_lib.zeros(L().cloudaae_fc_forward_tickets(int(M), int(N)), dtype=torch.int32, device=dev),
            _lib.empty(n, dtype=torch.float32, device=dev))


class FcFn(torch.autograd.Function):
    """tf_util.fully_connected as a whole (utils/tf_util.py:321-365): matmul + bias [+ batch norm + ReLU]
    for a batch of at most 128 rows, one launch forward and one backward (csrc/fc.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, ema_mean, ema_var, decay, training, relu):
        require(x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[0], "FcFn: shape mismatch")
        ctx.set_materialize_grads(False)
        xp, ldx = rows_ptr(x)
        M, K = x.shape
        N = w.shape[1]
        dev = x.device
        bn = gamma is not None
        save_mean = save_var = out = None
        if bn:
            out = _lib.empty((M, N), dtype=torch.float32, device=dev)
            save_mean = _lib.empty(N, dtype=torch.float32, device=dev)
            save_var = _lib.empty(N, dtype=torch.float32, device=dev)
        # a product cut over K is summed in slice order by its last slice (bit-reproducible forward pass):
        # arrival counters of the column tiles (zero before and after every launch) + the slices' partial tiles
        tickets, partials = _fc_scratch(M, K, N, bn, dev)
        y = _lib.empty((M, N), dtype=torch.float32, device=dev)
        _lib.check(L().cloudaae_fc_forward(
            M, K, N, xp, ldx, ptr(w), ptr(b), ptr(gamma), ptr(beta), int(bool(training)), ptr(decay), ptr(ema_mean),
            ptr(ema_var), ptr(save_mean), ptr(save_var), int(bool(relu)), ptr(y), ptr(out), ptr(tickets),
            ptr(partials), 0 if partials is None else partials.numel(), stream()), "cloudaae_fc_forward")
        ctx.save_for_backward(x, w, y if bn else None, gamma, beta, save_mean, save_var)
        ctx.cfg = (int(bool(training)), int(bool(relu)))
        ctx.bvar = b
        return out if bn else y

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 10
        x, w, y, gamma, beta, save_mean, save_var = ctx.saved_tensors
        training, relu = ctx.cfg
        M, K = x.shape
        N = w.shape[1]
        dout = dout.contiguous() if dout.stride(-1) != 1 else dout
        dop, lddo = rows_ptr(dout)
        xp, ldx = rows_ptr(x)
        dx = _lib.zeros((M, K), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[0] else None
        gw = _ParamGrad(w, ctx.needs_input_grad[1])
        gbias = _ParamGrad(ctx.bvar, ctx.bvar is not None and ctx.needs_input_grad[2])
        gg = _ParamGrad(gamma, gamma is not None and ctx.needs_input_grad[3])
        gbeta = _ParamGrad(beta, beta is not None and ctx.needs_input_grad[4])
        grads = [g for g in (gbias, gg, gbeta) if g.needed]
        acc = 1 if any(g.accumulate for g in grads) else 0
        if acc:                                             # mixed freshness: zero the fresh ones
            for g in grads:
                if not g.accumulate:
                    g.buf.zero_()
        _lib.check(L().cloudaae_fc_backward(
            M, K, N, xp, ldx, ptr(w), ptr(y), ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_var), training, relu,
            dop, lddo, ptr(dx), K, ptr(gw.buf), int(gw.accumulate), ptr(gg.buf), ptr(gbeta.buf), ptr(gbias.buf),
            acc, stream()), "cloudaae_fc_backward")
        return (dx, gw.done() if gw.needed else None, gbias.done() if gbias.needed else None,
                gg.done() if gg.needed else None, gbeta.done() if gbeta.needed else None, None, None, None, None,
                None)


class FcGroupFn(torch.autograd.Function):
    """Several independent fully connected layers of one batch (<= 128 rows) in ONE launch per direction
    (cloudaae_fc_forward_group / _backward_group): depth by depth, the decoder and the two pose heads
    (models/pointnet_ycb_23_decoder_4.py:413-455).

    apply(cfg, decay, *tensors): cfg = (n_inputs, x_index per layer, training, relu per layer);
    tensors = the n_inputs input matrices, then per layer (w, b, gamma, beta, ema_mean, ema_var) with
    gamma .. ema_var None for a layer without batch norm.  Returns one output per layer.  Layers that
    read the same input add their input gradients into one buffer (no separate sum)."""

    @staticmethod
    def forward(ctx, cfg, decay, *tensors):
        n_in, x_index, training, relus = cfg[:4]
        rowvecs = cfg[4] if len(cfg) > 4 else (None,) * len(x_index)     # per layer: None or a [M, d] tensor added to y
        ctx.set_materialize_grads(False)
        xs = tensors[:n_in]
        per = [tensors[n_in + 6 * i:n_in + 6 * i + 6] for i in range(len(x_index))]
        dev = xs[0].device
        M = xs[0].shape[0]
        layers = (_lib.FcLayer * len(per))()
        outs, keep = [], []
        for i, (w, b, gamma, beta, ema_mean, ema_var) in enumerate(per):
            x = xs[x_index[i]]
            require(x.dim() == 2 and x.shape[0] == M and x.shape[1] == w.shape[0], "FcGroupFn: shape mismatch")
            xp, ldx = rows_ptr(x)
            K, N = w.shape
            bn = gamma is not None
            y = _lib.empty((M, N), dtype=torch.float32, device=dev)
            out = save_mean = save_var = None
            if bn:
                out = _lib.empty((M, N), dtype=torch.float32, device=dev)
                save_mean = _lib.empty(N, dtype=torch.float32, device=dev)
                save_var = _lib.empty(N, dtype=torch.float32, device=dev)
            # a product cut over K is summed in slice order by its last slice (see _fc_scratch)
            tickets, partials = _fc_scratch(M, K, N, bn, dev)
            l = layers[i]
            l.K, l.N, l.x, l.ldx, l.w, l.bias = K, N, xp, ldx, ptr(w), ptr(b)
            l.gamma, l.beta, l.ema_mean, l.ema_var = ptr(gamma), ptr(beta), ptr(ema_mean), ptr(ema_var)
            l.save_mean, l.save_var, l.relu = ptr(save_mean), ptr(save_var), int(bool(relus[i]))
            l.y, l.out, l.tickets, l.partials = ptr(y), ptr(out), ptr(tickets), ptr(partials)
            l.partials_floats = 0 if partials is None else partials.numel()
            if rowvecs[i] is not None:
                require(not bn and rowvecs[i].dim() == 2 and rowvecs[i].shape[0] == M, "FcGroupFn: bad row vector")
                l.out_rowvec, l.out_rowvec_d = ptr(rowvecs[i]), int(rowvecs[i].shape[1])
            outs.append(out if bn else y)
            keep.append((y if bn else None, save_mean, save_var, tickets, partials))
        _lib.check(L().cloudaae_fc_forward_group(M, len(per), layers, int(bool(training)), ptr(decay), stream()),
                   "cloudaae_fc_forward_group")
        ctx.cfg, ctx.xs, ctx.per, ctx.keep, ctx.fwd_layers = cfg, xs, per, keep, layers
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        n_in, x_index, training, relus = ctx.cfg[:4]
        xs, per, keep = ctx.xs, ctx.per, ctx.keep
        M = xs[0].shape[0]
        live = [i for i, d in enumerate(douts) if d is not None]
        grads = [None] * (2 + n_in + 6 * len(per))
        # a layer nobody took a gradient of: its slots in the flat gradient buffer are not cleared at the
        # start of a step (see VariableStore.begin_step: zero_limit), so they are cleared here
        for i in range(len(per)):
            if i not in live:
                for t in per[i][:4]:
                    v = _var(t) if t is not None else None
                    if v is not None and v.grad is not None and v.fresh:
                        v.grad.zero_()
        if not live:
            return tuple(grads)
        dxs = [None] * n_in
        for j in range(n_in):
            if ctx.needs_input_grad[2 + j] and any(x_index[i] == j for i in live):
                dxs[j] = _lib.zeros(xs[j].shape, dtype=torch.float32, device=xs[j].device)
        layers = (_lib.FcLayer * len(live))()
        hold, done = [], []
        for n, i in enumerate(live):
            w, b, gamma, beta, _, _ = per[i]
            y, save_mean, save_var = keep[i][:3]
            x = xs[x_index[i]]
            dout = douts[i]
            dout = dout.contiguous() if dout.stride(-1) != 1 else dout
            dop, lddo = rows_ptr(dout)
            xp, ldx = rows_ptr(x)
            base = 2 + n_in + 6 * i
            gw = _ParamGrad(w, ctx.needs_input_grad[base])
            gbias = _ParamGrad(b, b is not None and ctx.needs_input_grad[base + 1])
            gg = _ParamGrad(gamma, gamma is not None and ctx.needs_input_grad[base + 2])
            gbeta = _ParamGrad(beta, beta is not None and ctx.needs_input_grad[base + 3])
            col = [g for g in (gbias, gg, gbeta) if g.needed]
            acc = 1 if any(g.accumulate for g in col) else 0
            if acc:                                             # mixed freshness: zero the fresh ones
                for g in col:
                    if not g.accumulate:
                        g.buf.zero_()
            l = layers[n]
            l.K, l.N, l.x, l.ldx, l.w = w.shape[0], w.shape[1], xp, ldx, ptr(w)
            l.gamma, l.beta, l.save_mean, l.save_var = ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_var)
            l.relu, l.y = int(bool(relus[i])), ptr(y)
            l.dout, l.lddo = dop, lddo
            dx = dxs[x_index[i]]
            l.dx, l.lddx = ptr(dx), (x.shape[1] if dx is not None else 0)
            l.dw, l.accumulate_dw = ptr(gw.buf), int(gw.accumulate)
            l.dgamma, l.dbeta, l.dbias, l.accumulate_param_grads = ptr(gg.buf), ptr(gbeta.buf), ptr(gbias.buf), acc
            hold.append(dout)
            done.append((base, gw, gbias, gg, gbeta))
        _lib.check(L().cloudaae_fc_backward_group(M, len(live), layers, int(bool(training)), stream()),
                   "cloudaae_fc_backward_group")
        ctx.bwd_layers = layers
        for base, gw, gbias, gg, gbeta in done:
            for k, g in enumerate((gw, gbias, gg, gbeta)):
                grads[base + k] = g.done() if g.needed else None
        for j in range(n_in):
            grads[2 + j] = dxs[j]
        return tuple(grads)


# Hand-off from a product that computed the column sums of its output to the batch norm that consumes it:
# the product's output tensor y carries `_cloudaae_colstats` = (workspace holding the sums, number of tile
# rows, M, N); BatchNormFn.forward uses it only for that very tensor (same object, same shape) in training
# mode.  Nothing global: a batch norm that is skipped leaves nothing behind for a later one to pick up.


class ConcatSlot(object):
    """The one [B,N,Ctot] buffer the encoder's layer outputs are written into as column slices
    (the tf.concat of models/...:410 without a copy) and, in backward, its gradient twin: the agg
    GEMM deposits d(concat) here, and each edge-conv layer whose INPUT is a slice accumulates its dx
    straight into that slice -- the sum autograd would otherwise make with an extra add kernel per
    layer (and which a recorded step could not replay)."""

    def __init__(self, buf):
        self.buf = buf
        self.dcat = None
        # bfloat16 twin of buf (activations kept in bfloat16, BASELINE configs[2]): each edge-conv layer stores its slice
        # there too, and the aggregation product reads it instead of converting the whole concat (cols16: columns written)
        self.buf16 = None
        self.cols16 = 0
        # (nn_idx, reverse-list buffer) of every edge-conv layer writing into this slot: the first backward
        # call builds all their reverse neighbour lists with one launch
        self.revs = []
        self.revs_built = False
        # weight-gradient products of those layers, deferred to ONE grouped launch (cloudaae_gemm_f32_tn_group) when the
        # last of them has run its backward pass: (job fields, tensors kept alive, the parameter's _ParamGrad)
        self.dw_jobs = []


_PENDING_DW = []        # ConcatSlots holding deferred weight-gradient products


def flush_deferred_dw(slot=None):
    """Launch the deferred weight-gradient products of `slot` (or of every slot that still holds some: the end of a
    backward pass in which not every layer took part) as one grouped launch."""
    slots = [slot] if slot is not None else list(_PENDING_DW)
    for sl in slots:
        if sl in _PENDING_DW:
            _PENDING_DW.remove(sl)
        jobs, sl.dw_jobs = sl.dw_jobs, []
        if not jobs:
            continue
        arr = (_lib.GemmTnJob * len(jobs))()
        for a, (f, _, _) in zip(arr, jobs):
            a.M, a.N, a.K, a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.fold_c, a.zeroed = f
        _lib.check(L().cloudaae_gemm_f32_tn_group(len(jobs), arr, stream()), "cloudaae_gemm_f32_tn_group")
        for _, _, gw in jobs:
            gw.done()


def drop_deferred_dw():
    """Forget deferred weight-gradient products that were never launched (the step that queued them raised)."""
    for sl in list(_PENDING_DW):
        sl.dw_jobs = []
    del _PENDING_DW[:]


class ConcatLinearFn(torch.autograd.Function):
    """Linear over the channel-concatenation of several [M,Ci] inputs
    (models/pointnet_ycb_23_decoder_4.py:410: conv2d(tf.concat([net1..net4], -1))).
    When the inputs are adjacent column slices of ONE row-major buffer (what the fused
    encoder produces) no concat copy is made and the input gradients come back as
    column slices of one [M, sum Ci] buffer."""

    @staticmethod
    def forward(ctx, slot, w, b, bias_grad_by_bn, *nets):
        M = nets[0].shape[0]
        ctx.slot = slot
        ctx.bias_here = b is not None and not (int(bias_grad_by_bn) & 1)
        widths = [t.shape[1] for t in nets]
        Ktot = sum(widths)
        require(w.shape[0] == Ktot, "ConcatLinearFn: weight rows != total input channels")
        base = nets[0]
        adjacent = base.stride(0) == Ktot and all(t.stride(1) == 1 for t in nets)
        off = 0
        for t, wd in zip(nets, widths):
            adjacent = adjacent and t.shape[0] == M and t.stride(0) == Ktot and \
                t.data_ptr() == base.data_ptr() + 4 * off
            off += wd
        if adjacent:
            xp, ldx = base.data_ptr(), Ktot
            ctx.cat = None
        else:
            cat = torch.cat([t.contiguous() for t in nets], dim=1)
            xp, ldx = cat.data_ptr(), Ktot
            ctx.cat = cat
        N = w.shape[1]
        ctx.bf16 = gemm_is_bf16()
        ctx.x16 = ctx.w16 = None
        ctx.x3 = False
        if (ctx.bf16 and ACT_BF16 and not DETERMINISTIC and BN_SYNC is None and (int(bias_grad_by_bn) & 6) == 6 and
                xp % 16 == 0 and w.is_contiguous() and N % 256 == 0 and
                int(L().cloudaae_gemm_b16_colstats_parts(M, N, Ktot)) > 0 and
                L().cloudaae_gemm_b16_supported(0, 1, M, Ktot, N) and L().cloudaae_gemm_b16_supported(1, 0, Ktot, N, M)):
            # activations in bf16: x, W rounded once, y stored as bf16, column sums from the fp32 accumulators
            parts = int(L().cloudaae_gemm_b16_colstats_parts(M, N, Ktot))
            if (isinstance(slot, ConcatSlot) and slot.buf16 is not None and slot.cols16 == Ktot and adjacent and
                    xp == slot.buf.data_ptr() and slot.buf16.numel() == M * Ktot):
                x16 = slot.buf16.view(M, Ktot)      # every layer stored its bfloat16 slice already (EdgeConvFn)
            else:
                x16 = _lib.empty((M, Ktot), dtype=torch.bfloat16, device=w.device)
                _lib.check(L().cloudaae_to_bf16(M * Ktot, xp, x16.data_ptr(), stream()), "cloudaae_to_bf16")
            w16 = to_bf16(w)
            y = _lib.empty((M, N), dtype=torch.bfloat16, device=w.device)
            ws = _lib.empty(parts * 2 * N, dtype=torch.float64, device=w.device)
            rec = TIMED_SITES.get("agg_fwd")
            if rec is not None:
                _lib.host(_mark, rec)
            _lib.check(L().cloudaae_gemm_b16(0, 0, M, N, Ktot, x16.data_ptr(), Ktot, w16.data_ptr(), N, y.data_ptr(), N, 1,
                                             ptr(b) if b is not None else None, 0, ptr(ws), stream()), "cloudaae_gemm_b16")
            if rec is not None:
                _lib.host(_mark, rec)
            y._cloudaae_colstats = (ws, parts, M, N)
            ctx.x16, ctx.w16 = x16, w16
            ctx.save_for_backward(w, *nets)
            ctx.widths, ctx.xp, ctx.bvar = widths, xp, b
            return y
        y = _lib.empty((M, N), dtype=torch.float32, device=w.device)
        ctx.x3 = (gemm_is_x3() and xp % 16 == 0 and w.is_contiguous() and Ktot % 32 == 0 and N % 32 == 0 and
                  bool(L().cloudaae_gemm_bf16x3p_supported(M, N, Ktot)) and
                  bool(L().cloudaae_gemm_bf16x3p_supported(M, Ktot, N)) and
                  bool(L().cloudaae_gemm_bf16x3_supported(1, 0, Ktot, N, M)))
        if gemm_is_x3() and not ctx.x3:
            _note_x3_fallback(M, N, Ktot)
        if ctx.x3:
            # the weight is split into its bfloat16 planes ONCE per step, for the forward and the input-gradient product
            pbytes = int(L().cloudaae_x3_planes_bytes(N, Ktot))
            ctx.planes_bwd = _lib.empty(pbytes // 2, dtype=torch.bfloat16, device=w.device)
            planes_fwd = _lib.empty(pbytes // 2, dtype=torch.bfloat16, device=w.device)
            _lib.check(L().cloudaae_x3_split_weight(Ktot, N, ptr(w), N, ptr(planes_fwd), ptr(ctx.planes_bwd), stream()),
                       "cloudaae_x3_split_weight")
            parts = int(L().cloudaae_gemm_bf16x3p_colstats_parts(M, N, Ktot)) if int(bias_grad_by_bn) & 2 else 0
            ws = _lib.empty(parts * 2 * N, dtype=torch.float64, device=w.device) if parts > 0 else None
            rec = TIMED_SITES.get("agg_fwd")
            if rec is not None:
                _lib.host(_mark, rec)
            _lib.check(L().cloudaae_gemm_bf16x3p(M, N, Ktot, xp, ldx, ptr(planes_fwd), ptr(y), N, ptr(b) if b is not None else None,
                                                 0, ptr(ws), stream()), "cloudaae_gemm_bf16x3p")
            if rec is not None:
                _lib.host(_mark, rec)
            if parts > 0:
                y._cloudaae_colstats = (ws, parts, M, N)
            ctx.save_for_backward(w, *nets)
            ctx.widths, ctx.xp, ctx.bvar = widths, xp, b
            return y
        # bias_grad_by_bn & 2: a training-mode batch norm consumes y next -- the product leaves the column
        # sums of its tiles in the batch norm's workspace and the statistics pass over y is skipped
        parts_fn = L().cloudaae_gemm_bf16_colstats_parts if ctx.bf16 else L().cloudaae_gemm_f32_colstats_parts
        parts = int(parts_fn(M, N, Ktot)) if int(bias_grad_by_bn) & 2 else 0
        if parts > 0:
            ws = _lib.empty(parts * 2 * N, dtype=torch.float64, device=w.device)      # colstats[parts][2][N]
            rec = TIMED_SITES.get("agg_fwd")
            if rec is not None:
                _lib.host(_mark, rec)
            fn = L().cloudaae_gemm_bf16_colstats if ctx.bf16 else L().cloudaae_gemm_f32_colstats
            _lib.check(fn(0, 0, M, N, Ktot, xp, ldx, ptr(w), N, ptr(y), N, ptr(b) if b is not None else None, ptr(ws),
                          stream()), "cloudaae_gemm_colstats")
            if rec is not None:
                _lib.host(_mark, rec)
            y._cloudaae_colstats = (ws, parts, M, N)
        else:
            gemm_forward(M, N, Ktot, xp, ldx, ptr(w), N, ptr(y), N, ptr(b) if b is not None else None, w.device,
                         site="agg_fwd", bf16=ctx.bf16)
        ctx.save_for_backward(w, *nets)
        ctx.widths, ctx.xp, ctx.bvar = widths, xp, b
        return y

    @staticmethod
    def backward(ctx, dy):
        w = ctx.saved_tensors[0]
        nets = ctx.saved_tensors[1:]
        M = nets[0].shape[0]
        Ktot, N = w.shape
        dy = dy.contiguous()
        if ctx.x16 is not None:
            return ConcatLinearFn._backward16(ctx, dy, w, M, Ktot, N)
        xp = ctx.cat.data_ptr() if ctx.cat is not None else ctx.xp
        dcat = None
        if any(ctx.needs_input_grad[4:]):
            dcat = _lib.empty((M, Ktot), dtype=torch.float32, device=w.device)
            if ctx.x3:
                _lib.check(L().cloudaae_gemm_bf16x3p(M, Ktot, N, ptr(dy), N, ptr(ctx.planes_bwd), ptr(dcat), Ktot, None, 0, None,
                                                     stream()), "cloudaae_gemm_bf16x3p")
            else:
                gemm(0, 1, M, Ktot, N, ptr(dy), N, ptr(w), N, ptr(dcat), Ktot, bf16=ctx.bf16, device=w.device)
            if ctx.slot is not None and ctx.cat is None:
                ctx.slot.dcat = dcat
        gw = _ParamGrad(w, ctx.needs_input_grad[1])
        if gw.needed:
            side = SIDE_STREAM if (gw.own is None and SIDE_AGG) else None   # (a returned gradient is consumed at once)
            if side is not None:
                _lib.stream_wait(side, stream())                    # dy is complete
            if ctx.x3:
                _lib.check(L().cloudaae_gemm_bf16x3(1, 0, Ktot, N, M, xp, Ktot, ptr(dy), N, ptr(gw.buf), N, None, gw.gemm_acc,
                                                    None, stream() if side is None else side), "cloudaae_gemm_bf16x3")
            else:
                gemm(1, 0, Ktot, N, M, xp, Ktot, ptr(dy), N, ptr(gw.buf), N, None, gw.gemm_acc, bf16=ctx.bf16, on=side,
                     device=w.device)
        gb_ret = None
        if ctx.bias_here and ctx.needs_input_grad[2]:
            gb = _ParamGrad(ctx.bvar, True)
            ws = _ws(L().cloudaae_bn_workspace_bytes(N), w.device)
            _lib.check(L().cloudaae_colsum_f32(M, N, ptr(dy), N, ptr(gb.buf), gb.accumulate, ptr(ws), stream()),
                       "cloudaae_colsum_f32")
            gb_ret = gb.done()
        grads, off = [], 0
        for i, wd in enumerate(ctx.widths):
            grads.append(dcat[:, off:off + wd] if (dcat is not None and ctx.needs_input_grad[4 + i]) else None)
            off += wd
        return (None, gw.done(), gb_ret, None) + tuple(grads)


def _concat_linear_backward16(ctx, dy, w, M, Ktot, N):
    """ConcatLinearFn.backward with bf16 storage: dy, x and W are bfloat16 in memory (csrc/gemm_b16.hip)."""
    require(dy.dtype == torch.bfloat16, "ConcatLinearFn: a bfloat16 output takes a bfloat16 gradient")
    dcat = None
    if any(ctx.needs_input_grad[4:]):
        dcat = _lib.empty((M, Ktot), dtype=torch.float32, device=w.device)
        _lib.check(L().cloudaae_gemm_b16(0, 1, M, Ktot, N, dy.data_ptr(), N, ctx.w16.data_ptr(), N, dcat.data_ptr(), Ktot,
                                         0, None, 0, None, stream()), "cloudaae_gemm_b16")
        if ctx.slot is not None and ctx.cat is None:
            ctx.slot.dcat = dcat
    gw = _ParamGrad(w, ctx.needs_input_grad[1])
    if gw.needed:
        side = SIDE_STREAM if (gw.own is None and SIDE_AGG) else None
        if side is not None:
            _lib.stream_wait(side, stream())                    # dy is complete
        _lib.check(L().cloudaae_gemm_b16(1, 0, Ktot, N, M, ctx.x16.data_ptr(), Ktot, dy.data_ptr(), N, ptr(gw.buf), N, 0,
                                         None, gw.gemm_acc, None, stream() if side is None else side),
                   "cloudaae_gemm_b16")
    require(not (ctx.bias_here and ctx.needs_input_grad[2]), "ConcatLinearFn: with bf16 activations the batch norm "
                                                             "writes the bias gradient")
    grads, off = [], 0
    for i, wd in enumerate(ctx.widths):
        grads.append(dcat[:, off:off + wd] if (dcat is not None and ctx.needs_input_grad[4 + i]) else None)
        off += wd
    return (None, gw.done(), None, None) + tuple(grads)


ConcatLinearFn._backward16 = staticmethod(_concat_linear_backward16)


class BatchNormFn(torch.autograd.Function):
    """batch_norm_template (+ReLU) on rows [M,C], optionally pooled over groups of
    `pool_rows` rows (mean/max over the points of a cloud)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, ema_mean, ema_var, decay, training, relu, pool_rows, pool_mode,
                want_activation, lin_bias=None):
        # lin_bias: the bias the producing Linear added to y; its gradient (column sums of dy) is
        # written by this function's backward (see LinearFn.forward: bias_grad_by_bn)
        ctx.set_materialize_grads(False)
        ctx.lin_bias = lin_bias
        yp, ldy = rows_ptr(y)
        M, C = y.shape
        dev = y.device
        save_mean = _lib.empty(C, dtype=torch.float32, device=dev)
        save_var = _lib.empty(C, dtype=torch.float32, device=dev)
        out = _lib.empty((M, C), dtype=torch.float32, device=dev) if (want_activation or pool_mode == 0) else None
        pooled = ties = None
        if pool_mode != 0:
            pooled = _lib.empty((M // pool_rows, C), dtype=torch.float32, device=dev)
            if pool_mode == 2:
                ties = _lib.empty_like(pooled)
        # mean pool + ReLU in training mode: the apply pass counts per group what backward needs (see
        # cloudaae_bn_backward: pool_stats), so backward starts without a pass over y
        pstats = None
        if pool_mode == 1 and training and relu and not want_activation:
            pstats = _lib.empty((M // pool_rows) * 3 * C, dtype=torch.float64, device=dev)
        pre = getattr(y, "_cloudaae_colstats", None)
        if not (pre is not None and pre[2] == M and pre[3] == C and ldy == C and training):
            pre = None
        ws = _ws(L().cloudaae_bn_workspace_bytes(C), dev)
        ctx.y16 = y.dtype == torch.bfloat16
        if ctx.y16:
            # y stored as bfloat16 by the product that made it (ConcatLinearFn with ACT_BF16): csrc/bn16.hip
            require(pre is not None and pstats is not None and BN_SYNC is None,
                    "BatchNormFn: a bfloat16 y takes the training-mode mean-pool path fed with the product's column sums")
            _lib.check(L().cloudaae_bn_meanpool_forward16(
                M, C, yp, ldy, ptr(gamma), ptr(beta), ptr(decay), ptr(ema_mean), ptr(ema_var), ptr(save_mean),
                ptr(save_var), int(pool_rows), ptr(pooled), ptr(pstats), ptr(ws), ptr(pre[0]), int(pre[1]), stream()),
                "cloudaae_bn_meanpool_forward16")
            ctx.sync = None
            ctx.save_for_backward(y, gamma, beta, save_mean, save_var, pooled, ties)
            ctx.pstats = pstats
            ctx.cfg = (int(training), int(relu), int(pool_rows), int(pool_mode))
            ctx.mark_non_differentiable(save_mean, save_var)
            return pooled, save_mean, save_var
        ctx.sync = BN_SYNC if training else None
        if ctx.sync is not None:
            _lib.check(L().cloudaae_bn_forward_sync(
                M, C, yp, ldy, ptr(gamma), ptr(beta), int(training), ptr(decay), ptr(ema_mean), ptr(ema_var),
                ptr(save_mean), ptr(save_var), int(relu), ptr(out), C, int(pool_rows), int(pool_mode), ptr(pooled),
                ptr(ties), ptr(pstats), ptr(ws), ptr(pre[0]) if pre is not None else None,
                int(pre[1]) if pre is not None else 0, ctx.sync.arg(C, dev), stream()), "cloudaae_bn_forward_sync")
        elif pre is not None:
            _lib.check(L().cloudaae_bn_forward_colstats(
                M, C, yp, ldy, ptr(gamma), ptr(beta), int(training), ptr(decay), ptr(ema_mean), ptr(ema_var),
                ptr(save_mean), ptr(save_var), int(relu), ptr(out), C, int(pool_rows), int(pool_mode), ptr(pooled),
                ptr(ties), ptr(pstats), ptr(ws), ptr(pre[0]), int(pre[1]), stream()), "cloudaae_bn_forward_colstats")
        else:
            _lib.check(L().cloudaae_bn_forward(
                M, C, yp, ldy, ptr(gamma), ptr(beta), int(training), ptr(decay), ptr(ema_mean), ptr(ema_var),
                ptr(save_mean), ptr(save_var), int(relu), ptr(out), C, int(pool_rows), int(pool_mode), ptr(pooled),
                ptr(ties), ptr(pstats), ptr(ws), stream()), "cloudaae_bn_forward")
        ctx.save_for_backward(y, gamma, beta, save_mean, save_var, pooled, ties)
        ctx.pstats = pstats
        ctx.cfg = (int(training), int(relu), int(pool_rows), int(pool_mode))
        ctx.mark_non_differentiable(save_mean, save_var)
        if pool_mode == 0:
            return out, save_mean, save_var
        if want_activation:
            return pooled, out, save_mean, save_var
        return pooled, save_mean, save_var

    @staticmethod
    def backward(ctx, *gouts):
        y, gamma, beta, save_mean, save_var, pooled, ties = ctx.saved_tensors
        training, relu, pool_rows, pool_mode = ctx.cfg
        M, C = y.shape
        yp, ldy = rows_ptr(y)
        if pool_mode == 0:
            dout, dpooled = gouts[0], None
        elif len(gouts) == 4:
            dpooled, dout = gouts[0], gouts[1]
        else:
            dpooled, dout = gouts[0], None
        if dout is not None:
            dout = dout.contiguous()
        if pool_mode != 0:
            dpooled = torch.zeros_like(pooled) if dpooled is None else dpooled.contiguous()
        if dout is None and pool_mode == 0:
            return (None,) * 12
        dy = _lib.empty((M, C), dtype=torch.bfloat16 if ctx.y16 else torch.float32, device=y.device)
        gg = _ParamGrad(gamma, ctx.needs_input_grad[1])
        gb = _ParamGrad(beta, ctx.needs_input_grad[2])
        glb = _ParamGrad(ctx.lin_bias, ctx.lin_bias is not None and ctx.needs_input_grad[11])
        grads = [g for g in (gg, gb, glb) if g.needed]
        acc = 1 if any(g.accumulate for g in grads) else 0
        if acc:                                             # mixed freshness: zero the fresh ones
            for g in grads:
                if not g.accumulate:
                    g.buf.zero_()
        ws = _ws(L().cloudaae_bn_workspace_bytes(C), y.device)
        if ctx.y16:
            require(dout is None and pool_mode == 1, "BatchNormFn: bfloat16 y is the mean-pool path")
            _lib.check(L().cloudaae_bn_meanpool_backward16(
                M, C, yp, ldy, ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_var), pool_rows, ptr(dpooled), dy.data_ptr(),
                C, ptr(gg.buf), ptr(gb.buf), ptr(glb.buf), acc, ptr(ctx.pstats), ptr(ws), stream()),
                "cloudaae_bn_meanpool_backward16")
        elif ctx.sync is not None:
            _lib.check(L().cloudaae_bn_backward_sync(
                M, C, yp, ldy, ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_var), training, relu, ptr(dout), C,
                pool_rows, pool_mode, ptr(dpooled), ptr(pooled), ptr(ties), ptr(dy), C, ptr(gg.buf), ptr(gb.buf),
                ptr(glb.buf), acc, ptr(ctx.pstats) if dout is None else None, ptr(ws), ctx.sync.arg(C, y.device),
                stream()), "cloudaae_bn_backward_sync")
        else:
            _lib.check(L().cloudaae_bn_backward(
                M, C, yp, ldy, ptr(gamma), ptr(beta), ptr(save_mean), ptr(save_var), training, relu, ptr(dout), C,
                pool_rows, pool_mode, ptr(dpooled), ptr(pooled), ptr(ties), ptr(dy), C, ptr(gg.buf), ptr(gb.buf),
                ptr(glb.buf), acc, ptr(ctx.pstats) if dout is None else None, ptr(ws), stream()), "cloudaae_bn_backward")
        return (dy, gg.done(), gb.done()) + (None,) * 8 + (glb.done() if glb.needed else None,)


class EdgeConvFn(torch.autograd.Function):
    """Fused get_edge_feature + conv2d(1x1)+bias + batch norm + ReLU + pool over k."""

    @staticmethod
    def forward(ctx, x, nn_idx, w, b, gamma, beta, ema_mean, ema_var, decay, training, pool_mode, out_slot,
                in_slot=None):
        # x: [B, N, Cin] whose rows are contiguous (row stride may exceed Cin)
        B, N, cin = x.shape
        require(x.stride(2) == 1 and x.stride(0) == N * x.stride(1), "EdgeConvFn: x rows must be contiguous")
        ldx = x.stride(1)
        cout = w.shape[1]
        require(w.shape[0] == 2 * cin, "EdgeConvFn: weights must be [2*Cin, Cout]")
        require(nn_idx.dtype == torch.int32 and tuple(nn_idx.shape[:2]) == (B, N), "EdgeConvFn: nn_idx [B,N,k] int32")
        k = nn_idx.shape[2]
        dev = x.device
        nn_idx = nn_idx.contiguous()
        if out_slot is None:
            out = _lib.empty((B, N, cout), dtype=torch.float32, device=dev)
        else:
            # (buffer [B,N,Ctot], channel offset): the output is written in place as a
            # column slice of a wider row-major buffer (saves the later concat copy).
            # The view is created HERE so autograd sees a fresh output, not an input.
            buf, off = out_slot
            if isinstance(buf, ConcatSlot):
                buf = buf.buf
            out = buf[:, :, off:off + cout]
            require(tuple(out.shape) == (B, N, cout) and out.stride(2) == 1 and
                    out.stride(0) == N * out.stride(1), "EdgeConvFn: bad output slot")
        ldo = out.stride(1)
        slot16 = None
        if (CONCAT_BF16 and out_slot is not None and isinstance(out_slot[0], ConcatSlot) and BN_SYNC is None and gemm_is_bf16() and ACT_BF16
                and not DETERMINISTIC and out_slot[0].buf.dim() == 3 and out_slot[0].buf.is_contiguous()):
            slot16 = out_slot[0]
            if slot16.buf16 is None:
                slot16.buf16 = _lib.empty(tuple(slot16.buf.shape), dtype=torch.bfloat16, device=dev)
                slot16.cols16 = 0
        pq = _lib.empty((B * N, 2 * cout), dtype=torch.float32, device=dev)
        save_mean = _lib.empty(cout, dtype=torch.float32, device=dev)
        save_var = _lib.empty(cout, dtype=torch.float32, device=dev)
        ties = _lib.empty((B * N, cout), dtype=torch.float32, device=dev) if pool_mode == 2 else None
        ws = _ws(L().cloudaae_edgeconv_workspace_bytes(cout), dev)
        # mean pool, training: per point what backward's statistics need of its k edges (see the C header)
        estats = _lib.empty((B * N, 3, cout), dtype=torch.float32, device=dev) if (training and pool_mode == 1) else None
        ctx.sync = BN_SYNC if training else None
        rec = TIMED_SITES.get("edgeconv")
        if rec is not None:
            _lib.host(_mark, rec, "edgeconv")
        if ctx.sync is not None:
            _lib.check(L().cloudaae_edgeconv_forward_sync(
                B, N, k, cin, cout, x.data_ptr(), ldx, ptr(nn_idx), ptr(w), ptr(b), ptr(gamma), ptr(beta),
                int(training), ptr(decay), ptr(ema_mean), ptr(ema_var), int(pool_mode), ptr(pq), ptr(save_mean),
                ptr(save_var), out.data_ptr(), ldo, ptr(ties), ptr(estats), int(gemm_is_bf16()), ptr(ws),
                ctx.sync.arg(cout, dev), stream()), "cloudaae_edgeconv_forward_sync")
        elif slot16 is not None:
            ctot = slot16.buf.shape[2]
            _lib.check(L().cloudaae_edgeconv_forward_b16out(
                B, N, k, cin, cout, x.data_ptr(), ldx, ptr(nn_idx), ptr(w), ptr(b), ptr(gamma), ptr(beta),
                int(training), ptr(decay), ptr(ema_mean), ptr(ema_var), int(pool_mode), ptr(pq), ptr(save_mean),
                ptr(save_var), out.data_ptr(), ldo, ptr(ties), ptr(estats), int(gemm_is_bf16()), ptr(ws),
                slot16.buf16.data_ptr() + 2 * out_slot[1], ctot, stream()), "cloudaae_edgeconv_forward_b16out")
            slot16.cols16 += cout
        else:
            _lib.check(L().cloudaae_edgeconv_forward(
                B, N, k, cin, cout, x.data_ptr(), ldx, ptr(nn_idx), ptr(w), ptr(b), ptr(gamma), ptr(beta),
                int(training), ptr(decay), ptr(ema_mean), ptr(ema_var), int(pool_mode), ptr(pq), ptr(save_mean),
                ptr(save_var), out.data_ptr(), ldo, ptr(ties), ptr(estats), int(gemm_is_bf16()), ptr(ws), stream()),
                "cloudaae_edgeconv_forward")
        if rec is not None:
            _lib.host(_mark, rec, "edgeconv")
        ctx.bf16 = gemm_is_bf16()
        ctx.estats = estats
        ctx.rev, ctx.rev_slot = None, None
        if training and out_slot is not None and isinstance(out_slot[0], ConcatSlot):
            ctx.rev = _lib.empty(B * (N + 1) + B * N * k, dtype=torch.int32, device=dev)
            ctx.rev_slot = out_slot[0]
            ctx.rev_slot.revs.append((nn_idx, ctx.rev, (B, N, k)))
        ctx.save_for_backward(x, nn_idx, w, b, gamma, beta, pq, save_mean, save_var, ties,
                              out if pool_mode == 2 else None)
        ctx.cfg = (int(training), int(pool_mode))
        # x is a column slice of a ConcatSlot buffer: backward may add dx into the slot's gradient twin
        ctx.in_slot = None
        if in_slot is not None and isinstance(in_slot[0], ConcatSlot):
            slot, ioff = in_slot
            if x.data_ptr() == slot.buf.data_ptr() + 4 * ioff and ldx == slot.buf.shape[2]:
                ctx.in_slot = (slot, int(ioff))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, nn_idx, w, b, gamma, beta, pq, save_mean, save_var, ties, fwd_out = ctx.saved_tensors
        training, pool_mode = ctx.cfg
        B, N, cin = x.shape
        cout = w.shape[1]
        k = nn_idx.shape[2]
        dev = x.device
        if not (dout.stride(2) == 1 and dout.stride(0) == N * dout.stride(1)):
            dout = dout.contiguous()
        dpq = _lib.empty((B * N, 2 * cout), dtype=torch.float32, device=dev)
        rev, rev_ready = ctx.rev, 0
        slot_r = ctx.rev_slot
        if slot_r is not None and len(slot_r.revs) <= 8 and len(set(r[2] for r in slot_r.revs)) == 1:
            if not slot_r.revs_built:
                cnt = len(slot_r.revs)
                idxs = (ctypes.c_void_p * cnt)(*[r[0].data_ptr() for r in slot_r.revs])
                revs = (ctypes.c_void_p * cnt)(*[r[1].data_ptr() for r in slot_r.revs])
                _lib.check(L().cloudaae_edgeconv_revlists(cnt, B, N, k, idxs, revs, stream()), "cloudaae_edgeconv_revlists")
                slot_r.revs_built = True
            rev_ready = 1
        else:
            rev = _lib.empty(B * (N + 1) + B * N * k, dtype=torch.int32, device=dev)
        dx, dx_ptr, lddx, acc_dx = None, None, cin, 0
        if ctx.needs_input_grad[0]:
            slot = ctx.in_slot[0] if ctx.in_slot is not None else None
            if slot is not None and slot.dcat is not None:
                # the agg GEMM's d(concat) already holds this slice's other gradient: add ours in place
                lddx = slot.dcat.shape[1]
                dx_ptr, acc_dx = slot.dcat.data_ptr() + 4 * ctx.in_slot[1], 1
            else:
                dx = _lib.empty((B, N, cin), dtype=torch.float32, device=dev)
                dx_ptr = dx.data_ptr()
        gw = _ParamGrad(w, ctx.needs_input_grad[2])
        gb = _ParamGrad(b, ctx.needs_input_grad[3])
        gg = _ParamGrad(gamma, ctx.needs_input_grad[4])
        gbe = _ParamGrad(beta, ctx.needs_input_grad[5])
        # the kernel stores; shared (non-fresh) parameters would need accumulation
        shared = [g for g in (gw, gb, gg, gbe) if g.needed and g.accumulate]
        tmp = {}
        for g in shared:
            tmp[id(g)] = g.buf
            g.buf = _lib.empty_like(g.buf)
        ws = _ws(L().cloudaae_edgeconv_workspace_bytes(cout), dev)
        side = SIDE_STREAM if (SIDE_EDGE and not shared and gw.needed and gw.own is None) else None
        # the weight gradient (nothing reads it before the optimiser) joins the other layers' in ONE grouped launch
        # at the end of the encoder's backward pass: alone it is a single wave of short split-K workgroups
        defer = (slot_r is not None and rev_ready and gw.needed and gw.own is None and not gw.accumulate and side is None
                 and not ctx.bf16 and not DETERMINISTIC)
        # deterministic mode: the weight gradient is issued here as a slice-ordered product (the kernel's own one would
        # stay whole over the B*N rows of K on a few CUs)
        det_dw = DETERMINISTIC and gw.needed and not gw.accumulate and not ctx.bf16 and side is None
        head = (B, N, k, cin, cout, x.data_ptr(), x.stride(1), ptr(nn_idx), ptr(w), ptr(b), ptr(gamma), ptr(beta),
                training, pool_mode, ptr(pq), ptr(save_mean), ptr(save_var),
                fwd_out.data_ptr() if fwd_out is not None else None, fwd_out.stride(1) if fwd_out is not None else 0,
                ptr(ties), dout.data_ptr(), dout.stride(1), ptr(dpq), ptr(rev), rev_ready, dx_ptr, lddx, acc_dx,
                None if (defer or det_dw) else ptr(gw.buf), 1 if (gw.needed and gw.gemm_acc == 2) else 0, ptr(gb.buf), ptr(gg.buf),
                ptr(gbe.buf), ptr(ctx.estats), int(ctx.bf16), ptr(ws))
        rec = TIMED_SITES.get("edgeconv")
        if rec is not None:
            _lib.host(_mark, rec, "edgeconv")
        if ctx.sync is not None:
            _lib.check(L().cloudaae_edgeconv_backward_sync(*(head + (ctx.sync.arg(cout, dev), stream(), side))),
                       "cloudaae_edgeconv_backward_sync")
        else:
            _lib.check(L().cloudaae_edgeconv_backward(*(head + (stream(), side))), "cloudaae_edgeconv_backward")
        if rec is not None:
            _lib.host(_mark, rec, "edgeconv")
        for g in shared:
            L().cloudaae_add_f32(g.buf.numel(), ptr(tmp[id(g)]), ptr(g.buf), ptr(tmp[id(g)]), stream())
            g.buf = tmp[id(g)]
        if det_dw:
            n_ws = int(L().cloudaae_gemm_f32_ordered_workspace(cin, 2 * cout, B * N))
            ws2 = _lib.empty(n_ws, dtype=torch.float32, device=dev) if n_ws else None
            _lib.check(L().cloudaae_gemm_f32_ordered_fold(1, 0, cin, 2 * cout, B * N, x.data_ptr(), x.stride(1), ptr(dpq),
                                                          2 * cout, ptr(gw.buf), cout, cout, ptr(ws2), n_ws, stream()),
                       "cloudaae_gemm_f32_ordered_fold")
        if defer:
            # [dW_c | dW_n] = X^T [dP' | dQ] over the folded kernel (as cloudaae_edgeconv_backward would issue it)
            slot_r.dw_jobs.append(((cin, 2 * cout, B * N, x.data_ptr(), x.stride(1), ptr(dpq), 2 * cout, ptr(gw.buf), cout,
                                    cout, 1 if gw.gemm_acc == 2 else 0), (x, dpq), gw))
            if slot_r not in _PENDING_DW:
                _PENDING_DW.append(slot_r)
            if len(slot_r.dw_jobs) == len(slot_r.revs):
                flush_deferred_dw(slot_r)
            return (dx, None, None, gb.done(), gg.done(), gbe.done()) + (None,) * 7
        return (dx, None, gw.done(), gb.done(), gg.done(), gbe.done()) + (None,) * 7


class FanOutFn(torch.autograd.Function):
    """n aliases of one tensor for n consumers; the gradients are summed HERE by our add kernel.
    (Autograd would sum them with its own kernel, which a recorded step cannot replay.)"""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g if g.is_contiguous() else g.contiguous() for g in grads if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            out = _lib.empty_like(acc)
            _lib.check(L().cloudaae_add_f32(acc.numel(), ptr(acc), ptr(g), ptr(out), stream()), "cloudaae_add_f32")
            acc = out
        return acc, None


class AddRowVecFn(torch.autograd.Function):
    """out[b,r,:] = x[b,r,:] + v[b,:]  (train_cloudAAE_ycbv.py:232-233); v carries no gradient path
    to the parameters (it is a statistic of the input), so only dx is produced."""

    @staticmethod
    def forward(ctx, x, v):
        x = x.contiguous()
        v = v.contiguous()
        B, R, D = x.shape
        out = _lib.empty_like(x)
        _lib.check(L().cloudaae_add_rowvec(B, R, D, ptr(x), ptr(v), ptr(out), stream()), "cloudaae_add_rowvec")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        require(a.shape == b.shape, "AddFn: shapes differ (chamfer_loss.py:12 needs n == m)")
        out = _lib.empty_like(a)
        _lib.check(L().cloudaae_add_f32(a.numel(), ptr(a), ptr(b), ptr(out), stream()), "cloudaae_add_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def _chamfer_grad_uniform(b, n, pred, label, gscalar, i1, i2, g1, g2, zeroed):
    """Chamfer gradient when every distance has the upstream gradient gscalar / (b n): fp32 atomics into zeroed
    outputs, or (deterministic mode) the reference's CPU summation order without atomics."""
    if DETERMINISTIC:
        _lib.check(L().cloudaae_nn_distance_grad_ordered(b, n, ptr(pred), n, ptr(label), None, ptr(i1), None, ptr(i2),
                                                         ptr(gscalar), 1.0 / (b * n), ptr(g1), ptr(g2), stream()),
                   "cloudaae_nn_distance_grad_ordered")
    else:
        _lib.check(L().cloudaae_nn_distance_grad_uniform(
            b, n, ptr(pred), n, ptr(label), ptr(gscalar), 1.0 / (b * n), ptr(i1), ptr(i2), ptr(g1), ptr(g2),
            1 if zeroed else 0, stream()), "cloudaae_nn_distance_grad_uniform")


def nn_search(b, n, pred, m, label, d1, i1, d2, i2, distinct=None):
    """cloudaae_nn_distance, or cloudaae_nn_distance_prefix when the caller knows that cloud c's `label` is
    distinct[0][c] distinct points followed by copies of them, row j a copy of row distinct[1][c, j] (the reference's
    Chamfer targets: visible points, then re-draws -- utils/hidden_point_removal.py:38-43): same results, the cost of the
    distinct points only."""
    if distinct is None:
        _lib.check(L().cloudaae_nn_distance(b, n, ptr(pred), m, ptr(label), ptr(d1), ptr(i1), ptr(d2), ptr(i2), stream()),
                   "cloudaae_nn_distance")
        return
    count, src = distinct
    require(count.dtype == torch.int64 and tuple(count.shape) == (b,) and count.is_contiguous(),
            "nn_distance: distinct counts must be a contiguous int64 [batch]")
    require(src.dtype == torch.int32 and tuple(src.shape) == (b, m) and src.is_contiguous(),
            "nn_distance: distinct row sources must be a contiguous int32 [batch, #points_2]")
    _lib.check(L().cloudaae_nn_distance_prefix(b, n, ptr(pred), m, ptr(label), ptr(count), ptr(src), ptr(d1), ptr(i1),
                                               ptr(d2), ptr(i2), stream()), "cloudaae_nn_distance_prefix")


class ChamferLossFn(torch.autograd.Function):
    """losses/chamfer_loss.py:8-14 as one node: nn_distance both ways, loss_per_sample = forward + backward
    distances, loss = their mean.  Returns (loss, loss_per_sample).  When only the loss is differentiated
    (the training step) the backward pass needs no per-point gradient arrays: every distance has the upstream
    gradient d(loss)/N (cloudaae_nn_distance_grad_uniform)."""

    @staticmethod
    def forward(ctx, pred, label, count2=None, row_src2=None):
        ctx.set_materialize_grads(False)
        require(pred.dim() == 3 and label.dim() == 3 and pred.shape[2] == 3 and label.shape[2] == 3,
                "NnDistance requires clouds of shape (batch,#points,3)")
        require(pred.shape[0] == label.shape[0], "NnDistance expects xyz1 and xyz2 have same batch size")
        require(pred.shape[1] == label.shape[1], "chamfer_loss: dists_forward + dists_backward needs clouds of "
                                                 "equal size (the reference fails the same way, chamfer_loss.py:12)")
        pred, label = pred.contiguous(), label.contiguous()
        b, n, _ = pred.shape
        dev = pred.device
        d1 = _lib.empty((b, n), dtype=torch.float32, device=dev)
        d2 = _lib.empty((b, n), dtype=torch.float32, device=dev)
        i1 = _lib.empty((b, n), dtype=torch.int32, device=dev)
        i2 = _lib.empty((b, n), dtype=torch.int32, device=dev)
        nn_search(b, n, pred, n, label, d1, i1, d2, i2, None if count2 is None else (count2, row_src2))
        per = _lib.empty((b, n), dtype=torch.float32, device=dev)
        loss = _lib.empty((), dtype=torch.float32, device=dev)
        ws = _ws(L().cloudaae_mean_workspace_bytes(), dev)
        _lib.check(L().cloudaae_add_mean_f32(b * n, ptr(d1), ptr(d2), ptr(per), ptr(loss), ptr(ws), stream()),
                   "cloudaae_add_mean_f32")
        ctx.save_for_backward(pred, label, i1, i2)
        return loss, per

    @staticmethod
    def backward(ctx, gloss, gper):
        pred, label, i1, i2 = ctx.saved_tensors
        b, n, _ = pred.shape
        need1, need2 = ctx.needs_input_grad[:2]
        if gloss is None and gper is None:
            return None, None, None, None
        if gper is None:
            rec = _lib.recording() is not None
            mk = (lambda t: _lib.zeros(t.shape, dtype=torch.float32, device=t.device)) if rec else _lib.empty_like
            if DETERMINISTIC:
                mk = _lib.empty_like
            g1 = mk(pred) if need1 else None
            g2 = mk(label) if need2 else None
            _chamfer_grad_uniform(b, n, pred, label, gloss.contiguous(), i1, i2, g1, g2, rec)
            return g1, g2, None, None
        gd = gper.contiguous() if gloss is None else gper + gloss / (b * n)
        g1 = _lib.empty_like(pred) if need1 else None
        g2 = _lib.empty_like(label) if need2 else None
        if DETERMINISTIC:
            _lib.check(L().cloudaae_nn_distance_grad_ordered(b, n, ptr(pred), n, ptr(label), ptr(gd), ptr(i1), ptr(gd),
                                                             ptr(i2), None, 1.0, ptr(g1), ptr(g2), stream()),
                       "cloudaae_nn_distance_grad_ordered")
        else:
            _lib.check(L().cloudaae_nn_distance_grad(b, n, ptr(pred), n, ptr(label), ptr(gd), ptr(i1), ptr(gd), ptr(i2),
                                                     ptr(g1), ptr(g2), stream()), "cloudaae_nn_distance_grad")
        return g1, g2, None, None


class MeanFn(torch.autograd.Function):
    """tf.reduce_mean over all elements -> 0-dim fp32."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        out = _lib.empty((), dtype=torch.float32, device=x.device)
        ws = _ws(L().cloudaae_mean_workspace_bytes(), x.device)
        _lib.check(L().cloudaae_mean_f32(x.numel(), ptr(x), ptr(out), ptr(ws), stream()), "cloudaae_mean_f32")
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        n = 1
        for s in ctx.shape:
            n *= s
        out = _lib.empty(ctx.shape, dtype=torch.float32, device=g.device)
        _lib.check(L().cloudaae_fill_scaled(n, ptr(g.contiguous()), 1.0 / n, None, ptr(out), stream()),
                   "cloudaae_fill_scaled")
        return out


class TransErrorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label):
        pred, label = pred.contiguous(), label.contiguous()
        B = pred.shape[0]
        per = _lib.empty(B, dtype=torch.float32, device=pred.device)
        _lib.check(L().cloudaae_trans_error(B, ptr(pred), ptr(label), ptr(per), stream()), "cloudaae_trans_error")
        ctx.save_for_backward(pred, label, per)
        return per

    @staticmethod
    def backward(ctx, gper):
        pred, label, per = ctx.saved_tensors
        d = _lib.empty_like(pred)
        _lib.check(L().cloudaae_trans_error_grad(pred.shape[0], ptr(pred), ptr(label), ptr(per),
                                                 ptr(gper.contiguous()), ptr(d), stream()),
                   "cloudaae_trans_error_grad")
        return d, None


class RotationErrorFn(torch.autograd.Function):
    """(mean angle fp32, per-sample angle fp64); gradient flows through the mean only,
    which is all the training graph uses (train_cloudAAE_ycbv.py:251-268)."""

    @staticmethod
    def forward(ctx, pred, label):
        ctx.set_materialize_grads(False)
        pred = pred.contiguous()
        label = label.to(torch.float64).contiguous()
        B = pred.shape[0]
        per = _lib.empty(B, dtype=torch.float64, device=pred.device)
        jac = _lib.empty((B, 3), dtype=torch.float64, device=pred.device)
        loss = _lib.empty((), dtype=torch.float32, device=pred.device)
        _lib.check(L().cloudaae_rotation_error(B, ptr(pred), ptr(label), ptr(per), ptr(jac), ptr(loss), stream()),
                   "cloudaae_rotation_error")
        ctx.save_for_backward(jac)
        ctx.mark_non_differentiable(per)
        return loss, per

    @staticmethod
    def backward(ctx, gloss, gper):
        (jac,) = ctx.saved_tensors
        if gloss is None:
            return None, None
        B = jac.shape[0]
        d = _lib.empty((B, 3), dtype=torch.float32, device=jac.device)
        _lib.check(L().cloudaae_rotation_error_grad(B, ptr(jac), ptr(gloss.contiguous()), ptr(d), stream()),
                   "cloudaae_rotation_error_grad")
        return d, None


class PoseLossFn(torch.autograd.Function):
    """The loss tail of the step in one launch each way (train_cloudAAE_ycbv.py:241-268):
    (total, trans_loss, trans_per [B], axag_loss, axag_per [B] float64) from the Chamfer loss scalar
    and the two pose predictions; only `total` carries a gradient (it is what the optimiser
    minimises, :268-273)."""

    @staticmethod
    def forward(ctx, xyz_loss, trans_pred, trans_label, rot_pred, rot_label, w0, w1, w2):
        ctx.set_materialize_grads(False)
        B = trans_pred.shape[0]
        dev = trans_pred.device
        trans_pred = trans_pred.contiguous()
        rot_pred = rot_pred.contiguous()
        trans_label = trans_label.to(torch.float32).contiguous()
        rot_label = rot_label.to(torch.float64).contiguous()
        tper = _lib.empty(B, dtype=torch.float32, device=dev)
        rper = _lib.empty(B, dtype=torch.float64, device=dev)
        jac = _lib.empty((B, 3), dtype=torch.float64, device=dev)
        tloss = _lib.empty((), dtype=torch.float32, device=dev)
        rloss = _lib.empty((), dtype=torch.float32, device=dev)
        total = _lib.empty((), dtype=torch.float32, device=dev)
        _lib.check(L().cloudaae_pose_losses(B, ptr(trans_pred), ptr(trans_label), ptr(rot_pred), ptr(rot_label),
                                            ptr(xyz_loss), w0, w1, w2, ptr(tper), ptr(tloss), ptr(rper), ptr(jac),
                                            ptr(rloss), ptr(total), stream()), "cloudaae_pose_losses")
        ctx.save_for_backward(trans_pred, trans_label, tper, jac)
        ctx.w = (w0, w1, w2)
        ctx.mark_non_differentiable(tloss, tper, rloss, rper)
        return total, tloss, tper, rloss, rper

    @staticmethod
    def backward(ctx, g, *unused):
        if g is None:
            return (None,) * 8
        trans_pred, trans_label, tper, jac = ctx.saved_tensors
        B = trans_pred.shape[0]
        dev = trans_pred.device
        dxyz = _lib.empty((), dtype=torch.float32, device=dev)
        dtp = _lib.empty((B, 3), dtype=torch.float32, device=dev)
        drp = _lib.empty((B, 3), dtype=torch.float32, device=dev)
        _lib.check(L().cloudaae_pose_losses_grad(B, ptr(trans_pred), ptr(trans_label), ptr(tper), ptr(jac),
                                                 ptr(g.contiguous()), ctx.w[0], ctx.w[1], ctx.w[2], ptr(dxyz),
                                                 ptr(dtp), ptr(drp), stream()), "cloudaae_pose_losses_grad")
        return dxyz, dtp, None, drp, None, None, None, None


class StepLossFn(torch.autograd.Function):
    """The loss side of the training graph as ONE node (train_cloudAAE_ycbv.py:236-268): Chamfer nn_distance both
    ways, loss_per_sample = forward + backward distances and its mean (losses/chamfer_loss.py:8-14), translation
    error, SO(3) error, weighted total -- two launches (the search, then cloudaae_loss_tail).  Returns
    (total, xyz_loss, xyz_per [B,n], trans_loss, trans_per [B], axag_loss, axag_per [B] float64); only `total`
    carries a gradient (it is what the optimiser minimises).  unit: the tensor that backward() will be given as
    d(total) -- a constant known now -- so the tail kernel already leaves the three gradients backward starts from
    and backward's first launch is the Chamfer gradient; any other upstream gradient takes cloudaae_pose_losses_grad."""

    @staticmethod
    def forward(ctx, pred, label, trans_pred, trans_label, rot_pred, rot_label, w0, w1, w2, unit, count2=None,
                row_src2=None):
        ctx.set_materialize_grads(False)
        require(pred.dim() == 3 and label.dim() == 3 and pred.shape[2] == 3 and label.shape[2] == 3,
                "NnDistance requires clouds of shape (batch,#points,3)")
        require(pred.shape[0] == label.shape[0], "NnDistance expects xyz1 and xyz2 have same batch size")
        require(pred.shape[1] == label.shape[1], "chamfer_loss: dists_forward + dists_backward needs clouds of "
                                                 "equal size (the reference fails the same way, chamfer_loss.py:12)")
        pred, label = pred.contiguous(), label.contiguous()
        b, n, _ = pred.shape
        dev = pred.device
        trans_pred = trans_pred.contiguous()
        rot_pred = rot_pred.contiguous()
        trans_label = trans_label.to(torch.float32).contiguous()
        rot_label = rot_label.to(torch.float64).contiguous()
        d1 = _lib.empty((b, n), dtype=torch.float32, device=dev)
        d2 = _lib.empty((b, n), dtype=torch.float32, device=dev)
        i1 = _lib.empty((b, n), dtype=torch.int32, device=dev)
        i2 = _lib.empty((b, n), dtype=torch.int32, device=dev)
        nn_search(b, n, pred, n, label, d1, i1, d2, i2, None if count2 is None else (count2, row_src2))
        per = _lib.empty((b, n), dtype=torch.float32, device=dev)
        xyz = _lib.empty((), dtype=torch.float32, device=dev)
        tper = _lib.empty(b, dtype=torch.float32, device=dev)
        rper = _lib.empty(b, dtype=torch.float64, device=dev)
        jac = _lib.empty((b, 3), dtype=torch.float64, device=dev)
        tloss = _lib.empty((), dtype=torch.float32, device=dev)
        rloss = _lib.empty((), dtype=torch.float32, device=dev)
        total = _lib.empty((), dtype=torch.float32, device=dev)
        dxyz = dtp = drp = None
        if unit is not None:
            dxyz = _lib.empty((), dtype=torch.float32, device=dev)
            dtp = _lib.empty((b, 3), dtype=torch.float32, device=dev)
            drp = _lib.empty((b, 3), dtype=torch.float32, device=dev)
        ws = _ws(L().cloudaae_loss_tail_workspace_bytes(), dev)
        ticket = _lib.zeros(1, dtype=torch.int32, device=dev)       # arrival counter: zero before and after
        _lib.check(L().cloudaae_loss_tail(b * n, ptr(d1), ptr(d2), ptr(per), ptr(xyz), b, ptr(trans_pred), ptr(trans_label),
                                          ptr(rot_pred), ptr(rot_label), w0, w1, w2, ptr(tper), ptr(tloss), ptr(rper),
                                          ptr(jac), ptr(rloss), ptr(total), ptr(unit), ptr(dxyz), ptr(dtp), ptr(drp),
                                          ptr(ws), ptr(ticket), stream()), "cloudaae_loss_tail")
        ctx.save_for_backward(pred, label, i1, i2, trans_pred, trans_label, tper, jac)
        ctx.w = (w0, w1, w2)
        ctx.unit_ptr = unit.data_ptr() if unit is not None else None
        ctx.ready = (dxyz, dtp, drp)
        ctx.mark_non_differentiable(xyz, per, tloss, tper, rloss, rper)
        return total, xyz, per, tloss, tper, rloss, rper

    @staticmethod
    def backward(ctx, g, *unused):
        if g is None:
            return (None,) * 12
        pred, label, i1, i2, trans_pred, trans_label, tper, jac = ctx.saved_tensors
        b, n, _ = pred.shape
        dev = pred.device
        if ctx.unit_ptr is not None and g.data_ptr() == ctx.unit_ptr:
            dxyz, dtp, drp = ctx.ready                  # written by the forward pass's tail kernel
        else:
            dxyz = _lib.empty((), dtype=torch.float32, device=dev)
            dtp = _lib.empty((b, 3), dtype=torch.float32, device=dev)
            drp = _lib.empty((b, 3), dtype=torch.float32, device=dev)
            _lib.check(L().cloudaae_pose_losses_grad(b, ptr(trans_pred), ptr(trans_label), ptr(tper), ptr(jac),
                                                     ptr(g.contiguous()), ctx.w[0], ctx.w[1], ctx.w[2], ptr(dxyz),
                                                     ptr(dtp), ptr(drp), stream()), "cloudaae_pose_losses_grad")
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g1 = g2 = None
        if need1 or need2:
            rec = _lib.recording() is not None
            mk = (lambda t: _lib.zeros(t.shape, dtype=torch.float32, device=t.device)) if rec else _lib.empty_like
            if DETERMINISTIC:
                mk = _lib.empty_like
            g1 = mk(pred) if need1 else None
            g2 = mk(label) if need2 else None
            _chamfer_grad_uniform(b, n, pred, label, dxyz, i1, i2, g1, g2, rec)
        return (g1, g2, dtp if ctx.needs_input_grad[2] else None, None, drp if ctx.needs_input_grad[4] else None,
                None, None, None, None, None, None, None)


class LossMixFn(torch.autograd.Function):
    """total = w0*a + w1*b + w2*c on device scalars (train_cloudAAE_ycbv.py:268)."""

    @staticmethod
    def forward(ctx, a, b, c, w0, w1, w2):
        out = _lib.empty((), dtype=torch.float32, device=a.device)
        _lib.check(L().cloudaae_loss_mix(ptr(a), ptr(b), ptr(c), w0, w1, w2, ptr(out), stream()),
                   "cloudaae_loss_mix")
        ctx.w = (w0, w1, w2)
        return out

    @staticmethod
    def backward(ctx, g):
        ga, gb, gc = (_lib.empty((), dtype=torch.float32, device=g.device) for _ in range(3))
        _lib.check(L().cloudaae_loss_mix_grad(ptr(g.contiguous()), ctx.w[0], ctx.w[1], ctx.w[2], ptr(ga), ptr(gb),
                                              ptr(gc), stream()), "cloudaae_loss_mix_grad")
        return ga, gb, gc, None, None, None


class EdgeFeatureFn(torch.autograd.Function):
    """Unfused get_edge_feature / get_edge_feature_wo_center (utils/tf_util.py:635-706)."""

    @staticmethod
    def forward(ctx, x, nn_idx, with_center):
        B, N, C = x.shape
        require(x.stride(2) == 1 and (B == 1 or x.stride(0) == N * x.stride(1)), "EdgeFeatureFn: x rows must be contiguous")
        nn_idx = nn_idx.contiguous()
        k = nn_idx.shape[2]
        out = _lib.empty((B, N, k, (2 if with_center else 1) * C), dtype=torch.float32, device=x.device)
        _lib.check(L().cloudaae_edge_feature(B, N, k, C, int(with_center), x.data_ptr(), x.stride(1), ptr(nn_idx),
                                             ptr(out), stream()), "cloudaae_edge_feature")
        ctx.save_for_backward(nn_idx)
        ctx.cfg = (B, N, k, C, int(with_center))
        return out

    @staticmethod
    def backward(ctx, g):
        (nn_idx,) = ctx.saved_tensors
        B, N, k, C, wc = ctx.cfg
        dx = _lib.empty((B, N, C), dtype=torch.float32, device=g.device)
        _lib.check(L().cloudaae_edge_feature_grad(B, N, k, C, wc, ptr(g.contiguous()), ptr(nn_idx), ptr(dx), stream()),
                   "cloudaae_edge_feature_grad")
        return dx, None, None


class PoolRowsFn(torch.autograd.Function):
    """tf.reduce_mean / tf.reduce_max over `rows` consecutive rows of x[groups*rows, C]."""

    @staticmethod
    def forward(ctx, x, rows, mode):
        x = x.contiguous()
        M, C = x.shape
        G = M // rows
        out = _lib.empty((G, C), dtype=torch.float32, device=x.device)
        ties = _lib.empty_like(out) if mode == 2 else None
        _lib.check(L().cloudaae_pool_rows(G, rows, C, mode, ptr(x), ptr(out), ptr(ties), stream()), "cloudaae_pool_rows")
        ctx.save_for_backward(x, out, ties)
        ctx.cfg = (G, rows, C, mode)
        return out

    @staticmethod
    def backward(ctx, g):
        x, out, ties = ctx.saved_tensors
        G, rows, C, mode = ctx.cfg
        dx = _lib.empty_like(x)
        _lib.check(L().cloudaae_pool_rows_grad(G, rows, C, mode, ptr(x), ptr(out), ptr(ties), ptr(g.contiguous()),
                                               ptr(dx), stream()), "cloudaae_pool_rows_grad")
        return dx, None, None


class MulAddFn(torch.autograd.Function):
    """out = a + b * c with c a constant (noise): z_mean + z_std * eps, models/...:953."""

    @staticmethod
    def forward(ctx, a, b, c):
        a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
        out = _lib.empty_like(a)
        _lib.check(L().cloudaae_mul_add_f32(a.numel(), ptr(a), ptr(b), ptr(c), ptr(out), stream()), "cloudaae_mul_add_f32")
        ctx.save_for_backward(c)
        return out

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        g = g.contiguous()
        db = _lib.empty_like(g)
        _lib.check(L().cloudaae_mul_add_f32(g.numel(), None, ptr(g), ptr(c), ptr(db), stream()), "cloudaae_mul_add_f32")
        return g, db, None
