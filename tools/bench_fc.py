"""Time the fully connected stack of the decoder / pose heads (csrc/fc.hip) the way the step launches it: depth by
depth, the three chains' layers of one depth in ONE grouped launch per direction, at a batch of M rows; next to it
the per-layer kernels against the gemm + batch-norm composition they replace.
    python tools/bench_fc.py [--rows 32 128] [--iters 200] [--knob NAME=VALUE ...] [--layers]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib  # noqa: E402


_FLUSH = None


def timeit_cold(fn, iters):
    """Every launch timed on its own, behind a pass over 1 GB of other data: the weights come from HBM as they do inside a
    training step (back to back the 63 MB of the stack sit in the 256 MB memory-side cache and the wide layer reads
    40 % faster than it ever does in the step)."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
    iters = max(8, iters // 8)
    tot = 0.0
    for _ in range(3):
        fn()
    for _ in range(iters):
        _FLUSH.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b) * 1e3
    return tot / iters


def timeit(fn, iters):
    if COLD:
        return timeit_cold(fn, iters)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


COLD = False
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731


class Layer(object):
    def __init__(self, L, M, K, N, bn):
        self.M, self.K, self.N, self.bn = M, K, N, bn
        d = "cuda"
        self.x = torch.randn(M, K, device=d)
        self.W = torch.randn(K, N, device=d) / K ** 0.5
        self.b = torch.randn(N, device=d)
        self.gamma, self.beta = (torch.ones(N, device=d), torch.zeros(N, device=d)) if bn else (None, None)
        self.sm, self.sv = torch.zeros(N, device=d), torch.ones(N, device=d)
        self.mean, self.var = torch.empty(N, device=d), torch.empty(N, device=d)
        self.y, self.out = torch.zeros(M, N, device=d), torch.empty(M, N, device=d)
        self.dout = torch.randn(M, N, device=d)
        self.dx, self.dw = torch.zeros(M, K, device=d), torch.empty(K, N, device=d)
        self.dg, self.db, self.dbias = torch.empty(N, device=d), torch.empty(N, device=d), torch.empty(N, device=d)
        self.dy = torch.empty(M, N, device=d)
        self.tk = torch.zeros(max(L.cloudaae_fc_forward_tickets(M, N), 1), dtype=torch.int32, device=d)
        self.nparts = int(L.cloudaae_fc_forward_partials(M, K, N, int(bn)))
        self.parts = torch.empty(max(self.nparts, 1), device=d)

    def fill(self, l):
        l.K, l.N, l.x, l.ldx, l.w, l.bias = self.K, self.N, P(self.x), self.K, P(self.W), P(self.b)
        l.gamma, l.beta, l.ema_mean, l.ema_var = P(self.gamma), P(self.beta), P(self.sm), P(self.sv)
        l.save_mean, l.save_var, l.relu = P(self.mean), P(self.var), int(self.bn)
        l.y, l.out, l.tickets = P(self.y), P(self.out) if self.bn else None, P(self.tk)
        l.partials, l.partials_floats = P(self.parts), self.nparts
        l.dout, l.lddo, l.dx, l.lddx, l.dw, l.accumulate_dw = P(self.dout), self.N, P(self.dx), self.K, P(self.dw), 0
        l.dgamma, l.dbeta, l.dbias = P(self.dg) if self.bn else None, P(self.db) if self.bn else None, P(self.dbias)
        l.accumulate_param_grads = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rows", type=int, nargs="*", default=[32, 128])
    ap.add_argument("--points", type=int, default=1024, help="N of the model: the output layer has 12 N columns")
    ap.add_argument("--knob", action="append", default=[], help="NAME=VALUE development knob (repeatable)")
    ap.add_argument("--layers", action="store_true", help="also time every layer alone against gemm + bn")
    ap.add_argument("--cold", action="store_true", help="every launch behind a 1 GB pass over other data (weights from HBM, "
                    "as in the step; an event pair around one launch adds ~2 us)")
    ap.add_argument("--no-dx", action="store_true", help="backward without the input gradient (what the atomics cost)")
    ap.add_argument("--no-dw", action="store_true", help="backward without the weight gradient")
    args = ap.parse_args()
    global COLD
    COLD = args.cold
    L = _lib.lib()
    s = _lib.stream()
    for kv in args.knob:
        k, v = kv.split("=")
        _lib.set_knob(k, int(v))
    decay = torch.full((1,), 0.9, device="cuda")
    depths = [[(1024, 1024, True), (1024, 512, True), (1024, 512, True)],
              [(1024, 1024, True), (512, 256, True), (512, 256, True)],
              [(1024, 12 * args.points, False), (256, 3, False), (256, 3, False)]]
    for M in args.rows:
        tot_f = tot_b = 0.0
        for d, shapes in enumerate(depths):
            layers = [Layer(L, M, K, N, bn) for K, N, bn in shapes]
            arr = (_lib.FcLayer * len(layers))()
            for l, rec in zip(layers, arr):
                l.fill(rec)
                if args.no_dx:
                    rec.dx = None
                if args.no_dw:
                    rec.dw = None

            def fwd():
                rc = L.cloudaae_fc_forward_group(M, len(layers), arr, 1, P(decay), s)
                assert rc == 0, L.cloudaae_last_error()

            def bwd():
                rc = L.cloudaae_fc_backward_group(M, len(layers), arr, 1, s)
                assert rc == 0, L.cloudaae_last_error()
            fwd()
            tf, tb = timeit(fwd, args.iters), timeit(bwd, args.iters)
            mb = sum(K * N for K, N, _ in shapes) * 4 / 1e6
            gf = sum(2.0 * M * K * N for K, N, _ in shapes) / 1e9
            tot_f += tf
            tot_b += tb
            print("M=%3d depth %d  W=%5.1f MB %5.2f GFLOP | fwd %6.1f us (%.2f TB/s, %5.1f TF) | bwd %6.1f us (%.2f TB/s, %5.1f TF)"
                  % (M, d + 1, mb, gf, tf, mb / tf, gf / tf * 1e-3 * 1e3, tb, 2 * mb / tb, 2 * gf / tb * 1e-3 * 1e3), flush=True)
        print("M=%3d stack: forward %.1f us + backward %.1f us = %.1f us in 6 launches" % (M, tot_f, tot_b, tot_f + tot_b),
              flush=True)
        if not args.layers:
            continue
        ws = torch.empty(int(L.cloudaae_bn_workspace_bytes(12 * args.points)) // 8 + 1, dtype=torch.float64, device="cuda")
        for K, N, bn in [(1024, 1024, True), (1024, 512, True), (512, 256, True), (256, 3, False), (1024, 12 * args.points, False)]:
            l = Layer(L, M, K, N, bn)
            gp, bp = P(l.gamma), P(l.beta)

            def fused_fwd():
                L.cloudaae_fc_forward(M, K, N, P(l.x), K, P(l.W), P(l.b), gp, bp, 1, P(decay), P(l.sm), P(l.sv), P(l.mean),
                                      P(l.var), 1, P(l.y), P(l.out), P(l.tk), P(l.parts) if l.nparts else None, l.nparts, s)

            def old_fwd():
                L.cloudaae_gemm_f32(0, 0, M, N, K, P(l.x), K, P(l.W), N, P(l.y), N, P(l.b), 0, s)
                if bn:
                    L.cloudaae_bn_forward(M, N, P(l.y), N, gp, bp, 1, P(decay), P(l.sm), P(l.sv), P(l.mean), P(l.var), 1,
                                          P(l.out), N, 0, 0, None, None, None, P(ws), s)

            def fused_bwd():
                L.cloudaae_fc_backward(M, K, N, P(l.x), K, P(l.W), P(l.y), gp, bp, P(l.mean) if bn else None,
                                       P(l.var) if bn else None, 1, 1, P(l.dout), N, P(l.dx), K, P(l.dw), 0,
                                       P(l.dg) if bn else None, P(l.db) if bn else None, P(l.dbias), 0, s)

            def old_bwd():
                src = l.dout
                if bn:
                    L.cloudaae_bn_backward(M, N, P(l.y), N, gp, bp, P(l.mean), P(l.var), 1, 1, P(l.dout), N, 0, 0, None, None,
                                           None, P(l.dy), N, P(l.dg), P(l.db), P(l.dbias), 0, None, P(ws), s)
                    src = l.dy
                else:
                    L.cloudaae_colsum_f32(M, N, P(l.dout), N, P(l.dbias), 0, P(ws), s)
                L.cloudaae_gemm_f32(0, 1, M, K, N, P(src), N, P(l.W), N, P(l.dx), K, None, 0, s)
                L.cloudaae_gemm_f32(1, 0, K, N, M, P(l.x), K, P(src), N, P(l.dw), N, None, 0, s)

            old_fwd()
            mb = K * N * 4 / 1e6
            t = [timeit(f, args.iters) for f in (fused_fwd, old_fwd, fused_bwd, old_bwd)]
            print("  M=%3d K=%5d N=%5d bn=%d  W=%.1f MB | fwd %6.1f us (gemm+bn %6.1f) %.2f TB/s | bwd %6.1f us (gemm+bn %6.1f) %.2f TB/s"
                  % (M, K, N, bn, mb, t[0], t[1], mb / t[0], t[2], t[3], 2 * mb / t[2]), flush=True)


if __name__ == "__main__":
    main()
