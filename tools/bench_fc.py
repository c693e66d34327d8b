"""Time the fused fully connected kernels (csrc/fc.hip) against the gemm + batch-norm composition
they replace, per layer shape of the decoder / pose heads at a batch of 32.
    python tools/bench_fc.py [--iters 200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloudaae_amd import _lib  # noqa: E402


def timeit(fn, iters):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    L = _lib.lib()
    s = _lib.stream()
    M = 32
    for K, N, bn in [(1024, 1024, True), (1024, 512, True), (512, 256, True), (256, 3, False), (1024, 12288, False)]:
        x = torch.randn(M, K, device="cuda")
        W = torch.randn(K, N, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        sm, sv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        mean, var = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        decay = torch.full((1,), 0.9, device="cuda")
        y, out = torch.zeros(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        dout = torch.randn(M, N, device="cuda")
        dx, dw = torch.zeros(M, K, device="cuda"), torch.empty(K, N, device="cuda")
        dg, db, dbias = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        dy = torch.empty(M, N, device="cuda")
        tk = torch.zeros(L.cloudaae_fc_forward_tickets(N), dtype=torch.int32, device="cuda")
        ws = torch.empty(int(L.cloudaae_bn_workspace_bytes(N)) // 8 + 1, dtype=torch.float64, device="cuda")
        P = lambda t: t.data_ptr()  # noqa: E731
        gp, bp = (P(gamma), P(beta)) if bn else (None, None)

        nparts = int(L.cloudaae_fc_forward_partials(K, N, int(bn)))
        parts = torch.empty(max(nparts, 1), device="cuda")

        def fused_fwd():        # slices summed in a fixed order by the last one to arrive (what the package uses)
            L.cloudaae_fc_forward(M, K, N, P(x), K, P(W), P(b), gp, bp, 1, P(decay), P(sm), P(sv), P(mean), P(var), 1,
                                  P(y), P(out), 1, P(tk), P(parts) if nparts else None, nparts, s)

        def atomic_fwd():       # slices added with fp32 atomics (y counted as cleared: kernel time only)
            L.cloudaae_fc_forward(M, K, N, P(x), K, P(W), P(b), gp, bp, 1, P(decay), P(sm), P(sv), P(mean), P(var), 1,
                                  P(y), P(out), 1, P(tk), None, 0, s)

        def old_fwd():
            L.cloudaae_gemm_f32(0, 0, M, N, K, P(x), K, P(W), N, P(y), N, P(b), 0, s)
            if bn:
                L.cloudaae_bn_forward(M, N, P(y), N, gp, bp, 1, P(decay), P(sm), P(sv), P(mean), P(var), 1, P(out), N,
                                      0, 0, None, None, None, P(ws), s)

        def fused_bwd():
            L.cloudaae_fc_backward(M, K, N, P(x), K, P(W), P(y), gp, bp, P(mean) if bn else None,
                                   P(var) if bn else None, 1, 1, P(dout), N, P(dx), K, P(dw), 0,
                                   P(dg) if bn else None, P(db) if bn else None, P(dbias), 0, s)

        def old_bwd():
            src = dout
            if bn:
                L.cloudaae_bn_backward(M, N, P(y), N, gp, bp, P(mean), P(var), 1, 1, P(dout), N, 0, 0, None, None, None,
                                       P(dy), N, P(dg), P(db), P(dbias), 0, None, P(ws), s)
                src = dy
            else:
                L.cloudaae_colsum_f32(M, N, P(dout), N, P(dbias), 0, P(ws), s)
            L.cloudaae_gemm_f32(0, 1, M, K, N, P(src), N, P(W), N, P(dx), K, None, 0, s)
            L.cloudaae_gemm_f32(1, 0, K, N, M, P(x), K, P(src), N, P(dw), N, None, 0, s)

        old_fwd()
        mb = K * N * 4 / 1e6
        t = [timeit(f, args.iters) for f in (fused_fwd, old_fwd, fused_bwd, old_bwd, atomic_fwd)]
        print("K=%5d N=%5d bn=%d  W=%.1f MB | fwd %6.1f us (atomics %6.1f, gemm+bn %6.1f) %.2f TB/s | bwd %6.1f us (was %6.1f) %.2f TB/s"
              % (K, N, bn, mb, t[0], t[4], t[1], mb / t[0], t[2], t[3], 2 * mb / t[2]))


if __name__ == "__main__":
    main()
