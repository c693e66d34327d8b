"""Where a fully connected launch spends its time: the profiling build of the library (make -C cloudaae_amd/csrc prof;
CLOUDAAE_HIP_LIB=cloudaae_amd/libcloudaae_hip_prof.so) leaves the 100 MHz wall clock of every workgroup at a few points
of fc_fwd_kernel / fc_bwd_kernel; this prints, per depth and direction, when the workgroups start, how long each phase
takes (median / max over workgroups) and when the last one ends, all relative to the first workgroup's start.
    CLOUDAAE_HIP_LIB=$PWD/cloudaae_amd/libcloudaae_hip_prof.so python tools/dev/fc_phases.py [--rows 32 128]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloudaae_amd import _lib  # noqa: E402
from tools.bench_fc import Layer, P  # noqa: E402


def read(L, clear=True):
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    rc = L._cdll.cloudaae_fc_profile_read(buf.ctypes.data_as(ctypes.c_void_p), int(clear))
    assert rc == 0
    return buf.reshape(8192, 8).astype(np.int64)


def report(name, st, slots, labels):
    live = st[:, 0] > 0
    st = st[live]
    t0 = st[:, 0].min()
    us = lambda v: (v - t0) / 100.0  # noqa: E731
    line = "%s: %d workgroups, starts %.1f..%.1f us" % (name, len(st), us(st[:, 0]).min(), us(st[:, 0]).max())
    prev = 0
    for sl, lab in zip(slots, labels):
        ok = st[:, sl] > 0
        if not ok.any():
            continue
        d = (st[ok, sl] - st[ok, prev]) / 100.0
        line += " | %s %.1f/%.1f (n=%d)" % (lab, np.median(d), d.max(), ok.sum())
        prev = sl
    last = max(st[:, s][st[:, s] > 0].max() for s in slots if (st[:, s] > 0).any())
    print(line + " | last end %.1f us" % us(last), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[32, 128])
    ap.add_argument("--knob", action="append", default=[])
    args = ap.parse_args()
    L = _lib.lib()
    L._cdll.cloudaae_fc_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for kv in args.knob:
        k, v = kv.split("=")
        _lib.set_knob(k, int(v))
    s = _lib.stream()
    decay = torch.full((1,), 0.9, device="cuda")
    depths = [[(1024, 1024, True), (1024, 512, True), (1024, 512, True)],
              [(1024, 1024, True), (512, 256, True), (512, 256, True)],
              [(1024, 12288, False), (256, 3, False), (256, 3, False)]]
    for M in args.rows:
        for d, shapes in enumerate(depths):
            layers = [Layer(L, M, K, N, bn) for K, N, bn in shapes]
            arr = (_lib.FcLayer * len(layers))()
            for l, rec in zip(layers, arr):
                l.fill(rec)
            for direction in ("fwd", "bwd"):
                fn = L.cloudaae_fc_forward_group if direction == "fwd" else L.cloudaae_fc_backward_group
                call = (lambda: fn(M, len(layers), arr, 1, P(decay), s)) if direction == "fwd" else \
                    (lambda: fn(M, len(layers), arr, 1, s))
                for _ in range(5):
                    assert call() == 0, L.cloudaae_last_error()
                torch.cuda.synchronize()
                read(L)
                assert call() == 0
                torch.cuda.synchronize()
                st = read(L)
                if direction == "fwd":
                    report("M=%d depth %d fwd" % (M, d + 1), st, [1, 2, 3, 4, 5],
                           ["stream", "combine+publish", "ticket", "reduce", "finish"])
                else:
                    report("M=%d depth %d bwd" % (M, d + 1), st, [1, 2, 3], ["dY", "products", "drain"])


if __name__ == "__main__":
    main()
