"""Loader of libcloudaae_hip.so (the C-ABI of include/cloudaae_hip.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  If the
shared object is missing, or a tensor is not on the GPU, the call fails loudly.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLOUDAAE_HIP_LIB", os.path.join(_HERE, "libcloudaae_hip.so"))   # env: kernel A/B builds
CSRC = os.path.join(_HERE, "csrc")

_lib = None

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float
_U = ctypes.c_ulonglong
# argument types of every entry point of include/cloudaae_hip.h (stream last)
_SIGNATURES = {
    "cloudaae_nn_distance": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P],
    "cloudaae_nn_distance_prefix": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "cloudaae_nn_distance_grad": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "cloudaae_nn_distance_grad_ordered": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P],
    "cloudaae_farthest_point_sample": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_gather_point": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_gather_point_grad": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_prob_sample": [_I, _I, _I, _P, _P, _P, _P, _P],
    "cloudaae_knn": [_I, _I, _I, _I, _I, _P, _P, _P],
    "cloudaae_knn_hinted": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "cloudaae_selftest_div_by": [_F, _I, _P, _P],
    "cloudaae_gemm_f32": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "cloudaae_gemm_bf16": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "cloudaae_gemm_f32_ordered": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _L, _P],
    "cloudaae_gemm_bf16_ordered": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _L, _P],
    "cloudaae_gemm_f32_ordered_fold": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _L, _P],
    "cloudaae_gemm_f32_tn_group": [_I, _P, _P],
    "cloudaae_bn_forward": [_I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P,
                            _P, _P, _P],
    "cloudaae_bn_backward": [_I, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I,
                             _P, _P, _P, _I, _P, _P, _P],
    "cloudaae_colsum_f32": [_I, _I, _P, _I, _P, _I, _P, _P],
    # SyncBN variants: the plain argument lists + (colstats, parts for the forward) + cloudaae_bn_sync* + stream(s)
    "cloudaae_bn_forward_sync": [_I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P,
                                 _P, _P, _I, _P, _P],
    "cloudaae_bn_backward_sync": [_I, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I,
                                  _P, _P, _P, _I, _P, _P, _P, _P],
    "cloudaae_edgeconv_forward_sync": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P,
                                       _P, _P, _P, _I, _P, _P, _I, _P, _P, _P],
    "cloudaae_edgeconv_backward_sync": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P,
                                        _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P,
                                        _P, _P],
    "cloudaae_gemm_f32_colstats": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P],
    "cloudaae_gemm_bf16_colstats": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P],
    "cloudaae_gemm_b16": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _P],
    "cloudaae_to_bf16": [ctypes.c_longlong, _P, _P, _P],
    "cloudaae_gemm_bf16x3": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P],
    "cloudaae_x3_split": [_I, _I, _P, _I, _I, _P, _P],
    "cloudaae_x3_split_weight": [_I, _I, _P, _I, _P, _P, _P],
    "cloudaae_gemm_bf16x3p": [_I, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P],
    "cloudaae_bn_meanpool_forward16": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "cloudaae_bn_meanpool_backward16": [_I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P],
    "cloudaae_bn_forward_colstats": [_I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P,
                                     _P, _P, _I, _P],
    "cloudaae_fc_forward": [_I, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P],
    "cloudaae_fc_backward": [_I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P,
                             _P, _I, _P],
    "cloudaae_stream_wait": [_P, _P],
    "cloudaae_fc_forward_group": [_I, _I, _P, _I, _P, _P],
    "cloudaae_fc_backward_group": [_I, _I, _P, _I, _P],
    "cloudaae_edgeconv_forward": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P,
                                  _P, _P, _P, _I, _P, _P, _I, _P, _P],
    "cloudaae_edgeconv_forward_b16out": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P,
                                         _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P],
    "cloudaae_edgeconv_backward": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P,
                                   _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P],
    "cloudaae_edgeconv_revlists": [_I, _I, _I, _I, _P, _P, _P],
    "cloudaae_input_assemble": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "cloudaae_input_assemble_noise": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _U, _P, _P],
    "cloudaae_zero_buffers": [_I, _P, _P, _P],
    "cloudaae_loss_tail": [_L, _P, _P, _P, _P, _I, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                           _P, _P, _P],
    "cloudaae_add_rowvec": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_add_f32": [_L, _P, _P, _P, _P],
    "cloudaae_fill_scaled": [_L, _P, _F, _P, _P, _P],
    "cloudaae_mul_add_f32": [_L, _P, _P, _P, _P, _P],
    "cloudaae_edge_feature": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P],
    "cloudaae_edge_feature_grad": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cloudaae_pool_rows": [_I, _I, _I, _I, _P, _P, _P, _P],
    "cloudaae_pool_rows_grad": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cloudaae_mean_f32": [_L, _P, _P, _P, _P],
    "cloudaae_add_mean_f32": [_L, _P, _P, _P, _P, _P, _P],
    "cloudaae_nn_distance_grad_uniform": [_I, _I, _P, _I, _P, _P, _F, _P, _P, _P, _P, _I, _P],
    "cloudaae_trans_error": [_I, _P, _P, _P, _P],
    "cloudaae_trans_error_grad": [_I, _P, _P, _P, _P, _P, _P],
    "cloudaae_rotation_error": [_I, _P, _P, _P, _P, _P, _P],
    "cloudaae_rotation_error_grad": [_I, _P, _P, _P, _P],
    "cloudaae_exponential_map": [_I, _P, _P, _P],
    "cloudaae_pose_losses": [_I, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "cloudaae_pose_losses_grad": [_I, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P],
    "cloudaae_loss_mix": [_P, _P, _P, _F, _F, _F, _P, _P],
    "cloudaae_loss_mix_grad": [_P, _F, _F, _F, _P, _P, _P, _P],
    "cloudaae_adam_tf": [_L, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _F, _I, _P],
    "cloudaae_adam_tf_step": [_L, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _F, _P, _F, _F, _F, _F, _F, _F, _P, _P, _P],
    "cloudaae_sgd": [_L, _P, _P, _F, _F, _P],
    "cloudaae_bn_decay_schedule": [_P, _F, _F, _F, _F, _F, _P, _P],
    "cloudaae_increment": [_P, _F, _P],
    "cloudaae_transform_object_model": [_I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cloudaae_random_spherical_occluder": [_I, _I, _P, _F, _F, _F, _F, _U, _P, _P],
    "cloudaae_spherical_flip": [_I, _I, _P, _I, _P, _P, _F, _P, _P, _P],
    "cloudaae_hidden_point_removal": [_I, _I, _P, _P, _U, _P, _P, _P, _P, _P],
    "cloudaae_hidden_point_removal_rows": [_I, _I, _P, _P, _U, _I, _P, _P, _P, _P, _P, _P],
}


class GemmTnJob(ctypes.Structure):
    """struct cloudaae_gemm_tn_job (include/cloudaae_hip.h): one product of a grouped weight-gradient launch."""
    _fields_ = [("M", _I), ("N", _I), ("K", _I), ("A", _P), ("lda", _I), ("B", _P), ("ldb", _I), ("C", _P), ("ldc", _I),
                ("fold_c", _I), ("zeroed", _I)]


class FcLayer(ctypes.Structure):
    """struct cloudaae_fc_layer (include/cloudaae_hip.h): one layer of a grouped fully connected launch."""
    _fields_ = [("K", _I), ("N", _I), ("x", _P), ("ldx", _I), ("w", _P), ("bias", _P), ("gamma", _P), ("beta", _P),
                ("ema_mean", _P), ("ema_var", _P), ("save_mean", _P), ("save_var", _P), ("relu", _I), ("y", _P),
                ("out", _P), ("tickets", _P), ("dout", _P), ("lddo", _I), ("dx", _P), ("lddx", _I), ("dw", _P),
                ("accumulate_dw", _I), ("dgamma", _P), ("dbeta", _P), ("dbias", _P), ("accumulate_param_grads", _I),
                ("partials", _P), ("partials_floats", _L), ("out_rowvec", _P), ("out_rowvec_d", _I)]


# int (*cloudaae_allreduce_fn)(void *ctx, double *buf, int count, cloudaae_stream_t stream)
ALLREDUCE_FN = ctypes.CFUNCTYPE(_I, _P, _P, _I, _P)


class BnSyncStruct(ctypes.Structure):
    """struct cloudaae_bn_sync (include/cloudaae_hip.h): the host's all-reduce for batch-norm sums."""
    _fields_ = [("allreduce", ALLREDUCE_FN), ("ctx", _P), ("world", _I), ("buf", _P)]


_LONGLONG_RESULTS = ["cloudaae_x3_planes_bytes", "cloudaae_loss_tail_workspace_bytes", "cloudaae_bn_workspace_bytes", "cloudaae_edgeconv_workspace_bytes",
                     "cloudaae_mean_workspace_bytes", "cloudaae_gemm_f32_ordered_workspace",
                     "cloudaae_gemm_bf16_ordered_workspace"]


ABI_VERSION = 601     # CLOUDAAE_ABI_VERSION of include/cloudaae_hip.h (tests/test_capi_symbols.py compares the two)


class HipLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    jobs = str(min(8, os.cpu_count() or 1))
    subprocess.run(["make", "-C", CSRC, "-j", jobs, "all"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


class StepPlan(object):
    """A recorded step: the C-ABI calls (and host callbacks) of one pass through the hot path, in
    issue order, with their arguments frozen -- device pointers included.  `replay()` re-issues them
    on the same stream without going back through Python layers or autograd: the MI355X-native
    stand-in for a captured graph (a hipGraph of the same ~330 kernel nodes replays SLOWER than
    eager launches on ROCm 7.2: 5.3 ms vs 4.0 ms per step, tools/try_graph.py).

    Pointers stay valid because every buffer the step allocates while recording comes from the
    plan's arena (`empty()` below) and the arena lives as long as the plan."""

    CHUNK = 256 << 20

    def __init__(self, device):
        self.device = torch.device(device)
        self.entries = []          # (callable, args tuple, name or None)
        self.chunks = []           # arena: uint8 tensors
        self.offset = 0
        self.bytes = 0
        self.stream = None
        self.foreign_ops = []      # aten kernels seen while recording (must stay empty)
        # zero zone: buffers that must hold zeros when the step reaches them (outputs of split-K
        # products) sit together, so ONE fill at the start of a replay clears them all instead of
        # one ~4 us clear pass per product
        self.zchunks = []          # [tensor, used bytes]
        self.zextra = []           # other buffers to clear at the start of a replay: (tensor, bytes)
        self.internal = False      # allocator-internal torch calls are not "foreign"
        self.poison = os.environ.get("CLOUDAAE_POISON_ARENA") == "1"

    # -- arena ------------------------------------------------------------------------------
    def alloc(self, shape, dtype, device):
        if device is not None and torch.device(device) != self.device:
            raise HipLibraryError("plan arena is on %s, allocation asked for %s" % (self.device, device))
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        n = 1
        for x in shape:
            n *= x
        item = torch.empty((), dtype=dtype).element_size()
        need = max(n * item, item)
        padded = (need + 255) // 256 * 256
        if not self.chunks or self.offset + padded > self.chunks[-1].numel():
            self.chunks.append(torch.empty(max(padded, self.CHUNK), dtype=torch.uint8, device=self.device))
            self.offset = 0
        view = self.chunks[-1][self.offset:self.offset + max(n, 1) * item].view(dtype)
        self.offset += padded
        self.bytes += padded
        if self.poison and n and dtype.is_floating_point:
            # debugging aid (CLOUDAAE_POISON_ARENA=1, used by the tests): every buffer starts as NaN, so a
            # kernel that reads what no kernel wrote shows up in the losses instead of seeing stale data
            # (through the library, not torch: an in-place torch op would bump the version counter of
            # the whole chunk, which autograd views of other buffers in it would trip over)
            if getattr(self, "_nan", None) is None:
                self.internal = True
                try:
                    self._nan = torch.full((1,), float("nan"), dtype=torch.float32, device=self.device)
                finally:
                    self.internal = False
            check(lib()._cdll.cloudaae_fill_scaled(n * item // 4, self._nan.data_ptr(), 1.0, None, view.data_ptr(),
                                                   stream()), "cloudaae_fill_scaled")
        return view[:n].view(shape) if n else view[:0].view(shape)

    ZCHUNK = 32 << 20

    def alloc_zero(self, shape, dtype, device):
        """A buffer that holds zeros now and again at the start of every replay."""
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        n = 1
        for x in shape:
            n *= x
        item = torch.empty((), dtype=dtype).element_size()
        padded = (max(n, 1) * item + 255) // 256 * 256
        if not self.zchunks or self.zchunks[-1][1] + padded > self.zchunks[-1][0].numel():
            self.internal = True
            try:
                self.zchunks.append([torch.zeros(max(padded, self.ZCHUNK), dtype=torch.uint8, device=self.device), 0])
            finally:
                self.internal = False
        chunk, used = self.zchunks[-1]
        view = chunk[used:used + max(n, 1) * item].view(dtype)
        self.zchunks[-1][1] = used + padded
        return view[:n].view(shape)

    def clear_at_replay(self, tensor, nbytes):
        """`tensor`'s first nbytes (a multiple of 16, 16-byte aligned) hold zeros when a replay starts: cleared by
        the same launch as the zero zones."""
        if tensor.data_ptr() % 16 or nbytes % 16:
            raise HipLibraryError("clear_at_replay: buffer must be 16-byte aligned and a multiple of 16 bytes")
        self.zextra.append((tensor, int(nbytes)))
        self._zargs = None

    # -- replay -----------------------------------------------------------------------------
    def replay(self):
        if stream() != self.stream:
            raise HipLibraryError("a recorded step must be replayed on the stream it was recorded on")
        if self.zchunks or self.zextra:
            # ONE launch clears the zero zones and whatever else the step wants cleared when it starts
            if getattr(self, "_zargs", None) is None:
                segs = [(c.data_ptr(), (used + 15) // 16 * 16) for c, used in self.zchunks if used] + \
                       [(t.data_ptr(), nbytes) for t, nbytes in self.zextra]
                self._zargs = (len(segs), (ctypes.c_void_p * len(segs))(*[p for p, _ in segs]),
                               (ctypes.c_longlong * len(segs))(*[n for _, n in segs]))
            check(lib()._cdll.cloudaae_zero_buffers(self._zargs[0], self._zargs[1], self._zargs[2], self.stream),
                  "cloudaae_zero_buffers")
        for fn, args, name in self.entries:
            rc = fn(*args)
            if name is not None and rc != 0:
                check(rc, name)


_recording = None       # the StepPlan being recorded (module-global: the autograd thread sees it too)


class _Entry(object):
    """One launching entry point: calls through, and appends itself to the plan being recorded."""
    __slots__ = ("fn", "name")

    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *args):
        rc = self.fn(*args)
        if _recording is not None:
            _recording.entries.append((self.fn, args, self.name))
        return rc


class _Library(object):
    """What lib() returns: attribute access gives the ctypes function (non-launching queries) or
    its recording wrapper (everything in _SIGNATURES)."""

    def __init__(self, cdll):
        self._cdll = cdll
        for fn in _SIGNATURES:
            setattr(self, fn, _Entry(getattr(cdll, fn), fn))

    def __getattr__(self, name):          # only reached for names not set above
        return getattr(self._cdll, name)


def lib():
    """The loaded library.  torch is imported first so that the HIP runtime torch
    ships (SONAME libamdhip64.so.7) is the one the library binds to -- one runtime
    per process, so torch's streams and allocations are valid handles here."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C cloudaae_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        cdll = ctypes.CDLL(LIB_PATH)
        cdll.cloudaae_last_error.restype = ctypes.c_char_p
        # the signature table below describes ONE revision of the C ABI (CLOUDAAE_ABI_VERSION of include/cloudaae_hip.h):
        # a library of another revision would take these calls with shifted arguments and corrupt memory
        cdll.cloudaae_version.argtypes = []
        cdll.cloudaae_version.restype = ctypes.c_int
        have = int(cdll.cloudaae_version())
        if have != ABI_VERSION:
            raise HipLibraryError("%s speaks ABI revision %d, this package expects %d: rebuild it (`make -C cloudaae_amd/csrc`)"
                                  % (LIB_PATH, have, ABI_VERSION))
        for fn, sig in _SIGNATURES.items():
            f = getattr(cdll, fn)
            f.argtypes = sig
            f.restype = ctypes.c_int
        for fn in _LONGLONG_RESULTS:
            getattr(cdll, fn).restype = ctypes.c_longlong
        cdll.cloudaae_hpr_workspace_bytes.restype = ctypes.c_longlong
        cdll.cloudaae_hpr_workspace_bytes.argtypes = [_I, _I]
        cdll.cloudaae_gemm_f32_splits.argtypes = [_I, _I, _I]
        cdll.cloudaae_gemm_f32_splits.restype = ctypes.c_int
        cdll.cloudaae_gemm_bf16_splits.argtypes = [_I, _I, _I]
        cdll.cloudaae_gemm_bf16_splits.restype = ctypes.c_int
        cdll.cloudaae_gemm_f32_ordered_workspace.argtypes = [_I, _I, _I]
        cdll.cloudaae_gemm_bf16_ordered_workspace.argtypes = [_I, _I, _I]
        cdll.cloudaae_bn_workspace_bytes.argtypes = [_I]
        cdll.cloudaae_gemm_f32_colstats_parts.argtypes = [_I, _I, _I]
        cdll.cloudaae_gemm_f32_colstats_parts.restype = ctypes.c_int
        cdll.cloudaae_gemm_bf16_colstats_parts.argtypes = [_I, _I, _I]
        cdll.cloudaae_gemm_bf16_colstats_parts.restype = ctypes.c_int
        cdll.cloudaae_gemm_b16_colstats_parts.argtypes = [_I, _I, _I]
        cdll.cloudaae_gemm_b16_colstats_parts.restype = ctypes.c_int
        cdll.cloudaae_gemm_b16_supported.argtypes = [_I, _I, _I, _I, _I]
        cdll.cloudaae_gemm_b16_supported.restype = ctypes.c_int
        cdll.cloudaae_gemm_bf16x3_colstats_parts.argtypes = [_I, _I, _I]
        cdll.cloudaae_gemm_bf16x3_colstats_parts.restype = ctypes.c_int
        cdll.cloudaae_gemm_bf16x3_supported.argtypes = [_I, _I, _I, _I, _I]
        cdll.cloudaae_gemm_bf16x3_supported.restype = ctypes.c_int
        for q in ("cloudaae_gemm_bf16x3p_supported", "cloudaae_gemm_bf16x3p_colstats_parts"):
            getattr(cdll, q).argtypes = [_I, _I, _I]
            getattr(cdll, q).restype = ctypes.c_int
        cdll.cloudaae_x3_planes_bytes.argtypes = [_I, _I]
        for q in ("cloudaae_fc_max_rows", "cloudaae_fc_max_group"):
            getattr(cdll, q).argtypes = []
            getattr(cdll, q).restype = ctypes.c_int
        cdll.cloudaae_side_stream.argtypes = []
        cdll.cloudaae_side_stream.restype = ctypes.c_void_p
        cdll.cloudaae_fc_forward_tickets.argtypes = [_I, _I]
        cdll.cloudaae_fc_forward_tickets.restype = ctypes.c_int
        cdll.cloudaae_loss_tail_workspace_bytes.argtypes = []
        cdll.cloudaae_fc_forward_partials.argtypes = [_I, _I, _I, _I]
        cdll.cloudaae_fc_forward_partials.restype = ctypes.c_longlong
        cdll.cloudaae_edgeconv_workspace_bytes.argtypes = [_I]
        for q in ("cloudaae_set_knob", "cloudaae_unset_knob"):
            getattr(cdll, q).restype = ctypes.c_int
        cdll.cloudaae_set_knob.argtypes = [ctypes.c_char_p, _I]
        cdll.cloudaae_unset_knob.argtypes = [ctypes.c_char_p]
        _lib = _Library(cdll)
    return _lib


class record(object):
    """`with record(plan): step()` -- every launching C-ABI call and host() callback of the block
    goes into `plan`, every empty() comes from its arena, and any OTHER GPU kernel torch launches
    in the block (on this thread or the autograd thread) is noted in plan.foreign_ops: such a step
    cannot be replayed faithfully, and the caller refuses it."""

    def __init__(self, plan):
        self.plan = plan
        self.mode = None

    def __enter__(self):
        global _recording
        if _recording is not None:
            raise HipLibraryError("nested step recording")
        self.plan.stream = stream()
        _recording = self.plan
        self.mode = _ForeignOps(self.plan)
        self.mode.__enter__()
        return self.plan

    def __exit__(self, *exc):
        global _recording
        self.mode.__exit__(*exc)
        _recording = None
        return False


def recording():
    return _recording


def host(fn, *args):
    """Run a host-side callback now and, while recording, at the same position of every replay
    (event records of bench.py, the early gradient all-reduce)."""
    fn(*args)
    if _recording is not None:
        _recording.entries.append((fn, args, None))


def empty(shape, dtype=torch.float32, device=None):
    """torch.empty, or a slice of the recording plan's arena (stable address across replays)."""
    if _recording is None:
        return torch.empty(shape, dtype=dtype, device=device)
    return _recording.alloc(shape, dtype, device)


def empty_like(t):
    return empty(t.shape, t.dtype, t.device)


def zeros(shape, dtype=torch.float32, device=None):
    """torch.zeros, or a slice of the recording plan's zero zone (cleared once per replay)."""
    if _recording is None:
        return torch.zeros(shape, dtype=dtype, device=device)
    return _recording.alloc_zero(shape, dtype, device)


def gemm_splits(M, N, K, bf16=False):
    """K slices cloudaae_gemm_f32 / _bf16 uses for this shape; > 1 means the output is built with atomics."""
    fn = lib().cloudaae_gemm_bf16_splits if bf16 else lib().cloudaae_gemm_f32_splits
    return int(fn(int(M), int(N), int(K)))


from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

# aten ops that touch no device memory (views, metadata) or only allocate
_HARMLESS = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.detach", "aten.alias", "aten.slice",
             "aten.select", "aten.squeeze", "aten.unsqueeze", "aten.expand", "aten.transpose", "aten.t.",
             "aten.permute", "aten.as_strided", "aten.empty", "aten.new_empty", "aten.empty_like",
             "aten.empty_strided", "aten.lift_fresh", "aten.sym_", "aten.is_", "aten.unbind", "aten.split",
             "aten.narrow", "aten._reshape_alias", "aten.view_as")


class _ForeignOps(TorchDispatchMode):
    def __init__(self, plan):
        super().__init__()
        self.plan = plan

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not self.plan.internal and not name.startswith(_HARMLESS) and any(isinstance(a, torch.Tensor) and a.is_cuda for a in args):
            self.plan.foreign_ops.append(name)      # a kernel (or a sync) the plan would not replay
        return func(*args, **(kwargs or {}))


def set_knob(name, value):
    """A development knob of the library (include/cloudaae_hip.h: cloudaae_set_knob); value None = unset."""
    if value is None:
        check(lib()._cdll.cloudaae_unset_knob(name.encode()), "cloudaae_unset_knob")
    else:
        check(lib()._cdll.cloudaae_set_knob(name.encode(), int(value)), "cloudaae_set_knob")


def check(rc, what):
    if rc != 0:
        msg = lib().cloudaae_last_error().decode("utf-8", "replace")
        raise HipLibraryError("%s failed with hipError %d: %s" % (what, rc, msg))


def side_stream():
    """The library's low-priority stream for work off the critical path (raw hipStream_t)."""
    p = lib()._cdll.cloudaae_side_stream()
    if not p:
        raise HipLibraryError("cloudaae_side_stream failed: %s" % lib().cloudaae_last_error().decode("utf-8", "replace"))
    return p


def stream_wait(waiter, signaller):
    """Everything enqueued on `signaller` so far precedes what `waiter` gets from now on (recorded
    into the plan like any launching call)."""
    check(lib().cloudaae_stream_wait(waiter, signaller), "cloudaae_stream_wait")


def stream():
    """Raw hipStream_t of torch's current stream on the current device."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("cloudaae_amd ops run on the GPU only; got a %s tensor" % t.device)
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def rows_ptr(t):
    """(pointer, row stride) of a 2-D tensor whose rows are contiguous (column slices
    of a wider row-major buffer are fine)."""
    if not t.is_cuda:
        raise HipLibraryError("cloudaae_amd ops run on the GPU only; got a %s tensor" % t.device)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("expected a 2-D tensor with contiguous rows")
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


def require(cond, msg):
    if not cond:
        raise ValueError(msg)
