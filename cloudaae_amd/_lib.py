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
    "cloudaae_nn_distance_grad": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "cloudaae_farthest_point_sample": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_gather_point": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_gather_point_grad": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_knn": [_I, _I, _I, _I, _I, _P, _P, _P],
    "cloudaae_gemm_f32": [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    "cloudaae_bn_forward": [_I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P,
                            _P, _P],
    "cloudaae_bn_backward": [_I, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I,
                             _P, _P, _I, _P, _P],
    "cloudaae_colsum_f32": [_I, _I, _P, _I, _P, _I, _P, _P],
    "cloudaae_edgeconv_forward": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P,
                                  _P, _P, _P, _I, _P, _P, _P],
    "cloudaae_edgeconv_backward": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P,
                                   _P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "cloudaae_input_assemble": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "cloudaae_add_rowvec": [_I, _I, _I, _P, _P, _P, _P],
    "cloudaae_add_f32": [_L, _P, _P, _P, _P],
    "cloudaae_fill_scaled": [_L, _P, _F, _P, _P, _P],
    "cloudaae_mul_add_f32": [_L, _P, _P, _P, _P, _P],
    "cloudaae_edge_feature": [_I, _I, _I, _I, _I, _P, _I, _P, _P, _P],
    "cloudaae_edge_feature_grad": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "cloudaae_pool_rows": [_I, _I, _I, _I, _P, _P, _P, _P],
    "cloudaae_pool_rows_grad": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cloudaae_mean_f32": [_L, _P, _P, _P, _P],
    "cloudaae_trans_error": [_I, _P, _P, _P, _P],
    "cloudaae_trans_error_grad": [_I, _P, _P, _P, _P, _P, _P],
    "cloudaae_rotation_error": [_I, _P, _P, _P, _P, _P, _P],
    "cloudaae_rotation_error_grad": [_I, _P, _P, _P, _P],
    "cloudaae_exponential_map": [_I, _P, _P, _P],
    "cloudaae_loss_mix": [_P, _P, _P, _F, _F, _F, _P, _P],
    "cloudaae_loss_mix_grad": [_P, _F, _F, _F, _P, _P, _P, _P],
    "cloudaae_adam_tf": [_L, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _F, _I, _P],
    "cloudaae_sgd": [_L, _P, _P, _F, _F, _P],
    "cloudaae_bn_decay_schedule": [_P, _F, _F, _F, _F, _F, _P, _P],
    "cloudaae_increment": [_P, _F, _P],
    "cloudaae_transform_object_model": [_I, _I, _I, _P, _P, _P, _P, _P, _P],
    "cloudaae_random_spherical_occluder": [_I, _I, _P, _F, _F, _F, _F, _U, _P, _P],
    "cloudaae_spherical_flip": [_I, _I, _P, _I, _P, _P, _F, _P, _P, _P],
    "cloudaae_hidden_point_removal": [_I, _I, _P, _P, _U, _P, _P, _P, _P, _P],
}
_LONGLONG_RESULTS = ["cloudaae_bn_workspace_bytes", "cloudaae_edgeconv_workspace_bytes",
                     "cloudaae_mean_workspace_bytes"]


class HipLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    jobs = str(min(8, os.cpu_count() or 1))
    subprocess.run(["make", "-C", CSRC, "-j", jobs, "all"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


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
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.cloudaae_last_error.restype = ctypes.c_char_p
        for fn, sig in _SIGNATURES.items():
            f = getattr(_lib, fn)
            f.argtypes = sig
            f.restype = ctypes.c_int
        for fn in _LONGLONG_RESULTS:
            getattr(_lib, fn).restype = ctypes.c_longlong
        _lib.cloudaae_hpr_workspace_bytes.restype = ctypes.c_longlong
        _lib.cloudaae_hpr_workspace_bytes.argtypes = [_I, _I]
        _lib.cloudaae_bn_workspace_bytes.argtypes = [_I]
        _lib.cloudaae_edgeconv_workspace_bytes.argtypes = [_I]
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().cloudaae_last_error().decode("utf-8", "replace")
        raise HipLibraryError("%s failed with hipError %d: %s" % (what, rc, msg))


def stream():
    return torch.cuda.current_stream().cuda_stream


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
