"""Loader of libcloudaae_hip.so (the C-ABI of include/cloudaae_hip.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  If the
shared object is missing, or a tensor is not on the GPU, the call fails loudly.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcloudaae_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None


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
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().cloudaae_last_error().decode("utf-8", "replace")
        raise HipLibraryError("%s failed with hipError %d: %s" % (what, rc, msg))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise HipLibraryError("cloudaae_amd ops run on the GPU only; got a %s tensor" % t.device)
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def require(cond, msg):
    if not cond:
        raise ValueError(msg)
