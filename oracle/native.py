"""ctypes/numpy front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under cloudaae_amd/ does.

  liboracle.so               our C restatement (oracle/cloudaae_oracle.c)
  _ref/libref_nndistance.so  the reference's own Chamfer lines, when built
                             (oracle/build_ref.sh); `ref_*` raise if absent.
  _ref/libref_gpu.so         the reference's own GPU kernels (tf_sampling_g.cu, tf_nndistance_g.cu) compiled for gfx950
                             (oracle/ref_gpu_shim.hip); `ref_gpu_*` take torch CUDA tensors, GPU tests only.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c_f = ctypes.POINTER(ctypes.c_float)
_c_i = ctypes.POINTER(ctypes.c_int32)


def build(quiet=True):
    """Compile liboracle.so (and _ref when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load(path):
    if not os.path.exists(path):
        return None
    return ctypes.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        p = os.environ.get("CLOUDAAE_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # (env: the sanitizer build)
        if not os.path.exists(p):
            build()
        _lib = ctypes.CDLL(p)
    return _lib


def ref():
    """The reference-lines library, or None when it was not built."""
    global _ref
    if _ref is None:
        _ref = _load(os.path.join(_HERE, "_ref", "libref_nndistance.so"))
    return _ref


def have_ref():
    return ref() is not None


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_c_f)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_c_i)


def max_threads():
    return int(lib().oracle_max_threads())


def nn_distance(xyz1, xyz2, threads=1):
    """(dist1, idx1, dist2, idx2) -- tf_nndistance.cpp:21-43,79-80."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), np.float32)
    i1 = np.zeros((b, n), np.int32)
    d2 = np.zeros((b, m), np.float32)
    i2 = np.zeros((b, m), np.int32)
    lib().oracle_nn_distance(b, n, m, p1, p2, d1.ctypes.data_as(_c_f), i1.ctypes.data_as(_c_i),
                             d2.ctypes.data_as(_c_f), i2.ctypes.data_as(_c_i), int(threads))
    return d1, i1, d2, i2


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2, threads=1):
    """(grad_xyz1, grad_xyz2) -- tf_nndistance.cpp:126-163."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    g1, pg1 = _f(grad_dist1)
    g2, pg2 = _f(grad_dist2)
    i1, pi1 = _i(idx1)
    i2, pi2 = _i(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    o1 = np.zeros((b, n, 3), np.float32)
    o2 = np.zeros((b, m, 3), np.float32)
    lib().oracle_nn_distance_grad(b, n, m, p1, p2, pg1, pi1, pg2, pi2,
                                  o1.ctypes.data_as(_c_f), o2.ctypes.data_as(_c_f), int(threads))
    return o1, o2


def ref_nn_distance(xyz1, xyz2):
    r = ref()
    if r is None:
        raise RuntimeError("oracle/_ref/libref_nndistance.so not built")
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), np.float32)
    i1 = np.zeros((b, n), np.int32)
    d2 = np.zeros((b, m), np.float32)
    i2 = np.zeros((b, m), np.int32)
    r.ref_nn_distance(b, n, m, p1, p2, d1.ctypes.data_as(_c_f), i1.ctypes.data_as(_c_i),
                      d2.ctypes.data_as(_c_f), i2.ctypes.data_as(_c_i))
    return d1, i1, d2, i2


def ref_nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    r = ref()
    if r is None:
        raise RuntimeError("oracle/_ref/libref_nndistance.so not built")
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    g1, pg1 = _f(grad_dist1)
    g2, pg2 = _f(grad_dist2)
    i1, pi1 = _i(idx1)
    i2, pi2 = _i(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    o1 = np.zeros((b, n, 3), np.float32)
    o2 = np.zeros((b, m, 3), np.float32)
    r.ref_nn_distance_grad(b, n, m, p1, p2, pg1, pi1, pg2, pi2,
                           o1.ctypes.data_as(_c_f), o2.ctypes.data_as(_c_f))
    return o1, o2


def farthest_point_sample(npoint, inp, threads=1):
    """idx [b, npoint] int32 -- tf_sampling_g.cu:105-170."""
    inp, p = _f(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    lib().oracle_farthest_point_sample(b, n, int(npoint), p, out.ctypes.data_as(_c_i), int(threads))
    return out


def gather_point(inp, idx):
    inp, p = _f(inp)
    idx, pi = _i(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.zeros((b, m, 3), np.float32)
    lib().oracle_gather_point(b, n, m, p, pi, out.ctypes.data_as(_c_f))
    return out


def gather_point_grad(inp_shape, idx, out_g):
    idx, pi = _i(idx)
    out_g, pg = _f(out_g)
    b, n, _ = inp_shape
    m = idx.shape[1]
    inp_g = np.zeros((b, n, 3), np.float32)
    lib().oracle_gather_point_grad(b, n, m, pg, pi, inp_g.ctypes.data_as(_c_f))
    return inp_g


def prob_sample(inp, inpr, return_cumsum=False):
    """out [b, m] int32 -- tf_sampling_g.cu:7-104 (ProbSample): inp [b, n] weights, inpr [b, m] in [0,1)."""
    inp, pp = _f(inp)
    inpr, pr = _f(inpr)
    b, n_ = inp.shape
    m = inpr.shape[1]
    cum = np.zeros((b, n_), np.float32)
    out = np.zeros((b, m), np.int32)
    lib().oracle_prob_sample(b, n_, m, pp, pr, cum.ctypes.data_as(_c_f), out.ctypes.data_as(_c_i))
    return (out, cum) if return_cumsum else out


def knn(x, k, channels=None, threads=1, return_dist=False):
    """nn_idx [b, n, k] int32 -- tf_util.py:597-632 on the first `channels` of x[b,n,ld]."""
    x, p = _f(x)
    b, n, ld = x.shape
    c = ld if channels is None else int(channels)
    idx = np.zeros((b, n, k), np.int32)
    dist = np.zeros((b, n, k), np.float32) if return_dist else None
    lib().oracle_knn(b, n, c, ld, int(k), p, idx.ctypes.data_as(_c_i),
                     dist.ctypes.data_as(_c_f) if return_dist else None, int(threads))
    return (idx, dist) if return_dist else idx


def pairwise_distance(x, channels=None):
    """D [n, n] of ONE cloud x[n, ld] -- tf_util.py:597-618."""
    x, p = _f(x)
    n, ld = x.shape
    c = ld if channels is None else int(channels)
    D = np.zeros((n, n), np.float32)
    lib().oracle_pairwise_distance(n, c, ld, p, D.ctypes.data_as(_c_f))
    return D


# ---- the reference's GPU kernels, compiled for gfx950 (oracle/ref_gpu_shim.hip; needs a GPU) --------------------------
_ref_gpu = None


def ref_gpu():
    """oracle/_ref/libref_gpu.so (tf_sampling_g.cu whole, tf_nndistance_g.cu:5-151, built by oracle/build_ref.sh where
    /root/reference exists), or None."""
    global _ref_gpu
    if _ref_gpu is None:
        _ref_gpu = _load(os.path.join(_HERE, "_ref", "libref_gpu.so"))
    return _ref_gpu


def have_ref_gpu():
    return ref_gpu() is not None


def _dev(t, dtype):
    import torch
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.dtype)
    return ctypes.c_void_p(t.data_ptr())


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s: the reference kernel's launch failed (%d)" % (what, rc))


def ref_gpu_farthest_point_sample(npoint, inp):
    """farthestpointsamplingLauncher (tf_sampling_g.cu:203) on torch CUDA tensors: inp [b,n,3] f32 -> [b,npoint] i32"""
    import torch
    b, n, _ = inp.shape
    temp = torch.empty((32, n), dtype=torch.float32, device=inp.device)          # tf_sampling.cpp:115
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    _ok(ref_gpu().ref_gpu_farthest_point_sample(b, n, int(npoint), _dev(inp, torch.float32), _dev(temp, torch.float32),
                                                _dev(out, torch.int32)), "farthestpointsamplingLauncher")
    return out


def ref_gpu_gather_point(inp, idx):
    import torch
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
    _ok(ref_gpu().ref_gpu_gather_point(b, n, m, _dev(inp, torch.float32), _dev(idx, torch.int32), _dev(out, torch.float32)),
        "gatherpointLauncher")
    return out


def ref_gpu_gather_point_grad(n, idx, out_g):
    import torch
    b, m = idx.shape
    inp_g = torch.zeros((b, n, 3), dtype=torch.float32, device=idx.device)       # tf_sampling.cpp:174 clears it
    _ok(ref_gpu().ref_gpu_scatter_add_point(b, n, m, _dev(out_g, torch.float32), _dev(idx, torch.int32),
                                            _dev(inp_g, torch.float32)), "scatteraddpointLauncher")
    return inp_g


def ref_gpu_prob_sample(inp, inpr):
    """probsampleLauncher (tf_sampling_g.cu:198): inp [b,n] weights, inpr [b,m] uniform numbers -> ([b,m] i32, cumsum [b,n])"""
    import torch
    b, n = inp.shape
    m = inpr.shape[1]
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    _ok(ref_gpu().ref_gpu_prob_sample(b, n, m, _dev(inp, torch.float32), _dev(inpr, torch.float32), _dev(temp, torch.float32),
                                      _dev(out, torch.int32)), "probsampleLauncher")
    return out, temp


def ref_gpu_nn_distance(xyz1, xyz2):
    """NmDistanceKernelLauncher (tf_nndistance_g.cu:128)"""
    import torch
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = torch.empty((b, n), dtype=torch.float32, device=xyz1.device)
    i1 = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
    d2 = torch.empty((b, m), dtype=torch.float32, device=xyz1.device)
    i2 = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    _ok(ref_gpu().ref_gpu_nn_distance(b, n, _dev(xyz1, torch.float32), m, _dev(xyz2, torch.float32), _dev(d1, torch.float32),
                                      _dev(i1, torch.int32), _dev(d2, torch.float32), _dev(i2, torch.int32)),
        "NmDistanceKernelLauncher")
    return d1, i1, d2, i2


def ref_gpu_nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """NmDistanceGradKernel (tf_nndistance_g.cu:132-151) as its launcher issues it (:153-156)"""
    import torch
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    _ok(ref_gpu().ref_gpu_nn_distance_grad(b, n, _dev(xyz1, torch.float32), m, _dev(xyz2, torch.float32),
                                           _dev(grad_dist1, torch.float32), _dev(idx1, torch.int32),
                                           _dev(grad_dist2, torch.float32), _dev(idx2, torch.int32), _dev(g1, torch.float32),
                                           _dev(g2, torch.float32)), "NmDistanceGradKernel")
    return g1, g2
