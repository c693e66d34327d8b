"""Sampling ops -- mirror of the reference's tf_ops/sampling/tf_sampling.py:14-57,
backed by cloudaae_prob_sample / cloudaae_farthest_point_sample / cloudaae_gather_point[_grad]
(include/cloudaae_hip.h).
"""
import torch

from ... import _lib
from ..._lib import ptr, require, stream


def prob_sample(inp, inpr):
    """
input:
    batch_size * ncategory float32
    batch_size * npoints   float32
returns:
    batch_size * npoints   int32
    """
    # tf_sampling.cpp:77-82: rank-2 inputs with equal batch; no gradient (tf_sampling.py:22)
    require(inp.dim() == 2, "ProbSample expects (batch_size,num_choices) inp shape")
    require(inpr.dim() == 2 and inpr.shape[0] == inp.shape[0],
            "ProbSample expects (batch_size,num_points) inpr shape")
    require(inp.dtype == torch.float32 and inpr.dtype == torch.float32, "ProbSample: float32 inputs")
    require(inp.shape[1] >= 1, "ProbSample needs at least one category")
    inp = inp.detach().contiguous()
    inpr = inpr.detach().contiguous()
    b, n = inp.shape
    m = inpr.shape[1]
    out = _lib.empty((b, m), dtype=torch.int32, device=inp.device)
    temp = _lib.empty((b, n), dtype=torch.float32, device=inp.device)
    _lib.check(_lib.lib().cloudaae_prob_sample(b, n, m, ptr(inp), ptr(inpr), ptr(temp), ptr(out), stream()),
               "cloudaae_prob_sample")
    return out


def farthest_point_sample(npoint, inp):
    """
input:
    int32
    batch_size * ndataset * 3   float32
returns:
    batch_size * npoint         int32
    """
    # tf_sampling.cpp:98,105: npoint > 0, inp (batch_size,num_points,3); no gradient (tf_sampling.py:57)
    require(int(npoint) > 0, "FarthestPointSample expects positive npoint")
    require(inp.dim() == 3 and inp.shape[2] == 3,
            "FarthestPointSample expects (batch_size,num_points,3) inp shape")
    require(inp.dtype == torch.float32, "FarthestPointSample: inp must be float32")
    inp = inp.detach().contiguous()
    b, n, _ = inp.shape
    out = _lib.empty((b, int(npoint)), dtype=torch.int32, device=inp.device)
    temp = None
    if n > 16384:  # the reference's 32*n workspace (tf_sampling.cpp:115)
        temp = _lib.empty((32, n), dtype=torch.float32, device=inp.device)
    _lib.check(_lib.lib().cloudaae_farthest_point_sample(b, n, int(npoint), ptr(inp), ptr(temp),
                                                         ptr(out), stream()),
               "cloudaae_farthest_point_sample")
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        require(inp.dim() == 3 and inp.shape[2] == 3,
                "GatherPoint expects (batch_size,num_points,3) inp shape")
        require(idx.dim() == 2 and idx.shape[0] == inp.shape[0],
                "GatherPoint expects (batch_size,num_result) idx shape")
        require(inp.dtype == torch.float32 and idx.dtype == torch.int32,
                "GatherPoint: inp float32, idx int32")
        inp = inp.contiguous()
        idx = idx.contiguous()
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = _lib.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        _lib.check(_lib.lib().cloudaae_gather_point(b, n, m, ptr(inp), ptr(idx), ptr(out), stream()),
                   "cloudaae_gather_point")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        b, m = idx.shape
        inp_g = _lib.empty((b, ctx.n, 3), dtype=torch.float32, device=out_g.device)
        _lib.check(_lib.lib().cloudaae_gather_point_grad(b, ctx.n, m, ptr(out_g.contiguous()),
                                                         ptr(idx), ptr(inp_g), stream()),
                   "cloudaae_gather_point_grad")
        return inp_g, None


def gather_point(inp, idx):
    """
input:
    batch_size * ndataset * 3   float32
    batch_size * npoints        int32
returns:
    batch_size * npoints * 3    float32
    """
    return _GatherPoint.apply(inp, idx)


def gather_point_grad(inp, idx, out_g):
    """The GatherPointGrad op itself (tf_sampling.cpp:55-63,150-178)."""
    require(inp.dim() == 3 and inp.shape[2] == 3, "GatherPointGradGpuOp expects (batch_size,num_points,3) inp")
    b, n, _ = inp.shape
    require(idx.dim() == 2 and idx.shape[0] == b, "GatherPointGradGpuOp expects (batch_size,num_result) idx shape")
    m = idx.shape[1]
    require(tuple(out_g.shape) == (b, m, 3), "GatherPointGradGpuOp expects (batch_size,num_result,3) out_g shape")
    inp_g = _lib.empty((b, n, 3), dtype=torch.float32, device=out_g.device)
    _lib.check(_lib.lib().cloudaae_gather_point_grad(b, n, m, ptr(out_g.contiguous()),
                                                     ptr(idx.contiguous()), ptr(inp_g), stream()),
               "cloudaae_gather_point_grad")
    return inp_g
