"""Chamfer nearest-neighbour op -- mirror of the reference's
tf_ops/nn_distance/tf_nndistance.py:14-37 (`nn_distance` + its registered gradient),
backed by cloudaae_nn_distance / cloudaae_nn_distance_grad (include/cloudaae_hip.h).
"""
import torch

from ... import _lib
from ..._lib import ptr, require, stream


def _check_cloud(name, t):
    # same conditions as the OP_REQUIREs of tf_nndistance.cpp:51-58
    require(t.dim() == 3, "NnDistance requires %s be of shape (batch,#points,3)" % name)
    require(t.shape[2] == 3, "NnDistance only accepts 3d point set %s" % name)
    require(t.dtype == torch.float32, "NnDistance: %s must be float32" % name)


class _NnDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, count2=None, row_src2=None):
        ctx.set_materialize_grads(False)
        _check_cloud("xyz1", xyz1)
        _check_cloud("xyz2", xyz2)
        require(xyz1.shape[0] == xyz2.shape[0],
                "NnDistance expects xyz1 and xyz2 have same batch size")
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist1 = _lib.empty((b, n), dtype=torch.float32, device=xyz1.device)
        idx1 = _lib.empty((b, n), dtype=torch.int32, device=xyz1.device)
        dist2 = _lib.empty((b, m), dtype=torch.float32, device=xyz1.device)
        idx2 = _lib.empty((b, m), dtype=torch.int32, device=xyz1.device)
        from ...utils import _functions as F
        F.nn_search(b, n, xyz1, m, xyz2, dist1, idx1, dist2, idx2, None if count2 is None else (count2, row_src2))
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, idx1, dist2, idx2

    @staticmethod
    def backward(ctx, grad_dist1, grad_idx1, grad_dist2, grad_idx2):
        # grad_idx* are ignored, as in tf_nndistance.py:31-37
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        need1, need2 = ctx.needs_input_grad[:2]
        if grad_dist1 is None:
            grad_dist1 = torch.zeros((b, n), dtype=torch.float32, device=xyz1.device)
        if grad_dist2 is None:
            grad_dist2 = torch.zeros((b, m), dtype=torch.float32, device=xyz1.device)
        g1 = _lib.empty_like(xyz1) if need1 else None
        g2 = _lib.empty_like(xyz2) if need2 else None
        _grad(b, n, xyz1, m, xyz2, grad_dist1.contiguous(), idx1, grad_dist2.contiguous(), idx2, g1, g2)
        return g1, g2, None, None


def nn_distance(xyz1, xyz2, distinct2=None):
    """
Computes the distance of nearest neighbors for a pair of point clouds
input: xyz1: (batch_size,#points_1,3)  the first point cloud
input: xyz2: (batch_size,#points_2,3)  the second point cloud
output: dist1: (batch_size,#point_1)   distance from first to second
output: idx1:  (batch_size,#point_1)   nearest neighbor from first to second
output: dist2: (batch_size,#point_2)   distance from second to first
output: idx2:  (batch_size,#point_2)   nearest neighbor from second to first
distinct2 (extra, optional): (count [batch] int64, row_src [batch,#points_2] int32) -- the caller's knowledge that cloud
    b of xyz2 is count[b] distinct points followed by bitwise copies of them, row j a copy of row row_src[b, j] < count[b]
    (the reference's Chamfer targets: visible points, then random re-draws, utils/hidden_point_removal.py:38-43;
    hidden_point_removal.convexHull(return_src=True) gives both).  The outputs are the same bit for bit; the search
    visits the distinct points only.
    """
    count2, row_src2 = distinct2 if distinct2 is not None else (None, None)
    return _NnDistance.apply(xyz1, xyz2, count2, row_src2)


def _grad(b, n, xyz1, m, xyz2, gd1, idx1, gd2, idx2, g1, g2, ordered=None):
    """cloudaae_nn_distance_grad (fp32 atomics, as the reference's GPU kernel) or, in deterministic mode
    (utils._functions.DETERMINISTIC, or ordered=True), cloudaae_nn_distance_grad_ordered: the summation order of
    the reference's sequential CPU loops (tf_nndistance.cpp:126-163), bit for bit."""
    from ...utils import _functions as F
    if F.DETERMINISTIC if ordered is None else ordered:
        _lib.check(_lib.lib().cloudaae_nn_distance_grad_ordered(
            b, n, ptr(xyz1), m, ptr(xyz2), ptr(gd1), ptr(idx1), ptr(gd2), ptr(idx2), None, 1.0, ptr(g1), ptr(g2),
            stream()), "cloudaae_nn_distance_grad_ordered")
    else:
        _lib.check(_lib.lib().cloudaae_nn_distance_grad(
            b, n, ptr(xyz1), m, ptr(xyz2), ptr(gd1), ptr(idx1), ptr(gd2), ptr(idx2), ptr(g1), ptr(g2), stream()),
            "cloudaae_nn_distance_grad")


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2, ordered=None):
    """The NnDistanceGrad op itself (tf_nndistance.cpp:10-18): (grad_xyz1, grad_xyz2).  ordered=True: accumulated in
    the order of the reference's CPU loops (bit-identical to them); default: the mode of utils._functions.DETERMINISTIC."""
    _check_cloud("xyz1", xyz1)
    _check_cloud("xyz2", xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    require(xyz2.shape[0] == b, "NnDistanceGrad expects xyz1 and xyz2 have same batch size")
    require(tuple(grad_dist1.shape) == (b, n), "NnDistanceGrad requires grad_dist1 be of shape(batch,#points)")
    require(tuple(idx1.shape) == (b, n), "NnDistanceGrad requires idx1 be of shape(batch,#points)")
    require(tuple(grad_dist2.shape) == (b, m), "NnDistanceGrad requires grad_dist2 be of shape(batch,#points)")
    require(tuple(idx2.shape) == (b, m), "NnDistanceGrad requires idx2 be of shape(batch,#points)")
    require(idx1.dtype == torch.int32 and idx2.dtype == torch.int32, "idx must be int32")
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    g1 = _lib.empty_like(xyz1)
    g2 = _lib.empty_like(xyz2)
    _grad(b, n, xyz1, m, xyz2, grad_dist1.contiguous(), idx1.contiguous(), grad_dist2.contiguous(), idx2.contiguous(),
          g1, g2, ordered=ordered)
    return g1, g2
