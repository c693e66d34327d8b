"""Layer library -- mirror of the reference's utils/tf_util.py for the functions the
CloudAAE scripts call (conv2d :111-179, fully_connected :321-365,
batch_norm_* :473-570, pairwise_xyz_distance :597-618, knn :621-632,
get_edge_feature :635-669), backed by libcloudaae_hip.so.
"""
import torch

from .. import _lib
from .._lib import ptr, require, stream


class PairwiseDistance(object):
    """What `pairwise_xyz_distance` returns: the [B,N,N] matrix of tf_util.py:618
    in LAZY form.  The reference materialises it only to feed `knn`; here `knn`
    consumes the lazy form with the fused kernel (cloudaae_knn) and the matrix is
    never written."""

    def __init__(self, points, channels):
        self.points = points      # [B, N, ld] contiguous fp32
        self.channels = channels  # leading channels that form the metric

    @property
    def shape(self):
        b, n, _ = self.points.shape
        return (b, n, n)


def pairwise_xyz_distance(point_cloud):
    """Compute pairwise distance of a point cloud.

    Args:
      point_cloud: tensor (batch_size, num_points, num_dims)  or
                   (batch_size, num_points, 1, num_dims)
    Returns:
      pairwise distance: (batch_size, num_points, num_points)  [lazy]

    tf_util.py:608 slices `[:, :, 0:3]`: on a 3-D input that keeps xyz only; on the
    4-D `[B,N,1,C]` tensors the later layers pass it hits the size-1 axis and keeps
    all C channels.  Both behaviours are reproduced.
    """
    require(point_cloud.dtype == torch.float32, "pairwise_xyz_distance: float32 expected")
    if point_cloud.dim() == 4:
        require(point_cloud.shape[2] == 1, "pairwise_xyz_distance: expected [B,N,1,C]")
        pts = point_cloud.reshape(point_cloud.shape[0], point_cloud.shape[1], point_cloud.shape[3])
        channels = pts.shape[2]
    else:
        require(point_cloud.dim() == 3, "pairwise_xyz_distance: expected [B,N,C]")
        pts = point_cloud
        channels = min(3, pts.shape[2])
    return PairwiseDistance(pts.detach().contiguous(), channels)


def knn(adj_matrix, k=9):
    """Get KNN based on the pairwise distance.
    Args:
      pairwise distance: (batch_size, num_points, num_points)
      k: int

    Returns:
      nearest neighbors: (batch_size, num_points, k)   int32, ascending distance,
      ties -> lower index (tf.nn.top_k of the negated matrix, tf_util.py:630-631)
    """
    require(isinstance(adj_matrix, PairwiseDistance),
            "knn expects the result of pairwise_xyz_distance")
    x = adj_matrix.points
    b, n, ld = x.shape
    nn_idx = torch.empty((b, n, int(k)), dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().cloudaae_knn(b, n, adj_matrix.channels, ld, int(k), ptr(x), ptr(nn_idx),
                                       stream()), "cloudaae_knn")
    return nn_idx
