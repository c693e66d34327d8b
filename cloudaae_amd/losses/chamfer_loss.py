"""Chamfer loss -- mirror of the reference's losses/chamfer_loss.py:8-14."""
from ..tf_ops.nn_distance import tf_nndistance
from ..utils import _functions as F


def get_loss(pred, label, distinct=None):
    """ pred: BxNx3,
        label: BxNx3,
    returns (loss, loss_per_sample) where loss_per_sample = dists_forward + dists_backward is
    [B,N] (elementwise, so both clouds must have the same number of points) and
    loss = reduce_mean(loss_per_sample).
    distinct (extra, optional): (count [B] int64, row_src [B,N] int32) when `label` is count[b] distinct points
    followed by copies of them (tf_nndistance.nn_distance has the details): same results, a cheaper search. """
    count2, row_src2 = distinct if distinct is not None else (None, None)
    if pred.dim() == 3 and label.dim() == 3 and pred.shape[1] == label.shape[1]:
        # one node: nn_distance, the sum of the two distance arrays and its mean
        return F.ChamferLossFn.apply(pred, label, count2, row_src2)
    dists_forward, _, dists_backward, _ = tf_nndistance.nn_distance(pred, label, distinct2=distinct)
    loss_per_sample = F.AddFn.apply(dists_forward, dists_backward)      # fails for n != m like the reference
    loss = F.MeanFn.apply(loss_per_sample)
    return loss, loss_per_sample
