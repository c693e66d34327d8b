"""Translation error -- mirror of the reference's losses/trans_distance.py:4-9."""
from ..utils import _functions as F


def get_translation_error(pred, label):
    """loss_perSample = sqrt(reduce_sum(square(label - pred), axis=1)); loss = its mean."""
    loss_perSample = F.TransErrorFn.apply(pred, label)
    loss = F.MeanFn.apply(loss_perSample)
    return loss, loss_perSample
