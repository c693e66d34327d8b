"""SO(3) geodesic error -- mirror of the reference's losses/angular_distance_taylor.py.

get_rotation_error (:103-116) runs as one float64 HIP kernel per batch
(cloudaae_rotation_error): exponential_map of both axis-angles (:30-66, Taylor branch
for theta^2 < 1e-2), R = R_label R_pred^T, theta = acos(clip((tr R - 1)/2, +-0.9999999))
(:69-84); its gradient w.r.t. `pred` is carried by forward-mode duals through the same
op sequence.
"""
import torch

from .. import _lib
from .._lib import ptr, require, stream
from ..utils import _functions as F


def get_rotation_error(pred, label):
    '''
    Return (mean) rotation error in form of angular distance in SO(3)
    :param pred: B,3 tensor (float32 network output; cast to float64 inside, train...:249)
    :param label: B,3 tensor (float64 axis-angle)
    :return: (scalar mean [float32, as train...:253 casts it], per-sample angles [float64])
    '''
    return F.RotationErrorFn.apply(pred.to(torch.float32), label)


def exponential_map(axag, EPS=1e-2):
    """Rodrigues' formula with the Taylor branch for small angles (:30-66), float64.
    :param axag: B, 3 tensor
    :return: B, 3, 3 tensor
    (used by the data pipeline, train_cloudAAE_ycbv.py:79-85)"""
    require(EPS == 1e-2, "exponential_map: the kernel implements the reference's EPS = 1e-2")
    a = axag.detach().to(torch.float64).contiguous()
    B = a.shape[0]
    R = _lib.empty((B, 3, 3), dtype=torch.float64, device=a.device)
    _lib.check(_lib.lib().cloudaae_exponential_map(B, ptr(a), ptr(R), stream()), "cloudaae_exponential_map")
    return R
