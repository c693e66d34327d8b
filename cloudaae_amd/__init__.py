"""cloudaae_amd -- MI355X (gfx950) native hot path of CloudAAE.

Host side mirrors the reference's module layout so its call sites read the same:
  tf_ops/nn_distance/tf_nndistance.py   nn_distance
  tf_ops/sampling/tf_sampling.py        farthest_point_sample, gather_point
  utils/tf_util.py                      conv2d, fully_connected, knn, ...
  models/pointnet_ycb_23_decoder_4.py   get_model_dgcnn_mean_6d, ...
  losses/{chamfer_loss,trans_distance,angular_distance_taylor}.py
All arithmetic runs in libcloudaae_hip.so (csrc/, C-ABI in include/cloudaae_hip.h);
PyTorch provides device memory, streams, autograd plumbing and torch.distributed.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
