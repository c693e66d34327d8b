"""Inference step -- mirror of the graph the reference's evaluate_cloudAAE_ycbv.py builds
(:405-477), MI355X-native.

What is mirrored: the eval-mode forward (batch-norm statistics from the moving averages, no noise),
the 4N -> N farthest point sampling of the reconstruction + gather + Chamfer against the first N
observed points (:449-451), the translation error of the prediction and of the plain centroid
(:455-460), the SO(3) error (:468-474).  What is NOT: reading the YCB-Video test frames, the ICP
refinement with open3d and the result files (out of scope).

    graph = T.TrainGraph(...); graph.restore("model.ckpt")
    out = evaluate_batch(graph, element)       # element: xyz_inlier, visiblePoints_org, class_id,
                                               #          translation, axisangle (device tensors)
"""
import torch

from . import _lib
from ._lib import ptr, require, stream
from .losses import angular_distance_taylor, chamfer_loss, trans_distance
from .tf_ops.sampling import tf_sampling
from .train_cloudAAE_ycbv import NUM_CLASS
from .utils import _functions as F


def evaluate_batch(graph, element, replay=False):
    """One pass of evaluate_cloudAAE_ycbv.py:421-477 on a batch.  Returns the tensors its loop
    fetches (:546-560): xyz_recon [B,4N,3], xyz_recon_FPS [B,N,3], rot_pred, trans_pred, the three
    losses with their per-sample values, mean_dist_loss, element_mean.
    replay=True: the pass is recorded once per input shape (_lib.StepPlan) and re-issued afterwards
    without Python layers in between -- the batch-1 latency is then the kernels', not the host's;
    the returned tensors are the same objects every call, overwritten in place."""
    if replay:
        return _replayed(graph, element)
    return _evaluate(graph, element)


def _replayed(graph, element):
    N = graph.NUM_POINT
    src = {'xyz_inlier': (element['xyz_inlier'], torch.float32),
           'visiblePoints_org': (element['visiblePoints_org'][:, 0:N, :], torch.float32),
           'class_id': (element['class_id'], torch.int64), 'translation': (element['translation'], torch.float32),
           'axisangle': (element['axisangle'], torch.float64)}
    key = tuple((k, tuple(v.shape)) for k, (v, _) in src.items())
    plans = graph.__dict__.setdefault('_eval_plans', {})
    if key not in plans:
        static = {k: torch.empty(tuple(v.shape), dtype=dt, device=graph.device) for k, (v, dt) in src.items()}
        plans[key] = [None, None, static]
        while len(plans) > 4:
            plans.pop(next(iter(plans)))
    plan, out, static = plans[key]
    for k, (v, _) in src.items():
        static[k].copy_(v, non_blocking=True)
    if plan is None:
        plan = _lib.StepPlan(graph.device)
        with _lib.record(plan):
            out = _evaluate(graph, static)
        if plan.foreign_ops:
            import warnings
            warnings.warn("evaluation pass not replayable (torch kernels inside: %s)" % sorted(set(plan.foreign_ops)))
            plans.pop(key)
            return out
        plans[key][0], plans[key][1] = plan, out
        return out
    plan.replay()
    return out


def _evaluate(graph, element):
    N = graph.NUM_POINT
    xyz = element['xyz_inlier']
    require(xyz.dim() == 3 and xyz.shape[1] >= N and xyz.shape[2] == 3, "xyz_inlier must be [B, >=num_point, 3]")
    xyz = xyz.to(torch.float32).contiguous()
    B, P, _ = xyz.shape
    cls = element['class_id'].to(torch.int64).contiguous()
    with torch.no_grad():
        # :421-438 -- first N inlier points, centroid, centring, one-hot class; no noise in evaluation
        pc = _lib.empty((B, N, 3 + NUM_CLASS), dtype=torch.float32, device=xyz.device)
        element_mean = _lib.empty((B, 3), dtype=torch.float32, device=xyz.device)
        _lib.check(_lib.lib().cloudaae_input_assemble(B, P, N, NUM_CLASS, ptr(xyz), None, ptr(cls), ptr(pc),
                                                      ptr(element_mean), None, stream()), "cloudaae_input_assemble")
        xyz_recon_res, rot_pred, trans_pred_res, end_points = graph._call_model(pc, False)      # :441-444
        xyz_recon = F.AddRowVecFn.apply(xyz_recon_res, element_mean)                             # :446
        trans_pred = F.AddRowVecFn.apply(trans_pred_res.unsqueeze(1), element_mean).squeeze(1)   # :447
        # :450 -- for all decoder: FPS of the 4N reconstructed points down to N, then Chamfer (:452)
        xyz_recon_FPS = tf_sampling.gather_point(xyz_recon, tf_sampling.farthest_point_sample(N, xyz_recon))
        visiblePoints_final = element['visiblePoints_org'][:, 0:N, :].to(torch.float32).contiguous()   # :431-433
        xyz_loss, xyz_per = chamfer_loss.get_loss(xyz_recon_FPS, visiblePoints_final)
        translation = element['translation'].to(torch.float32)
        trans_loss, trans_per = trans_distance.get_translation_error(trans_pred, translation)            # :455
        mean_dist_loss, mean_dist_per = trans_distance.get_translation_error(element_mean, translation)  # :457
        axag_loss, axag_per = angular_distance_taylor.get_rotation_error(rot_pred, element['axisangle'])  # :470-474
    return dict(xyz_recon=xyz_recon, xyz_recon_FPS=xyz_recon_FPS, rot_pred=rot_pred, trans_pred=trans_pred,
                xyz_loss=xyz_loss, xyz_loss_per_sample=xyz_per, trans_loss=trans_loss,
                trans_loss_perSample=trans_per, mean_dist_loss=mean_dist_loss,
                mean_dist_loss_perSample=mean_dist_per, axag_loss=axag_loss, axag_loss_perSample=axag_per,
                element_mean=element_mean, end_points=end_points)
