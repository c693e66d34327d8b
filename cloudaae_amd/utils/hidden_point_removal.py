"""Spherical flip + hidden point removal -- mirror of the reference's
utils/hidden_point_removal.py (sphericalFlip :6-24, convexHull :27-43, hidden_point_removal
:46-48 and the `_org` twins :51-73), batched on the GPU.  The reference calls qhull through
tf.py_func, twice per sample under the GIL; here one launch tests every point of every cloud
(csrc/synth.hip)."""
import torch

from .. import _lib
from .._lib import ptr, require, stream


def _flip(a, b, center, param):
    a = a.to(torch.float32).contiguous()
    B, na, _ = a.shape
    nb = 0
    if b is not None:
        b = b.to(torch.float32).contiguous()
        nb = b.shape[1]
    flipped = torch.empty((B, na + nb + 1, 3), dtype=torch.float32, device=a.device)
    org = torch.empty_like(flipped)
    _lib.check(_lib.lib().cloudaae_spherical_flip(B, na, ptr(a), nb, ptr(b), ptr(center.contiguous())
                                                  if center is not None else None, float(param), ptr(flipped),
                                                  ptr(org), stream()), "cloudaae_spherical_flip")
    return flipped, org


def sphericalFlip(x, center, param):
    """points = concat(model_xyz_rot_trans, occluder) (:7); adds 'flippedPoints', 'orgPoints'
    (both [B, 2048+400+1, 3], last row = the viewpoint)."""
    x['flippedPoints'], x['orgPoints'] = _flip(x['model_xyz_rot_trans'], x['occluder'], center, float(param))
    return x


def sphericalFlip_org(x, center, param):
    """The same without the occluder (:51-68): 'flippedPoints_org', 'orgPoints_org'."""
    x['flippedPoints_org'], x['orgPoints_org'] = _flip(x['model_xyz_rot_trans'], None, center, float(param))
    return x


def convexHull(points, orgPoints, seed=0, return_ids=False, rows=None, return_src=False):
    """(:27-43) points [B,n+1,3] flipped (+viewpoint row); returns (visiblePoints [B,n+1,3],
    num_vis_point [B] int64): rows [0,num_vis) are the visible points in ascending index, the
    rest random re-draws of visible points.  rows: another number of output rows (default n+1, the
    reference's), filled by the same rule.  return_src: also row_src [B,rows] int32 -- the row < num_vis every
    output row equals (what tf_nndistance.nn_distance(..., distinct2=) takes)."""
    points = points.to(torch.float32).contiguous()
    orgPoints = orgPoints.to(torch.float32).contiguous()
    B, n1, _ = points.shape
    require(orgPoints.shape == points.shape, "convexHull: points and orgPoints differ in shape")
    rows = n1 if rows is None else int(rows)
    vis = torch.empty((B, rows, 3), dtype=torch.float32, device=points.device)
    num = torch.empty((B,), dtype=torch.int64, device=points.device)
    ids = torch.empty((B, rows), dtype=torch.int32, device=points.device) if return_ids else None
    src = torch.empty((B, rows), dtype=torch.int32, device=points.device) if return_src else None
    ws = torch.empty(int(_lib.lib().cloudaae_hpr_workspace_bytes(B, n1)), dtype=torch.uint8, device=points.device)
    _lib.check(_lib.lib().cloudaae_hidden_point_removal_rows(B, n1, ptr(points), ptr(orgPoints), int(seed), rows,
                                                             ptr(vis), ptr(num), ptr(ids), ptr(src), ptr(ws), stream()),
               "cloudaae_hidden_point_removal")
    out = (vis, num)
    if return_ids:
        out = out + (ids,)
    if return_src:
        out = out + (src,)
    return out


def hidden_point_removal(x, seed=0, rows=None):
    x['visiblePoints'], x['num_vis_point'] = convexHull(x['flippedPoints'], x['orgPoints'], seed, rows=rows)
    return x


def hidden_point_removal_org(x, seed=0, rows=None):
    # (+ 'visiblePoints_org_src': the row each row of the Chamfer target equals -- an extra key the train step hands
    #  to the nearest-neighbour search, which then looks at the distinct points only.  The key describes THESE rows:
    #  a pipeline that shuffles, subsamples or perturbs 'visiblePoints_org' afterwards must drop the key (or the
    #  Chamfer loss is computed against rows that no longer are what the key says; CLOUDAAE_NN_PREFIX_VERIFY=1 makes
    #  cloudaae_nn_distance_prefix check every copy against its original and return NaN distances on a mismatch))
    x['visiblePoints_org'], x['num_vis_point_org'], x['visiblePoints_org_src'] = convexHull(
        x['flippedPoints_org'], x['orgPoints_org'], seed + 1, rows=rows, return_src=True)
    return x
