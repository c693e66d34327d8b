"""CPU restatement of the reference's on-line synthesis -- TEST INFRASTRUCTURE ONLY.
numpy fp32 arithmetic in the reference's op order + scipy.spatial.ConvexHull (the qhull the
reference itself calls, utils/hidden_point_removal.py:4,32).  Random parts (occluder draws, the
padding re-draws) are NOT restated: tests feed explicit occluders and check the padded rows by
membership."""
import math

import numpy as np
from scipy.spatial import ConvexHull


def exponential_map(axag):
    """losses/angular_distance_taylor.py:30-66, float64, one vector."""
    a = np.asarray(axag, np.float64)
    ss = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    tsq = (a * a).sum()
    if tsq < 1e-2:
        t1 = 1 - tsq / 6 + tsq ** 2 / 120 - tsq ** 3 / 5040 + tsq ** 4 / 362880
        t2 = 0.5 - tsq / 24 + tsq ** 2 / 720 - tsq ** 3 / 40320 + tsq ** 4 / 3628800
    else:
        th = math.sqrt(tsq)
        t1, t2 = math.sin(th) / th, (1 - math.cos(th)) / tsq
    return np.eye(3) + t1 * ss + t2 * (ss @ ss)


def transform_object_model(model_xyz, axisangle, translation):
    """train_cloudAAE_ycbv.py:79-93 for one sample: R = float32(exp_map(float64 axis-angle));
    out = p R^T + t, three-term dot product left to right in fp32 (Eigen's order is unpinned)."""
    R = exponential_map(np.asarray(axisangle, np.float32).astype(np.float64)).astype(np.float32)
    p = np.asarray(model_xyz, np.float32)
    t = np.asarray(translation, np.float32)
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((p[:, 0] * R[r, 0] + p[:, 1] * R[r, 1]) + p[:, 2] * R[r, 2]) + t[r]
    return out


def spherical_flip(points, param=0.8 * math.pi):
    """utils/hidden_point_removal.py:6-24 for one cloud (center = 0): returns (flipped, org), both
    with the zero viewpoint row appended."""
    p = np.asarray(points, np.float32)
    norm = np.sqrt((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]).astype(np.float32)
    R = np.float32(norm.max() * np.float32(np.float32(10.0) ** np.float32(param)))
    tmp = (np.float32(2) * (R - norm))[:, None] * p
    f = (tmp / norm[:, None] + p).astype(np.float32)
    z = np.zeros((1, 3), np.float32)
    return np.concatenate([f, z], 0), np.concatenate([p, z], 0)


def convex_hull_visible(points):
    """utils/hidden_point_removal.py:27-36 for one cloud: the visible ids (ascending) after the
    reference's two `[:-1]`, and all hull vertex ids."""
    hull = ConvexHull(np.asarray(points, np.float64))
    flag = np.zeros(len(points), int)
    flag[hull.vertices[:-1]] = 1
    visible = np.where(flag == 1)[0][:-1]
    return visible, np.sort(hull.vertices)


def get_frustum_near(vertical_fov=45., near=0.5, ratio=58. / 45.):
    """utils/sample_pose_in_frustum.py:45-46 (tan of 22.5 *radians*)."""
    h = 2 * math.tan(vertical_fov / 2) * near
    return h, h * ratio
