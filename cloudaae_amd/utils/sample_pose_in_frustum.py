"""get_frustum -- mirror of the reference's utils/sample_pose_in_frustum.py:42-70 (host-side
constants only: four scalars and eight corner points)."""
import math

import torch


def get_frustum(vertical_fov, nearDist, farDist, ratio):
    """Returns (frustum_corners [3,8], Hnear, Wnear, Hfar, Wfar).
    NOTE (kept on purpose): the reference applies tf.math.tan to
    vertical_fov/2 = 22.5 *as radians* (it never converts degrees), so for ycbv
    Hnear = 2*tan(22.5 rad)*0.5 = 0.55785 and Wnear = 0.71901."""
    t = math.tan(float(vertical_fov) / 2)
    Hnear = 2 * t * nearDist
    Wnear = Hnear * ratio
    Hfar = 2 * t * farDist
    Wfar = Hfar * ratio
    cam_direction = torch.tensor([0., 0., 1.])
    up = torch.tensor([0., 1., 0.])
    right = torch.linalg.cross(up, cam_direction)
    fc = cam_direction * farDist
    nc = cam_direction * nearDist
    corners = torch.stack([fc + up * Hfar / 2 - right * Wfar / 2, fc + up * Hfar / 2 + right * Wfar / 2,
                           fc - up * Hfar / 2 - right * Wfar / 2, fc - up * Hfar / 2 + right * Wfar / 2,
                           nc + up * Hnear / 2 - right * Wnear / 2, nc + up * Hnear / 2 + right * Wnear / 2,
                           nc - up * Hnear / 2 - right * Wnear / 2, nc - up * Hnear / 2 + right * Wnear / 2], dim=1)
    return corners, Hnear, Wnear, Hfar, Wfar
