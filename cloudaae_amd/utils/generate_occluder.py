"""Random spherical occluders -- mirror of the reference's utils/generate_occluder.py:38-81,
batched and generated on the GPU (cloudaae_random_spherical_occluder)."""
import torch

from .. import _lib
from .._lib import ptr, require, stream
from .sample_pose_in_frustum import get_frustum

_CAMERAS = {  # generate_occluder.py:40-52
    'linemod': dict(vertical_fov=45., nearDist=0.4, farDist=1.5, ratio=57.5 / 45.),
    'ycbv': dict(vertical_fov=45., nearDist=0.5, farDist=1., ratio=58. / 45.),
}


def get_random_spherical_occluder(x, dataset, seed=0):
    """x: dict with 'translation' [B,3] (device).  Adds x['occluder'] [B,400,3] (two Gaussian
    blobs of 200 points, sigma 0.01, between the camera's near plane and the object) and
    x['frustum_corners']."""
    require(dataset in _CAMERAS, "dataset must be 'linemod' or 'ycbv'")
    cam = _CAMERAS[dataset]
    corners, Hnear, Wnear, _, _ = get_frustum(cam['vertical_fov'], cam['nearDist'], cam['farDist'], cam['ratio'])
    t = x['translation'].to(torch.float32).contiguous()
    B = t.shape[0]
    occ = torch.empty((B, 400, 3), dtype=torch.float32, device=t.device)
    _lib.check(_lib.lib().cloudaae_random_spherical_occluder(B, 200, ptr(t), float(Wnear), float(Hnear),
                                                             float(cam['nearDist']), 0.01, int(seed), ptr(occ),
                                                             stream()), "cloudaae_random_spherical_occluder")
    x['occluder'] = occ
    x['frustum_corners'] = corners
    return x
