"""Synthetic stand-ins for the benchmark configurations.

The reference ships neither datasets nor camera files (no `dataset/`, `demo_data/`), so
cfg-1 and the "sear_steak-like" cfg-2/3 scenes are generated from fixed seeds with the
distributions recorded here.  Parameters are returned RAW (pre-activation), in the
layout the refine loop optimises (igs/models/gaussian_model.py:265-348):
xyz [P,3], shs [P,16,3], opacity logit [P,1], log-scale [P,3], rotation [P,4] (w,x,y,z).
"""
import math

import torch

from .camera import Camera, look_at_c2w, focal2fov

SEAR_STEAK_BBOX = ((-14.0, -3.0, 9.0), (7.0, 8.0, 17.0))   # configs/bbox.json, key "sear_steak"


def activate(raw):
    """Activations applied by the caller, outside the rasterizer (gaussian_model.py:90-127)."""
    return dict(
        means3D=raw["xyz"],
        shs=raw["shs"],
        opacities=torch.sigmoid(raw["opacity"]),
        scales=torch.exp(raw["scaling"]),
        rotations=torch.nn.functional.normalize(raw["rotation"]),
    )


def _sh(gen, P):
    shs = torch.randn(P, 16, 3, generator=gen) * 0.05
    shs[:, 0, :] = torch.randn(P, 3, generator=gen) / 0.2821 * 0.3
    return shs


def cfg1_scene(P=10000, seed=0, size=256):
    """cfg-1: P random Gaussians in [-1.5,1.5]^3, one camera at (0,0,-5) looking +z, FoV 50 deg, size x size."""
    gen = torch.Generator().manual_seed(seed)
    raw = dict(
        xyz=(torch.rand(P, 3, generator=gen) * 3.0 - 1.5),
        scaling=(torch.rand(P, 3, generator=gen) * 2.0 - 4.0),
        rotation=torch.randn(P, 4, generator=gen),
        opacity=torch.randn(P, 1, generator=gen) * 1.5,
    )
    raw["shs"] = _sh(gen, P)
    c2w = torch.eye(4)
    c2w[2, 3] = -5.0
    fov = math.radians(50.0)
    cam = Camera.from_c2w(c2w, (fov, fov), (size, size))
    return raw, [cam], torch.zeros(3)


def sear_steak_like_scene(P=200000, seed=1, n_cams=10, width=1352, height=1014, focal=730.0, scale_mean=-4.0, held_out=False):
    """cfg-2/3 stand-in: half the Gaussians uniform in the sear_steak dynamic bbox, half on a background
    shell at distance 15-40 m in front of the rig; `n_cams` cameras on a +-20 deg arc around the bbox centre,
    starting at the origin, fx = fy = `focal` px (N3DV-like half resolution; builder's choice).
    `scale_mean`: mean of the log-scales (SURVEY 8d prescribes -4.0: ~1 px splats, R/P = 2.6; -3.0 gives the "dense" diagnostic
    scene, R in the millions as in SURVEY 8d's own example).  `held_out=True` appends one more camera half-way between the two
    middle training cameras (the evaluation view of bench.py's PSNR figures; it is never trained on)."""
    gen = torch.Generator().manual_seed(seed)
    lo = torch.tensor(SEAR_STEAK_BBOX[0])
    hi = torch.tensor(SEAR_STEAK_BBOX[1])
    n_in = P // 2
    n_bg = P - n_in
    xyz_in = lo + (hi - lo) * torch.rand(n_in, 3, generator=gen)
    # shell: directions inside a cone of +-50 deg (horizontal) / +-40 deg (vertical) about +z, radius 15..40
    az = (torch.rand(n_bg, generator=gen) * 2 - 1) * math.radians(50.0)
    el = (torch.rand(n_bg, generator=gen) * 2 - 1) * math.radians(40.0)
    rad = 15.0 + 25.0 * torch.rand(n_bg, generator=gen)
    xyz_bg = torch.stack([rad * torch.sin(az) * torch.cos(el), rad * torch.sin(el), rad * torch.cos(az) * torch.cos(el)], 1)
    perm = torch.randperm(P, generator=gen)
    raw = dict(
        xyz=torch.cat([xyz_in, xyz_bg])[perm].contiguous(),
        scaling=(torch.randn(P, 3, generator=gen) * 0.8 + scale_mean).clamp(-7.0, -1.0),
        rotation=torch.randn(P, 4, generator=gen),
        opacity=torch.randn(P, 1, generator=gen) * 2.0 + 0.5,
    )
    raw["shs"] = _sh(gen, P)
    centre = 0.5 * (lo + hi)
    fov = (focal2fov(focal, width), focal2fov(focal, height))
    cams = []
    angles = [-20.0 + 40.0 * (i / max(1, n_cams - 1)) for i in range(n_cams)]
    if held_out:
        angles.append(-20.0 + 40.0 * ((max(0, n_cams // 2 - 1) + 0.5) / max(1, n_cams - 1)))
    for ang in angles:
        th = math.radians(ang)
        # rotate the origin about the vertical axis through the bbox centre
        d = -centre
        eye = centre + torch.tensor([math.cos(th) * d[0] + math.sin(th) * d[2], d[1],
                                     -math.sin(th) * d[0] + math.cos(th) * d[2]])
        cams.append(Camera.from_c2w(look_at_c2w(eye, centre), fov, (height, width)))
    return raw, cams, torch.zeros(3)


def perturbed_copy(raw, sigma=0.02, seed=123):
    """Ground-truth generator for the refine loop: the same scene with xyz + N(0, sigma)."""
    gen = torch.Generator().manual_seed(seed)
    out = {k: v.clone() for k, v in raw.items()}
    out["xyz"] = out["xyz"] + torch.randn(out["xyz"].shape, generator=gen) * sigma
    return out
