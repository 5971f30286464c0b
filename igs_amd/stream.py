"""Frame-by-frame streaming refinement (the per-frame loop of `infer_batch.py:245-357` around the rasterizer; AGM-Net, which
predicts the motion between key frames in the reference, is outside this repository's scope -- SURVEY.md 2).

For every frame of a stream the Gaussians of the previous frame are refined against that frame's training views for
`refine_iterations` steps (one random view per step without replacement, loss 0.8 L1 + 0.2 (1 - SSIM), Adam with a fresh state per
frame as `load_fromstream` does, optional densify-and-prune), then evaluated (PSNR as `infer_batch.py:350-353`).

`SyntheticStream` stands in for the N3DV frames (absent): a target Gaussian set that drifts from frame to frame
(xyz += N(0, 0.01) on the dynamic-bbox subset), rendered to ground-truth images by the same renderer.
"""
import time

import torch

from .refine import GaussianParams, Refiner, render, psnr
from .scenes import activate, SEAR_STEAK_BBOX


class SyntheticStream:
    """Ground truth for frame f = render of a drifting copy of the scene's Gaussians."""

    def __init__(self, raw, cams, bg, device, motion_sigma=0.01, seed=7, start_sigma=0.02):
        self.device, self.cams, self.bg = device, cams, bg
        self.gen = torch.Generator().manual_seed(seed)
        self.target = {k: v.clone().to(device) for k, v in raw.items()}
        lo, hi = torch.tensor(SEAR_STEAK_BBOX[0]), torch.tensor(SEAR_STEAK_BBOX[1])
        inside = ((raw["xyz"] >= lo) & (raw["xyz"] <= hi)).all(dim=1)
        self.dynamic = inside.to(device) if bool(inside.any()) else torch.ones(raw["xyz"].shape[0], dtype=torch.bool, device=device)
        self.motion_sigma = motion_sigma
        if start_sigma:
            self.target["xyz"] = self.target["xyz"] + (torch.randn(raw["xyz"].shape, generator=self.gen) * start_sigma).to(device)

    def next_frame(self):
        """Moves the dynamic subset and returns the frame's ground-truth images (one per camera)."""
        step = (torch.randn(self.target["xyz"].shape, generator=self.gen) * self.motion_sigma).to(self.device)
        self.target["xyz"] = self.target["xyz"] + step * self.dynamic.unsqueeze(1).float()
        with torch.no_grad():
            return [render(activate(self.target), c, self.bg)["images_pred"].clone() for c in self.cams]


def run_stream(raw, cams, bg, frames, refine_iterations=50, device="cuda", loss="l1_ssim", densify=None, source=None, lrs=None,
               world_size=1, rank=0, spatial_sort=True, log=None, lambda_depth_normal=0.0):
    """Refines `raw` through `frames` frames; returns a list of per-frame dicts {psnr_before, psnr_after, seconds, num_gaussians}.
    `source.next_frame()` supplies each frame's ground-truth images (default: SyntheticStream)."""
    dev = torch.device(device)
    cams = [c.to(dev) for c in cams]
    bg = bg.to(dev)
    source = source or SyntheticStream(raw, cams, bg, dev)
    cur = {k: v.clone() for k, v in raw.items()}
    out = []
    for f in range(frames):
        gts = source.next_frame()
        # load_fromstream (gaussian_model.py:265-348): new leaves and a NEW optimizer for every frame
        params = GaussianParams(cur, dev, lrs=lrs)
        if spatial_sort:
            params.spatial_sort()
        ref = Refiner(params, cams, gts, bg, loss=loss, world_size=world_size, rank=rank, seed=f, densify=densify,
                      lambda_depth_normal=lambda_depth_normal)
        ref.start_frame()
        with torch.no_grad():
            p0 = float(psnr(render(params.activated(), cams[0], bg)["images_pred"], gts[0]))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(refine_iterations):
            ref.step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        with torch.no_grad():
            p1 = float(psnr(render(params.activated(), cams[0], bg)["images_pred"], gts[0]))
        rec = dict(frame=f, psnr_before=p0, psnr_after=p1, seconds=dt, num_gaussians=params.P,
                   gaussians_per_s=params.P * refine_iterations * world_size / dt)
        out.append(rec)
        if log:
            log(rec)
        cur = {k: v.detach().clone() for k, v in params.leaves.items()}          # convert2stream: next frame starts from here
    return out
