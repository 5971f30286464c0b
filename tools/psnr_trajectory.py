"""Held-out-view PSNR along the refine steps of bench.py's cfg3 workload (every 10 steps), twice from the same start: how much of the
`psnr.after` figure is trajectory noise (one view per step, float-atomic ordering) and how much is signal."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.refine import GaussianParams, Refiner, render, psnr
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate


def main():
    dev = torch.device("cuda:0")
    rasterizer.NAN_CHECKS = False
    raw, cams, bg = sear_steak_like_scene(held_out=True)
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    cams, test_cam = cams[:-1], cams[-1]
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
        gt_test = render(activate(gt_raw), test_cam, bg)["images_pred"].clone()
    for run in range(2):
        p = GaussianParams(raw, dev); p.spatial_sort()
        ref = Refiner(p, cams, gts, bg, loss="l1", seed=0)
        traj = []
        for it in range(401):
            if it % 10 == 0:
                with torch.no_grad():
                    ph = float(psnr(render(p.activated(), test_cam, bg)["images_pred"], gt_test))
                    pt = sum(float(psnr(render(p.activated(), c, bg)["images_pred"], g)) for c, g in zip(cams, gts)) / len(cams)
                traj.append((it, round(ph, 2), round(pt, 2)))
            ref.step()
        print("run %d (step, held-out PSNR, mean train PSNR):" % run, traj)


if __name__ == "__main__":
    main()
