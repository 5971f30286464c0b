"""Workload for rocprofv3: N refine steps of the bench scene (forward + L1 + backward + Adam), nothing else."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    dev = torch.device("cuda:0")
    rasterizer.NAN_CHECKS = False
    raw, cams, bg = sear_steak_like_scene(P=P)
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    ref = Refiner(GaussianParams(raw, dev), cams, gts, bg, loss="l1")
    for _ in range(steps):
        ref.step()
    torch.cuda.synchronize()
    print("done", steps)

if __name__ == "__main__":
    main()
