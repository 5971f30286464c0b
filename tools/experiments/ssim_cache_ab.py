"""A/B: ground-truth statistics cache of the SSIM loss on / off (igs_refine_step, cfg4's per-step workload), interleaved on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate
dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene()
cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
with torch.no_grad():
    gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]


def run(cache, steps=200):
    p = GaussianParams(raw, dev); p.spatial_sort()
    r = Refiner(p, cams, gts, bg, loss="l1_ssim", seed=7)
    r.cache_gt_stats = cache
    for _ in range(30):
        r.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        r.step()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t) / steps


for i in range(3):
    print("cache on %.4f  off %.4f ms/step" % (run(True), run(False)), flush=True)
