"""How long do the first refine steps of a process take?  (bench.py with the driver's 20 steps measures 0.29-0.30 ms per step where
200 steps measure 0.266: which steps are slow?)  Per-step wall time with a synchronise after every step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate
dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene()
cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
with torch.no_grad():
    gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
p = GaussianParams(raw, dev); p.spatial_sort()
r = Refiner(p, cams, gts, bg, loss="l1", seed=0)
torch.cuda.synchronize()
ts = []
for i in range(80):
    t = time.perf_counter(); r.step(); torch.cuda.synchronize(); ts.append(1000 * (time.perf_counter() - t))
print("per-step ms (synchronised):", " ".join("%.3f" % x for x in ts))
# and unsynchronised blocks of 10
for b in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): r.step()
    torch.cuda.synchronize(); print("block of 10: %.4f ms per step" % (100 * (time.perf_counter() - t)))
