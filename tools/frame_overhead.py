"""Where a stream frame's time goes outside its refine iterations (bench.py --config cfg4 / cfg5: per-frame set-up is inside the
timed region): parameter store + optimiser, Morton sort, Refiner, the two PSNR renders, the clone that hands the result on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igs_amd.refine import GaussianParams, Refiner, render, psnr
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate

dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene()
cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
with torch.no_grad():
    gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
cur = {k: v.clone() for k, v in raw.items()}
T = {}
def tick(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + 1000 * (time.perf_counter() - t); return r
for f in range(6):
    if f == 1: T.clear()
    params = tick("GaussianParams", lambda: GaussianParams(cur, dev))
    tick("spatial_sort", params.spatial_sort)
    ref = tick("Refiner + start_frame", lambda: (lambda r: (r.start_frame(), r)[1])(Refiner(params, cams, gts, bg, loss="l1_ssim", seed=f)))
    tick("psnr render x2", lambda: [psnr(render(params.activated(), cams[0], bg)["images_pred"], gts[0]) for _ in range(2)])
    tick("50 steps", lambda: [ref.step() for _ in range(50)])
    cur = tick("clone", lambda: {k: v.detach().clone() for k, v in params.leaves.items()})
for k, v in T.items():
    print("%-24s %.3f ms per frame" % (k, v / 5))
