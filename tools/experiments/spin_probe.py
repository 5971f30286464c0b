"""How long does a fresh process need before the SAME 20 refine steps (steps 6-25 from the same start) run at their steady speed?
python tools/experiments/spin_probe.py     (one GPU; prints the wall clock since the first GPU work and the per-step time of every repeat)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, activate, perturbed_copy
dev = torch.device("cuda", 0)
raw, cams, bg = sear_steak_like_scene(P=200000, n_cams=10, width=1352, height=1014)
cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
t00 = time.perf_counter()
with torch.no_grad():
    gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]


def run():
    p = GaussianParams(raw, dev); p.spatial_sort(); r = Refiner(p, cams, gts, bg, loss="l1", seed=3)
    for _ in range(5): r.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): r.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 20 * 1e3


busy = 0.0
for i in range(14):
    ms = run()
    print("repeat %2d at %.2f s: %.4f ms per step over steps 6-25" % (i, time.perf_counter() - t00, ms), flush=True)
    if i == 8:
        time.sleep(2.0); print("(2 s idle)")
