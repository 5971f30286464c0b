"""Probe for GraphedLoop crashes: python tools/debug/graph_loop_probe.py <P> <width> <height> <loss> <optimizer> <losses> <psnr 0|1> <sync 0|1>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import faulthandler; faulthandler.enable()
import torch
from igs_amd.graphs import GraphedLoop
from igs_amd.refine import render, DEFAULT_LRS
from igs_amd.scenes import perturbed_copy, sear_steak_like_scene, activate
from tools.dropin_loop import CallerModel, refine_iteration, make_losses

P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
loss, optimizer, losses, psnr, sync = sys.argv[4], sys.argv[5], sys.argv[6], int(sys.argv[7]), int(sys.argv[8])
dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene(P=P, n_cams=2, width=W, height=H, focal=90.0 * W / 160, scale_mean=-2.0 if W < 500 else -4.0)
cams = [c.to(dev) for c in cams]; bgd = bg.to(dev)
gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw, sigma=0.03).items()}
with torch.no_grad():
    gts = [render(activate(gt_raw), c, bgd)["images_pred"].clone() for c in cams]
lf = make_losses(losses, ssim_fn=None)
gg = CallerModel(raw, dev, DEFAULT_LRS, optimizer=optimizer)
loop = GraphedLoop(lambda v: refine_iteration(gg, cams[v], gts[v], bgd, loss=loss, losses=lf, psnr_line=bool(psnr)))
for i in range(9):
    pkg, total = loop(i % 2)
    if sync:
        print(i, float(total), flush=True)
torch.cuda.synchronize()
print("ok", float(total), flush=True)
