"""N views per optimiser step against the reference's one view per step (hard part 4).

From the same start, same loss (0.8 L1 + 0.2 (1 - SSIM)), same learning rates, same seeded view permutation:
  A  the reference's schedule: 1 view per Adam step (infer_batch.py:279-288), S steps;
  B  N-view steps: the gradients of N different views averaged (each view's loss pre-scaled by 1 / N, what N ranks do), ONE Adam step --
     run on ONE GPU by accumulating N gradients-only calls of igs_refine_step (no collectives); k steps for several k.
Prints the held-out PSNR of each; `schedules()` is what tests/test_gpu_parity.py asserts on.
usage: python tools/psnr_schedules.py [N=8] [S=50]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd.refine import GaussianParams, Refiner, render, psnr
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate


def build(dev, P=200000, width=1352, height=1014, focal=730.0, n_cams=10):
    raw, cams, bg = sear_steak_like_scene(P=P, n_cams=n_cams, width=width, height=height, focal=focal, held_out=True)
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    cams, test_cam = cams[:-1], cams[-1]
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
        gt_test = render(activate(gt_raw), test_cam, bg)["images_pred"].clone()
    return raw, cams, bg, gts, test_cam, gt_test


def held_out(p, test_cam, bg, gt_test):
    with torch.no_grad():
        return float(psnr(render(p.activated(), test_cam, bg)["images_pred"], gt_test))


def single_view(raw, cams, bg, gts, dev, steps, loss="l1_ssim", seed=0):
    p = GaussianParams(raw, dev); p.spatial_sort()
    r = Refiner(p, cams, gts, bg, loss=loss, seed=seed)
    for _ in range(steps):
        r.step()
    return p


def n_view(raw, cams, bg, gts, dev, steps, N, loss="l1_ssim", seed=0, at=(), lr_scale=1.0):
    """`steps` optimiser steps, each on the averaged gradient of N views.  Returns the store and {k: callback result} for k in `at`."""
    from igs_amd.refine import DEFAULT_LRS
    p = GaussianParams(raw, dev, lrs={k: v * lr_scale for k, v in DEFAULT_LRS.items()}); p.spatial_sort()
    r = Refiner(p, cams, gts, bg, loss=loss, world_size=N, rank=0, seed=seed)      # (world_size: the 1 / N loss scale and the N picks per step)
    acc = torch.zeros_like(p.grad)
    marks = {}
    for s in range(steps):
        r._next_view()
        acc.zero_()
        for v in r.last_picks:                       # what the N ranks of a step render, one after the other on this GPU
            r._view = v
            r._fused_step(cams[v], gts[v], grads_only=True)
            acc += p.grad
        p.grad.copy_(acc)
        p.adam_step()
        if (s + 1) in at:
            marks[s + 1] = at[s + 1](p) if isinstance(at, dict) else None
    return p, marks


def schedules(dev, N=8, S=50, scene=None, ks=None):
    raw, cams, bg, gts, test_cam, gt_test = scene or build(dev)
    ho = lambda p: held_out(p, test_cam, bg, gt_test)
    p0 = GaussianParams(raw, dev)
    out = {"start": ho(p0), "single_view_%d_steps" % S: ho(single_view(raw, cams, bg, gts, dev, S))}
    ks = ks or sorted({math.ceil(S / N), math.ceil(S / 4), math.ceil(S / 2), S})
    _, marks = n_view(raw, cams, bg, gts, dev, max(ks), N, at={k: ho for k in ks})
    for k in ks:
        out["%d_view_%d_steps" % (N, k)] = marks[k]
    return out


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    res = schedules(torch.device("cuda:0"), N, S, ks=sorted({math.ceil(S / N), 10, 13, 17, 25, 35, S}))
    for k, v in res.items():
        print("%-28s held-out PSNR %.2f dB" % (k, v))
    # learning rates scaled with the number of views per step (Adam: a step moves a parameter by ~lr whatever N is)
    scene = build(torch.device("cuda:0"))
    raw, cams, bg, gts, test_cam, gt_test = scene
    for ls in (1.5, 2.0, 3.0):
        ks = sorted({math.ceil(S / N), 10, 13, 17, 25})
        _, marks = n_view(raw, cams, bg, gts, torch.device("cuda:0"), max(ks), N, at={k: (lambda p: held_out(p, test_cam, bg, gt_test)) for k in ks}, lr_scale=ls)
        print("lr x %.1f: " % ls + ", ".join("%d steps %.2f dB" % (k, marks[k]) for k in ks))
