"""Times the UNCHANGED caller loop (tools/dropin_loop.py = infer_batch.py:279-324) on the bench scene, per variant.
usage: python tools/dropin_bench.py [steps] [variant ...]      variant = <loss>:<optimizer>:<nan 0|1>[:<losses igs|torch>[:<flags: fa (fused activations), np (no PSNR line)>]]
IGS_DROPIN_MORTON=1: the Gaussians in Morton order of their positions, as bench.py's legs have them (GaussianParams.spatial_sort).
Prints one line per variant as it finishes and a JSON list of [variant, ms] pairs at the end (bench.py runs its graph-replay legs through
this script in a CHILD process: an invalid capture ends this ROCm's hipStreamEndCapture in a segmentation fault, not an error code, and the
driver's one JSON line must not die with it)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.refine import render, DEFAULT_LRS
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate
from tools.dropin_loop import CallerModel, refine_iteration, make_losses


def setup(dev, P=200000):
    raw, cams, bg = sear_steak_like_scene(P=P)
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    return raw, cams, bg, gts


def run_variant(raw, cams, bg, gts, dev, loss, optimizer, nan, losses="igs", steps=100, warm=20, flags=""):
    """flags: "fa" = the caller uses igs_amd.activations.activate in its three activation properties, "np" = no per-iteration PSNR line."""
    rasterizer.NAN_CHECKS = bool(nan)
    gs = CallerModel(raw, dev, DEFAULT_LRS, optimizer=optimizer, fused_activations="fa" in flags)
    lf = make_losses(losses)
    kw = dict(loss=loss, losses=lf, psnr_line="np" not in flags)
    body = lambda v: refine_iteration(gs, cams[v], gts[v], bg, **kw)
    if optimizer.endswith("_capturable"):          # the loop body replayed from one hipGraph per view (igs_amd/graphs.py)
        from igs_amd.graphs import GraphedLoop
        body = GraphedLoop(body)
    for i in range(max(warm, 2 * len(cams))):
        body(i % len(cams))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        body(i % len(cams))
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / steps


def main():
    dev = torch.device("cuda:0")
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    variants = sys.argv[2:] or ["l1:fused:0", "l1:fused:1", "l1_ssim:fused:0", "l1:torch_fused:0", "l1:torch:0", "l1:fused:0:torch", "l1_ssim:fused:1",
                                "l1:fused_capturable:1", "l1_ssim:fused_capturable:1", "l1:torch_capturable:1"]
    raw, cams, bg, gts = setup(dev)
    if os.environ.get("IGS_DROPIN_MORTON") == "1":
        from igs_amd.refine import GaussianParams
        p = GaussianParams(raw, dev)
        p.spatial_sort()
        raw = {k: v.detach().clone() for k, v in p.leaves.items()}
    out = {}
    pairs = []
    for v in variants:
        f = v.split(":")
        out[v] = round(run_variant(raw, cams, bg, gts, dev, f[0], f[1], int(f[2]), f[3] if len(f) > 3 else "igs", steps=steps,
                                   flags=f[4] if len(f) > 4 else ""), 4)
        pairs.append([v, out[v]])
        print(v, out[v], flush=True)
    print(json.dumps(out))
    print("PAIRS " + json.dumps(pairs), flush=True)


if __name__ == "__main__":
    main()
