"""What an unchanged IGS caller sees: the step driven through the reference's Python API (GaussianRasterizer autograd Function,
torch activations, torch loss, loss.backward()), against the library's fused single-call step.  Same scene as bench.py."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate

def run(ref, steps=100, warm=20):
    for _ in range(warm):
        ref.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ref.step()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / steps

def main():
    dev = torch.device("cuda:0")
    raw, cams, bg = sear_steak_like_scene()
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    out = {}
    for name, kw, nan in (("python_api_autograd_nan_checks_on", dict(native=False), True),
                          ("python_api_autograd", dict(native=False), False),
                          ("python_api_autograd_fused_activations", dict(native=False), False),
                          ("c_abi_unfused", dict(native=True, fused=False), False),
                          ("igs_refine_step", dict(native=True, fused=True), False)):
        rasterizer.NAN_CHECKS = nan
        for loss in ("l1", "l1_ssim"):
            p = GaussianParams(raw, dev); p.spatial_sort()
            r = Refiner(p, cams, gts, bg, loss=loss, **kw)
            r.fused_activations = name.endswith("fused_activations")
            r.direct_adam = True         # autograd path: gradients straight from autograd into the fused Adam (set_to_none semantics, infer_batch.py:324)
            out["%s/%s" % (name, loss)] = round(run(r), 4)
    print(json.dumps(out))

if __name__ == "__main__":
    main()
