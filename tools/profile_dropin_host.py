"""Where does the host time of the UNCHANGED-caller path go?  cProfile over refine steps driven through the reference's Python API
(GaussianRasterizer autograd Function, torch activations / loss / backward; Refiner(native=False)); also the GPU-side time of the
same steps, to tell host-bound from device-bound."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate


def main():
    dev = torch.device("cuda:0")
    rasterizer.NAN_CHECKS = False
    raw, cams, bg = sear_steak_like_scene()
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    p = GaussianParams(raw, dev); p.spatial_sort()
    ref = Refiner(p, cams, gts, bg, loss="l1", native=False)
    ref.direct_adam = True
    for _ in range(30):
        ref.step()
    torch.cuda.synchronize()
    # device time of a step: events around 100 steps, host far ahead?  (if host-bound the two agree)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(200):
        ref.step()
    e1.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("200 steps: host enqueue %.4f ms/step, device span %.4f ms/step" % (1000 * t_host / 200, e0.elapsed_time(e1) / 200))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        ref.step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue())


if __name__ == "__main__":
    main()
