"""Timing probe (round 3) for the two-stream plan of DESIGN.md section 8: how much of the optimiser's streaming kernel hides under the
NEXT view's preprocess + tile sort when it runs on a second stream?  TIMING ONLY -- mode "nowait" lets the next step start before the
update has landed (a data race on purpose; all learning rates are 0 so the workload stays the one measured).

  fused    : igs_refine_step as bench.py runs it (one stream, Adam inside the per-Gaussian kernel)
  serial   : gradients-only step, then igs_adam_exchange_step (SH Adam from the view's colour gradients + the small groups) behind it
  side     : the same, the optimiser kernel on a side stream that waits for the step and that the next step waits for (no overlap: control)
  nowait   : the optimiser kernel on the side stream from the colour event on; the next step does NOT wait for it

python tools/experiments/two_stream_probe.py    (one GPU)"""
import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from igs_amd import _cabi, rasterizer as _rast
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, activate, perturbed_copy


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    raw, cams, bg = sear_steak_like_scene(P=200000, n_cams=10, width=1352, height=1014)
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    L = _cabi.lib()
    for mode in ("fused", "serial", "side", "nowait"):
        p = GaussianParams(raw, dev); p.spatial_sort()
        r = Refiner(p, cams, gts, bg, loss="l1", seed=3)
        P = p.P
        gc = torch.zeros((1, P, 3), dtype=torch.float32, device=dev)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
        done = torch.cuda.Event()
        side = torch.cuda.Stream(device=dev)
        lrs = dict(p.lrs) if mode == "fused" else {k: 0.0 for k in p.lrs}
        sp = p.spans
        b1, b2 = p.betas

        def adam(stream, view):
            campos = (C.c_float * 3)(*[float(x) for x in cams[view].camera_center.reshape(3).tolist()]) if view not in adam.cache else adam.cache[view]
            adam.cache[view] = campos
            t = p.step_count + 1
            rc = L.igs_adam_exchange_step(stream, P, 3, 16, 1, C.cast(campos, C.c_void_p), gc.data_ptr(), 0.0, p.flat.data_ptr(),
                                          p.exp_avg.data_ptr(), p.exp_avg_sq.data_ptr(), p.grad.data_ptr(), sp["xyz"][0], sp["rotation"][0],
                                          sp["shs"][0], sp["opacity"][0], sp["scaling"][0], lrs["xyz"], lrs["rotation"], lrs["shs"],
                                          lrs["opacity"], lrs["scaling"], b1, b2, p.eps, 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t))
            _rast._check(rc, "igs_adam_exchange_step")
            p.step_count += 1
        adam.cache = {}

        def one():
            view = r._next_view()
            main = torch.cuda.current_stream(dev)
            if mode == "fused":
                r._fused_step(cams[view], gts[view])
            elif mode == "serial":
                r._fused_step(cams[view], gts[view], grads_only=True, color_out=gc[0])
                adam(main.cuda_stream, view)
            elif mode == "side":
                r._fused_step(cams[view], gts[view], grads_only=True, color_out=gc[0])
                side.wait_stream(main)
                adam(side.cuda_stream, view)
                main.wait_stream(side)
            else:
                r._fused_step(cams[view], gts[view], grads_only=True, color_out=gc[0], color_event=ev)
                side.wait_event(ev)
                adam(side.cuda_stream, view)
        for _ in range(30):
            one()
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 200
        for _ in range(n):
            one()
        torch.cuda.synchronize()
        print("%-7s: %.4f ms per step" % (mode, (time.perf_counter() - t) * 1e3 / n), flush=True)


if __name__ == "__main__":
    main()
