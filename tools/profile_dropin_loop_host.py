"""Host side of the UNCHANGED caller loop (tools/dropin_loop.py): enqueue time vs device span per step, then cProfile.
usage: python tools/profile_dropin_loop_host.py [variant]     (variant as tools/dropin_bench.py, default l1:fused:0)"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.refine import DEFAULT_LRS
from tools.dropin_bench import setup
from tools.dropin_loop import CallerModel, refine_iteration, make_losses


def main():
    dev = torch.device("cuda:0")
    f = (sys.argv[1] if len(sys.argv) > 1 else "l1:fused:0").split(":")
    rasterizer.NAN_CHECKS = bool(int(f[2]))
    raw, cams, bg, gts = setup(dev)
    gs = CallerModel(raw, dev, DEFAULT_LRS, optimizer=f[1])
    lf = make_losses("igs")
    it = lambda i: refine_iteration(gs, cams[i % len(cams)], gts[i % len(cams)], bg, loss=f[0], losses=lf)
    for i in range(30):
        it(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(200):
        it(i)
    e1.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("200 steps: host enqueue %.4f ms/step, device span %.4f ms/step" % (1000 * t_host / 200, e0.elapsed_time(e1) / 200))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(200):
        it(i)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
    print(s.getvalue())


if __name__ == "__main__":
    main()
