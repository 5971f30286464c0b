"""Where does a preprocess wave spend its life?  Debug build only:
    IGS_EXTRA_FLAGS=-DPRE_TIMELINE python -c "import igs_amd.build as b; b.build()" && python tools/debug/preprocess_timeline.py
Every wave stamps the 100 MHz clock at the marks of its chain (preprocess.hip, TL(k)); this prints, per role, the time of each mark
since the first wave of the launch started (median / 90 % / last wave), for the waves that get that far."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from igs_amd import _cabi
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, activate, perturbed_copy

MARKS, WAVES = 10, 16384
NAMES = ["start", "xyz loaded, view test", "EWA (+ eigen-solver in the record role)", "plane fit, radius, rectangle", "SH -> RGB",
         "record stores issued / culled waves rejoin", "count scan + atomic", "workgroup prefix + bounding box", "binning done"]


def main():
    dev = torch.device("cuda", 0)
    raw, cams, bg = sear_steak_like_scene(P=200000, n_cams=10, width=1352, height=1014)
    cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    p = GaussianParams(raw, dev); p.spatial_sort()
    r = Refiner(p, cams, gts, bg, loss="l1", seed=3)
    for _ in range(40):
        r.step()
    torch.cuda.synchronize()
    L = _cabi.lib()
    buf = (C.c_ulonglong * (MARKS * WAVES))()
    rc = L.igs_debug_preprocess_timeline(buf, MARKS * WAVES)
    assert rc == 0, rc
    t = np.frombuffer(buf, dtype=np.uint64).reshape(WAVES, MARKS).astype(np.int64)
    nw = 4 * 2 * ((p.P + 255) // 256)
    t = t[:nw]
    t0 = t[:, 0].min()
    role = (np.arange(nw) // 4) & 1          # even workgroups: binning role, odd: record role
    for ro, name in ((0, "binning role"), (1, "record role")):
        tt = t[role == ro]
        print("%s: %d waves" % (name, len(tt)))
        for k in range(9):
            ok = tt[:, k] >= tt[:, 0]          # (a stale stamp of an earlier launch is older than this launch's start stamp)
            if k in (2, 3, 4):
                ok &= tt[:, k] >= t0
            v = (tt[ok, k] - t0) * 0.01        # us
            if len(v):
                print("  mark %d %-52s n=%5d  median %6.2f  p90 %6.2f  last %6.2f us" % (k, NAMES[k], len(v), np.median(v), np.percentile(v, 90), v.max()))
    print("launch span (first start to last stamp): %.2f us" % ((t[t >= t0].max() - t0) * 0.01))


if __name__ == "__main__":
    main()
