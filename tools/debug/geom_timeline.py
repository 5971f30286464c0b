"""Where does a geom_bwd (+ Adam) workgroup spend its life?  Debug build only:
    IGS_EXTRA_FLAGS=-DGEOM_TIMELINE python -c "import igs_amd.build as b; b.build()" && python tools/debug/geom_timeline.py
Every workgroup (one wave, 64 Gaussians) stamps the 100 MHz clock at the marks of its chain (geom_bwd.hip, GTL(k))."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from igs_amd import _cabi
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, activate, perturbed_copy

MARKS, WGS = 8, 8192
NAMES = ["start", "record + accumulator row loaded", "geometry gradients done", "SH backward done, rows in LDS", "small groups updated, first SH batch back",
         "SH span updated (end)"]


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
    buf = (C.c_ulonglong * (MARKS * WGS))()
    assert L.igs_debug_geom_timeline(buf, MARKS * WGS) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, MARKS).astype(np.int64)
    nw = (p.P + 63) // 64
    t = t[:nw]
    t0 = t[:, 0].min()
    seen = t[:, 1] >= t[:, 0]            # workgroups with at least one visible Gaussian (marks 1, 2 are stamped inside that branch by thread 0 only)
    print("%d workgroups, thread 0 visible in %d" % (nw, int(seen.sum())))
    for k in range(6):
        ok = t[:, k] >= t[:, 0]
        v = (t[ok, k] - t0) * 0.01
        d = (t[ok, k] - t[ok, 0]) * 0.01
        print("  mark %d %-46s n=%5d  since launch: median %6.2f p90 %6.2f last %6.2f | since own start: median %6.2f p90 %6.2f us"
              % (k, NAMES[k], len(v), np.median(v), np.percentile(v, 90), v.max(), np.median(d), np.percentile(d, 90)))
    print("launch span: %.2f us" % ((t[:, 5].max() - t0) * 0.01))


if __name__ == "__main__":
    main()
