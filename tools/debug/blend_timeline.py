"""Life of the fused blend kernel's workgroups.  Debug build only:
    IGS_EXTRA_FLAGS=-DBLEND_TIMELINE python -c "import igs_amd.build as b; b.build()" && python tools/debug/blend_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from igs_amd import _cabi
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, activate, perturbed_copy

MARKS, WGS = 4, 16384


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
    assert L.igs_debug_blend_timeline(buf, MARKS * WGS) == 0
    gx, gy = (1352 + 15) // 16, (1014 + 15) // 16
    nb = 8 * ((gy + 7) // 8) * gx
    t = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, MARKS).astype(np.int64)[:nb]
    t0 = t[:, 0].min()
    live = t[:, 2] >= t[:, 0]
    tl = t[live]
    n = tl[:, 3]
    st, mid, en = (tl[:, 0] - t0) * 0.01, (tl[:, 1] - t0) * 0.01, (tl[:, 2] - t0) * 0.01
    print("%d workgroups (%d with a tile); instances per tile: mean %.1f max %d" % (nb, len(tl), n.mean(), n.max()))
    print("start since launch: median %.1f p90 %.1f last %.1f us" % (np.median(st), np.percentile(st, 90), st.max()))
    print("forward pass of a tile: median %.1f p90 %.1f max %.1f us; backward: median %.1f p90 %.1f max %.1f us"
          % (np.median(mid - st), np.percentile(mid - st, 90), (mid - st).max(), np.median(en - mid), np.percentile(en - mid, 90), (en - mid).max()))
    span = en.max()
    print("launch span %.1f us; workgroups running at t = " % span + ", ".join("%d us: %d" % (x, int(((st <= x) & (en > x)).sum())) for x in range(0, int(span) + 1, 10)))
    order = np.argsort(en)[-8:]
    print("the last eight to finish: " + "; ".join("n=%d start %.1f fwd %.1f bwd %.1f end %.1f" % (n[i], st[i], mid[i] - st[i], en[i] - mid[i], en[i]) for i in order))
    for lo, hi in ((0, 0), (1, 32), (33, 64), (65, 128), (129, 192), (193, 256), (257, 512)):
        m = (n >= lo) & (n <= hi)
        if m.any():
            print("  tiles of %3d-%3d instances: %5d   forward median %5.1f  backward median %5.1f us" % (lo, hi, int(m.sum()), np.median((mid - st)[m]), np.median((en - mid)[m])))


if __name__ == "__main__":
    main()
