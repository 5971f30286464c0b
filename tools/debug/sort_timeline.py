"""Where does tile_sort spend its time?  Debug build only:
    IGS_EXTRA_FLAGS=-DSORT_TIMELINE python -c "import igs_amd.build as b; b.build()" && python tools/debug/sort_timeline.py
Every workgroup (= tile) stamps the 100 MHz clock at its start, after the count, and when its sorted list is written (sort.hip, STL)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from igs_amd import _cabi
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, activate, perturbed_copy

MARKS, TILES = 6, 8192


def main():
    dev = torch.device("cuda", 0)
    dense = "--dense" in sys.argv          # the diagnostic scene of `bench.py --scene dense`
    raw, cams, bg = sear_steak_like_scene(P=200000, n_cams=10, width=1352, height=1014, **({"scale_mean": -2.7} if dense else {}))
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
    buf = (C.c_ulonglong * (MARKS * TILES))()
    assert L.igs_debug_sort_timeline(buf, MARKS * TILES) == 0
    T = ((1352 + 15) // 16) * ((1014 + 15) // 16)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(TILES, MARKS).astype(np.int64)[:T]
    t0 = t[:, 0].min()
    n = t[:, 5]
    print("%d tiles; instances per tile: mean %.1f, median %d, p90 %d, p99 %d, max %d; tiles > 256: %d, > 512: %d"
          % (T, n.mean(), np.median(n), np.percentile(n, 90), np.percentile(n, 99), n.max(), int((n > 256).sum()), int((n > 512).sum())))
    st = (t[:, 0] - t0) * 0.01
    print("workgroup start since launch: median %.2f p90 %.2f last %.2f us" % (np.median(st), np.percentile(st, 90), st.max()))
    c1 = (t[:, 1] - t[:, 0]) * 0.01
    print("count read (+ reset): median %.2f p90 %.2f max %.2f us" % (np.median(c1), np.percentile(c1, 90), c1.max()))
    for lo, hi in ((1, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 2048), (2049, 1 << 20)):
        m = (n >= lo) & (n <= hi) & (t[:, 4] >= t[:, 0])
        if m.any():
            life = (t[m, 4] - t[m, 0]) * 0.01
            end = (t[m, 4] - t0) * 0.01
            print("  tiles of %4d-%4d instances: %5d   life median %5.2f p90 %5.2f max %5.2f us   end since launch: median %5.2f last %5.2f us"
                  % (lo, hi, int(m.sum()), np.median(life), np.percentile(life, 90), life.max(), np.median(end), end.max()))
    ok = t[:, 4] >= t[:, 0]
    print("launch span: %.2f us" % ((t[ok, 4].max() - t0) * 0.01))


if __name__ == "__main__":
    main()
