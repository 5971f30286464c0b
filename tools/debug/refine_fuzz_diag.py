"""Where do the fused refine step's forward and the unfused native step's forward part?  For one seed of tests/test_gpu_fuzz.py's refine
fuzz: the per-Gaussian records of both (debug dump), the pixels whose colour differs, and the splat at fault.
usage: python tools/debug/refine_fuzz_diag.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fuzz as F
from igs_amd import rasterizer as R
from igs_amd.refine import GaussianParams, Refiner

seed = int(sys.argv[1])
dev = torch.device("cuda:0")
raw, cam, bg, req, deg, ks = F.random_case(5000 + seed)
P = raw["xyz"].shape[0]
rng = np.random.default_rng(9000 + seed)
gen = torch.Generator().manual_seed(9000 + seed)
raw["scaling"] = torch.rand(P, 3, generator=gen) * 2.0 - 5.5
cams = [cam.to(dev)]
gts = [torch.rand(3, cam.height, cam.width, generator=gen).to(dev)]
loss = ["l1", "l1_ssim"][int(rng.integers(0, 2))]
ldn = float(rng.choice([0.0, 0.0, 0.05]))
print("seed", seed, "P", P, cam.width, cam.height, loss, ldn)
recs, imgs = [], []
for fused in (True, False):
    p = GaussianParams(raw, dev)
    r = Refiner(p, cams, gts, bg.to(dev), loss=loss, native=True, fused=fused, lambda_depth_normal=ldn)
    r.adam_fn = lambda: None
    pk = r.step(view=0)
    torch.cuda.synchronize()
    bufs = [v for v in r.__dict__.values() if isinstance(v, R.RasterBuffers)]
    if not bufs:
        print("no RasterBuffers on this Refiner (mode %s); attributes: %s" % (r._mode(), sorted(r.__dict__)))
        sys.exit(0)
    ss = bufs[0].scratch
    nr = int(pk.get("num_rendered", 0)) if isinstance(pk, dict) and "num_rendered" in pk else 0
    d = R.debug_dump(P, 0, cam.width, cam.height, ss.geom, ss.binning, ss.img)
    recs.append(d["rec"].cpu().numpy()); imgs.append(pk["images_pred"].detach().cpu().numpy())
a, b = recs
vis = (pk["radii"].cpu().numpy() > 0)
names = ["x", "y", "conic.x", "conic.y", "conic.z", "op*coef", "r", "g", "b", "ts", "ray.x", "ray.y", "vp.x", "vp.y", "vp.z", "n.x"] + ["cp%d" % i for i in range(4)] + ["cp4", "cp5", "n.y", "n.z", "cov0", "cov1", "cov2", "cov3", "cov4", "cov5", "clamped", "dkey"]
diff = (a.view(np.uint32) != b.view(np.uint32)) & vis[:, None]
print("visible Gaussians", int(vis.sum()), "; records with a differing word:", int(diff.any(1).sum()))
for j in np.nonzero(diff.any(0))[0]:
    g = np.nonzero(diff[:, j])[0]
    k = g[0]
    print("  word %2d %-8s differs in %5d records; e.g. Gaussian %d: fused %.9g unfused %.9g" % (j, names[j], len(g), k, a[k, j], b[k, j]))
di = np.abs(imgs[0] - imgs[1])
print("pixels whose colour differs by > 2e-5:", np.argwhere(di.max(0) > 2e-5).tolist(), "max", di.max())
