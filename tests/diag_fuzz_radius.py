"""For one fuzz seed: the Gaussians whose radius differs from the oracle's, with the per-Gaussian record of both sides."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_fuzz as F
from test_gpu_parity import hip_forward, oracle_forward
from igs_amd import rasterizer as R
from igs_amd.scenes import activate
seed = int(sys.argv[1])
dev = torch.device("cuda:0")
raw, cam, bg, req, deg, ks = F.random_case(seed)
a = activate(raw)
P = a["means3D"].shape[0]
out, ad, mats = hip_forward(a, cam, bg, dev, req, deg=deg, kernel_size=ks)
nr_o, oo, st = oracle_forward(a, cam, bg, req, deg=deg, kernel_size=ks)
it = st.intermediates()
d = R.debug_dump(P, out[0], cam.width, cam.height, out[9], out[10], out[11])
rec = d["rec"].cpu().numpy()
radii = out[8].cpu().numpy()
bad = np.nonzero(radii != oo["radii"])[0]
print("seed", seed, "ks", ks, "tanfov", cam.tanfovx, cam.tanfovy, "W,H", cam.width, cam.height, "mismatches", bad.tolist())
np.set_printoptions(precision=9, floatmode="maxprec")
for g in bad[:4]:
    print("g", g, "radius hip/oracle", radii[g], oo["radii"][g])
    print("  hip    xy", rec[g, 0:2], "conic", rec[g, 2:5], "opac*coef", rec[g, 5], "cov3D", rec[g, 24:30])
    print("  oracle xy", it["means2D"][g], "conic", it["conic_opacity"][g])
    print("  mean", a["means3D"][g].numpy(), "scale", a["scales"][g].numpy(), "rot", a["rotations"][g].numpy())
    for k in it:
        if k.startswith("cov"):
            print("  oracle", k, it[k][g])
