"""One forward of a fuzz seed, then the backward N times in the same process: is a wrong result per call or per process?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_fuzz as F
from test_gpu_parity import GNAMES, E, hip_forward, rand_grads, oracle_forward
from igs_amd import rasterizer as R, _cabi
from igs_amd.scenes import activate
seed, w = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
raw, cam, bg, req, deg, ks = F.random_case(seed)
a = activate(raw)
P = a["means3D"].shape[0]
nr_o, oo, st = oracle_forward(a, cam, bg, req, deg=deg, kernel_size=ks)
grads = rand_grads(oo, seed)
gt = {k: torch.from_numpy(v).to(dev) for k, v in grads.items()}
for fwd in range(3):
    out, ad, mats = hip_forward(a, cam, bg, dev, req, deg=deg, kernel_size=ks)
    nr, color, coord, mcoord, alpha, normal, depth, mdepth, radii, gb, bb, ib = out
    V, Pm, cc = mats
    for rep in range(5):
        gout = R.rasterize_gaussians_backward(bg.to(dev), ad["means3D"], radii, E, ad["scales"], ad["rotations"], 1.0, E, V, Pm, cam.tanfovx, cam.tanfovy, ks,
                                              gt["color"], gt["coord"], gt["mcoord"], gt["depth"], gt["mdepth"], gt["alpha"], gt["normal"],
                                              normal, ad["shs"], deg, cc, gb, nr, bb, ib, alpha, req[0], req[1], bool(int(os.environ.get("DBG", "1"))))
        print(fwd, rep, "means3D", gout[3][w].cpu().numpy(), "cov3D", gout[4][w].cpu().numpy()[:3])
