"""Prints, for one fuzz seed, where the HIP gradients leave the float32 / float64 oracle (debug aid for tests/test_gpu_fuzz.py).
usage: python tests/diag_fuzz_case.py SEED [REPEATS]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_fuzz as F
from test_gpu_parity import GNAMES, hip_forward, hip_backward, oracle_forward, oracle_backward, rand_grads
from igs_amd.scenes import activate

seed = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
raw, cam, bg, req, deg, ks = F.random_case(seed)
a = activate(raw)
P = a["means3D"].shape[0]
nr_o, oo, st = oracle_forward(a, cam, bg, req, deg=deg, kernel_size=ks)
grads = rand_grads(oo, seed)
gr = oracle_backward(st, oo, a, cam, bg, grads, deg=deg)
g64 = F.oracle_gradients_f64(a, cam, bg, req, deg, ks, grads)
print("seed", seed, "P", P, cam.width, cam.height, req, deg, ks, "nr", nr_o)
for rep in range(reps):
    out, ad, mats = hip_forward(a, cam, bg, dev, req, deg=deg, kernel_size=ks)
    gout = hip_backward(out, ad, mats, cam, bg, dev, grads, req, deg=deg, kernel_size=ks)
    for n, t in zip(GNAMES, gout):
        A = t.cpu().numpy().reshape(gr[n].shape).astype(np.float64)
        e_hip, e_or = np.abs(A - g64[n]), np.abs(gr[n] - g64[n])
        flat = e_hip.reshape(P, -1).max(1)
        worst = np.argsort(-flat)[:3]
        print(rep, n, "||hip-f64|| %.3g ||f32-f64|| %.3g ||f64|| %.3g; worst Gaussians %s" % (np.linalg.norm(e_hip), np.linalg.norm(e_or), np.linalg.norm(g64[n]), worst.tolist()))
        for g in worst[:2]:
            print("    g%d hip %s\n       f32 %s\n       f64 %s  radius %d" % (g, A.reshape(P, -1)[g][:6], gr[n].reshape(P, -1)[g][:6], g64[n].reshape(P, -1)[g][:6], oo["radii"][g]))
