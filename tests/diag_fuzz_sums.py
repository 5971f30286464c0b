"""For one fuzz seed and Gaussian: the blend-backward sums of the HIP path (accumulator row) next to the oracle's.
usage: python tests/diag_fuzz_sums.py SEED GAUSSIAN"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_fuzz as F
from test_gpu_parity import GNAMES, KEYS, E, hip_forward, oracle_forward, rand_grads
from igs_amd import rasterizer as R, _cabi
from igs_amd.scenes import activate
from oracle import c_oracle as co
seed, w = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
raw, cam, bg, req, deg, ks = F.random_case(seed)
a = activate(raw)
P = a["means3D"].shape[0]
out, ad, mats = hip_forward(a, cam, bg, dev, req, deg=deg, kernel_size=ks)
nr_o, oo, st = oracle_forward(a, cam, bg, req, deg=deg, kernel_size=ks)
grads = rand_grads(oo, seed)
nr, color, coord, mcoord, alpha, normal, depth, mdepth, radii, gb, bb, ib = out
V, Pm, cc = mats
gt = {k: torch.from_numpy(v).to(dev) for k, v in grads.items()}
ws = torch.empty(_cabi.lib().igs_rast_backward_workspace_bytes(P), dtype=torch.uint8, device=dev)
gout = R.rasterize_gaussians_backward(bg.to(dev), ad["means3D"], radii, E, ad["scales"], ad["rotations"], 1.0, E, V, Pm, cam.tanfovx, cam.tanfovy, ks,
                                      gt["color"], gt["coord"], gt["mcoord"], gt["depth"], gt["mdepth"], gt["alpha"], gt["normal"],
                                      normal, ad["shs"], deg, cc, gb, nr, bb, ib, alpha, req[0], req[1], True, workspace=ws)
gr = co.rasterize_backward(st, bg, a["means3D"], None, a["scales"], a["rotations"], None, cam.world_view_transform, cam.full_proj_transform,
                           cam.camera_center, a["shs"], oo["alpha"], oo["normal"], *[grads[k] for k in KEYS], debug_intermediates=True)
base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
g = ws[base:base + P * 32 * 4].view(torch.float32).view(P, 32).cpu().numpy().astype(np.float64)
compact = not (req[0] or req[1])
print("seed", seed, "P", P, "req", req, "compact", compact, "radius", oo["radii"][w], "xy", st.intermediates()["means2D"][w], "conic_opacity", st.intermediates()["conic_opacity"][w])
if compact:     # {c0 c1 c2 Q0} {Qx Qy Qxx Qxy} {Qyy Z}
    Qxx, Qxy, Qyy, Q0 = g[:, 6], g[:, 7], g[:, 8], g[:, 3]
else:
    Qxx, Qxy, Qyy, Q0 = g[:, 21], g[:, 22], g[:, 23], g[:, 18]
hc = np.stack([-0.5 * Qxx, -0.5 * Qxy, -0.5 * Qyy], 1)
oc = gr["_dbg"]["conic"][:, [0, 1, 3]].astype(np.float64)
print("dL_dconic hip   ", hc[w]); print("dL_dconic oracle", oc[w])
r = np.abs(hc - oc).max(1) / (np.abs(oc).max(1) + 1e-20)
worst = np.argsort(-r)[:8]
print("largest relative dL_dconic differences:", [(int(i), int(oo["radii"][i]), float("%.3g" % r[i])) for i in worst])
print("opacity sum hip", Q0[w], "oracle dL_dopacity*? ", gr["opacity"][w])
for n, t in zip(GNAMES, gout):
    print(n, "hip", t.cpu().numpy().reshape(P, -1)[w][:6], "oracle", gr[n].reshape(P, -1)[w][:6])
