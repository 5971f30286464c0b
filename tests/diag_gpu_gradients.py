"""Find the worst pixel of the cfg-1 full-size comparison and list what the oracle blends there."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from igs_amd.scenes import cfg1_scene, activate
from igs_amd import rasterizer as R
from oracle import c_oracle as co
dev = torch.device("cuda:0")
raw, cams, bg = cfg1_scene()
cam = cams[0]; a = activate(raw); ad = {k: v.to(dev) for k, v in a.items()}
V, Pm, cc = cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), cam.camera_center.to(dev)
e = torch.Tensor([])
out = R.rasterize_gaussians(bg.to(dev), ad["means3D"], e, ad["opacities"], ad["scales"], ad["rotations"], 1.0, e, V, Pm, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, ad["shs"], 3, cc, False, True, True, True)
nr, color, coord, mcoord, alpha, normal, depth, mdepth, radii, gb, bb, ib = out
nr_o, oo, st = co.rasterize_forward(bg, a["means3D"], None, a["opacities"], a["scales"], a["rotations"], 1.0, None, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, a["shs"], 3, cam.camera_center)
it = st.intermediates()
d = R.debug_dump(10000, nr, 256, 256, gb, bb, ib)
nc = d["n_contrib"].cpu().numpy().astype(np.int64); nco = it["n_contrib"].astype(np.int64)
da = np.abs(alpha.cpu().numpy()[0] - oo["alpha"][0])
ys, xs = np.where(da > 1e-4)
print("bad alpha pixels", len(ys))
order = np.argsort(-da[ys, xs])[:6]
rec = d["rec"].cpu().numpy()
for o in order:
    y, x = ys[o], xs[o]
    t = (y // 16) * 16 + (x // 16)
    r0, r1 = it["ranges"][t]
    print("pixel", x, y, "tile", t, "range", r0, r1, "n", r1 - r0, "alpha hip/ora", alpha[0, y, x].item(), oo["alpha"][0, y, x], "last hip/ora", nc[0, y, x], nco[0, y, x], "max", nc[1, y, x], nco[1, y, x])
    T = np.float32(1.0); k = 0
    for idx in range(r0, r1):
        g = it["point_list"][idx]; k += 1
        X, Y = it["means2D"][g]; cx, cy, cz, op = it["conic_opacity"][g]
        dx = np.float32(X - x); dy = np.float32(Y - y)
        power = np.float32(-0.5) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy
        if power > 0: continue
        al = min(np.float32(0.99), op * np.exp(power))
        if al < 1 / 255: continue
        tt = T * (1 - al)
        if tt < 1e-4: print("   terminate at k", k); break
        # device-side mask for this instance
        det = cx * cz - cy * cy
        two_tau = 2 * np.log(255 * op) * 1.002 + 1e-3
        ex = np.sqrt(two_tau * cz / det) * 1.001 + 0.02; ey = np.sqrt(two_tau * cx / det) * 1.001 + 0.02
        tx0, ty0 = (x // 16) * 16, (y // 16) * 16
        qx, qy = (x % 16) // 8, (y % 16) // 8
        lx, hx = X - ex - tx0, X + ex - tx0; ly, hy = Y - ey - ty0, Y + ey - ty0
        okx = (hx >= 8 * qx) and (lx <= 8 * qx + 7); oky = (hy >= 8 * qy) and (ly <= 8 * qy + 7)
        flag = "" if (okx and oky) else "  <<< CULLED BY MASK"
        if al * T > 2e-3 or flag:
            print("   k %4d g %5d alpha %.5f aT %.5f T %.5f power %.4g op %.4f%s" % (k, g, al, al * T, T, power, op, flag))
        T = tt
