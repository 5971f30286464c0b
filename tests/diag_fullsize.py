import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from igs_amd import rasterizer as R
from igs_amd.scenes import sear_steak_like_scene, activate
from oracle import c_oracle as co
dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene()
a = activate(raw)
E = torch.Tensor([])
for ci in range(len(cams)):
    cam = cams[ci]
    ad = {k: v.to(dev) for k, v in a.items()}
    V, Pm, cc = cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), cam.camera_center.to(dev)
    out = R.rasterize_gaussians(bg.to(dev), ad["means3D"], E, ad["opacities"], ad["scales"], ad["rotations"], 1.0, E, V, Pm, cam.tanfovx, cam.tanfovy,
                                0.0, cam.height, cam.width, ad["shs"], 3, cc, False, True, True, False)
    co.set_precision("float32")
    nr_o, oo, st = co.rasterize_forward(bg, a["means3D"], None, a["opacities"], a["scales"], a["rotations"], 1.0, None, cam.world_view_transform,
                                        cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, a["shs"], 3, cam.camera_center)
    it = st.intermediates()
    d = R.debug_dump(200000, out[0], cam.width, cam.height, out[9], out[10], out[11])
    rh = out[8].cpu().numpy(); ro = oo["radii"]
    th = d["tiles_touched"].cpu().numpy().astype(np.int64); to = it["tiles_touched"].astype(np.int64)
    bad = np.nonzero((rh != ro) | (th != to))[0]
    print("cam", ci, "nr", out[0], nr_o, "mismatches", len(bad))
    rec = d["rec"].cpu().numpy()
    for i in bad[:6]:
        print("  idx", i, "radii", rh[i], ro[i], "tiles", th[i], to[i], "xy hip", rec[i, 0:2], "xy oracle", it["means2D"][i],
              "conic hip", rec[i, 2:5], "oracle", it["conic_opacity"][i, :3], "depth", it["depths"][i] if "depths" in it else None)
