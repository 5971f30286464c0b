import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from igs_amd import _cabi, rasterizer as R
from igs_amd.scenes import sear_steak_like_scene, activate
dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene()
a = {k: v.to(dev) for k, v in activate(raw).items()}; cam = cams[0].to(dev); bg = bg.to(dev)
e = torch.Tensor([])
for skip in (0, 3, 7, 11, 15):
    os.environ["IGS_DBG_SKIP"] = str(skip)
    _cabi.profile_enable(True); _cabi.profile_read(True)
    for _ in range(20):
        R.rasterize_gaussians(bg, a["means3D"], e, a["opacities"], a["scales"], a["rotations"], 1.0, e, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, a["shs"], 3, cam.camera_center, False, True, True, False)
    torch.cuda.synchronize()
    st, r, c = _cabi.profile_read(True)
    print("skip", skip, "preprocess ms", st["preprocess"][0] / max(1, st["preprocess"][1]))
