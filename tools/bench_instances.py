"""Times the forward / backward blend instances on the bench scene through the drop-in API (not the fused refine step):
   colour-only backward (what a colour loss needs) vs the full backward (all seven upstream gradients present)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer as R, _cabi
from igs_amd.scenes import sear_steak_like_scene, activate

def main():
    dev = torch.device("cuda:0")
    R.NAN_CHECKS = False
    raw, cams, bg = sear_steak_like_scene()
    a = {k: v.to(dev) for k, v in activate(raw).items()}
    cam = cams[0].to(dev); bg = bg.to(dev)
    E = torch.Tensor([])
    bufs = R.RasterBuffers()
    def fwd():
        return R.rasterize_gaussians(bg, a["means3D"], E, a["opacities"], a["scales"], a["rotations"], 1.0, E, cam.world_view_transform,
                                     cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, a["shs"], 3,
                                     cam.camera_center, False, True, True, False, buffers=bufs)
    out = fwd()
    nr, color, coord, mcoord, alpha, normal, depth, mdepth, radii, gb, bb, ib = out
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda t: (torch.randn(t.shape, generator=g) * 1e-3).to(dev)
    full = [rnd(color), rnd(coord), rnd(mcoord), rnd(depth), rnd(mdepth), rnd(alpha), rnd(normal)]
    col = [full[0], None, None, None, None, None, None]
    def bwd(gr):
        return R.rasterize_gaussians_backward(bg, a["means3D"], radii, E, a["scales"], a["rotations"], 1.0, E, cam.world_view_transform,
                                              cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, *gr, normal, a["shs"], 3,
                                              cam.camera_center, gb, nr, bb, ib, alpha, True, True, False, workspace=bufs.workspace)
    res = {}
    for name, gr in (("colour_only", col), ("full", full)):
        for _ in range(3):
            bwd(gr)
        torch.cuda.synchronize()
        _cabi.profile_enable(True, every=1); _cabi.profile_read(reset=True)
        for _ in range(20):
            bwd(gr)
        torch.cuda.synchronize()
        st, _, _ = _cabi.profile_read(reset=True)
        _cabi.profile_enable(False)
        res[name] = {k: round(ms / c, 4) for k, (ms, c) in st.items() if c}
    _cabi.profile_enable(True, every=1); _cabi.profile_read(reset=True)
    for _ in range(20):
        fwd()
    torch.cuda.synchronize()
    st, _, _ = _cabi.profile_read(reset=True)
    res["forward"] = {k: round(ms / c, 4) for k, (ms, c) in st.items() if c}
    res["num_rendered"] = nr
    # BASELINE cfg-2: forward-only render, wall clock (includes the one host wait per frame of igs_rast_forward)
    import time
    _cabi.profile_enable(False)
    for _ in range(10):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fwd()
    torch.cuda.synchronize()
    res["forward_only_ms"] = round(1000 * (time.perf_counter() - t0) / 200, 4)
    res["forward_only_gaussians_per_s"] = round(200000 / (res["forward_only_ms"] * 1e-3))
    print(json.dumps(res))

if __name__ == "__main__":
    main()
