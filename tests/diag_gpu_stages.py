"""GPU diagnostic: run the HIP path on a small scene and compare every stage with the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from igs_amd.scenes import cfg1_scene, activate
from igs_amd import rasterizer as R
from oracle import c_oracle as co

def rel(A, B):
    A = np.asarray(A, np.float64); B = np.asarray(B, np.float64)
    return np.abs(A - B) / (np.abs(B) + 1e-3 * max(np.abs(B).max(), 1e-30))

def main(P=3000, size=128, req=(True, True)):
    dev = torch.device("cuda:0")
    raw, cams, bg = cfg1_scene(P=P, size=size)
    bg = torch.tensor([0.2, 0.4, 0.6])
    cam = cams[0]
    a = activate(raw)
    ad = {k: v.to(dev) for k, v in a.items()}
    V, Pm, cc = cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), cam.camera_center.to(dev)
    e = torch.Tensor([])
    t0 = time.time()
    out = R.rasterize_gaussians(bg.to(dev), ad["means3D"], e, ad["opacities"], ad["scales"], ad["rotations"], 1.0, e, V, Pm,
                                cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, ad["shs"], 3, cc, False, req[0], req[1], True)
    torch.cuda.synchronize()
    nr, color, coord, mcoord, alpha, normal, depth, mdepth, radii, gb, bb, ib = out
    print("HIP fwd ok, R =", nr, "t=%.3f" % (time.time() - t0))
    nr_o, oo, st = co.rasterize_forward(bg, a["means3D"], None, a["opacities"], a["scales"], a["rotations"], 1.0, None,
                                        cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0,
                                        cam.height, cam.width, a["shs"], 3, cam.camera_center, require_coord=req[0], require_depth=req[1])
    it = st.intermediates()
    d = R.debug_dump(P, nr, cam.width, cam.height, gb, bb, ib)
    rec = d["rec"].cpu().numpy()
    print("R oracle", nr_o, "radii mismatch", int((radii.cpu().numpy() != oo["radii"]).sum()),
          "tiles mismatch", int((d["tiles_touched"].cpu().numpy().astype(np.uint32) != it["tiles_touched"]).sum()))
    vis = oo["radii"] > 0
    def cmp(name, A, B):
        A = A[vis]; B = B[vis]
        print("  %-12s max abs %.3e  max rel %.3e" % (name, np.abs(A - B).max(), rel(A, B).max()))
    cmp("means2D", rec[:, 0:2], it["means2D"]); cmp("conic", np.stack([rec[:, 2], rec[:, 3], rec[:, 4]], 1), it["conic_opacity"][:, :3])
    cmp("opacity", rec[:, 5], it["conic_opacity"][:, 3]); cmp("rgb", np.stack([rec[:, 6], rec[:, 7], rec[:, 8]], 1), it["rgb"])
    cmp("ts", rec[:, 9], it["ts"]); cmp("ray_plane", rec[:, 10:12], it["ray_planes"]); cmp("view_points", rec[:, 12:15], it["view_points"])
    cmp("normals", np.stack([rec[:, 15], rec[:, 22], rec[:, 23]], 1), it["normals"])
    cmp("cam_plane", np.concatenate([rec[:, 16:20], rec[:, 20:22]], 1), it["camera_planes"]); cmp("cov3D", np.concatenate([rec[:, 24:28], rec[:, 28:30]], 1), it["cov3D"])
    if nr == nr_o:
        pl = d["point_list"].cpu().numpy().astype(np.uint32)
        print("point_list mismatch", int((pl != it["point_list"]).sum()), "ranges mismatch", int((d["ranges"].cpu().numpy().astype(np.uint32) != it["ranges"]).sum()))
        print("n_contrib mismatch", int((d["n_contrib"].cpu().numpy().astype(np.uint32) != it["n_contrib"]).sum()), "of", it["n_contrib"].size)
    for k, v in [("color", color), ("coord", coord), ("mcoord", mcoord), ("depth", depth), ("mdepth", mdepth), ("alpha", alpha), ("normal", normal)]:
        dd = np.abs(v.cpu().numpy() - oo[k])
        print("  img %-7s max abs %.3e  #>1e-4 %d / %d" % (k, dd.max(), int((dd > 1e-4).sum()), dd.size))
    # backward
    rng = np.random.default_rng(0)
    g = {k: rng.standard_normal(oo[k].shape).astype(np.float32) for k in ["color", "coord", "mcoord", "depth", "mdepth", "alpha", "normal"]}
    gt = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    t0 = time.time()
    gb_out = R.rasterize_gaussians_backward(bg.to(dev), ad["means3D"], radii, e, ad["scales"], ad["rotations"], 1.0, e, V, Pm, cam.tanfovx,
                                            cam.tanfovy, 0.0, gt["color"], gt["coord"], gt["mcoord"], gt["depth"], gt["mdepth"], gt["alpha"],
                                            gt["normal"], normal, ad["shs"], 3, cc, gb, nr, bb, ib, alpha, req[0], req[1], True)
    torch.cuda.synchronize()
    print("HIP bwd ok t=%.3f" % (time.time() - t0))
    # feed the ORACLE backward with the HIP forward's own alpha/normal? no: oracle uses its own state
    gr = co.rasterize_backward(st, bg, a["means3D"], None, a["scales"], a["rotations"], None, cam.world_view_transform,
                               cam.full_proj_transform, cam.camera_center, a["shs"], oo["alpha"], oo["normal"], g["color"], g["coord"],
                               g["mcoord"], g["depth"], g["mdepth"], g["alpha"], g["normal"])
    names = ["means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations"]
    for n, t in zip(names, gb_out):
        A = t.cpu().numpy().reshape(gr[n].shape); B = gr[n]
        r = rel(A, B)
        print("  grad %-9s max|ref| %.3e  rel: median %.2e  99%% %.2e  max %.2e  frac>1e-3 %.4f  nan %d" %
              (n, np.abs(B).max(), np.median(r), np.quantile(r, 0.99), r.max(), (r > 1e-3).mean(), int(np.isnan(A).sum())))

if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    main(P, size)
    if len(sys.argv) > 3:
        for req in [(True, False), (False, True), (False, False)]:
            print("=== require_coord, require_depth =", req)
            main(P, size, req)
