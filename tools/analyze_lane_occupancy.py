"""How full are the 8x8 quad rows of the blend kernels, and how many depth-adjacent rows of a quad touch disjoint pixel sets?
(the upper bound of what pairing two splats into one wave row could save.)  One camera of the bench scene;
termination by transmittance is ignored (every row whose splat reaches alpha >= 1/255 on some pixel of the quad counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from igs_amd import rasterizer as R
from igs_amd.scenes import sear_steak_like_scene, activate

dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene()
cam = cams[int(sys.argv[1]) if len(sys.argv) > 1 else 0].to(dev)
a = {k: v.to(dev) for k, v in activate(raw).items()}
E = torch.Tensor([])
out = R.rasterize_gaussians(bg.to(dev), a["means3D"], E, a["opacities"], a["scales"], a["rotations"], 1.0, E, cam.world_view_transform,
                            cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, a["shs"], 3, cam.camera_center,
                            False, True, True, False)
nr, radii, gb, bb, ib = out[0], out[8], out[9], out[10], out[11]
P = a["means3D"].shape[0]
d = R.debug_dump(P, nr, cam.width, cam.height, gb, bb, ib)
rec, pl, ranges = d["rec"], d["point_list"].long(), d["ranges"].long()
gx = (cam.width + 15) // 16
T = ranges.shape[0]
py, px = torch.meshgrid(torch.arange(8, device=dev), torch.arange(8, device=dev), indexing="ij")
rows = lanes = pairs_disjoint = pairs = rows_after = 0
for t in range(T):
    r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
    if r1 <= r0:
        continue
    ids = pl[r0:r1]
    g = rec[ids]                                   # [n, 32]: xy, conic a b c, opacity
    tx, ty = (t % gx) * 16, (t // gx) * 16
    for q in range(4):
        X = (tx + (q & 1) * 8 + px).reshape(1, 64).float(); Y = (ty + (q >> 1) * 8 + py).reshape(1, 64).float()
        inside = ((X < cam.width) & (Y < cam.height))
        dx, dy = g[:, 0:1] - X, g[:, 1:2] - Y
        power = -0.5 * (g[:, 2:3] * dx * dx + g[:, 4:5] * dy * dy) - g[:, 3:4] * dx * dy
        alpha = torch.clamp(g[:, 5:6] * torch.exp(power), max=0.99)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0) & inside
        keep = valid.any(dim=1)
        v = valid[keep]
        n = v.shape[0]
        if n == 0:
            continue
        rows += n; lanes += int(v.sum())
        if n > 1:
            dis = ~((v[:-1] & v[1:]).any(dim=1))
            pairs += n - 1; pairs_disjoint += int(dis.sum())
            dl = dis.tolist(); i = 0; cnt = 0
            while i < n:                          # greedy: pair row i with i+1 when disjoint
                if i + 1 < n and dl[i]:
                    i += 2
                else:
                    i += 1
                cnt += 1
            rows_after += cnt
        else:
            rows_after += 1
print("quad rows %d, mean useful lanes %.1f of 64; depth-adjacent rows with disjoint pixel sets: %.1f %%; rows after greedy pairing: %.3f x"
      % (rows, lanes / rows, 100.0 * pairs_disjoint / max(pairs, 1), rows_after / rows))
