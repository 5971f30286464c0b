"""How many wave trips would the blend kernels make under different lane <-> (pixel, splat) assignments?  CPU only (oracle forward
of one bench-scene view).  A (group, splat) entry counts when some pixel of the group has alpha >= 1/255 and lies in front of the
pixel's last contributor (the backward's validity; the forward's is the same up to saturation).

  quad      : now -- wave = 8x8 quad, one splat per trip            trips = sum_quads n_q
  half2     : two 8x4 halves walking their own lists              trips = sum_quads max(n_A, n_B)
  blk4      : four 4x4 blocks walking their own lists             trips = sum_quads max_b n_b
  half_x2   : wave = 8x4 half x 2 consecutive splats              trips = sum_halves ceil(n_h / 2)
  blk_x4    : wave = 4x4 block x 4 consecutive splats             trips = sum_blocks ceil(n_b / 4)
  blk8x2_x4 : wave = 8x2 block x 4 consecutive splats
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from igs_amd.scenes import sear_steak_like_scene, activate
from oracle import c_oracle as co

scale_mean = float(sys.argv[2]) if len(sys.argv) > 2 else -4.0
raw, cams, bg = sear_steak_like_scene(scale_mean=scale_mean)
cam = cams[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
a = activate(raw)
nr, out, st = co.rasterize_forward(bg, a["means3D"], None, a["opacities"], a["scales"], a["rotations"], 1.0, None,
                                   cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0,
                                   cam.height, cam.width, a["shs"], 3, cam.camera_center)
im = st.intermediates()
pl, ranges, ncon = im["point_list"].astype(np.int64), im["ranges"].astype(np.int64), im["n_contrib"][0]
m2d, co4 = im["means2D"], im["conic_opacity"]
W, H = cam.width, cam.height
gx, gy = (W + 15) // 16, (H + 15) // 16
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
tot = dict(quad=0, half2=0, blk4=0, half_x2=0, blk_x4=0, blk8x2_x4=0, half_x2_seq=0)
lanes = 0
entries = dict(quad=0, half=0, blk=0)
for t in range(gx * gy):
    r0, r1 = ranges[t]
    if r1 <= r0:
        continue
    ids = pl[r0:r1]
    tx, ty = (t % gx) * 16, (t // gx) * 16
    X = (tx + xx).astype(np.float32)[None]; Y = (ty + yy).astype(np.float32)[None]
    inside = (X < W) & (Y < H)
    dx = m2d[ids, 0][:, None, None] - X; dy = m2d[ids, 1][:, None, None] - Y
    c = co4[ids]
    power = -0.5 * (c[:, 0, None, None] * dx * dx + c[:, 2, None, None] * dy * dy) - c[:, 1, None, None] * dx * dy
    alpha = np.minimum(0.99, c[:, 3, None, None] * np.exp(power))
    lc = np.zeros((16, 16), np.int64)
    hh, ww = min(16, H - ty), min(16, W - tx)
    lc[:hh, :ww] = ncon[ty:ty + hh, tx:tx + ww]
    eidx = np.arange(len(ids))[:, None, None]
    valid = (power <= 0) & (alpha >= 1.0 / 255.0) & inside & (eidx < lc[None])
    lanes += int(valid.sum())
    v = valid.reshape(len(ids), 2, 8, 2, 8)                      # [n, qy, y, qx, x]
    q = v.any(axis=(2, 4))                                         # [n, qy, qx]
    nq = q.sum(axis=0)                                             # [2,2]
    tot["quad"] += int(nq.sum()); entries["quad"] += int(nq.sum())
    hv = valid.reshape(len(ids), 2, 2, 4, 2, 8).any(axis=(3, 5))  # [n, qy, half, qx]
    nh = hv.sum(axis=0)                                            # [2,2,2]
    entries["half"] += int(nh.sum())
    tot["half2"] += int(nh.max(axis=1).sum())
    tot["half_x2"] += int(((nh + 1) // 2).sum())
    bv = valid.reshape(len(ids), 4, 4, 4, 4).any(axis=(2, 4))     # [n, by, bx]
    nb = bv.sum(axis=0)                                            # [4,4]
    entries["blk"] += int(nb.sum())
    tot["blk_x4"] += int(((nb + 3) // 4).sum())
    nb_q = nb.reshape(2, 2, 2, 2)                                  # [qy, by, qx, bx]
    tot["blk4"] += int(nb_q.max(axis=(1, 3)).sum())
    b82 = valid.reshape(len(ids), 8, 2, 2, 8).any(axis=(2, 4)).sum(axis=0)
    tot["blk8x2_x4"] += int(((b82 + 3) // 4).sum())
print("R = %d, useful (pixel, splat) pairs = %d, entries: quad %d (%.1f lanes/row), half %d, 4x4 block %d"
      % (nr, lanes, entries["quad"], lanes / max(1, entries["quad"]), entries["half"], entries["blk"]))
for k, v in tot.items():
    if v:
        print("%-10s trips %9d   %.3f x of now, %.1f useful lanes per trip" % (k, v, v / tot["quad"], lanes / v))
