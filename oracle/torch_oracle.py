"""Independent pure-PyTorch restatement of the rasterizer FORWARD (differentiable by autograd).

TEST INFRASTRUCTURE ONLY (see oracle/rast_oracle.c header).  Purpose:
  * a second, independently derived forward (standard row/column matrix notation instead of the
    reference's glm column-major literals) to catch transcription errors in the C oracle;
  * autograd gradients as an independent check of the analytic backward
    (DGR/cuda_rasterizer/backward.cu) on well-conditioned inputs;
  * it is also the "pure-PyTorch projection + per-pixel alpha-blend" CPU baseline that
    BASELINE.json's north_star names (timed by bench.py --cpu-baseline torch).

Follows DGR/cuda_rasterizer/forward.cu:23-423 (per-Gaussian stage, vectorised over P) and
forward.cu:428-693 (blend; vectorised per tile as a [256 pixels, K instances] masked cumulative product).

Known, documented departures of autograd from the reference's hand-written backward (so tests compare
with these in mind): the reference ignores d(coef)/d(cov) consistently only up to a wrong operand
(rasterizer_impl.cu:569) -- negligible at kernel_size = 0; the reference differentiates through the
min(0.99, .) clamp as if it were the identity (backward.cu:852,979) -- reproduced here with a
straight-through clamp; the ill-conditioned covariance branch (forward.cu:151-155) is not
differentiable here (inverse is used), so only well-conditioned inputs are compared.
"""
import math

import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]
TILE = 16


def sh_basis(deg, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    B = [torch.full_like(x, C0)]
    if deg > 0:
        B += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        B += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        B += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(B, 1)          # [P, (deg+1)^2]


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]       # (w,x,y,z), NOT normalised (forward.cu:279)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    return R                            # standard rotation matrix, rows as written


def per_gaussian(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_modifier,
                 viewmatrix, projmatrix, campos, tanfovx, tanfovy, kernel_size, W, H, degree, detach_coef=False):
    dt = means3D.dtype
    P = means3D.shape[0]
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    Rwc = viewmatrix[:3, :3].t()        # viewmatrix is w2c transposed (row-vector convention)
    twc = viewmatrix[3, :3]
    t = means3D @ Rwc.t() + twc         # view-space point
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], 1) @ projmatrix
    p_w = 1.0 / (hom[:, 3] + 0.0000001)
    ndc = hom[:, :2] * p_w[:, None]
    valid = t[:, 2] > 0.2               # auxiliary.h:170

    if cov3D_precomp is None:
        R = quat_to_rot(rotations)
        S = scale_modifier * scales
        Mm = R * S[:, None, :]          # R diag(s)
        Sigma = Mm @ Mm.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)

    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    u = torch.clamp(t[:, 0] / tz, -limx, limx)
    v = torch.clamp(t[:, 1] / tz, -limy, limy)
    tx, ty = u * tz, v * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    Scam = Rwc @ Sigma @ Rwc.t()
    cov2 = J @ Scam @ J.transpose(1, 2)
    a, b, c = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det = a * c - b * b
    det0 = torch.clamp(det, min=1e-6)
    det1 = torch.clamp((a + kernel_size) * (c + kernel_size) - b * b, min=1e-6)
    coef = torch.sqrt(det0 / (det1 + 1e-6) + 1e-6)
    coef = torch.where((det0 <= 1e-6) | (det1 <= 1e-6), torch.zeros_like(coef), coef)
    if detach_coef:
        coef = coef.detach()

    A = torch.linalg.inv(Scam)          # well-conditioned branch of forward.cu:144-150
    uvh = torch.stack([u, v, torch.ones_like(u)], 1)
    m = (A @ uvh[:, :, None])[:, :, 0]
    mn = m / m.norm(dim=1, keepdim=True)
    vbn = (mn * uvh).sum(1)
    u2, v2, uv = u * u, v * v, u * v
    l = torch.sqrt(tx * tx + ty * ty + tz * tz)
    q = mn / torch.clamp(vbn, min=0.0000001)[:, None]
    plane0 = (v2 + 1) * q[:, 0] - uv * q[:, 1] - u * q[:, 2]
    plane1 = -uv * q[:, 0] + (u2 + 1) * q[:, 1] - v * q[:, 2]
    nl = u2 + v2 + 1
    cam_plane = torch.stack([
        (-(v2 + 1) * tz + plane0 * tx) / nl / fx, (uv * tz + plane1 * tx) / nl / fy,
        (uv * tz + plane0 * ty) / nl / fx, (-(u2 + 1) * tz + plane1 * ty) / nl / fy,
        (tx + plane0 * tz) / nl / fx, (ty + plane1 * tz) / nl / fy], 1)
    ray_plane = torch.stack([plane0 * l / nl / fx, plane1 * l / nl / fy], 1)
    fn = l / nl
    rn = torch.stack([-plane0 * fn, -plane1 * fn, -torch.ones_like(fn)], 1)
    cn = torch.stack([rn[:, 0] / tz + tx / l * rn[:, 2], rn[:, 1] / tz + ty / l * rn[:, 2],
                      -tx / (tz * tz) * rn[:, 0] - ty / (tz * tz) * rn[:, 1] + tz / l * rn[:, 2]], 1)
    normal = cn / cn.norm(dim=1, keepdim=True)
    ts = t.norm(dim=1)

    det_inv = 1.0 / det
    conic = torch.stack([c * det_inv, -b * det_inv, a * det_inv], 1)
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        lam2 = mid - torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2)))
    xy = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)

    if colors_precomp is None:
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        B = sh_basis(degree, d)
        rgb = (B[:, :, None] * shs[:, :B.shape[1], :]).sum(1) + 0.5
        rgb = torch.clamp(rgb, min=0.0)
    else:
        rgb = colors_precomp
    valid = valid & (det != 0)
    return dict(valid=valid, xy=xy, conic=conic, opac=opacities.reshape(-1) * coef, rgb=rgb, view_points=t,
                cam_plane=cam_plane, ts=ts, ray_plane=ray_plane, normal=normal, depth=t[:, 2], radius=radius,
                cov2=torch.stack([a, b, c], 1), coef=coef)


def _rects(xy, radius, gx, gy):
    r = radius
    x0 = np.clip(np.trunc((xy[:, 0] - r) / TILE), 0, gx).astype(np.int64)
    y0 = np.clip(np.trunc((xy[:, 1] - r) / TILE), 0, gy).astype(np.int64)
    x1 = np.clip(np.trunc((xy[:, 0] + r + TILE - 1) / TILE), 0, gx).astype(np.int64)
    y1 = np.clip(np.trunc((xy[:, 1] + r + TILE - 1) / TILE), 0, gy).astype(np.int64)
    return x0, y0, x1, y1


def render(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_modifier, viewmatrix,
           projmatrix, campos, tanfovx, tanfovy, kernel_size, W, H, degree, bg, require_coord=True,
           require_depth=True, g=None, detach_coef=False):
    """Returns dict(color, coord, mcoord, depth, mdepth, alpha, normal, radii, num_rendered).
    `g` may carry a precomputed per_gaussian() result (then the first arguments are ignored except for dtype)."""
    dt = bg.dtype
    if g is None:
        g = per_gaussian(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_modifier,
                         viewmatrix, projmatrix, campos, tanfovx, tanfovy, kernel_size, W, H, degree, detach_coef)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    COORD, DEPTH = bool(require_coord), bool(require_depth)
    NORMAL = COORD or DEPTH

    xy_np = g["xy"].detach().to(torch.float32).numpy().astype(np.float32)
    rad = g["radius"].numpy().astype(np.float32)
    valid = g["valid"].numpy()
    x0, y0, x1, y1 = _rects(xy_np, np.where(valid, rad, 0), gx, gy)
    cnt = np.where(valid, (x1 - x0) * (y1 - y0), 0)
    radii = np.where(cnt > 0, rad, 0).astype(np.int32)
    total = int(cnt.sum())
    gid = np.repeat(np.arange(len(cnt)), cnt)
    start = np.cumsum(cnt) - cnt
    local = np.arange(total) - np.repeat(start, cnt)
    wrect = (x1 - x0)[gid]
    tile = (y0[gid] + local // np.maximum(wrect, 1)) * gx + (x0[gid] + local % np.maximum(wrect, 1))
    dbits = g["depth"].detach().to(torch.float32).numpy().view(np.uint32).astype(np.int64)
    key = tile.astype(np.int64) * (1 << 32) + dbits[gid]
    order = np.argsort(key, kind="stable")
    gid_s, tile_s = gid[order], tile[order]
    bounds = np.searchsorted(tile_s, np.arange(gx * gy + 1))

    HW = (H, W)
    out = dict(color=torch.zeros(3, *HW, dtype=dt), coord=torch.zeros(3, *HW, dtype=dt), mcoord=torch.zeros(3, *HW, dtype=dt),
               depth=torch.zeros(1, *HW, dtype=dt), mdepth=torch.zeros(1, *HW, dtype=dt), alpha=torch.zeros(1, *HW, dtype=dt),
               normal=torch.zeros(3, *HW, dtype=dt))
    pieces = {k: [] for k in out}
    where = []
    ly, lx = np.meshgrid(np.arange(TILE), np.arange(TILE), indexing="ij")
    for t in range(gx * gy):
        tx, ty = t % gx, t // gx
        px = (tx * TILE + lx).reshape(-1)
        py = (ty * TILE + ly).reshape(-1)
        inside = (px < W) & (py < H)
        px, py = px[inside], py[inside]
        n = len(px)
        ids = torch.from_numpy(gid_s[bounds[t]:bounds[t + 1]])
        K = len(ids)
        pxf = torch.tensor(px, dtype=dt)
        pyf = torch.tensor(py, dtype=dt)
        pnx, pny = (pxf - W / 2.0) / fx, (pyf - H / 2.0) / fy
        ln = torch.sqrt(pnx * pnx + pny * pny + 1)
        if K == 0:
            col = bg[:, None].expand(3, n).clone()
            z3, z1 = torch.zeros(3, n, dtype=dt), torch.zeros(1, n, dtype=dt)
            res = dict(color=col, coord=z3, mcoord=z3, depth=z1, mdepth=z1, alpha=z1, normal=z3)
        else:
            dx = g["xy"][ids, 0][None, :] - pxf[:, None]
            dy = g["xy"][ids, 1][None, :] - pyf[:, None]
            con = g["conic"][ids]
            power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
            a_raw = g["opac"][ids][None, :] * torch.exp(power)
            alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()      # straight-through clamp
            with torch.no_grad():
                ok = (power <= 0) & (alpha >= 1.0 / 255.0)
            am = torch.where(ok, alpha, torch.zeros_like(alpha))
            one_m = 1 - am
            Tincl = torch.cumprod(one_m, 1)
            Tbefore = torch.cat([torch.ones(n, 1, dtype=dt), Tincl[:, :-1]], 1)
            with torch.no_grad():
                stop = ok & (Tincl < 0.0001)
                dead = torch.cumsum(stop.to(torch.int32), 1) > 0          # this one and everything after
                live = ok & ~dead
                before_med = live & (Tbefore > 0.5)
            w = torch.where(live, am * Tbefore, torch.zeros_like(am))      # aT
            Tfin = torch.where(live, one_m, torch.ones_like(one_m)).prod(1)
            weight = w.sum(1)
            has = live.any(1)
            safe_w = torch.where(has, weight, torch.ones_like(weight))
            col = (w[:, :, None] * g["rgb"][ids][None]).sum(1).t() + Tfin[None, :] * bg[:, None]
            res = dict(color=col, alpha=weight[None, :])
            # index of the median contributor (last live one with T > 0.5), if any
            with torch.no_grad():
                kk = torch.arange(K)[None, :].expand(n, K)
                med_idx = torch.where(before_med, kk, torch.full_like(kk, -1)).max(1).values
                has_med = med_idx >= 0
                med_idx_c = med_idx.clamp(min=0)
            ar = torch.arange(n)
            if COORD:
                cp = g["cam_plane"][ids]
                vp = g["view_points"][ids]
                coord = torch.stack([vp[None, :, i] + cp[None, :, 2 * i] * dx + cp[None, :, 2 * i + 1] * dy for i in range(3)], 2)
                Coord = (w[:, :, None] * coord).sum(1)
                res["coord"] = torch.where(has[:, None], Coord / safe_w[:, None], torch.zeros_like(Coord)).t()
                mc = coord[ar, med_idx_c]
                res["mcoord"] = torch.where(has_med[:, None], mc, torch.zeros_like(mc)).t()
            else:
                res["coord"] = torch.zeros(3, n, dtype=dt); res["mcoord"] = torch.zeros(3, n, dtype=dt)
            if DEPTH:
                rp = g["ray_plane"][ids]
                tt = g["ts"][ids][None, :] + (rp[None, :, 0] * dx + rp[None, :, 1] * dy)
                Depth = (w * tt).sum(1)
                res["depth"] = torch.where(has, Depth / ln / safe_w, torch.zeros_like(Depth))[None, :]
                md = tt[ar, med_idx_c]
                res["mdepth"] = (torch.where(has_med, md, torch.zeros_like(md)) / ln)[None, :]
            else:
                res["depth"] = torch.zeros(1, n, dtype=dt); res["mdepth"] = torch.zeros(1, n, dtype=dt)
            if NORMAL:
                Nn = (w[:, :, None] * g["normal"][ids][None]).sum(1)
                ln_n = torch.sqrt((Nn * Nn).sum(1) + (~has).to(dt))          # avoid sqrt'(0) for empty pixels
                nrm = Nn / torch.clamp(ln_n, min=1e-12)[:, None]
                res["normal"] = torch.where(has[:, None], nrm, torch.zeros_like(nrm)).t()
            else:
                res["normal"] = torch.zeros(3, n, dtype=dt)
        where.append((torch.from_numpy(py), torch.from_numpy(px)))
        for k in out:
            pieces[k].append(res[k])
    pyc = torch.cat([w_[0] for w_ in where])
    pxc = torch.cat([w_[1] for w_ in where])
    for k in out:
        vals = torch.cat(pieces[k], 1)
        out[k] = _scatter(out[k], pyc, pxc, vals)
    out["radii"] = radii
    out["num_rendered"] = total
    return out


def _scatter(img, py, px, vals):
    C, H, W = img.shape
    flat = torch.zeros(C, H * W, dtype=img.dtype)
    idx = (py * W + px)
    flat = flat.index_copy(1, idx, vals)
    return flat.reshape(C, H, W)
