"""Densify-and-prune of the refine loop on the flat parameter store.

Host logic mirrors the reference step by step (`igs/models/gaussian_model.py`: densify_and_prune :640-663, densify_and_clone
:619-638, densify_and_split :586-617, prune_points :512-535, add_densification_stats :865-868; driver `infer_batch.py:308-321`);
what the reference does with five nn.Parameters, ten Adam-state tensors, boolean masks and torch.cat is ONE gather pass over the
flat buffers here (`igs_densify_remap`), and the per-step statistics are one small kernel (`igs_densify_stats`).
"""
import ctypes as C

import torch

from . import _cabi
from .refine import GROUPS


class DensifyConfig:
    """configs/demo.yaml:57-62 + the literals of infer_batch.py:308-321."""

    def __init__(self, until_iter=100, from_iter=0, interval=20, grad_threshold=0.00015, min_opacity=0.005, max_num=150000,
                 percent_dense=0.01, extent=1.0, max_screen_size=None, control_max=True):
        self.until_iter, self.from_iter, self.interval = until_iter, from_iter, interval
        self.grad_threshold, self.min_opacity, self.max_num = grad_threshold, min_opacity, max_num
        self.percent_dense, self.extent, self.max_screen_size, self.control_max = percent_dense, extent, max_screen_size, control_max


class DensifyState:
    """xyz_gradient_accum / denom / max_radii2D (gaussian_model.py:577-580)."""

    def __init__(self, P, device):
        self.reset(P, device)

    def reset(self, P, device):
        self.grad_accum = torch.zeros(P, dtype=torch.float32, device=device)
        self.denom = torch.zeros(P, dtype=torch.float32, device=device)
        self.max_radii = torch.zeros(P, dtype=torch.float32, device=device)

    def add(self, dL_dmean2D, radii):
        """infer_batch.py:311-312: max_radii2D update + add_densification_stats for the Gaussians with radii > 0."""
        P = radii.numel()
        rc = _cabi.lib().igs_densify_stats(torch.cuda.current_stream(radii.device).cuda_stream, P, dL_dmean2D.data_ptr(), radii.data_ptr(),
                                           self.grad_accum.data_ptr(), self.denom.data_ptr(), self.max_radii.data_ptr())
        if rc != 0:
            raise RuntimeError("igs_densify_stats failed: %d" % rc)


def reduce_state(state, world_size):
    """N > 1 (views sharded over ranks): every rank has accumulated the statistics of ITS views; before the densification decision
    the accumulators are summed over ranks and the radii maximised (`xyz_gradient_accum` / `denom` are sums of
    per-view terms, gaussian_model.py:865-868; `max_radii2D` is a running maximum, infer_batch.py:311), which leaves identical
    tensors on every rank -- so the plan below, with identically seeded split sampling, is identical everywhere and the replicas
    stay in lock-step without a broadcast.  A rank's loss carries the factor 1/N of the step's mean over views, so its screen-space
    gradient norms are multiplied back by N here: the accumulated statistic is then the per-view one the reference's
    `densify_grad_threshold` was chosen for."""
    if world_size <= 1:
        return state
    import torch.distributed as dist
    dist.all_reduce(state.grad_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(state.denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(state.max_radii, op=dist.ReduceOp.MAX)
    state.grad_accum.mul_(float(world_size))
    return state


def build_rotation(q):
    """submodules/RaDe-GS/utils/general_utils.py build_rotation (normalises the quaternion, (w, x, y, z))."""
    q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def plan(xyz, rotation, opacity_logit, log_scale, state, cfg, generator=None):
    """The selection logic of densify_and_prune, returned as a gather plan over the CURRENT Gaussians:
    src[i] (source row of new Gaussian i), fresh[i] (created now -> zero Adam moments), ovr[i] (row of ovr_xyz / ovr_scale
    for split children, else -1).  Order of the new rows = the order the reference's cat / mask sequence produces."""
    dev = xyz.device
    P = xyz.shape[0]
    grads = (state.grad_accum / state.denom).view(P, 1)
    grads[grads.isnan()] = 0.0
    max_num_add = cfg.max_num - P
    selected = torch.norm(grads, dim=-1) >= cfg.grad_threshold
    if cfg.control_max and int(selected.sum()) > max_num_add:            # max-points bounded densify (:648-654)
        if max_num_add > 0:
            vals, idx = torch.topk(grads, max_num_add, dim=0)
            g2 = torch.zeros_like(grads)
            g2.scatter_(0, idx, vals)
            grads = g2
        else:
            grads = torch.zeros_like(grads)
    scaling = torch.exp(log_scale)
    big = scaling.max(dim=1).values > cfg.percent_dense * cfg.extent
    # ---- clone (:619-638): small Gaussians with a large view-space gradient are duplicated as they are
    clone_mask = (torch.norm(grads, dim=-1) >= cfg.grad_threshold) & ~big
    idx_clone = torch.nonzero(clone_mask).squeeze(1)
    n_c = idx_clone.numel()
    # ---- split (:586-617): large ones are replaced by N = 2 children sampled inside them (mask over P + n_c rows; the clones
    #      have a padded gradient of zero and are never selected)
    split_mask = (grads.squeeze(1) >= cfg.grad_threshold) & big
    idx_split = torch.nonzero(split_mask).squeeze(1)
    n_s = idx_split.numel()
    N = 2
    stds = scaling[idx_split].repeat(N, 1)
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds, generator=generator) if n_s else stds
    rots = build_rotation(rotation[idx_split]).repeat(N, 1, 1)
    ovr_xyz = (torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + xyz[idx_split].repeat(N, 1)) if n_s else torch.zeros((0, 3), device=dev)
    ovr_scale = torch.log(scaling[idx_split].repeat(N, 1) / (0.8 * N)) if n_s else torch.zeros((0, 3), device=dev)
    ar = torch.arange(P, device=dev)
    src = torch.cat([ar, idx_clone, idx_split.repeat(N)])
    fresh = torch.cat([torch.zeros(P, dtype=torch.bool, device=dev), torch.ones(n_c + N * n_s, dtype=torch.bool, device=dev)])
    ovr = torch.cat([torch.full((P + n_c,), -1, dtype=torch.long, device=dev), torch.arange(N * n_s, device=dev)])
    keep = torch.cat([~split_mask, torch.ones(n_c + N * n_s, dtype=torch.bool, device=dev)])      # prune the split originals (:615-616)
    # ---- prune (:659-664): low opacity (+ optionally too large on screen / in the world); children inherit their source's opacity
    opac = torch.sigmoid(opacity_logit.view(-1))[src]
    prune = opac < cfg.min_opacity
    if cfg.max_screen_size:
        # (statistics were reset by the densification postfix, :577-580: max_radii2D is zero for every row at this point)
        cur_scale = torch.exp(log_scale)[src]
        cur_scale = torch.where(ovr.view(-1, 1) >= 0, torch.exp(ovr_scale)[ovr.clamp(min=0)], cur_scale)
        prune = prune | (cur_scale.max(dim=1).values > 0.1 * cfg.extent)
    keep = keep & ~prune
    sel = torch.nonzero(keep).squeeze(1)
    return dict(src=src[sel].to(torch.int32).contiguous(), fresh=fresh[sel].to(torch.int32).contiguous(),
                ovr=ovr[sel].to(torch.int32).contiguous(), ovr_xyz=ovr_xyz.contiguous().float(), ovr_scale=ovr_scale.contiguous().float(),
                n_clone=n_c, n_split=n_s, n_pruned=int((~keep).sum()) - n_s)


def densify_and_prune(params, state, cfg, generator=None):
    """Applies `plan` to a GaussianParams store in place (new flat param / exp_avg / exp_avg_sq buffers, statistics reset).
    Returns the plan (for logging / tests)."""
    L = _cabi.lib()
    lv = params.leaves
    pl = plan(lv["xyz"].detach(), lv["rotation"].detach(), lv["opacity"].detach(), lv["scaling"].detach(), state, cfg, generator)
    P_new = int(pl["src"].numel())
    dev = params.device
    per = sum(k for _, k in GROUPS)
    new_flat = torch.empty(per * P_new, dtype=torch.float32, device=dev)
    new_m = torch.empty_like(new_flat)
    new_v = torch.empty_like(new_flat)
    order = ("xyz", "rotation", "shs", "opacity", "scaling")
    kk = dict(GROUPS)
    off_old = (C.c_size_t * 5)(*[params.spans[n][0] for n in order])
    offs, o = {}, 0
    for name, k in GROUPS:
        offs[name] = o
        o += k * P_new
    off_new = (C.c_size_t * 5)(*[offs[n] for n in order])
    if P_new:
        with torch.cuda.device(dev):
            rc = L.igs_densify_remap(torch.cuda.current_stream(dev).cuda_stream, P_new, 16, pl["src"].data_ptr(), pl["fresh"].data_ptr(),
                                     pl["ovr"].data_ptr(), pl["ovr_xyz"].data_ptr() if pl["ovr_xyz"].numel() else pl["src"].data_ptr(),
                                     pl["ovr_scale"].data_ptr() if pl["ovr_scale"].numel() else pl["src"].data_ptr(),
                                     params.flat.data_ptr(), params.exp_avg.data_ptr(), params.exp_avg_sq.data_ptr(), off_old,
                                     new_flat.data_ptr(), new_m.data_ptr(), new_v.data_ptr(), off_new)
        if rc != 0:
            raise RuntimeError("igs_densify_remap failed: %d" % rc)
    params._bind(P_new, new_flat, new_m, new_v)
    state.reset(P_new, dev)
    return pl
