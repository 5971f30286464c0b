"""Per-frame refine loop: render -> loss -> backward -> (all-reduce) -> Adam, views sharded over ranks.

This is the build's own driver for the loop the reference runs in `infer_batch.py:245-357` (spec, not shipped code):
  * parameters and activations as `GaussianModel.load_fromstream` sets them up (igs/models/gaussian_model.py:265-348,
    90-127): raw xyz / rotation / shs / opacity-logit / log-scale, sigmoid / exp / L2-normalise applied outside the
    rasterizer, `Adam(lr=0, eps=1e-15)` with per-group learning rates (configs/demo.yaml:64-69);
  * `forward_single_view` settings (infer_batch.py:60-79): scale_modifier 1, kernel_size 0, require_coord = require_depth
    = True, prefiltered False, SH degree 3;
  * loss `lambda_l1 * L1 + (1 - lambda_l1) * (1 - SSIM)` with lambda_l1 = 0.8 (infer_batch.py:302-305), or L1 only
    (BASELINE.json config 3).

MI355X-first layout: all five parameter groups live in ONE flat fp32 buffer (59 floats per Gaussian, 47.2 MB at 200k)
with a matching flat gradient buffer, so the multi-GPU exchange is a single RCCL all-reduce over xGMI and the Adam update
is five contiguous fused-kernel launches.  Views are sharded over ranks: at step s rank r renders view perm[s*N + r];
every rank applies the same averaged gradient, so replicas stay identical without a broadcast.
"""
import math
import os

import torch
import torch.nn.functional as F      # (F.normalize of the raw quaternion: the caller-side activation)

from . import _cabi
from . import rasterizer as _rast
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

# the 11 "small" floats first, SH last: the multi-GPU exchange reduces the small groups as ONE contiguous range
GROUPS = (("xyz", 3), ("rotation", 4), ("opacity", 1), ("scaling", 3), ("shs", 48))
DEFAULT_LRS = dict(xyz=0.0016, rotation=0.01, shs=0.0025, opacity=0.05, scaling=0.005)      # configs/demo.yaml:64-69


class GaussianParams:
    """Flat SoA parameter store + Adam state (replica held by every rank)."""

    def __init__(self, raw, device, lrs=None, betas=(0.9, 0.999), eps=1e-15):
        P = raw["xyz"].shape[0]
        self.device = device
        self.lrs = dict(DEFAULT_LRS if lrs is None else lrs)
        self.betas, self.eps, self.step_count = betas, eps, 0
        total = sum(k for _, k in GROUPS) * P
        flat = torch.empty(total, dtype=torch.float32, device=device)
        o = 0
        for name, k in GROUPS:
            n = k * P
            flat[o:o + n].copy_(raw[name].reshape(-1).to(device))
            o += n
        self._bind(P, flat, torch.zeros(total, dtype=torch.float32, device=device),
                   torch.zeros(total, dtype=torch.float32, device=device))

    def _bind(self, P, flat, exp_avg, exp_avg_sq):
        """(Re)binds the store to flat buffers of P Gaussians: spans, aliasing leaves and a fresh gradient buffer."""
        self.P = P
        self.flat, self.exp_avg, self.exp_avg_sq = flat, exp_avg, exp_avg_sq
        self.grad = torch.zeros(flat.numel(), dtype=torch.float32, device=self.device)
        self.spans, self.leaves = {}, {}
        for key in ("_adam_groups", "_adam_groups_small"):
            if hasattr(self, key):
                delattr(self, key)
        o = 0
        for name, k in GROUPS:
            n = k * P
            self.spans[name] = (o, n)
            shape = (P, 16, 3) if name == "shs" else (P, k)
            leaf = self.flat[o:o + n].view(shape).requires_grad_(True)     # a leaf that aliases the flat buffer
            leaf.grad = self.grad[o:o + n].view(shape)                     # autograd accumulates in place into the flat gradient
            self.leaves[name] = leaf
            o += n

    def spatial_sort(self, bits=10):
        """Reorders the Gaussians along a Morton (Z-order) curve of their positions, in place (parameters and Adam moments, one
        gather pass).  Consecutive Gaussians then project to neighbouring tiles, which lets the binning stage reserve slots
        once per (workgroup, tile) instead of once per instance (preprocess.hip).  The rasterizer's results do not depend on
        the order (apart from which of two splats at EXACTLY the same depth comes first); `self.order[i]` is the index Gaussian i
        had when the store was created -- `original_order()` undoes every sort so far."""
        import ctypes as C
        xyz = self.leaves["xyz"].detach().contiguous()
        lohi = torch.cat([xyz.min(dim=0).values, xyz.max(dim=0).values]).contiguous()      # stays on the device: no host read-back
        Lb = _cabi.lib()
        perm = torch.empty(self.P, dtype=torch.int32, device=self.device)
        need = Lb.igs_morton_order_scratch_bytes(self.P)
        if getattr(self, "_morton_scratch", None) is None or self._morton_scratch.numel() < need:
            self._morton_scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            # keys, identity and the stable radix sort are the library's own (sort.hip); same order as torch.argsort(code, stable=True)
            rc = Lb.igs_morton_order(torch.cuda.current_stream(self.device).cuda_stream, self.P, xyz.data_ptr(), lohi.data_ptr(), bits,
                                     self._morton_scratch.data_ptr(), perm.data_ptr())
        if rc != 0:
            raise RuntimeError("igs_morton_order failed: %d" % rc)
        P = self.P
        new = [torch.empty_like(self.flat) for _ in range(3)]
        off = (C.c_size_t * 5)(*[self.spans[n][0] for n in ("xyz", "rotation", "shs", "opacity", "scaling")])
        with torch.cuda.device(self.device):
            rc = _cabi.lib().igs_densify_remap(torch.cuda.current_stream(self.device).cuda_stream, P, 16, perm.data_ptr(), None, None, None,
                                               None, self.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), off,
                                               new[0].data_ptr(), new[1].data_ptr(), new[2].data_ptr(), off)
        if rc != 0:
            raise RuntimeError("igs_densify_remap failed: %d" % rc)
        prev = getattr(self, "order", None)
        self._bind(P, *new)
        self.order = perm.long() if prev is None or prev.numel() != P else prev[perm.long()]
        return perm

    def original_order(self):
        """Raw leaves permuted back to the order the store was created with (valid while no densification changed the set)."""
        out = {}
        inv = None
        if getattr(self, "order", None) is not None and self.order.numel() == self.P:
            inv = torch.empty_like(self.order)
            inv[self.order] = torch.arange(self.P, device=self.order.device)
        for k, v in self.leaves.items():
            out[k] = v.detach()[inv] if inv is not None else v.detach()
        return out

    def activated(self, fused=False):
        """Activated parameters as the reference's model properties give them (gaussian_model.py:90-127); `fused`: the same from one
        autograd Function (igs_amd/activations.py) instead of separate PyTorch ops."""
        L = self.leaves
        if fused:
            from .activations import activate
            o, s, r = activate(L["opacity"], L["scaling"], L["rotation"])
            return dict(means3D=L["xyz"], shs=L["shs"], opacities=o, scales=s, rotations=r)
        return dict(means3D=L["xyz"], shs=L["shs"], opacities=torch.sigmoid(L["opacity"]), scales=torch.exp(L["scaling"]),
                    rotations=F.normalize(L["rotation"]))

    def raw(self):
        return {k: v.detach() for k, v in self.leaves.items()}

    def zero_grad(self):
        self.grad.zero_()

    def adam_step(self, skip_sh=False):
        """torch.optim.Adam semantics, all parameter groups in one fused HIP launch (`skip_sh`: every group but the SH
        coefficients -- the multi-GPU exchange updates those itself, igs_adam_sh_from_view_colors)."""
        L = _cabi.lib()
        self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.step_count
        bc2s = math.sqrt(1.0 - b2 ** self.step_count)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        key = "_adam_groups_small" if skip_sh else "_adam_groups"
        if not hasattr(self, key):
            import ctypes as C
            names = [n for n, _ in GROUPS if not (skip_sh and n == "shs")]
            k = len(names)
            setattr(self, key, ((C.c_size_t * k)(*[self.spans[n][0] for n in names]),
                                (C.c_size_t * k)(*[self.spans[n][1] for n in names]),
                                (C.c_float * k)(*[self.lrs[n] for n in names]), k))
        off, cnt, lrs, k = getattr(self, key)
        rc = L.igs_adam_step_groups(stream, k, off, cnt, lrs, self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                    self.exp_avg_sq.data_ptr(), b1, b2, self.eps, bc1, bc2s)
        if rc != 0:
            raise RuntimeError("igs_adam_step_groups failed: %d" % rc)


def render(params_act, cam, bg, sh_degree=3, require_coord=True, require_depth=True, means2D=None, debug=False, clamp=False):
    """`forward_single_view` (infer_batch.py:39-124) on an activated parameter dict."""
    settings = GaussianRasterizationSettings(
        image_height=int(cam.height), image_width=int(cam.width), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, kernel_size=0.0,
        bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, require_depth=require_depth,
        require_coord=require_coord, debug=debug)
    if means2D is None:
        means2D = torch.zeros_like(params_act["means3D"], requires_grad=True)
    Rast = GaussianRasterizer
    if clamp:        # igs/models/gs.py:39 imports the clamp package (gradients clamped to +-15)
        from .rasterizer import GaussianRasterizerClamp as Rast
    color, radii, coord, mcoord, depth, mdepth, alpha, normal = Rast(settings)(
        means3D=params_act["means3D"], means2D=means2D, opacities=params_act["opacities"], shs=params_act["shs"],
        scales=params_act["scales"], rotations=params_act["rotations"])
    return dict(images_pred=color, depth_pred=depth, radii=radii, visibility_filter=radii > 0, viewspace_points=means2D,
                coord=coord, mcoord=mcoord, mdepth=mdepth, alpha=alpha, normal=normal)


class L1Fused:
    """Fused L1 forward+backward on the HIP side: returns d(mean |pred-gt|)/dpred without autograd bookkeeping."""

    def __init__(self, device):
        self.device = device
        self.loss_sum = torch.zeros(1024, dtype=torch.float32, device=device)      # 64 shards, 16 floats apart

    def __call__(self, pred, gt, grad_out, weight=1.0):
        L = _cabi.lib()
        n = pred.numel()
        self.loss_sum.zero_()
        rc = L.igs_l1_loss_fwd_bwd(torch.cuda.current_stream(self.device).cuda_stream, n, pred.data_ptr(), gt.data_ptr(),
                                   grad_out.data_ptr(), self.loss_sum.data_ptr(), weight / n)
        if rc != 0:
            raise RuntimeError("igs_l1_loss_fwd_bwd failed: %d" % rc)
        return self.loss_sum        # 64 partial sums of |pred - gt| at [::16] (call .sum() when the value is needed)


class GtStatsCache:
    """blur(gt), blur(gt^2) of the SSIM loss per ground-truth image (`gt_stats` of include/igs_rast.h): the reference recomputes them in
    every iteration (loss_utils.py:34-63) although a view's ground truth does not change while a frame is refined (50 iterations
    over 10 views: every view comes back five times).  One buffer per view, recycled from frame to frame through a pool per
    (H, W, device).  `get(view, gt)` returns (pointer, valid): valid is 0 on the first visit of this ground-truth tensor -- the
    call that then FILLS the buffer -- and 1 afterwards."""
    _pool = {}          # (H, W, device) -> free buffers
    KEEP = 32

    def __init__(self, device):
        self.device = device
        self.entries = {}          # view -> [buffer, identity of the gt it holds, pool key]

    def get(self, view, gt):
        """(pointer, valid) for `gt` as view `view`'s ground truth.  The entry keeps a REFERENCE to the tensor it was filled from (the
        allocator cannot hand its address to another image while the entry lives) and its version counter; a first visit returns valid = 0
        and the entry only counts as filled once `confirm(view)` has been called after the launch that fills it succeeded."""
        H, W = int(gt.shape[-2]), int(gt.shape[-1])
        e = self.entries.get(view)
        if e is not None and e[1] is not None and e[1][0] is gt and e[1][1] == gt._version and e[3]:
            return e[0].data_ptr(), 1
        if e is None or e[2] != (H, W, str(self.device)):
            if e is not None:
                self._give_back(e)
            pk = (H, W, str(self.device))
            free = GtStatsCache._pool.setdefault(pk, [])
            buf = free.pop() if free else torch.empty(_cabi.lib().igs_ssim_gt_stats_bytes(W, H) // 4, dtype=torch.float32, device=self.device)
            e = self.entries[view] = [buf, None, pk, False]
        e[1], e[3] = (gt, gt._version), False
        return e[0].data_ptr(), 0

    def confirm(self, view):
        """The launch that fills view `view`'s buffer has been enqueued successfully: later visits may read it."""
        e = self.entries.get(view)
        if e is not None:
            e[3] = True

    @staticmethod
    def _give_back(e):
        free = GtStatsCache._pool.setdefault(e[2], [])
        if len(free) < GtStatsCache.KEEP:
            free.append(e[0])

    def __del__(self):
        try:
            for e in self.entries.values():
                self._give_back(e)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class L1SsimFused:
    """(1 - lambda) L1 + lambda (1 - SSIM), forward + backward in two HIP launches (igs_ssim_l1_loss_fwd_bwd)."""

    def __init__(self, device, lambda_dssim):
        self.device, self.lambda_dssim = device, float(lambda_dssim)
        self.scratch, self.key = None, None
        self.sums = torch.zeros(2048, dtype=torch.float32, device=device)

    def __call__(self, pred, gt, grad_out, weight=1.0, gt_stats=None):
        """`gt_stats`: (pointer, valid) from a GtStatsCache -- the ground truth's own SSIM statistics are then cached / reused."""
        L = _cabi.lib()
        H, W = int(pred.shape[-2]), int(pred.shape[-1])
        if self.key != (H, W):
            self.key = (H, W)
            self.scratch = torch.empty(L.igs_ssim_l1_scratch_bytes(W, H), dtype=torch.uint8, device=self.device)
        ptr, valid = gt_stats if gt_stats is not None else (None, 0)
        rc = L.igs_ssim_l1_loss_fwd_bwd_cached(torch.cuda.current_stream(self.device).cuda_stream, W, H, pred.data_ptr(), gt.data_ptr(),
                                               self.lambda_dssim, float(weight), self.scratch.data_ptr(), grad_out.data_ptr(),
                                               self.sums.data_ptr(), ptr, valid)
        if rc != 0:
            raise RuntimeError("igs_ssim_l1_loss_fwd_bwd failed: %d" % rc)
        return self.sums

    def value(self, n, weight=1.0):
        """Loss value of the last call (synchronises)."""
        ssim_mean = float(self.sums[:1024].sum().item()) / n
        l1_mean = float(self.sums[1024:].sum().item()) / n
        return weight * ((1.0 - self.lambda_dssim) * l1_mean + self.lambda_dssim * (1.0 - ssim_mean))


class Refiner:
    """One refine step = one view per rank: render, loss, backward, gradient all-reduce (N > 1), Adam."""

    def __init__(self, params, cams, gt_images, bg, loss="l1", lambda_l1=0.8, world_size=1, rank=0, seed=0,
                 render_fn=None, adam_fn=None, native=True, fused=True, densify=None, densify_seed=0,
                 lambda_depth_normal=0.0):
        self.params, self.cams, self.gt, self.bg = params, cams, gt_images, bg
        self.loss, self.lambda_l1 = loss, lambda_l1
        self.world_size, self.rank = world_size, rank
        # injectable for the CPU (gloo) tests of the sharding logic; the product path uses the HIP renderer / Adam
        self.render_fn = render if render_fn is None else render_fn
        self.adam_fn = params.adam_step if adam_fn is None else adam_fn
        self.l1 = L1Fused(params.device) if loss == "l1" else L1SsimFused(params.device, 1.0 - lambda_l1)
        self.gt_stats = GtStatsCache(params.device) if loss != "l1" else None      # the ground truth's SSIM statistics, per view (GtStatsCache)
        self.cache_gt_stats = True
        # RaDe-GS depth-normal regulariser (train.py:143-164; BASELINE cfg-5 uses 0.05): needs dL/d depth, mdepth, normal -- the
        # fused step evaluates it in one HIP launch and runs the <depth, normal> backward instance; the unfused native path does
        # not implement it (the autograd path does, through igs_amd.losses.depth_normal_loss = the same kernel)
        self.lambda_depth_normal = float(lambda_depth_normal)
        self.native = native          # drive the C ABI directly instead of going through autograd
        self.fused = fused            # ... and on a single GPU run the whole iteration as one library call (igs_refine_step)
        self.grad_img = None
        self.gen = torch.Generator().manual_seed(seed)      # same seed on every rank -> same view permutation
        self.order = []
        self.last_num_rendered = 0
        # optional densify-and-prune (configs/demo.yaml:57-62): a DensifyConfig; statistics and iteration counter per frame
        self.densify = densify
        # fused step: also return dL/d(screen-space mean) with its absolute-gradient column (what the densification statistics read);
        # off = the colour-only blend backward drops that moment
        self.want_viewspace_grad = densify is not None
        # N > 1: "colors" = gather the per-view colour gradients and rebuild dL/dSH locally (_colour_exchange_step);
        # "gradients" = one all-reduce of the flat gradient
        self.exchange = "colors"
        self.iteration = 0
        self.densify_state = None
        self.densify_gen = None
        self.densify_seed = densify_seed
        self.densify_log = []
        # bench.py: HIP-event pairs around the collectives of the N > 1 step (None = not recorded)
        self.exchange_events = None

    class _Timed:
        """Brackets a group of collectives with events on the current stream (the collective's own stream is joined to it)."""

        def __init__(self, owner):
            self.o = owner

        def __enter__(self):
            if self.o.exchange_events is not None:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.o.exchange_events is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.o.exchange_events.append((self.e0, e1))
            return False

    def exchange_ms_per_step(self, steps):
        """Mean time per step spent in the collectives since `exchange_events` was set to [] (synchronises)."""
        if not self.exchange_events or steps <= 0:
            return 0.0
        torch.cuda.synchronize(self.params.device)
        return sum(a.elapsed_time(b) for a, b in self.exchange_events) / steps

    def _next_view(self):
        """Without-replacement view sampling (infer_batch.py:280-288); a step consumes `world_size` views."""
        picks = []
        for _ in range(self.world_size):
            if not self.order:
                order = torch.randperm(len(self.cams), generator=self.gen).tolist()
                # a step that straddles two passes over the views (10 views on 8 ranks: the second step takes the 2 left over and 6 of
                # the next pass) must not hold a view twice while there are enough views: the ones this step already has are drawn
                # LAST from the new pass (pop() takes from the end).  Same generator on every rank -> same order everywhere.
                if picks and len(self.cams) >= self.world_size:
                    order = [v for v in order if v in picks] + [v for v in order if v not in picks]
                self.order = order
            picks.append(self.order.pop())
        self.last_picks = picks          # the views of every rank this step (same permutation everywhere)
        return picks[self.rank]

    def _native_step(self, cam, gt, defer=True):
        """render -> fused L1 -> backward -> activation backward, driving the C ABI directly (no autograd graph): the
        rasterizer writes dL/dxyz and dL/dsh straight into their spans of the flat gradient buffer."""
        p = self.params
        L = _cabi.lib()
        dev, P = p.device, p.P
        if not hasattr(self, "_bufs"):
            self._bufs = _rast.RasterBuffers()
        if getattr(self, "_native_P", None) != P:
            self._native_P = P
            self._act = torch.empty(8 * P, dtype=torch.float32, device=dev)       # opacity P | scale 3P | rot 4P
            self._dact = torch.empty(8 * P, dtype=torch.float32, device=dev)
            self._tmp = torch.empty(12 * P, dtype=torch.float32, device=dev)      # means2D 3 | colors 3 | cov3D 6
            self._empty = torch.Tensor([])
        stream = torch.cuda.current_stream(dev).cuda_stream
        raw = p.leaves
        opac, scal, rotn = self._act[:P].view(P, 1), self._act[P:4 * P].view(P, 3), self._act[4 * P:].view(P, 4)
        rc = L.igs_activate_fwd(stream, P, raw["opacity"].data_ptr(), raw["scaling"].data_ptr(), raw["rotation"].data_ptr(),
                                opac.data_ptr(), scal.data_ptr(), rotn.data_ptr())
        assert rc == 0
        e = self._empty
        with torch.no_grad():
            out = _rast.rasterize_gaussians(self.bg, raw["xyz"].detach(), e, opac, scal, rotn, 1.0, e, cam.world_view_transform,
                                            cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width,
                                            raw["shs"].detach(), 3, cam.camera_center, False, True, True, False, buffers=self._bufs,
                                            defer=defer)
            nr, color, coord, mcoord, alpha, normal, depth, mdepth, radii, gb, bb, ib = out
            if self.grad_img is None or self.grad_img.shape != color.shape:
                self.grad_img = torch.empty_like(color)
            if self.gt_stats is not None and self.cache_gt_stats and getattr(self, "_view", None) is not None:
                self.l1(color, gt, self.grad_img, weight=1.0 / self.world_size, gt_stats=self.gt_stats.get(self._view, gt))
                self.gt_stats.confirm(self._view)          # (l1 raises when the launch fails)
            else:
                self.l1(color, gt, self.grad_img, weight=1.0 / self.world_size)      # L1 or L1 + D-SSIM, fused fwd + bwd
            G = p.grad
            def span(name, shape):
                o, n = p.spans[name]
                return G[o:o + n].view(shape)
            d_op, d_sc, d_rt = self._dact[:P].view(P, 1), self._dact[P:4 * P].view(P, 3), self._dact[4 * P:].view(P, 4)
            outs = dict(means3D=span("xyz", (P, 3)), sh=span("shs", (P, 16, 3)), opacity=d_op, scales=d_sc, rotations=d_rt,
                        means2D=self._tmp[:3 * P].view(P, 3), colors=self._tmp[3 * P:6 * P].view(P, 3),
                        cov3D=self._tmp[6 * P:].view(P, 6))
            _rast.rasterize_gaussians_backward(self.bg, raw["xyz"].detach(), radii, e, scal, rotn, 1.0, e, cam.world_view_transform,
                                               cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, self.grad_img, None, None,
                                               None, None, None, None, normal, raw["shs"].detach(), 3, cam.camera_center, gb, nr,
                                               bb, ib, alpha, True, True, False, out=outs, workspace=self._bufs.workspace)
            rc = L.igs_activate_bwd(stream, P, opac.data_ptr(), scal.data_ptr(), raw["rotation"].data_ptr(), d_op.data_ptr(),
                                    d_sc.data_ptr(), d_rt.data_ptr(), span("opacity", (P, 1)).data_ptr(),
                                    span("scaling", (P, 3)).data_ptr(), span("rotation", (P, 4)).data_ptr())
            assert rc == 0
        if defer:
            # the whole frame was enqueued without the host knowing the instance count; look at it now (the 12-byte
            # read-back finished long ago on the GPU's timeline) and redo the frame in the rare case the list did not fit
            nr = _rast.rasterize_finish()
            if nr is None:
                return self._native_step(cam, gt, defer=False)
        self.last_num_rendered = nr
        return dict(images_pred=color, radii=radii, visibility_filter=None, viewspace_points=outs["means2D"], alpha=alpha,
                    depth_pred=depth, normal=normal)

    def _fused_step(self, cam, gt, grads_only=False, color_out=None, color_event=None):
        """Single-GPU step entirely inside the library: `igs_refine_step` (include/igs_rast.h) -- activations, render, L1,
        backward and the Adam update in 5 launches; no gradient array is materialised."""
        import ctypes as C
        p = self.params
        L = _cabi.lib()
        dev, P = p.device, p.P
        H, W = int(cam.height), int(cam.width)
        if not hasattr(self, "_bufs"):
            self._bufs = _rast.RasterBuffers()
        imgs, radii, ss = self._bufs.get(P, H, W, dev)
        if not hasattr(self, "_fused") or self._fused["m2d"].shape[0] != P:
            self._fused = dict(m2d=torch.zeros((P, 3), dtype=torch.float32, device=dev),
                               loss=torch.zeros(1, dtype=torch.float32, device=dev))
        a = _cabi.RefineStepArgs()
        a.stream = torch.cuda.current_stream(dev).cuda_stream
        if getattr(self, "_cb_of", None) is not ss:          # (callback address + the set's three `user` words, valid while `ss` lives)
            cb, ug, ub, ui = ss.callbacks()
            self._cb_of, self._cb = ss, (_cabi.ALLOC_FN(cb), ug, ub, ui)
        fn, ug, ub, ui = self._cb
        a.geometry_buffer, a.binning_buffer, a.image_buffer = fn, fn, fn
        a.geometry_user, a.binning_user, a.image_user = ug, ub, ui
        a.workspace = self._bufs.workspace.data_ptr()
        a.P, a.D, a.M, a.width, a.height = P, 3, 16, W, H
        a.background = self.bg.data_ptr()
        a.param, a.exp_avg, a.exp_avg_sq = p.flat.data_ptr(), p.exp_avg.data_ptr(), p.exp_avg_sq.data_ptr()
        a.grad_out = p.grad.data_ptr() if grads_only else None          # multi-GPU: gradients for the all-reduce, Adam afterwards
        a.off_xyz, a.off_rot, a.off_sh = p.spans["xyz"][0], p.spans["rotation"][0], p.spans["shs"][0]
        a.off_opacity, a.off_scale = p.spans["opacity"][0], p.spans["scaling"][0]
        a.lr_xyz, a.lr_rot, a.lr_sh = p.lrs["xyz"], p.lrs["rotation"], p.lrs["shs"]
        a.lr_opacity, a.lr_scale = p.lrs["opacity"], p.lrs["scaling"]
        a.beta1, a.beta2, a.eps = p.betas[0], p.betas[1], p.eps
        a.step = p.step_count + 1
        a.viewmatrix, a.projmatrix = cam.world_view_transform.data_ptr(), cam.full_proj_transform.data_ptr()
        a.cam_pos = cam.camera_center.data_ptr()
        a.tan_fovx, a.tan_fovy = cam.tanfovx, cam.tanfovy
        a.gt, a.loss_weight = gt.data_ptr(), getattr(self, "loss_scale", 1.0) / self.world_size      # gradients are averaged over the views of a step
        a.lambda_depth_normal, a.depth_ratio = self.lambda_depth_normal, 0.6
        if self.loss == "l1_ssim" or self.lambda_depth_normal > 0.0:
            if getattr(self, "_loss_scratch_key", None) != (H, W):
                self._loss_scratch_key = (H, W)
                self._loss_scratch = torch.empty(L.igs_refine_loss_scratch_bytes(W, H), dtype=torch.uint8, device=dev)
            a.lambda_dssim = (1.0 - self.lambda_l1) if self.loss == "l1_ssim" else 0.0
            a.loss_scratch = self._loss_scratch.data_ptr()
            if self.loss == "l1_ssim" and self.gt_stats is not None and self.cache_gt_stats and getattr(self, "_view", None) is not None:
                a.gt_stats, a.gt_stats_valid = self.gt_stats.get(self._view, gt)
        else:
            a.lambda_dssim, a.loss_scratch = 0.0, None
        a.out_images, a.radii = imgs.data_ptr(), radii.data_ptr()
        a.dL_dmean2D, a.loss_out = (self._fused["m2d"].data_ptr() if self.want_viewspace_grad else None), self._fused["loss"].data_ptr()
        rq = 1 if getattr(self, "require_geometry", True) else 0      # the reference's loop always renders coord / depth / normal
        a.require_coord, a.require_depth = rq, rq
        a.clamp_grads = 15.0 if getattr(self, "clamp", False) else 0.0
        a.color_grad_out = color_out.data_ptr() if color_out is not None else None
        if color_event is not None:
            h = color_event.cuda_event                         # ctypes.c_void_p of the hipEvent_t
            a.color_ready_event = getattr(h, "value", h)
        else:
            a.color_ready_event = None
        a.scratch_clean = 0 if os.environ.get("IGS_SCRATCH_CLEAN") == "0" else 1      # (RasterBuffers: zero-filled at allocation, touched by this library only)
        with torch.cuda.device(dev):
            nr = L.igs_refine_step(C.byref(a))
        _rast._check(nr, "igs_refine_step")
        if a.gt_stats:
            self.gt_stats.confirm(self._view)              # filled (or read) by a call that succeeded
        if not grads_only:
            p.step_count += 1
        self.last_num_rendered = nr
        return dict(images_pred=imgs[0:3], radii=radii, visibility_filter=None, viewspace_points=self._fused["m2d"] if self.want_viewspace_grad else None,
                    alpha=imgs[11:12], depth_pred=imgs[9:10], normal=imgs[12:15], loss=self._fused["loss"])

    def _ssim_value(self, img, gt):
        """Autograd path: mean SSIM from the fused HIP kernels (igs_amd.losses.ssim); `ssim_fn` lets a test inject another
        differentiable implementation (the PyTorch restatement of the test tree) for an independent comparison."""
        fn = getattr(self, "ssim_fn", None)
        if fn is not None:
            return fn(img, gt)
        from .losses import ssim as fused_ssim
        return fused_ssim(img, gt.unsqueeze(0), size_average=False).squeeze()

    def _depth_normal_value(self, pkg, cam):
        fn = getattr(self, "depth_normal_fn", None)
        if fn is not None:
            return fn(pkg, cam)
        from .losses import depth_normal_loss
        return depth_normal_loss(pkg, cam)

    def start_frame(self):
        """A new frame of the stream begins (infer_batch.py:270-278): iteration counter and densification statistics restart."""
        self.iteration = 0
        self.densify_state = None

    def _densify_hooks(self, pkg, did_adam):
        """infer_batch.py:308-321, after the backward of iteration `self.iteration`: statistics, then (every `interval`
        iterations) densify-and-prune.  Returns True when the Gaussian set was rebuilt."""
        from . import densify as _dn
        cfg, p = self.densify, self.params
        rebuilt = False
        if self.iteration < cfg.until_iter:
            if self.densify_state is None or self.densify_state.denom.numel() != p.P:
                self.densify_state = _dn.DensifyState(p.P, p.device)
            self.densify_state.add(pkg["viewspace_points"], pkg["radii"])
            if self.iteration > cfg.from_iter and self.iteration % cfg.interval == 0:
                assert not did_adam
                if self.densify_gen is None:
                    self.densify_gen = torch.Generator(device=p.device).manual_seed(self.densify_seed)      # same seed on every rank
                _dn.reduce_state(self.densify_state, self.world_size)      # N > 1: sum / max of the per-rank statistics
                pl = _dn.densify_and_prune(p, self.densify_state, cfg, self.densify_gen)
                self.densify_log.append((self.iteration, pl["n_clone"], pl["n_split"], pl["n_pruned"], p.P))
                rebuilt = True
        return rebuilt

    def _densify_due(self):
        cfg = self.densify
        return (cfg is not None and self.iteration < cfg.until_iter and self.iteration > cfg.from_iter
                and self.iteration % cfg.interval == 0)

    def _native_then_adam(self, cam, gt):
        pkg = self._native_step(cam, gt)
        self.adam_fn()
        return pkg

    def _colour_exchange_step(self, cam, gt, picks):
        """N > 1: the step's gradient with 3.4x fewer bytes on the wire than an all-reduce of the flat buffer.  dL/dSH (48 of the 59
        floats per Gaussian) is linear in the per-view colour gradient: sum_v basis(dir_v) x dL/dcolour_v.  So the ranks all-gather
        the [P,3] colour gradients (12 B per Gaussian and view), every rank rebuilds the SH gradient of the whole step itself
        (`igs_sh_grad_from_view_colors`, views in rank order: identical bits everywhere), and only the 11 small-group floats go
        through all-reduces.  At N = 8, 200k Gaussians: 19 MB gathered + 8.8 MB reduced instead of 47 MB reduced."""
        import torch.distributed as dist
        p = self.params
        L = _cabi.lib()
        P, N, dev = p.P, self.world_size, p.device
        if getattr(self, "_gc", None) is None or self._gc.shape[1] != P:
            self._gc = torch.zeros((N, P, 3), dtype=torch.float32, device=dev)
            self._gc_mine = torch.zeros((P, 3), dtype=torch.float32, device=dev)
            self._campos_host = {}
        # The view's colour gradients are final right after the blend backward, one kernel before the step ends: the library writes them
        # there and records an event (igs_refine_step_args::color_ready_event); the all-gather waits for THAT on a side stream and runs
        # underneath the per-Gaussian kernel (geom_bwd, ~70 us) instead of behind it.  `overlap_exchange = False`: gather after the step.
        overlap = getattr(self, "overlap_exchange", True) and dev.type == "cuda"
        if overlap and getattr(self, "_color_event", None) is None:
            self._color_event = torch.cuda.Event()
            self._color_event.record(torch.cuda.current_stream(dev))       # (creates the hipEvent_t the library will record)
            self._comm_stream = torch.cuda.Stream(device=dev)
        pkg = self._fused_step(cam, gt, grads_only=True, color_out=self._gc_mine, color_event=self._color_event if overlap else None)

        def gather():
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(self._gc.view(-1), self._gc_mine.view(-1))      # RCCL: straight into the [N,P,3] buffer
            else:
                dist.all_gather(list(self._gc.unbind(0)), self._gc_mine)
        if overlap:
            main = torch.cuda.current_stream(dev)
            self._comm_stream.wait_event(self._color_event)
            with torch.cuda.stream(self._comm_stream):
                gather()
            with self._Timed(self):
                main.wait_stream(self._comm_stream)                    # (what is left of the gather once geom_bwd is done)
        else:
            with self._Timed(self):
                gather()
        import ctypes as C
        for v in picks:
            if v not in self._campos_host:          # (one device read per camera, the first time it is used)
                self._campos_host[v] = [float(x) for x in self.cams[v].camera_center.reshape(3).tolist()]
        campos = (C.c_float * (3 * N))(*[x for v in picks for x in self._campos_host[v]])
        (sh0, shn) = p.spans["shs"]
        clamp = 15.0 if getattr(self, "clamp", False) else 0.0
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self.adam_fn == p.adam_step:
            # the library's own optimiser: the SH update straight from the gathered colours (no SH gradient in HBM), the 11 small
            # floats through the all-reduce and the grouped Adam launch
            b1, b2 = p.betas
            t = p.step_count + 1
            with self._Timed(self):
                if sh0 > 0:
                    dist.all_reduce(p.grad[:sh0], op=dist.ReduceOp.SUM)
                if sh0 + shn < p.grad.numel():
                    dist.all_reduce(p.grad[sh0 + shn:], op=dist.ReduceOp.SUM)
            # ONE launch for the whole optimiser step: SH coefficients from the gathered colours, the 11 small floats from their
            # all-reduced gradients (igs_adam_exchange_step)
            sp = p.spans
            rc = L.igs_adam_exchange_step(stream, P, 3, 16, N, C.cast(campos, C.c_void_p), self._gc.data_ptr(), clamp, p.flat.data_ptr(),
                                          p.exp_avg.data_ptr(), p.exp_avg_sq.data_ptr(), p.grad.data_ptr(), sp["xyz"][0], sp["rotation"][0],
                                          sp["shs"][0], sp["opacity"][0], sp["scaling"][0], p.lrs["xyz"], p.lrs["rotation"], p.lrs["shs"],
                                          p.lrs["opacity"], p.lrs["scaling"], b1, b2, p.eps, 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t))
            _rast._check(rc, "igs_adam_exchange_step")
            p.step_count += 1
            return pkg
        rc = L.igs_sh_grad_from_view_colors(stream, P, 3, 16, N, p.flat.data_ptr() + 4 * p.spans["xyz"][0],
                                            C.cast(campos, C.c_void_p), self._gc.data_ptr(), clamp,
                                            p.grad.data_ptr() + 4 * sh0)
        _rast._check(rc, "igs_sh_grad_from_view_colors")
        with self._Timed(self):
            if sh0 > 0:
                dist.all_reduce(p.grad[:sh0], op=dist.ReduceOp.SUM)             # xyz | rotation | opacity | scaling: 11 floats per Gaussian
            if sh0 + shn < p.grad.numel():
                dist.all_reduce(p.grad[sh0 + shn:], op=dist.ReduceOp.SUM)
        self.adam_fn()
        return pkg

    def _exchange_step(self, cam, gt, explicit_view):
        """N > 1 (or an injected optimiser): the same fused launches, ending in the flat gradient instead of the update; then the
        exchange over RCCL / xGMI and the identical Adam step on every rank."""
        p = self.params
        picks = getattr(self, "last_picks", None) if explicit_view is None else None
        if 1 < self.world_size <= 64 and self.fused and self.exchange == "colors" and picks is not None:      # (the library takes at most 64 views)
            return self._colour_exchange_step(cam, gt, picks)
        pkg = self._fused_step(cam, gt, grads_only=True) if self.fused else self._native_step(cam, gt)
        if self.world_size > 1:
            import torch.distributed as dist
            with self._Timed(self):
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)      # one flat 59*P-float buffer over RCCL / xGMI
        self.adam_fn()
        return pkg

    # ---- one refine iteration -------------------------------------------------------------------------------------------------
    # `step()` picks the view and dispatches to ONE of five small methods; which one follows from how the Refiner was set up
    # (`_mode()`), not from branches spread over the body:
    #   densify    native path with densify-and-prune hooks around the step (infer_batch.py:308-321)
    #   fused      one GPU, the library's optimiser: the whole iteration inside igs_refine_step
    #   exchange   N > 1 ranks, an injected optimiser, or fused=False: the native launches end in the flat gradient, then the exchange
    #              (colours / gradients / none) and Adam
    #   direct     autograd path (the reference's Python API), gradients handed straight to the fused Adam (zero_grad(set_to_none) semantics)
    #   autograd   autograd path accumulating into the flat gradient buffer (all-reduce for N > 1, injected render / optimiser functions)
    def _native_ok(self):
        return self.native and self.render_fn is render and (self.fused or self.lambda_depth_normal == 0.0)

    def _mode(self):
        p = self.params
        own_adam = self.adam_fn == p.adam_step
        if self.densify is not None:
            if not (self._native_ok() and own_adam):
                raise NotImplementedError("densify-and-prune is implemented for the native path with the library's optimiser only "
                                          "(native=True, no injected render_fn / adam_fn)")
            return "densify"
        if self._native_ok():
            return "fused" if (self.fused and self.world_size == 1 and own_adam) else "exchange"
        if (self.world_size == 1 and own_adam and self.render_fn is render and p.flat.is_cuda and getattr(self, "direct_adam", False)):
            return "direct"          # opt-in: tests and callers that read `.grad` keep the flat buffer
        return "autograd"

    def step(self, view=None):
        explicit_view = view
        if view is None:
            view = self._next_view()
        cam, gt = self.cams[view], self.gt[view]
        self._view = view
        mode = self._mode()
        if mode == "densify":
            pkg = self._step_densify(cam, gt, explicit_view)      # (its hooks read the iteration number BEFORE it is advanced, as the reference's loop does)
            self.iteration += 1
            return pkg
        self.iteration += 1
        if mode == "fused":
            return self._fused_step(cam, gt)
        if mode == "exchange":
            return self._exchange_step(cam, gt, explicit_view)
        return self._step_autograd(cam, gt, direct=(mode == "direct"))

    def _step_densify(self, cam, gt, explicit_view):
        """The reference updates the statistics after every backward and, on a densification iteration, rebuilds the Gaussians BEFORE
        optimizer.step() -- which then finds no gradients and does nothing (infer_batch.py:308-324).  N > 1: every rank adds the
        statistics of its own view; they are reduced over ranks right before the decision (densify.reduce_state), which every rank
        then takes identically."""
        if self._densify_due():
            pkg = self._fused_step(cam, gt, grads_only=True) if self.fused else self._native_step(cam, gt)    # gradients only, no Adam
            self._densify_hooks(pkg, did_adam=False)
            return pkg
        if self.world_size == 1:
            pkg = self._fused_step(cam, gt) if self.fused else self._native_then_adam(cam, gt)
        else:
            pkg = self._exchange_step(cam, gt, explicit_view)
        if self.iteration < self.densify.until_iter:
            self._densify_hooks(pkg, did_adam=True)
        return pkg

    def _autograd_loss(self, pkg, cam, gt):
        """The loss of the autograd path as a differentiable scalar: L1 (+ D-SSIM) (+ depth-normal regulariser), infer_batch.py:300-306."""
        img = pkg["images_pred"]
        loss = torch.abs(img - gt).mean()
        if self.loss != "l1":
            loss = self.lambda_l1 * loss + (1.0 - self.lambda_l1) * (1.0 - self._ssim_value(img, gt))
        if self.lambda_depth_normal > 0.0:
            self.last_depth_normal_loss = self._depth_normal_value(pkg, cam)
            loss = loss + self.lambda_depth_normal * self.last_depth_normal_loss
        return loss * getattr(self, "loss_scale", 1.0)

    def _step_autograd(self, cam, gt, direct):
        """The reference's own loop through the autograd Function.  `direct`: `optimizer.zero_grad(set_to_none=True)` semantics
        (infer_batch.py:324) -- torch.autograd.grad returns the five gradient tensors and one fused Adam launch per group consumes them
        (no 47 MB zero fill, no 47 MB accumulate into the flat gradient buffer); otherwise the gradients accumulate into the flat
        buffer (all-reduce for N > 1, injected optimisers, tests that read `.grad`)."""
        p = self.params
        if not direct:
            p.zero_grad()
        act = p.activated(fused=getattr(self, "fused_activations", False) and self.render_fn is render and p.flat.is_cuda)
        if getattr(self, "clamp", False) and self.render_fn is render:
            pkg = self.render_fn(act, cam, self.bg, clamp=True)
        else:
            pkg = self.render_fn(act, cam, self.bg)
        img = pkg["images_pred"]
        pure_l1 = self.loss == "l1" and self.lambda_depth_normal == 0.0
        if pure_l1:         # fused L1 forward + gradient in one launch, handed to autograd as the upstream gradient of the image
            if self.grad_img is None or self.grad_img.shape != img.shape:
                self.grad_img = torch.empty_like(img)
            self.l1(img, gt, self.grad_img, weight=1.0 if direct else 1.0 / self.world_size)      # (gradients are averaged over the views of a step)
        if direct:
            names = [n for n, _ in GROUPS]
            leaves = [p.leaves[n] for n in names]
            grads = (torch.autograd.grad([img], leaves, [self.grad_img]) if pure_l1
                     else torch.autograd.grad([self._autograd_loss(pkg, cam, gt)], leaves))
            L = _cabi.lib()
            p.step_count += 1
            b1, b2 = p.betas
            bc1, bc2s = 1.0 - b1 ** p.step_count, math.sqrt(1.0 - b2 ** p.step_count)
            stream = torch.cuda.current_stream(p.device).cuda_stream
            for n, g in zip(names, grads):
                o, cnt = p.spans[n]
                g = g.contiguous()
                rc = L.igs_adam_step(stream, cnt, p.flat.data_ptr() + 4 * o, g.data_ptr(), p.exp_avg.data_ptr() + 4 * o,
                                     p.exp_avg_sq.data_ptr() + 4 * o, p.lrs[n], b1, b2, p.eps, bc1, bc2s)
                if rc != 0:
                    raise RuntimeError("igs_adam_step failed: %d" % rc)
            return pkg
        if pure_l1:
            img.backward(gradient=self.grad_img)
        else:
            (self._autograd_loss(pkg, cam, gt) / self.world_size).backward()
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)      # one flat 59*P-float buffer over RCCL / xGMI
        self.adam_fn()
        return pkg


def psnr(img, gt):
    """infer_batch.py:350-353."""
    return -10.0 * torch.log10(torch.mean((torch.clamp(img, 0, 1) - gt) ** 2))
