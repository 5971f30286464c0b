"""Host-side mirror of the reference's rasterizer binding.

Layers (reference file:line in parentheses; DGR = submodules/RaDe-GS/submodules/diff-gaussian-rasterization):
  * `_C` = igs_amd/_C.*.so, the COMPILED module (igs_amd/csrc_torch/igs_torch_ext.cpp) with the four functions DGR/ext.cpp:15-20 exports
    -- `rasterize_gaussians`, `rasterize_gaussians_backward`, `mark_visible`, `integrate_gaussians_to_points` (raises) -- and the
    positional signatures of DGR/rasterize_points.h:18-107: torch glue over the C ABI of include/igs_rast.h (libigs_rast.so);
  * this module: `_RasterizeGaussians`, `GaussianRasterizationSettings`, `GaussianRasterizer`, `rasterize_gaussians_autograd`
    = DGR/diff_gaussian_rasterization_rade/__init__.py:21-243, same field order, argument validation messages, saved tensors and
    8-tuple output order `(color, radii, coord, mcoord, depth, mdepth, alpha, normal)`; plus thin wrappers of the `_C` functions
    with the extensions the refine loop uses (reusable buffers, deferred instance count, preallocated gradient destinations).

torch is used for device memory and the current stream only; all arithmetic runs in libigs_rast.so.
There is no CPU path: calling these functions without the HIP library / the compiled module / a GPU raises.
"""
import contextlib
import ctypes as C
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _cabi

_C = _cabi.ext()                       # builds (first use in a fresh tree) and imports igs_amd/_C.*.so; raises if that is impossible
RasterizerError = _C.RasterizerError   # RuntimeError subclass raised by the compiled module
ScratchSet = _C.ScratchSet


def _check(rc, what):
    if rc < 0:
        raise RasterizerError("%s failed (%d): %s" % (what, rc, _cabi.last_error()))
    return rc


class RasterBuffers:
    """Reusable output / scratch tensors for callers that render in a loop (the refine loop): one render's outputs are
    overwritten by the next, so only use it when the previous results are no longer needed."""

    def __init__(self):
        self.key, self.imgs, self.radii, self.scratch, self.workspace = None, None, None, None, None

    def get(self, P, H, W, dev):
        key = (P, H, W, dev)
        if key != self.key:
            self.key = key
            self.imgs = torch.zeros((15, H, W), dtype=torch.float32, device=dev)
            self.radii = torch.zeros((P,), dtype=torch.int32, device=dev)
            self.scratch = ScratchSet(dev, True)          # persistent: born zero-filled, used by this library only
            self.workspace = self.scratch.workspace(P)
        return self.imgs, self.radii, self.scratch


class _ScratchPool:
    """Scratch for the autograd binding.  The reference allocates three fresh byte tensors per forward
    (rasterize_points.cu:80-87); here a {geometry, binning, image, backward workspace} set is leased per
    (P, H, W, device, stream) and handed back when the autograd node that saved it dies (after `loss.backward()` /
    when the graph is dropped; at once under `torch.no_grad()`), so a render loop allocates nothing after its second
    iteration.  A set is never shared between two live graphs.  Outputs (images, radii, gradients) are always fresh tensors."""

    KEEP = 4

    def __init__(self):
        self.free = {}

    def acquire(self, key):
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        return ScratchSet(key[3], True)

    def release(self, key, item):
        lst = self.free.setdefault(key, [])
        if len(lst) < self.KEEP:
            lst.append(item)
        if len(self.free) > 8:            # sizes come and go (densification): forget the oldest keys
            for k in list(self.free)[:-8]:
                del self.free[k]


_POOL = _ScratchPool()


class CaptureScratch:
    """Scratch sets for a FAMILY of captured graphs that never run concurrently (igs_amd/graphs.py: one graph per view, replayed on one
    stream).  A forward recorded with a fresh, graph-owned set has to zero-fill that set inside the graph -- its memory goes back to the
    graph's pool between the backward and the next replay's forward -- which costs ~30 us per replay at 1352 x 1014.  Inside
    `capture_scratch(holder)` the n-th render of the body always takes the holder's n-th set instead: created and grown by the EAGER
    visit (ordinary allocator memory, born zero-filled, left clean by every forward + backward pair), then baked into every graph of
    the family.  The holder must outlive the graphs; a set that would have to grow while capturing (larger slabs than the eager
    visit needed) is refused loudly."""

    def __init__(self):
        self.sets, self._n = {}, {}

    def begin(self):
        self._n = {}

    def take(self, key, capturing):
        k = key[:4]
        i = self._n.get(k, 0)
        self._n[k] = i + 1
        lst = self.sets.setdefault(k, [])
        if i < len(lst):
            return lst[i]
        if capturing:
            return None                    # no eager visit has made this one: a graph-owned set (and its zero-fill) after all
        lst.append(ScratchSet(k[3], True))
        return lst[i]

    def discard(self, item):
        for lst in self.sets.values():
            if item in lst:
                lst.remove(item)


_TLS = threading.local()


@contextlib.contextmanager
def capture_scratch(holder):
    prev = getattr(_TLS, "scope", None)
    _TLS.scope = holder
    holder.begin()
    try:
        yield holder
    finally:
        _TLS.scope = prev


def _set_addresses(ss):
    return tuple(t.data_ptr() if t is not None and t.numel() else 0 for t in (ss.geom, ss.binning, ss.img))


class _Lease:
    """Returns its scratch set to the pool when the owning autograd context is garbage-collected.

    Under stream capture (`torch.cuda.graph`) the pool is NOT used: the captured kernels bake the raw scratch pointers, so the set
    must be memory the GRAPH owns -- it is allocated fresh while capturing (from the graph's private memory pool, which the
    graph keeps reserved for as long as it lives) and never handed to `_POOL`, where an eager call with the same key would
    share it and a pool eviction would let the allocator recycle it under later replays."""
    __slots__ = ("key", "item", "pooled", "scope", "baked")

    def __init__(self, key, capturing):
        self.key = key
        self.scope = getattr(_TLS, "scope", None)
        self.baked = None
        if self.scope is not None:
            item = self.scope.take(key, capturing)
            if item is not None:
                self.pooled, self.item = False, item
                if capturing:
                    self.baked = _set_addresses(item)
                return
        self.pooled = not capturing
        self.item = _POOL.acquire(key) if self.pooled else ScratchSet(key[3], True)

    def check_baked(self):
        """After a forward recorded into a graph with a set of a CaptureScratch holder: the set must not have grown (new memory
        would belong to the graph's pool while the holder hands the set to eager calls and other graphs)."""
        if self.baked is not None and _set_addresses(self.item) != self.baked:
            self.scope.discard(self.item)
            raise RasterizerError("the scratch set of this capture family had to grow while capturing (the eager visit needed less): "
                                  "run the body eagerly once more, then capture again")

    def __del__(self):
        try:
            if self.pooled:
                _POOL.release(self.key, self.item)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, kernel_size, image_height, image_width, sh, degree, campos,
                        prefiltered, require_coord, require_depth, debug, buffers=None, defer=False, scratch=None):
    """`_C.rasterize_gaussians` (RasterizeGaussiansCUDA, DGR/rasterize_points.cu:35-133) with this library's extensions.

    Returns (num_rendered, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geomBuffer, binningBuffer, imgBuffer).
    `buffers` is a RasterBuffers object whose image / radii / scratch tensors are reused across calls; `scratch` a ScratchSet to use
    instead of fresh byte tensors.  `defer=True` does not wait for the instance count: num_rendered is then an upper bound (accepted
    by rasterize_gaussians_backward) and `rasterize_finish()` must be called before the results are trusted.
    On a CAPTURING stream (torch.cuda.graph) the no-wait entry point igs_rast_forward_nowait is used by itself: the launches are
    recorded, num_rendered is the same upper bound, and `capture_status()` reports on a replay after the fact."""
    mode = 2 if (means3D.is_cuda and torch.cuda.is_current_stream_capturing()) else (1 if defer else 0)
    if buffers is not None:
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RasterizerError("means3D must have dimensions (num_points, 3)")
        imgs, radii, ss = buffers.get(means3D.size(0), int(image_height), int(image_width), means3D.device)
        return _C.rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                      projmatrix, tan_fovx, tan_fovy, kernel_size, image_height, image_width, sh, degree, campos,
                                      prefiltered, require_coord, require_depth, debug, scratch=ss, out_images=imgs, out_radii=radii,
                                      mode=mode, scratch_clean=True)
    return _C.rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                  projmatrix, tan_fovx, tan_fovy, kernel_size, image_height, image_width, sh, degree, campos,
                                  prefiltered, require_coord, require_depth, debug, scratch=scratch, mode=mode)


def rasterize_finish():
    """Completes a `rasterize_gaussians(..., defer=True)`: returns the true num_rendered, or None when the optimistic
    instance-list capacity was too small -- everything computed from that frame must then be discarded and the frame redone
    (igs_rast_forward_finish, include/igs_rast.h)."""
    return _C.forward_finish()


def capture_status(any_capture=False):
    """(num_rendered, overflow) of the last forward of this host thread on the current device -- for forwards that ran as part of
    a replayed graph (`torch.cuda.graph`), after the caller has synchronised.  overflow != 0: a tile needed that many instance
    slots and the slabs baked into the capture were smaller; the results of that replay are invalid -- capture again (the slab
    hint has been raised).  `any_capture=True`: whatever the last executed forward posted, for callers who replay several graphs
    in turn (the strict form only accepts the most recently captured one)."""
    n, ov, pf = C.c_int(0), C.c_uint(0), C.c_uint(0)
    fn = _cabi.lib().igs_rast_last_posted_status if any_capture else _cabi.lib().igs_rast_last_status
    _check(fn(C.byref(n), C.byref(ov), C.byref(pf)), "igs_rast_last_status")
    if pf.value:
        raise RasterizerError("Point is filtered although prefiltered is set. This shouldn't happen!")
    return n.value, ov.value


_OUT_NAMES = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")


def rasterize_gaussians_backward(*args, out=None, workspace=None):
    """`_C.rasterize_gaussians_backward` (RasterizeGaussiansBackwardCUDA, DGR/rasterize_points.cu:135-246), same 32 positional
    arguments (background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
    tan_fovx, tan_fovy, kernel_size, dL_dout_color, dL_dout_coord, dL_dout_mcoord, dL_dout_depth, dL_dout_mdepth, dL_dout_alpha,
    dL_dout_normal, normalmap, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, require_coord,
    require_depth, debug).

    Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations).
    Extensions over the reference: any upstream gradient may be None (= zeros: the output was not used by the loss);
    `out=` may name preallocated contiguous destination tensors by those eight names (e.g. spans of a flat gradient
    buffer), `workspace=` a reusable uint8 scratch tensor."""
    kw = {}
    if out:
        kw = {"out_" + k: v for k, v in out.items() if k in _OUT_NAMES}
    return _C.rasterize_gaussians_backward(*args, workspace=workspace, **kw)


last_backward_instance = _cabi.last_backward_instance
mark_visible = _C.mark_visible
integrate_gaussians_to_points = _C.integrate_gaussians_to_points


def debug_dump(P, R, W, H, geomBuffer, binningBuffer, imgBuffer):
    """Per-stage scratch contents as torch tensors (tests / roofline harness)."""
    L = _cabi.lib()
    dev = geomBuffer.device
    T = ((W + 15) // 16) * ((H + 15) // 16)
    ptr = lambda t: t.data_ptr() if t.numel() else None
    with torch.cuda.device(dev):
        rec = torch.zeros((P, 32), dtype=torch.float32, device=dev)
        tiles = torch.zeros((P,), dtype=torch.int32, device=dev)
        pl = torch.zeros((R,), dtype=torch.int32, device=dev)
        ranges = torch.zeros((T, 2), dtype=torch.int32, device=dev)
        nc = torch.zeros((2, H, W), dtype=torch.int32, device=dev)
        _check(L.igs_rast_debug_dump(torch.cuda.current_stream(dev).cuda_stream, P, R, W, H, ptr(geomBuffer), ptr(binningBuffer),
                                     ptr(imgBuffer), rec.data_ptr(), tiles.data_ptr(), ptr(pl), ranges.data_ptr(), nc.data_ptr()),
               "igs_rast_debug_dump")
    return dict(rec=rec, tiles_touched=tiles, point_list=pl, ranges=ranges, n_contrib=nc)


# ---------------------------------------------------------------------------------------------------------------
# autograd binding: DGR/diff_gaussian_rasterization_rade/__init__.py
# ---------------------------------------------------------------------------------------------------------------
def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def _make_function(clamp_grads):
    clamp_value = 15.0 if clamp_grads else 0.0      # DGRC/diff_gaussian_rasterization_rade_clamp/__init__.py:156-162

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
            rs = raster_settings
            args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                    rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.image_height, rs.image_width, sh, rs.sh_degree,
                    rs.campos, rs.prefiltered, rs.require_coord, rs.require_depth, rs.debug)
            # outputs that take no part in the loss arrive in backward as None (not as zero-filled tensors): the C ABI reads
            # NULL as zeros and then runs the cheapest blend-backward instance that covers the gradients actually present
            ctx.set_materialize_grads(False)
            lease, mode = None, 0
            if means3D.is_cuda and means3D.dim() == 2:
                dev = means3D.device
                capturing = torch.cuda.is_current_stream_capturing()
                mode = 2 if capturing else 0
                lease = _Lease((means3D.size(0), int(rs.image_height), int(rs.image_width), dev,
                                torch._C._cuda_getCurrentRawStream(dev.index)), capturing)      # (raw handle: 0.3 us; torch.cuda.current_stream() builds a Stream object, 5 us)
            ss = lease.item if lease else None
            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    out = _C.rasterize_gaussians(*args, scratch=ss, mode=mode, scratch_clean=ss is not None)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                # (pooled / capture-owned sets are born zero-filled and touched by this library only: no per-frame zero-fill launch)
                out = _C.rasterize_gaussians(*args, scratch=ss, mode=mode, scratch_clean=ss is not None)
            if lease is not None and lease.baked is not None:
                lease.check_baked()
            num_rendered, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geomBuffer, binningBuffer, imgBuffer = out
            ctx.raster_settings = rs
            ctx.num_rendered = num_rendered
            ctx.lease = lease               # the scratch set goes back to the pool when this context dies
            ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, normal, radii, sh, geomBuffer,
                                  binningBuffer, imgBuffer, alpha)
            ctx.mark_non_differentiable(radii)
            return color, radii, coord, mcoord, depth, mdepth, alpha, normal

        @staticmethod
        def backward(ctx, grad_color, grad_radii, grad_coord, grad_mcoord, grad_depth, grad_mdepth, grad_alpha, grad_normal):
            rs = ctx.raster_settings
            (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, normal, radii, sh, geomBuffer, binningBuffer, imgBuffer,
             alpha) = ctx.saved_tensors
            args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                    rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, grad_color, grad_coord, grad_mcoord, grad_depth,
                    grad_mdepth, grad_alpha, grad_normal, normal, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered,
                    binningBuffer, imgBuffer, alpha, rs.require_coord, rs.require_depth, rs.debug)
            lease = getattr(ctx, "lease", None)
            ws = lease.item.workspace(means3D.size(0)) if lease is not None else None
            # The reference asserts on seven gradient tensors with seven `.any()` reductions and host syncs (__init__.py:156-162); here
            # the per-Gaussian kernel tests what it writes and posts ONE word to the host (igs_rast_next_backward_options): no extra
            # launch.  The assert is raised from the end of the SAME backward pass (an engine callback: `loss.backward()` still raises
            # before the caller's optimizer.step()), when everything downstream of this node has been enqueued too -- waiting right
            # here would stall the stream behind the host on every iteration.  The clamp package's five torch.clamp calls are applied by
            # that kernel as well.
            nan_report = 0
            if NAN_CHECKS and not (means3D.is_cuda and torch.cuda.is_current_stream_capturing()):      # (a host-side assert cannot be captured)
                nan_report = 2 if NAN_CHECKS_AT_END_OF_PASS else 1
            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    out, has_nan, word, seq = _C.rasterize_gaussians_backward_ex(*args, workspace=ws, nan_report=nan_report, clamp=clamp_value)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                out, has_nan, word, seq = _C.rasterize_gaussians_backward_ex(*args, workspace=ws, nan_report=nan_report, clamp=clamp_value)
            grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales, grad_rotations = out
            assert not has_nan
            if nan_report == 2 and word:
                def _verdict(word=word, seq=seq):
                    assert not _C.nan_report_wait(word, seq)
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(_verdict)
                except Exception:  # noqa: BLE001  (no engine pass to attach to: wait here, as the reference does)
                    _verdict()
            # shapes autograd expects: the gradient of an absent (empty CPU) input is None
            return (grad_means3D, grad_means2D, grad_sh if sh.numel() else None, grad_colors_precomp if colors_precomp.numel() else None,
                    grad_opacities, grad_scales if scales.numel() else None, grad_rotations if rotations.numel() else None,
                    grad_cov3Ds_precomp if cov3Ds_precomp.numel() else None, None)

    return _RasterizeGaussians


NAN_CHECKS = True      # mirrors the reference's NaN asserts on every backward; the refine loop may turn them off
NAN_CHECKS_AT_END_OF_PASS = True      # False: wait for the kernel's verdict inside the Function's backward, where the reference asserts
_RasterizeGaussians = _make_function(False)
_RasterizeGaussiansClamp = _make_function(True)


class GaussianRasterizationSettings(NamedTuple):
    """Field order is API (DGR/diff_gaussian_rasterization_rade/__init__.py:177-192)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    require_depth: bool
    require_coord: bool
    debug: bool


def _make_api(fn_cls):
    def rasterize_gaussians_autograd(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings):
        return fn_cls.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                rs = self.raster_settings
                return mark_visible(positions, rs.viewmatrix, rs.projmatrix)

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            raster_settings = self.raster_settings
            if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
            if shs is None:
                shs = torch.Tensor([])
            if colors_precomp is None:
                colors_precomp = torch.Tensor([])
            if scales is None:
                scales = torch.Tensor([])
            if rotations is None:
                rotations = torch.Tensor([])
            if cov3D_precomp is None:
                cov3D_precomp = torch.Tensor([])
            return rasterize_gaussians_autograd(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                                cov3D_precomp, raster_settings)

        def integrate(self, *args, **kwargs):
            return integrate_gaussians_to_points(*args, **kwargs)

    return rasterize_gaussians_autograd, GaussianRasterizer


rasterize_gaussians_autograd, GaussianRasterizer = _make_api(_RasterizeGaussians)
rasterize_gaussians_autograd_clamp, GaussianRasterizerClamp = _make_api(_RasterizeGaussiansClamp)
