"""The UNCHANGED-caller path: the compiled `_C` module, the autograd binding on top of it, the kernel-side NaN report and clamp, the
one-launch L1 loss, the multi-tensor Adam -- and the caller's loop (tools/dropin_loop.py = infer_batch.py:279-324) against the
library's own fused step."""
import numpy as np
import pytest
import torch

from igs_amd.scenes import cfg1_scene, activate

pytestmark = pytest.mark.gpu
E = torch.Tensor([])


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _settings(mod, cam, bg, debug=False):
    return mod.GaussianRasterizationSettings(image_height=cam.height, image_width=cam.width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                             kernel_size=0.0, bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                             projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center,
                                             prefiltered=False, require_depth=True, require_coord=True, debug=debug)


def test_compiled_module_is_what_the_packages_bind(dev):
    """`diff_gaussian_rasterization_rade._C` carries the four names of DGR/ext.cpp:15-20 and they are the compiled module's functions
    (a built-in, not a Python def); its 12-tuple / 8-tuple orders are rasterize_points.cu:133 / :246."""
    import types
    import diff_gaussian_rasterization_rade as D
    import diff_gaussian_rasterization_rade_clamp as DC
    from igs_amd import _cabi
    m = _cabi.ext()
    assert m.__file__.endswith(".so")
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "integrate_gaussians_to_points"):
        f = getattr(D._C, name)
        assert f is getattr(m, name) and isinstance(f, types.BuiltinFunctionType), name
    assert DC._C is D._C
    raw, cams, bg = cfg1_scene(P=700, size=64)
    cam = cams[0].to(dev)
    a = {k: v.to(dev) for k, v in activate(raw).items()}
    out = D._C.rasterize_gaussians(bg.to(dev), a["means3D"], E, a["opacities"], a["scales"], a["rotations"], 1.0, E, cam.world_view_transform,
                                   cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, a["shs"], 3,
                                   cam.camera_center, False, True, True, False)
    assert len(out) == 12 and isinstance(out[0], int) and out[0] > 0
    nr, color, coord, mcoord, alpha, normal, depth, mdepth, radii, gb, bb, ib = out
    assert tuple(color.shape) == (3, 64, 64) and tuple(alpha.shape) == (1, 64, 64) and radii.dtype == torch.int32 and gb.dtype == torch.uint8
    g = torch.ones_like(color)
    grads = D._C.rasterize_gaussians_backward(bg.to(dev), a["means3D"], radii, E, a["scales"], a["rotations"], 1.0, E, cam.world_view_transform,
                                              cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, g, None, None, None, None, None, None,
                                              normal, a["shs"], 3, cam.camera_center, gb, nr, bb, ib, alpha, True, True, False)
    assert len(grads) == 8 and tuple(grads[5].shape) == (700, 16, 3) and tuple(grads[4].shape) == (700, 6)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        D._C.rasterize_gaussians(bg.to(dev), a["means3D"].reshape(-1), E, a["opacities"], a["scales"], a["rotations"], 1.0, E,
                                 cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width,
                                 a["shs"], 3, cam.camera_center, False, True, True, False)
    with pytest.raises(NotImplementedError):
        D._C.integrate_gaussians_to_points()
    assert D._C.mark_visible(a["means3D"], cam.world_view_transform, cam.full_proj_transform).dtype == torch.bool


def test_nan_report_from_the_kernel_and_kernel_side_clamp(dev):
    """The reference's seven NaN asserts (__init__.py:156-162) as ONE word posted by the per-Gaussian kernel: a NaN in an upstream
    gradient reaches the Gaussians it covers and trips the assert; clean gradients do not; with the checks off nothing is asserted.
    The clamp package's five torch.clamp calls as part of the same kernel: equal to clamping the plain package's gradients."""
    import diff_gaussian_rasterization_rade as D
    import diff_gaussian_rasterization_rade_clamp as DC
    from igs_amd import rasterizer
    raw, cams, bg = cfg1_scene(P=1500, size=96)
    cam = cams[0].to(dev)
    bgd = bg.to(dev)

    def run(mod, upstream_scale=1.0, poison=False, use=("color",)):
        leaf = {k: v.to(dev).clone().requires_grad_(True) for k, v in raw.items()}
        a = activate(leaf)
        ras = mod.GaussianRasterizer(raster_settings=_settings(mod, cam, bgd))
        m2d = torch.zeros_like(a["means3D"], requires_grad=True)
        res = ras(means3D=a["means3D"], means2D=m2d, opacities=a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
        g = torch.full_like(res[0], upstream_scale)
        if poison:
            g[0, 44:52, 44:52] = float("nan")
        outs, gs = [res[0]], [g]
        if "depth" in use:
            outs.append(res[4]); gs.append(torch.full_like(res[4], 0.3 * upstream_scale))
        raster_in = (a["means3D"], m2d, a["shs"], a["opacities"], a["scales"], a["rotations"])
        return torch.autograd.grad(outs, raster_in, gs)

    prev = rasterizer.NAN_CHECKS
    try:
        rasterizer.NAN_CHECKS = True
        for at_end in (True, False):                       # verdict collected by an engine callback at the end of the pass / inside the node
            rasterizer.NAN_CHECKS_AT_END_OF_PASS = at_end
            run(D)                                         # clean: no assert
            with pytest.raises(AssertionError):
                run(D, poison=True)
            with pytest.raises(AssertionError):
                run(DC, poison=True)                       # a NaN survives the clamp, as through torch.clamp
            run(D)                                         # ... and the next clean pass is clean again
        rasterizer.NAN_CHECKS_AT_END_OF_PASS = True
        # `loss.backward()` raises too (the caller's optimizer.step() is never reached), and two renders in ONE pass are both looked at
        leaf = {k: v.to(dev).clone().requires_grad_(True) for k, v in raw.items()}
        a = activate(leaf)
        imgs = []
        for _ in range(2):
            ras = D.GaussianRasterizer(raster_settings=_settings(D, cam, bgd))
            imgs.append(ras(means3D=a["means3D"], means2D=torch.zeros_like(a["means3D"], requires_grad=True), opacities=a["opacities"],
                            shs=a["shs"], scales=a["scales"], rotations=a["rotations"])[0])
        w = torch.ones_like(imgs[0]); w[1, 40:56, 40:56] = float("nan")
        with pytest.raises(AssertionError):
            (imgs[0].sum() + (imgs[1] * w).sum()).backward()
        rasterizer.NAN_CHECKS = False
        gp = run(D, poison=True)                           # checks off: the NaN simply arrives
        assert any(bool(torch.isnan(t).any()) for t in gp)
        # clamp: the kernel-side clamp == torch.clamp of the plain gradients, for the five tensors the clamp package touches
        for use in (("color",), ("color", "depth")):
            plain = run(D, upstream_scale=400.0, use=use)
            clamped = run(DC, upstream_scale=400.0, use=use)
            assert float(plain[0].abs().max()) > 15.0
            for i, name in ((0, "means3D"), (2, "sh"), (3, "opacities"), (4, "scales"), (5, "rotations")):
                want = torch.clamp(plain[i], -15, 15)
                # two separate backward runs: float atomics in the blend backward make them differ by rounding RELATIVE TO THE UNCLAMPED
                # magnitudes (hundreds here), so an element near +-15 may sit on the other side of the bound; the clamp itself is exact
                d = (clamped[i] - want).abs()
                scale = float(plain[i].abs().max())
                assert float(d.max()) <= 2e-4 * scale, (name, use, float(d.max()), scale)
                assert float((d > 1e-5 * scale).float().mean()) < 5e-3, (name, use)
                assert float(clamped[i].abs().max()) <= 15.0
            torch.testing.assert_close(clamped[1], plain[1], rtol=2e-3, atol=1e-3 * float(plain[1].abs().max()))      # means2D is NOT clamped
    finally:
        rasterizer.NAN_CHECKS = prev


@pytest.mark.parametrize("n", [5, 4096, 3 * 211 * 157])
def test_one_launch_l1_loss_matches_torch(dev, n):
    """igs_amd.losses.l1_loss (loss_utils.py:17-18) as one launch forward + one scale backward: value and gradient against PyTorch's
    sub / abs / mean; deterministic (partials are added in index order)."""
    from igs_amd.losses import l1_loss
    g = torch.Generator().manual_seed(n)
    a = torch.rand(n, generator=g).to(dev)
    b = torch.rand(n, generator=g).to(dev)
    b[:: 7] = a[:: 7]                                       # exact zeros: sign(0) = 0
    x = a.clone().requires_grad_(True)
    y = a.clone().requires_grad_(True)
    l1 = l1_loss(x, b)
    l2 = torch.abs(y - b).mean()
    assert l1.shape == l2.shape == ()
    torch.testing.assert_close(l1, l2, rtol=2e-6, atol=1e-8)
    (3.0 * l1).backward(); (3.0 * l2).backward()
    torch.testing.assert_close(x.grad, y.grad, rtol=1e-6, atol=0)
    assert float(l1_loss(a, b)) == float(l1_loss(a, b))
    # shapes PyTorch would broadcast, or CPU tensors: PyTorch's own ops on the caller's tensors
    assert abs(float(l1_loss(a.cpu(), b.cpu())) - float(l2)) < 1e-6


def test_multi_tensor_adam_matches_torch_adam(dev):
    """igs_amd.optim.Adam (ONE launch for the five groups of gaussian_model.py:303-348) against torch.optim.Adam(lr=0, eps=1e-15) with
    per-group learning rates: identical parameters and state over several steps, including a parameter that joins late (its own step
    count) and one without a gradient."""
    from igs_amd.optim import Adam
    gen = torch.Generator().manual_seed(3)
    shapes = [(1000, 3), (1000, 4), (1000, 16, 3), (1000, 1), (1000, 3), (7,)]
    lrs = [1.6e-3, 1e-2, 2.5e-3, 5e-2, 5e-3, 1e-3]
    init = [torch.randn(s, generator=gen) for s in shapes]

    def make(cls):
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        return ps, cls([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))], lr=0.0, eps=1e-15)
    pa, oa = make(Adam)
    pb, ob = make(torch.optim.Adam)
    for step in range(6):
        for i, (x, y) in enumerate(zip(pa, pb)):
            if i == 5 and step < 2:
                continue                                    # joins at step 2: its bias corrections run two steps behind
            if i == 3 and step == 4:
                x.grad = None; y.grad = None                # no gradient this step: untouched, step count not advanced
                continue
            g = torch.randn(x.shape, generator=gen).to(dev) * (10.0 ** (step - 3))
            x.grad = g.clone(); y.grad = g.clone()
        oa.step(); ob.step()
        oa.zero_grad(set_to_none=True); ob.zero_grad(set_to_none=True)
    for i, (x, y) in enumerate(zip(pa, pb)):
        torch.testing.assert_close(x.detach(), y.detach(), rtol=2e-6, atol=2e-4 * lrs[i])        # (a step moves a parameter by ~lr)
        sa, sb = oa.state[x], ob.state[y]
        assert int(sa["step"]) == int(sb["step"])
        # (torch forms exp_avg with lerp_, this kernel as b1 m + (1 - b1) g: last-bit differences where the two terms cancel)
        torch.testing.assert_close(sa["exp_avg"], sb["exp_avg"], rtol=2e-5, atol=2e-6 * float(sb["exp_avg"].abs().max()))
        torch.testing.assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-5, atol=1e-6 * float(sb["exp_avg_sq"].abs().max()))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p = torch.nn.Parameter(torch.zeros(3)); p.grad = torch.ones(3)
        Adam([p], lr=1e-3).step()


@pytest.mark.parametrize("loss", ["l1", "l1_ssim"])
def test_unchanged_caller_loop_equals_the_fused_step(dev, loss):
    """tools/dropin_loop.py drives the packages exactly as infer_batch.py:279-324 does (nn.Parameters, PyTorch activations,
    GaussianRasterizer, l1_loss / ssim, loss.backward(), optimizer.step(), zero_grad(set_to_none=True)) -- with torch.optim.Adam and
    with igs_amd.optim.Adam -- and must land where the library's own single-call step (igs_refine_step) lands from the same start."""
    from igs_amd.refine import GaussianParams, Refiner, render, DEFAULT_LRS
    from igs_amd.scenes import perturbed_copy
    from tools.dropin_loop import CallerModel, refine_iteration, make_losses
    raw, cams, bg = cfg1_scene(P=3000, size=128)
    cams = [cams[0].to(dev)]
    bgd = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw, sigma=0.03).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), cams[0], bgd)["images_pred"].clone()]
    steps = 4
    pf = GaussianParams(raw, dev)
    rf = Refiner(pf, cams, gts, bgd, loss=loss)
    for _ in range(steps):
        rf.step(view=0)
    want = {k: v.detach().clone() for k, v in pf.leaves.items()}
    lf = make_losses("igs")
    for optimizer in ("torch", "fused"):
        gs = CallerModel(raw, dev, DEFAULT_LRS, optimizer=optimizer)
        for _ in range(steps):
            pkg, total = refine_iteration(gs, cams[0], gts[0], bgd, loss=loss, losses=lf)
        assert torch.isfinite(total) and pkg["viewspace_points"].grad is not None and tuple(pkg["viewspace_points"].grad.shape) == (3000, 3)
        got = gs.raw()
        for k in want:
            # |dp| per step <= lr: compare against the distance travelled (a sign flip of a ~zero gradient moves a parameter by 2 lr)
            d = (got[k] - want[k]).abs()
            lr = DEFAULT_LRS[k]
            assert float(torch.quantile(d.flatten()[:100000], 0.98)) < 0.02 * lr * steps, (optimizer, k, float(torch.quantile(d.flatten()[:100000], 0.98)))
            assert float(d.max()) <= 2.05 * lr * steps, (optimizer, k, float(d.max()))


def test_capturable_adam_counts_its_steps_on_the_device(dev, monkeypatch):
    """igs_amd.optim.Adam(capturable=True): `state[p]["step"]` is a GPU scalar advanced by the launch (igs_adam_step_multi_dev), the bias
    corrections come from it -- eagerly and replayed from a hipGraph it must walk with torch.optim.Adam."""
    from igs_amd.optim import Adam
    gen = torch.Generator().manual_seed(5)
    shapes = [(500, 3), (500, 16, 3), (9,)]
    lrs = [1.6e-3, 2.5e-3, 5e-2]
    init = [torch.randn(s, generator=gen) for s in shapes]

    def make(cls, **kw):
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        return ps, cls([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)], lr=0.0, eps=1e-15, **kw)
    pa, oa = make(Adam, capturable=True)
    pb, ob = make(torch.optim.Adam)
    grads = [[torch.randn(sh, generator=gen).to(dev) * (10.0 ** (k - 2)) for sh in shapes] for k in range(7)]
    static = [torch.zeros_like(p) for p in pa]
    for x, st in zip(pa, static):
        x.grad = st                                          # the graph reads the gradients from fixed addresses
    oa.init_state()
    assert all(torch.is_tensor(oa.state[x]["step"]) and oa.state[x]["step"].is_cuda for x in pa)
    for k in range(3):                                       # three eager steps
        for st, y, g in zip(static, pb, grads[k]):
            st.copy_(g); y.grad = g.clone()
        oa.step(); ob.step()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        oa.step()
    for k in range(3, 7):                                    # four replayed steps
        for st, y, g in zip(static, pb, grads[k]):
            st.copy_(g); y.grad = g.clone()
        graph.replay(); ob.step()
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(pa, pb)):
        assert float(oa.state[x]["step"]) == 7.0 == float(ob.state[y]["step"])
        torch.testing.assert_close(x.detach(), y.detach(), rtol=2e-6, atol=2e-4 * lrs[i])
        torch.testing.assert_close(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"], rtol=2e-5, atol=1e-6 * float(ob.state[y]["exp_avg_sq"].abs().max()))
    # a step whose state does not exist yet cannot be captured: loud, not a graph that allocates (the capture state is faked: a capture
    # that ends in an exception leaves this ROCm's later captures of the process unusable)
    pc, oc = make(Adam, capturable=True)
    for x in pc:
        x.grad = torch.ones_like(x)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    with pytest.raises(RuntimeError, match="before a step is captured"):
        oc.step()


@pytest.mark.parametrize("loss", ["l1", "l1_ssim"])
def test_caller_loop_replayed_from_graphs_equals_the_eager_loop(dev, loss):
    """igs_amd.graphs.GraphedLoop around the unchanged loop body (tools/dropin_loop.py; optimizer = igs_amd.optim.Adam(capturable=True)):
    first visit of a view eager, second captured, then replays -- must land where the eager loop lands after the same schedule, and
    the values it returns must be those of the iteration just replayed."""
    from igs_amd.graphs import GraphedLoop
    from igs_amd.refine import render, DEFAULT_LRS
    from igs_amd.scenes import perturbed_copy
    from tools.dropin_loop import CallerModel, refine_iteration, make_losses
    from igs_amd.scenes import sear_steak_like_scene
    raw, cams, bg = sear_steak_like_scene(P=3000, n_cams=2, width=160, height=120, focal=90.0, scale_mean=-2.0)
    cams = [c.to(dev) for c in cams]
    bgd = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw, sigma=0.03).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bgd)["images_pred"].clone() for c in cams]
    lf = make_losses("igs")
    schedule = [i % len(cams) for i in range(9)]
    ge = CallerModel(raw, dev, DEFAULT_LRS, optimizer="fused")
    eager_losses = []
    for v in schedule:
        _, total = refine_iteration(ge, cams[v], gts[v], bgd, loss=loss, losses=lf)
        eager_losses.append(float(total))
    gg = CallerModel(raw, dev, DEFAULT_LRS, optimizer="fused_capturable")
    loop = GraphedLoop(lambda v: refine_iteration(gg, cams[v], gts[v], bgd, loss=loss, losses=lf))
    graph_losses = []
    for v in schedule:
        pkg, total = loop(v)
        graph_losses.append(float(total))
    assert len(loop._graphs) == len(cams)
    loop.check()                                             # (no replayed forward overflowed its tile slabs)
    assert sum(len(v) for v in loop._scratch.sets.values()) == 1          # one scratch set serves the eager visits and both graphs
    assert tuple(pkg["viewspace_points"].grad.shape) == (3000, 3) and float(pkg["viewspace_points"].grad.abs().max()) > 0
    for a, b in zip(eager_losses, graph_losses):
        assert abs(a - b) <= 2e-3 * abs(a), (eager_losses, graph_losses)
    assert graph_losses[-1] < graph_losses[0]
    want, got = ge.raw(), gg.raw()
    steps = len(schedule)
    for k in want:
        d = (got[k] - want[k]).abs()
        lr = DEFAULT_LRS[k]
        assert float(torch.quantile(d.flatten()[:100000], 0.98)) < 0.02 * lr * steps, (k, float(torch.quantile(d.flatten()[:100000], 0.98)))
        assert float(d.max()) <= 2.05 * lr * steps, (k, float(d.max()))


def test_graphed_loop_reports_a_replay_that_overflowed_its_slabs(dev):
    """A replayed forward cannot tell the host that a tile outgrew the instance slab baked into its graph; GraphedLoop.check() reads what
    the last replay posted (igs_rast_last_posted_status: no sequence check, several graphs take turns) and raises.  Provoked by
    capturing with small slabs on small splats, then inflating every splat before the next replay."""
    from igs_amd import _cabi
    from igs_amd.graphs import GraphedLoop
    from igs_amd.rasterizer import RasterizerError
    from igs_amd.refine import render, DEFAULT_LRS
    from igs_amd.scenes import sear_steak_like_scene
    from tools.dropin_loop import CallerModel, refine_iteration, make_losses
    L = _cabi.lib()
    raw, cams, bg = sear_steak_like_scene(P=4000, n_cams=1, width=160, height=120, focal=90.0, scale_mean=-4.5)
    cams = [c.to(dev) for c in cams]
    bgd = bg.to(dev)
    with torch.no_grad():
        gts = [render(activate({k: v.to(dev) for k, v in raw.items()}), cams[0], bgd)["images_pred"].clone()]
    old_hint = L.igs_rast_get_slab_hint()
    try:
        L.igs_rast_set_slab_hint(256)
        gg = CallerModel(raw, dev, {k: 0.0 for k in DEFAULT_LRS}, optimizer="fused_capturable")       # (learning rates 0: the test moves the splats itself)
        lf = make_losses("igs")
        loop = GraphedLoop(lambda v: refine_iteration(gg, cams[v], gts[v], bgd, loss="l1", losses=lf))
        for _ in range(3):
            loop(0)                                          # eager, capture + replay, replay
        loop.check()
        assert L.igs_rast_get_slab_hint() == 256             # nothing overflowed so far: the graph has 256-slot slabs baked in
        with torch.no_grad():
            gg._scaling.add_(3.0)                            # every splat 20 x larger: hundreds of instances per tile
        loop(0)
        with pytest.raises(RasterizerError, match="overflowed"):
            loop.check()
        assert L.igs_rast_get_slab_hint() > 256              # the hint has been raised: a new capture will fit
        loop.reset()
        for _ in range(3):
            pkg, total = loop(0)
        loop.check()
        assert torch.isfinite(total)
    finally:
        L.igs_rast_set_slab_hint(old_hint)


def test_inputs_are_converted_and_errors_are_loud(dev):
    """The compiled glue accepts what the reference's does -- non-contiguous and float64 inputs are made contiguous float32
    (rasterize_points.cu:98-130 calls .contiguous().data<float>()) -- and refuses what it cannot serve with a RasterizerError."""
    import diff_gaussian_rasterization_rade as D
    from igs_amd.rasterizer import RasterizerError
    raw, cams, bg = cfg1_scene(P=900, size=80)
    cam = cams[0].to(dev)
    a = {k: v.to(dev) for k, v in activate(raw).items()}
    st = _settings(D, cam, bg.to(dev))
    ras = D.GaussianRasterizer(raster_settings=st)
    m2d = torch.zeros_like(a["means3D"])
    ref = ras(means3D=a["means3D"], means2D=m2d, opacities=a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    # float64 positions, a transposed (non-contiguous) SH tensor, a strided opacity column
    shs_nc = a["shs"].transpose(1, 2).contiguous().transpose(1, 2)
    op_nc = torch.stack([a["opacities"][:, 0], a["opacities"][:, 0]], dim=1)[:, :1]
    assert not shs_nc.is_contiguous() and not op_nc.is_contiguous()
    got = ras(means3D=a["means3D"].double(), means2D=m2d, opacities=op_nc, shs=shs_nc, scales=a["scales"], rotations=a["rotations"])
    for x, y in zip(ref, got):
        assert torch.equal(x, y)
    # tensors on the wrong device / no GPU tensor at all
    with pytest.raises(RasterizerError, match="must live on"):
        ras(means3D=a["means3D"], means2D=m2d, opacities=a["opacities"].cpu(), shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    with pytest.raises(RasterizerError, match="no CPU fallback"):
        ras(means3D=a["means3D"].cpu(), means2D=m2d.cpu(), opacities=a["opacities"].cpu(), shs=a["shs"].cpu(), scales=a["scales"].cpu(),
            rotations=a["rotations"].cpu())
    # a destination tensor of the wrong size is refused before anything is launched
    from igs_amd import _cabi
    m = _cabi.ext()
    with pytest.raises(RasterizerError, match="out_images"):
        m.rasterize_gaussians(bg.to(dev), a["means3D"], E, a["opacities"], a["scales"], a["rotations"], 1.0, E, cam.world_view_transform,
                              cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height, cam.width, a["shs"], 3, cam.camera_center,
                              False, True, True, False, out_images=torch.zeros(15, 8, 8, device=dev))


def test_two_host_threads_render_and_differentiate_concurrently(dev):
    """include/igs_rast.h: "the library keeps a little state PER HOST THREAD ... calls from different threads do not interfere".  Two Python
    threads, each on its own stream, run render -> loss -> backward (NaN report on) for different cameras at the same time -- the compiled
    module releases the GIL, the scratch pool and the NaN tickets are shared process state -- and must reproduce what one thread computes."""
    import threading
    import diff_gaussian_rasterization_rade as D
    from igs_amd import rasterizer
    from igs_amd.scenes import sear_steak_like_scene
    raw, cams, bg = sear_steak_like_scene(P=20000, n_cams=4, width=400, height=300, focal=220.0)
    cams = [c.to(dev) for c in cams]
    bgd = bg.to(dev)
    leaves0 = {k: v.to(dev) for k, v in raw.items()}

    def one(cam, scale):
        leaf = {k: v.clone().requires_grad_(True) for k, v in leaves0.items()}
        a = activate(leaf)
        ras = D.GaussianRasterizer(raster_settings=_settings(D, cam, bgd))
        out = ras(means3D=a["means3D"], means2D=torch.zeros_like(a["means3D"], requires_grad=True), opacities=a["opacities"], shs=a["shs"],
                  scales=a["scales"], rotations=a["rotations"])
        (scale * out[0].sum() + out[4].sum()).backward()
        return out[0].detach().clone(), {k: v.grad.clone() for k, v in leaf.items()}

    prev = rasterizer.NAN_CHECKS
    rasterizer.NAN_CHECKS = True
    try:
        want = [one(cams[i], 1.0 + i) for i in range(4)]
        torch.cuda.synchronize()
        results, errors = {}, []

        def worker(tid):
            try:
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.default_stream(dev))
                with torch.cuda.stream(s):
                    for rep in range(6):
                        for i in (tid, tid + 2):
                            results[(tid, rep, i)] = one(cams[i], 1.0 + i)
                s.synchronize()
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors
        assert len(results) == 24
        for (tid, rep, i), (img, grads) in results.items():
            assert torch.allclose(img, want[i][0], rtol=0, atol=1e-6), (tid, rep, i)
            for k in grads:
                # two runs of the same backward differ in the last bits (float atomics in another order), and the reference's coef
                # backward turns those bits into O(1) changes for a few very large splats: compare the bulk
                B = want[i][1][k]
                close = ((grads[k] - B).abs() <= 1e-4 * B.abs() + 1e-5 * float(B.abs().max())).float().mean().item()
                assert close > 0.999, (tid, rep, i, k, close)
    finally:
        rasterizer.NAN_CHECKS = prev


@pytest.mark.parametrize("with_filter", [True, False])
def test_ply_table_to_parameter_store_on_the_gpu(dev, tmp_path, with_filter):
    """SURVEY 8f rank 3 on the device: igs_ply_to_params (column gathering, channel-major SH -> [P, K, 3], `filter_3D` folded into scale and
    opacity: gs.py:400-462, 480-490) against the host reader of igs_amd/io.py on the same file -- written by the host writer with the
    reference's property order plus, as real assets have them, columns in another order; and igs_params_to_ply against the host writer."""
    import numpy as np
    from igs_amd import io
    raw, _, _ = cfg1_scene(P=5000, size=64)
    gen = torch.Generator().manual_seed(4)
    filt = (0.002 + 0.05 * torch.rand(5000, 1, generator=gen)) if with_filter else None
    path = str(tmp_path / "start.ply")
    io.write_gaussian_ply(path, raw, filter_3D=filt)
    want = io.load_start_gaussians(path)
    got = io.load_start_gaussians_gpu(path, dev)
    for k in want:
        assert got[k].shape == want[k].shape and got[k].device.type == "cuda", k
        torch.testing.assert_close(got[k].cpu(), want[k], rtol=2e-6, atol=2e-6), k
    if with_filter:       # the fold really happened: scales grew, opacities shrank
        assert float((got["scaling"].cpu() - raw["scaling"]).min()) > 0 and float((got["opacity"].cpu() - raw["opacity"]).max()) < 0
    else:
        for k in raw:
            assert torch.equal(got[k].cpu(), raw[k].float()), k
    # a file whose columns come in another order (rot before scale, filter first): the header decides, not the position
    v = io.read_ply_vertices(path)
    names = list(v.dtype.names)
    perm = [n for n in names if n.startswith("rot")] + [n for n in names if n == "filter_3D"] + [n for n in names if not n.startswith("rot") and n != "filter_3D"]
    p2 = str(tmp_path / "shuffled.ply")
    with open(p2, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n" + ("element vertex %d\n" % len(v)).encode())
        for n in perm:
            f.write(("property float %s\n" % n).encode())
        f.write(b"end_header\n")
        f.write(np.stack([v[n] for n in perm], axis=1).astype("<f4").tobytes())
    got2 = io.load_start_gaussians_gpu(p2, dev)
    for k in got:
        assert torch.equal(got2[k], got[k]), k
    # the writer: same bytes as the host writer for leaves on the GPU
    p3, p4 = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    io.write_gaussian_ply(p3, raw)
    io.write_gaussian_ply_gpu(p4, {k: t.to(dev) for k, t in raw.items()})
    assert open(p3, "rb").read() == open(p4, "rb").read()
