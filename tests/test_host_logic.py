"""CPU-side tests: the C-ABI library builds/loads and exports what include/igs_rast.h declares, the product path fails
loudly without a GPU, host logic of the refine loop (flat parameter store, view sharding, gradient all-reduce over gloo)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    from igs_amd import _cabi
    L = _cabi.lib()
    hdr = open(os.path.join(ROOT, "include", "igs_rast.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(igs_[a-z0-9_]+)\s*\(", hdr)) - {"igs_rast_alloc_fn"}
    assert {"igs_rast_forward", "igs_rast_backward", "igs_rast_mark_visible", "igs_rast_backward_workspace_bytes"} <= names
    for n in sorted(names):
        assert hasattr(L, n), "libigs_rast.so does not export %s" % n
    assert L.igs_rast_version() == _cabi.VERSION == 4
    assert L.igs_refine_step_args_size() > 0
    assert L.igs_rast_backward_workspace_bytes(1000) >= 1000 * 25 * 4
    assert set(_cabi.EXPORTS) <= names | {"igs_rast_last_error", "igs_rast_version"}


def test_product_path_fails_loudly_without_gpu_and_validates_arguments():
    import diff_gaussian_rasterization_rade as D
    from igs_amd.rasterizer import RasterizerError
    st = D.GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, kernel_size=0.0,
                                         bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                                         sh_degree=0, campos=torch.zeros(3), prefiltered=False, require_depth=True,
                                         require_coord=True, debug=False)
    ras = D.GaussianRasterizer(raster_settings=st)
    assert ras.raster_settings is st
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        ras(means3D=m, means2D=m, opacities=torch.ones(4, 1))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        ras(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 1, 3), scales=torch.ones(4, 3))
    if not torch.cuda.is_available():
        # CPU tensors: no silent fallback, a loud error
        with pytest.raises(RasterizerError, match="no CPU fallback"):
            ras(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 1, 3), scales=torch.ones(4, 3),
                rotations=torch.ones(4, 4))
        with pytest.raises(RasterizerError):
            ras.markVisible(m)
    with pytest.raises(NotImplementedError):
        ras.integrate()


def test_round4_entry_points_validate_before_they_launch():
    """Argument checks of the entry points added in round 4 that can be exercised without a GPU: every one of them returns
    IGS_RAST_E_INVALID (or 0 for "nothing to do") before any HIP call, and the Python optimiser / graph helper refuse CPU work loudly."""
    import ctypes as C
    from igs_amd import _cabi
    L = _cabi.lib()
    INVALID = -1
    P8 = C.c_void_p * 8
    F8 = C.c_float * 8
    S8 = C.c_size_t * 8
    fake = P8(*[0x1000 * (k + 1) for k in range(8)])             # never dereferenced: the checks below all fail first
    same = P8(*[0x1000] * 8)
    cnt, lr = S8(*[4] * 8), F8(*[1e-3] * 8)
    scr = C.c_void_p(0x9000)
    assert L.igs_adam_step_multi_dev_scratch_words() >= 33
    assert L.igs_adam_step_multi_dev(None, 0, fake, fake, fake, fake, cnt, lr, fake, scr, 0.9, 0.999, 1e-15) == 0          # nothing to do
    assert L.igs_adam_step_multi_dev(None, 9, fake, fake, fake, fake, cnt, lr, fake, scr, 0.9, 0.999, 1e-15) == INVALID    # at most 8 tensors
    assert L.igs_adam_step_multi_dev(None, 2, fake, fake, fake, fake, cnt, lr, None, scr, 0.9, 0.999, 1e-15) == INVALID    # no step counts
    assert L.igs_adam_step_multi_dev(None, 2, fake, fake, fake, fake, cnt, lr, fake, None, 0.9, 0.999, 1e-15) == INVALID   # no done-counters
    assert L.igs_adam_step_multi_dev(None, 2, fake, fake, fake, fake, cnt, lr, same, scr, 0.9, 0.999, 1e-15) == INVALID    # one counter for two tensors
    assert L.igs_adam_step_multi(None, 2, fake, fake, fake, fake, cnt, lr, None, None, 0.9, 0.999, 1e-15) == INVALID  # no bias corrections
    assert L.igs_l1_mean_fwd_bwd(None, 0, fake, fake, fake, fake, fake, fake) == INVALID
    assert L.igs_l1_mean_fwd_bwd(None, 16, None, fake, fake, fake, fake, fake) == INVALID
    from igs_amd.optim import Adam
    p = torch.nn.Parameter(torch.zeros(3)); p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Adam([p], lr=1e-3, capturable=True).step()
    if not torch.cuda.is_available():
        from igs_amd.graphs import GraphedLoop
        with pytest.raises(RuntimeError, match="needs a GPU"):
            GraphedLoop(lambda k: None)(0)


def test_no_product_module_imports_the_oracle():
    for base in ("igs_amd", "diff_gaussian_rasterization_rade", "diff_gaussian_rasterization_rade_clamp"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    txt = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)
                    assert "rast_oracle" not in txt, os.path.join(dp, f)


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    ab = bench.algorithmic_bytes(2_000_000, 1352, 1014)
    assert ab["blend_fwd"] == 2_000_000 * 100 + 1352 * 1014 * 88 + 8 * 5440       # SURVEY.md 8(d) example: ~0.32 GB
    assert ab["blend_bwd"] == 2_000_000 * 100 + 1352 * 1014 * 104 + 2_000_000 * 100
    abc = bench.algorithmic_bytes(1000, 256, 256, True, True, False, False, False)
    assert abc["blend_bwd"] == 1000 * 40 + 256 * 256 * 28 + 1000 * 100 and abc["blend_fwd"] == 1000 * 100 + 256 * 256 * 88 + 8 * 256
    ab0 = bench.algorithmic_bytes(1000, 256, 256, coord=False, depth=False)
    assert ab0["blend_fwd"] == 1000 * 40 + 256 * 256 * 24 + 8 * 256 and ab0["blend_bwd"] == 1000 * 40 + 256 * 256 * 28 + 1000 * 100


def test_bench_refuses_to_mislabel_a_smaller_job():
    """`python bench.py --gpus N` without a launcher self-launches N ranks -- or refuses (rc != 0) when fewer GPUs are visible; it never
    runs one GPU under the label of N (round-1 finding).  Here no GPU is visible at all."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("needs a box without N GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""


def _fake_render(act, cam, bg):
    """Differentiable stand-in for the rasterizer (CPU): an 'image' that depends on every parameter group and the view."""
    w = cam["w"]
    img = (act["means3D"].sum(1) * w[0] + act["opacities"][:, 0] * w[1] + act["scales"].prod(1) * w[2]
           + act["rotations"][:, 0] * w[3] + act["shs"].sum((1, 2)) * w[4])
    return dict(images_pred=img.reshape(1, -1, 1))


def _torch_adam_on_flat(params):
    import math

    def step():
        params.step_count += 1
        b1, b2 = params.betas
        bc1 = 1 - b1 ** params.step_count
        bc2s = math.sqrt(1 - b2 ** params.step_count)
        with torch.no_grad():
            for name, (o, n) in params.spans.items():
                g = params.grad[o:o + n]
                m = params.exp_avg[o:o + n].mul_(b1).add_(g, alpha=1 - b1)
                v = params.exp_avg_sq[o:o + n].mul_(b2).addcmul_(g, g, value=1 - b2)
                params.flat[o:o + n].addcdiv_(m, v.sqrt() / bc2s + params.eps, value=-params.lrs[name] / bc1)
    return step


def _make_scene(P=64, ncams=6):
    from igs_amd.scenes import cfg1_scene
    raw, _, _ = cfg1_scene(P=P, size=16)
    gen = torch.Generator().manual_seed(7)
    cams = [dict(w=torch.randn(5, generator=gen)) for _ in range(ncams)]
    gts = [torch.randn(1, P, 1, generator=gen) for _ in range(ncams)]
    return raw, cams, gts


def test_flat_parameter_store_aliases_gradients():
    from igs_amd.refine import GaussianParams, GROUPS
    raw, cams, gts = _make_scene()
    p = GaussianParams(raw, torch.device("cpu"))
    assert p.flat.numel() == 59 * 64 and sum(k for _, k in GROUPS) == 59
    act = p.activated()
    (_fake_render(act, cams[0], None)["images_pred"] ** 2).sum().backward()
    for name, (o, n) in p.spans.items():
        g = p.grad[o:o + n]
        assert g.abs().sum() > 0, name
        assert p.leaves[name].grad.data_ptr() == g.data_ptr()        # autograd accumulated in place into the flat buffer
        assert p.leaves[name].data_ptr() == p.flat[o:o + n].data_ptr()
    p.zero_grad()
    assert float(p.grad.abs().sum()) == 0.0 and float(p.leaves["xyz"].grad.abs().sum()) == 0.0


def _rank_main(rank, world, port, q, ncams=6, steps=4):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from igs_amd.refine import GaussianParams, Refiner
    raw, cams, gts = _make_scene(ncams=ncams)
    p = GaussianParams(raw, torch.device("cpu"))
    r = Refiner(p, cams, gts, None, loss="l1_ssim_off", world_size=world, rank=rank, seed=3,
                render_fn=_fake_render, adam_fn=_torch_adam_on_flat(p))
    r.loss = "plain"
    views = []
    # 'plain' loss: use an L1 written with torch so that the CPU path needs no HIP kernel
    import types

    def step(self, view=None):
        pp = self.params
        if view is None:
            view = self._next_view()
        pp.zero_grad()
        img = self.render_fn(pp.activated(), self.cams[view], None)["images_pred"]
        (torch.abs(img - self.gt[view]).mean() / self.world_size).backward()
        if self.world_size > 1:
            dist.all_reduce(pp.grad, op=dist.ReduceOp.SUM)
        self.adam_fn()
        return view
    r.step = types.MethodType(step, r)
    for _ in range(steps):
        views.append(r.step())
    q.put((rank, views, p.flat.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharding_and_gradient_allreduce_gloo_world2():
    """N = 2 over gloo: ranks draw DIFFERENT views of the same shared permutation, all-reduce the flat gradient, and end with
    bit-identical replicas that match a single process applying the averaged gradient of the same two views per step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    (_, v0, f0), (_, v1, f1) = res
    assert all(a != b for a, b in zip(v0, v1))                 # different views in every step
    np.testing.assert_array_equal(f0, f1)                      # replicas stay identical without a broadcast
    # single-process reference: same two views per step, averaged gradient
    from igs_amd.refine import GaussianParams
    raw, cams, gts = _make_scene()
    p = GaussianParams(raw, torch.device("cpu"))
    adam = _torch_adam_on_flat(p)
    for a, b in zip(v0, v1):
        p.zero_grad()
        for v in (a, b):
            img = _fake_render(p.activated(), cams[v], None)["images_pred"]
            (torch.abs(img - gts[v]).mean() / 2).backward()
        adam()
    np.testing.assert_allclose(p.flat.numpy(), f0, rtol=1e-5, atol=1e-7)
    # without-replacement sampling: the first 3 steps (6 draws) cover all 6 views once
    assert sorted(v0[:3] + v1[:3]) == list(range(6))


def test_view_sharding_world8_with_10_views_refills_mid_step_gloo():
    """N = 8 ranks over gloo with TEN views (BASELINE configs[3]'s shape: the permutation of the views runs out in the middle of every
    other step): no step holds a view twice, every pass over the views uses each view once, the replicas stay bit-identical, and
    they equal ONE process that applies the averaged gradient of the same eight views per step."""
    import torch.multiprocessing as mp
    world, ncams, steps = 8, 10, 5                           # 40 draws = 4 passes over the 10 views
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, ncams, steps)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    per_step = [[res[r][1][s] for r in range(world)] for s in range(steps)]
    for s_, vs in enumerate(per_step):
        assert len(set(vs)) == world, (s_, vs)               # eight DIFFERENT views in every step, also across a refill
    flat_draws = [v for vs in per_step for v in vs]
    for k in range(0, len(flat_draws), ncams):
        assert sorted(flat_draws[k:k + ncams]) == list(range(ncams)), flat_draws[k:k + ncams]      # sampling without replacement per pass
    for r in range(1, world):
        np.testing.assert_array_equal(res[0][2], res[r][2])   # replicas stay identical without a broadcast
    # single-process reference: the same eight views per step, averaged gradient
    from igs_amd.refine import GaussianParams
    raw, cams, gts = _make_scene(ncams=ncams)
    p = GaussianParams(raw, torch.device("cpu"))
    adam = _torch_adam_on_flat(p)
    for vs in per_step:
        p.zero_grad()
        for v in vs:
            img = _fake_render(p.activated(), cams[v], None)["images_pred"]
            (torch.abs(img - gts[v]).mean() / world).backward()
        adam()
    np.testing.assert_allclose(p.flat.detach().numpy(), res[0][2], rtol=2e-5, atol=2e-6)


def _densify_rank_main(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from igs_amd import densify as dn
    raw, stats, cfg = _densify_case()
    P = raw["xyz"].shape[0]
    st = dn.DensifyState(P, torch.device("cpu"))
    for it in range(3):                                        # three steps; rank r sees view 2*it + r, its loss carries 1/world
        g, radii = stats[2 * it + rank]
        vis = radii > 0
        st.grad_accum += torch.where(vis, g / world, torch.zeros_like(g))      # what igs_densify_stats adds (CPU restatement)
        st.denom += vis.float()
        st.max_radii = torch.maximum(st.max_radii, torch.where(vis, radii.float(), torch.zeros_like(g)))
    dn.reduce_state(st, world)
    pl = dn.plan(raw["xyz"], raw["rotation"], raw["opacity"], raw["scaling"], st, cfg, torch.Generator().manual_seed(5))
    q.put((rank, st.grad_accum.numpy(), st.denom.numpy(), st.max_radii.numpy(),
           {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in pl.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _densify_case(P=400):
    from igs_amd import densify as dn
    from igs_amd.scenes import cfg1_scene
    raw, _, _ = cfg1_scene(P=P, size=16)
    gen = torch.Generator().manual_seed(21)
    stats = []
    for _ in range(6):                                          # per view: screen-space gradient norms and radii (0 = not visible)
        g = torch.rand(P, generator=gen) * 4e-4
        radii = torch.randint(0, 30, (P,), generator=gen) * (torch.rand(P, generator=gen) > 0.3)
        stats.append((g, radii.int()))
    cfg = dn.DensifyConfig(until_iter=100, from_iter=0, interval=3, grad_threshold=1.5e-4, min_opacity=0.05, max_num=P + 60,
                           percent_dense=0.01, extent=5.0)
    return raw, stats, cfg


def test_densification_statistics_reduce_over_ranks_gloo_world2():
    """N = 2 densification (gaussian_model.py:865-868, infer_batch.py:308-321): the per-rank accumulators are summed
    (and the radii maximised) over ranks before the decision; both ranks then hold the statistics a single process would have
    accumulated over the same six views at full loss weight, and derive the identical clone / split / prune plan (same RNG seed)."""
    import torch.multiprocessing as mp
    from igs_amd import densify as dn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_densify_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    (_, ga0, de0, mr0, pl0), (_, ga1, de1, mr1, pl1) = res
    np.testing.assert_array_equal(ga0, ga1); np.testing.assert_array_equal(de0, de1); np.testing.assert_array_equal(mr0, mr1)
    for k in pl0:
        np.testing.assert_array_equal(pl0[k], pl1[k])          # identical plan on both ranks, split samples included
    raw, stats, cfg = _densify_case()
    P = raw["xyz"].shape[0]
    st = dn.DensifyState(P, torch.device("cpu"))
    for g, radii in stats:
        vis = radii > 0
        st.grad_accum += torch.where(vis, g, torch.zeros_like(g))
        st.denom += vis.float()
        st.max_radii = torch.maximum(st.max_radii, torch.where(vis, radii.float(), torch.zeros_like(g)))
    np.testing.assert_allclose(ga0, st.grad_accum.numpy(), rtol=1e-6)
    np.testing.assert_array_equal(de0, st.denom.numpy()); np.testing.assert_array_equal(mr0, st.max_radii.numpy())
    assert pl0["n_clone"] + pl0["n_split"] > 0 and pl0["src"].shape[0] != P


def test_densify_with_an_unsupported_driver_raises():
    """Densification must not be skipped silently (round-1 finding): paths that cannot honour it raise."""
    from igs_amd.refine import GaussianParams, Refiner
    from igs_amd import densify as dn
    raw, cams, gts = _make_scene()
    p = GaussianParams(raw, torch.device("cpu"))
    r = Refiner(p, cams, gts, None, render_fn=_fake_render, adam_fn=_torch_adam_on_flat(p), densify=dn.DensifyConfig())
    with pytest.raises(NotImplementedError):
        r.step(view=0)


def test_depth_to_normal_on_an_analytic_plane():
    """RaDe-GS depth_double_to_normal (graphics_utils.py:97-126): a planar depth map gives the plane's normal at every interior
    pixel (sign convention of cross(d/dy, d/dx) as the reference indexes it), zero on the border."""
    import math
    import torch
    from igs_amd.camera import Camera
    from oracle.torch_losses import depth_pair_to_normals as depth_double_to_normal, backproject
    H, W = 24, 32
    cam = Camera(torch.eye(4), 2 * math.atan(W / (2 * 40.0)), 2 * math.atan(H / (2 * 40.0)), (H, W))
    # plane n . p = d in camera space, seen along the pixel rays: depth z = d / (n . ray)
    n = torch.tensor([0.2, -0.3, 0.93]); n = n / n.norm(); d = 4.0
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    rays = torch.stack([(xs - W / 2) / 40.0, (ys - H / 2) / 40.0, torch.ones_like(xs)], dim=0)
    z = d / (rays * n.view(3, 1, 1)).sum(0)
    p1, p2 = backproject(cam, z[None]), backproject(cam, 2 * z[None])
    torch.testing.assert_close(p1, rays * z, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(p2, 2 * p1)
    nm = depth_double_to_normal(cam, z[None], 2 * z[None])
    assert nm.shape == (2, 3, H, W)
    inner = nm[0][:, 1:-1, 1:-1]
    dots = (inner * n.view(3, 1, 1)).sum(0)
    assert torch.all(dots.abs() > 1 - 1e-4) and torch.all(dots.sign() == dots.flatten()[0].sign())
    assert float(nm[0][:, 0, :].abs().max()) == 0.0 and float(nm[1][:, :, -1].abs().max()) == 0.0
    torch.testing.assert_close(nm[1], nm[0], rtol=1e-4, atol=1e-5)          # scaling the depth does not change the normal
