#!/usr/bin/env python3
"""Headline benchmark: Gaussians rendered/sec (fwd+bwd) at 1352x1014  (BASELINE.json metric).

Every step is ONE library call per rank (igs_refine_step: activations, forward, loss, backward and -- for N = 1 -- the Adam
update, 6 launches); for N > 1 the call ends in the flat gradient, which is all-reduced and followed by one Adam launch.

Workload at every N (weak scaling): BASELINE.json configs[2] -- the sear_steak-like frame-0 scene (200k Gaussians,
synthetic stand-in), 10 train cameras at 1352x1014, and per step ONE view per rank:
forward render + L1 loss + backward + (N>1: RCCL all-reduce of the flat 59*P-float gradient) + Adam step.
value = P * views_processed / seconds, whole job.

  python bench.py --gpus N --steps K --warmup W         (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timing over the timed region) and
`cpu_baseline` (the CPU oracle timed on a bounded sample of the same workload; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(R, W, H, coord=True, depth=True, bwd_coord=None, bwd_depth=None, bwd_normal=None, l1_fused=False):
    """per-launch algorithmic bytes of the two tile-blend kernels.
    forward : R*g + H*W*p + 8*T          g = 40 + 36[coord] + 12[depth] + 12[normal], p = 24 + 36[coord] + 12[depth] + 16[normal]
    backward: R*g + H*W*px + R*100       px = 28 + 48[coord] + 12[depth] + 16[normal]   (104 with every branch on)
    The backward's branches are those of the instance that was launched: with a colour-only loss the geometry gradients
    are absent and the colour-only instance runs (g = 40, px = 28).  With the L1 loss fused in (igs_refine_step) the kernel reads
    the rendered colour and the ground truth (24 B/px) instead of dL_dpix (12 B/px): px += 12."""
    normal = coord or depth
    g = 40 + 36 * coord + 12 * depth + 12 * normal
    p = 24 + 36 * coord + 12 * depth + 16 * normal
    T = ((W + 15) // 16) * ((H + 15) // 16)
    bc = coord if bwd_coord is None else bwd_coord
    bd = depth if bwd_depth is None else bwd_depth
    bn = (bc or bd) if bwd_normal is None else bwd_normal
    gb = 40 + 36 * bc + 12 * bd + 12 * bn
    px_bwd = 28 + (36 + 12) * bc + 12 * bd + 16 * bn + 12 * bool(l1_fused)
    return dict(blend_fwd=R * g + H * W * p + 8 * T, blend_bwd=R * gb + H * W * px_bwd + R * 100)


def cpu_baseline(raw, cams, bg, gts, n_views):
    """Times the CPU oracle (scalar C restatement of the reference, 1 thread) on `n_views` views of the same workload:
    forward + L1 gradient + backward.  Checker code used as a baseline only; never on the product path."""
    import numpy as np
    from igs_amd.scenes import activate
    from oracle import c_oracle as co
    co.set_precision("float32")
    a = {k: v.detach().cpu() for k, v in activate(raw).items()}
    P = a["means3D"].shape[0]
    t0 = time.time()
    for v in range(n_views):
        cam = cams[v]
        nr, out, st = co.rasterize_forward(bg, a["means3D"], None, a["opacities"], a["scales"], a["rotations"], 1.0, None,
                                           cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0,
                                           cam.height, cam.width, a["shs"], 3, cam.camera_center)
        g = np.sign(out["color"] - gts[v]).astype(np.float32) / out["color"].size
        co.rasterize_backward(st, bg, a["means3D"], None, a["scales"], a["rotations"], None, cam.world_view_transform,
                              cam.full_proj_transform, cam.camera_center, a["shs"], out["alpha"], out["normal"], g, None, None,
                              None, None, None, None)
    dt = time.time() - t0
    return dict(value=P * n_views / dt, unit="Gaussians/s", cores=1, kind="port",
                sample="%d view(s) fwd + L1 grad + bwd of the same 200k-Gaussian 1352x1014 workload on the scalar C oracle "
                       "(no Adam), %.1f s" % (n_views, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--cams", type=int, default=10)
    ap.add_argument("--loss", default="l1", choices=["l1", "l1_ssim"])
    ap.add_argument("--lambda-depth-normal", type=float, default=0.0,
                    help="add the RaDe-GS depth-normal regulariser with this weight (BASELINE cfg-5 uses 0.05): the blend backward\n"
                         "then runs its <depth, normal> instance")
    ap.add_argument("--clamp", action="store_true", help="clamp variant of the rasterizer (gradients clamped to +-15; BASELINE cfg-5)")
    ap.add_argument("--cpu-views", type=int, default=3, help="views timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage HIP events")
    ap.add_argument("--colour-only-forward", action="store_true",
                    help="NOT the reference configuration: render colour only (require_coord = require_depth = False); the L1 / SSIM\n"
                         "losses never look at the other outputs, so the refined parameters are the same")
    ap.add_argument("--viewspace-grad", action="store_true",
                    help="also produce dL/d(screen-space mean) with its absolute-gradient column (only the densification statistics\n"
                         "read it; the refine loop without densification does not)")
    ap.add_argument("--no-spatial-sort", action="store_true", help="keep the Gaussians in the (random) order of the synthetic scene")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the\n"
                    "N > 1 code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="record the per-stage HIP events on every N-th step of the timed region (each event costs stream time)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    dev = torch.device("cuda", local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    from igs_amd import _cabi, rasterizer
    from igs_amd.refine import GaussianParams, Refiner, render
    from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate

    rasterizer.NAN_CHECKS = False          # the reference's 7 NaN asserts are host syncs; parity tests keep them on
    raw, cams, bg = sear_steak_like_scene(P=args.points, n_cams=args.cams, width=args.width, height=args.height)
    cams = [c.to(dev) for c in cams]
    bg = bg.to(dev)
    # ground truth: the same renderer on a perturbed copy (synthetic data)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    params = GaussianParams(raw, dev)
    if not args.no_spatial_sort:
        params.spatial_sort()              # once per frame, outside the timed region: Morton order of the positions
    ref = Refiner(params, cams, gts, bg, loss=args.loss, world_size=world, rank=rank, seed=0,
                  lambda_depth_normal=args.lambda_depth_normal)
    ref.require_geometry = not args.colour_only_forward
    ref.clamp = args.clamp
    ref.want_viewspace_grad = args.viewspace_grad

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        ref.step()
    torch.cuda.synchronize()
    if not args.no_profile:
        _cabi.profile_enable(True, every=max(1, args.profile_every))
        _cabi.profile_read(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ref.step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    stages, r_sum, calls = ({}, 0.0, 0)
    if not args.no_profile:
        stages, r_sum, calls = _cabi.profile_read(reset=True)
        _cabi.profile_enable(False)

    if rank == 0:
        P = args.points
        value = P * args.steps * world / elapsed
        out = {
            "metric": "Gaussians rendered/sec (fwd+bwd) at 1352x1014",
            "value": value, "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sear_steak-like frame-0 stand-in (BASELINE.json configs[2]): %d Gaussians, SH degree 3, "
                                   "%d train cams @%dx%d, per step and rank one view: fwd + %s loss + bwd + Adam"
                                   % (P, args.cams, args.width, args.height, "L1" if args.loss == "l1" else "0.8*L1+0.2*(1-SSIM)"),
                       "points": P, "width": args.width, "height": args.height, "views": args.cams, "loss": args.loss,
                       "parallelism": ("one GPU: the whole step inside igs_refine_step, no exchange" if world == 1 else
                                       "views sharded over %d ranks, one view per rank and step; RCCL all-gather of the per-view colour "
                                       "gradients + all-reduce of the 11 small-group gradients (DESIGN.md section 6)" % world)},
        }
        if stages and calls:
            R_avg = r_sum / calls
            geo_bwd = False      # L1 / SSIM only see the colour image: geometry gradients are absent
            fused = args.loss == "l1"      # pure L1: the loss is evaluated inside blend_bwd (reads colour + gt instead of dL_dpix)
            geo_fwd = not args.colour_only_forward
            dn = args.lambda_depth_normal > 0      # depth-normal regulariser: depth + normal gradients present, coord absent
            ab = algorithmic_bytes(R_avg, args.width, args.height, geo_fwd, geo_fwd, geo_bwd, dn, dn, l1_fused=fused)
            per = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in stages.items()}
            dom = "blend_bwd" if per.get("blend_bwd", 0) >= per.get("blend_fwd", 0) else "blend_fwd"
            ach = ab[dom] / (per[dom] * 1e-3) / 1e9 if per[dom] > 0 else 0.0
            traffic = None
            tf = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:  # noqa: BLE001
                    traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": per[dom],
                               "instance": (("blend_bwd<depth, normal gradients%s>: R*64 + H*W*%d + R*100 bytes" % ((", L1 fused", 68) if fused else ("", 56)) if dn else
                                             "blend_bwd<colour-only gradients%s>: R*40 + H*W*%d + R*100 bytes" % ((", L1 fused", 40) if fused else ("", 28))) if dom == "blend_bwd"
                                            else "blend_fwd<coord,depth,normal>: R*100 + H*W*88 + 8*T bytes"),
                               "num_rendered_avg": R_avg,
                               "other": {"blend_fwd" if dom == "blend_bwd" else "blend_bwd": {
                                   "achieved": (ab["blend_fwd" if dom == "blend_bwd" else "blend_bwd"]
                                                / (per["blend_fwd" if dom == "blend_bwd" else "blend_bwd"] * 1e-3) / 1e9)
                                   if per.get("blend_fwd" if dom == "blend_bwd" else "blend_bwd", 0) > 0 else 0.0}},
                               "stage_ms": {k: round(v, 4) for k, v in per.items()}}
        if world == 1 and args.cpu_views > 0:
            try:
                gts_cpu = [g.cpu().numpy() for g in gts[:args.cpu_views]]
                out["cpu_baseline"] = cpu_baseline(raw, [c.to("cpu") for c in cams], bg.cpu(), gts_cpu, args.cpu_views)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Gaussians/s", "cores": 1, "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
