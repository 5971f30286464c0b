#!/usr/bin/env python3
"""Headline benchmark: Gaussians rendered/sec (fwd+bwd) at 1352x1014; PSNR vs ref  (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg2|cfg4|cfg5] [--scene bench|dense]

N = 1 runs in this process.  N > 1 with no WORLD_SIZE in the environment SELF-LAUNCHES: before anything touches the GPU the parent
checks torch.cuda.device_count() >= N (refusing with a non-zero exit code otherwise), starts N ranks as child processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU over RCCL), relays
rank 0's JSON line and exits with the children's code.  Launched by torch.distributed.run directly (what the driver does for
N > 1) it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

Configurations (BASELINE.json `configs`; every scene is the synthetic "sear_steak-like" stand-in):
  cfg3 (default, the headline)  200k Gaussians, 10 train cams @1352x1014; one step = ONE view per rank: activations + forward
        (coord, depth, normal on) + L1 + backward + Adam -- one library call per rank (igs_refine_step, 5 launches); for N > 1 the
        call ends in the gradient, the ranks exchange it over RCCL and apply the identical Adam step.
  cfg2  forward-only render of the same scene through the drop-in entry point (one host wait per frame, as the reference);
        one step = one rendered view per rank.
  cfg4  stream: frames x 50 refine iterations (fresh optimiser per frame = load_fromstream, frame-to-frame drift of the target,
        reference loss 0.8 L1 + 0.2 (1 - SSIM)); one step = one refine iteration (one view per rank); --steps K runs K // 50
        frames (default 2500 = 50 frames).  Per-frame optimiser set-up is inside the timed region.
  cfg5  cfg4 with the RaDe-GS depth-normal regulariser (lambda 0.05: the <depth, normal> backward instance) and the clamp
        variant's +-15 gradient clamp; default 15000 steps = 300 frames.

Before the W warm-up steps cfg3 / cfg2 keep the GPU busy for --spinup-ms (default 200 ms, untimed) with the same step on a SCRATCH copy
of the scene: an idle MI355X runs the same 20 steps 13 % slower than 50 ms of work later (clock ramp; tools/experiments/spin_probe.py,
profiles/r03_spin_test.txt).  The real parameters, optimiser state and view sequence are not touched; `spinup` in the JSON line says what ran.

value = Gaussians x views processed / seconds, whole job (all ranks), inputs resident in HBM when the timed region starts.
Rank 0 prints ONE JSON line with `roofline` (dominant kernel: HIP-event stage timing + algorithmic bytes + pixel-Gaussian pairs/s),
`cpu_baseline` (N = 1 only: the pure-PyTorch restatement on BASELINE cfg-1 on all host cores, plus the scalar C port on a bounded
sample of this workload) and `psnr` (held-out camera before / after the refine steps that were timed).
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_FILE = "r04_pmc.json"   # committed rocprofv3 counter summary of this workload (tools/profile_all.sh); used only when its build fingerprint matches
ITERS_PER_FRAME = 50       # configs/demo.yaml refine_iterations


def usable_cores():
    """Host cores this process may actually use: the smaller of os.cpu_count(), the affinity mask and the cgroup CPU quota (a GPU box
    hands a one-GPU job a share of a many-core host; running one thread per HOST core on that share oversubscribes it badly)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:  # noqa: BLE001
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:  # noqa: BLE001
            continue
    return max(1, n)


def log(msg):
    """Progress on stderr (rank 0): a long run must show signs of life; the JSON line stays alone on stdout."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


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


# ----------------------------------------------------------------------------------------------------------------------------
# CPU baselines (checker code used as a baseline only; never on the product path)
# ----------------------------------------------------------------------------------------------------------------------------
def cpu_baseline_torch(repeats=5):
    """north_star / BASELINE.md 3: the pure-PyTorch projection + per-pixel alpha-blend restatement (oracle/torch_oracle.py) on
    BASELINE cfg-1 (10k Gaussians, 1 cam @256x256) with torch.set_num_threads(os.cpu_count()); 1 warm-up, median of `repeats`."""
    import torch
    from igs_amd.scenes import cfg1_scene, activate
    from oracle import torch_oracle as to
    cores = usable_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        raw, cams, bg = cfg1_scene()
        cam = cams[0]
        P = raw["xyz"].shape[0]

        def run(bwd):
            leaf = {k: v.clone().requires_grad_(bwd) for k, v in raw.items()}
            a = activate(leaf)
            t = time.perf_counter()
            with torch.set_grad_enabled(bwd):
                out = to.render(a["means3D"], a["shs"], None, a["opacities"], a["scales"], a["rotations"], None, 1.0,
                                cam.world_view_transform, cam.full_proj_transform, cam.camera_center, cam.tanfovx, cam.tanfovy,
                                0.0, cam.width, cam.height, 3, bg)
                if bwd:
                    out["color"].abs().mean().backward()
            return time.perf_counter() - t
        run(True)
        fb = statistics.median(run(True) for _ in range(repeats))
        f = statistics.median(run(False) for _ in range(repeats))
    finally:
        torch.set_num_threads(prev)
    return dict(value=P / fb, unit="Gaussians/s", cores=cores, kind="port",
                sample="BASELINE cfg-1 (10k Gaussians, 1 cam @256x256) forward + L1 + autograd backward on the pure-PyTorch restatement "
                       "(oracle/torch_oracle.py), torch.set_num_threads(%d) = the cores this job may use (os.cpu_count() = %d), median of %d "
                       "after 1 warm-up: %.2f s per view" % (cores, os.cpu_count() or 1, repeats, fb),
                forward_only_value=P / f, forward_only_s=f)


def cpu_baseline_c_port(raw, cams, bg, gts, n_views):
    """The scalar C restatement of the reference (oracle/rast_oracle.c, 1 thread) on `n_views` views of THIS workload:
    forward + L1 gradient + backward (no Adam)."""
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
                sample="%d view(s) fwd + L1 grad + bwd of this workload (%d Gaussians @%dx%d) on the scalar C oracle (no Adam), %.1f s"
                       % (n_views, P, cams[0].width, cams[0].height, dt))


# ----------------------------------------------------------------------------------------------------------------------------
# self-launch (N > 1 without a launcher)
# ----------------------------------------------------------------------------------------------------------------------------
def self_launch(n, backend="nccl"):
    """Runs before ANY GPU call in this process (torch.cuda.device_count() does not initialise the device on this image).
    With --backend gloo (a rehearsal of the N > 1 code path on a box with fewer GPUs) the ranks share the visible GPU(s)."""
    import torch
    have = torch.cuda.device_count()
    if have < n and backend == "nccl" or have < 1:
        print("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to run a smaller job under that label" % (n, have),
              file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)          # children inherit stdout / stderr: rank 0's JSON line goes straight through


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for cfg3 / cfg2, 2500 for cfg4, 15000 for cfg5)")
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--spinup-ms", type=float, default=200.0,
                    help="before the W warm-up steps: keep the GPU busy this long with the same step on a SCRATCH copy of the scene\n"
                         "(the real parameters are not touched).  A fresh or idle MI355X runs the same 20 steps at 0.277 ms each and,\n"
                         "~50 ms of work later, at 0.244 (clock ramp; tools/experiments/spin_probe.py, DESIGN.md section 5).  0 = off")
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5"], help="BASELINE.json configs[1..4] (see module docstring)")
    ap.add_argument("--mode", default=None, choices=["refine", "forward"], help="alias: --mode forward = --config cfg2")
    ap.add_argument("--scene", default="bench", choices=["bench", "dense"],
                    help="bench: the scene SURVEY.md 8d prescribes (log-scale mean -4, R/P = 2.6).  dense: DIAGNOSTIC, same Gaussians with\n"
                         "log-scale mean -2.7 (R in the millions, as SURVEY 8d's own example): the blend kernels where blending dominates")
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--width", type=int, default=1352)
    ap.add_argument("--height", type=int, default=1014)
    ap.add_argument("--cams", type=int, default=10)
    ap.add_argument("--loss", default=None, choices=["l1", "l1_ssim"], help="default: l1 for cfg3 (BASELINE configs[2]), l1_ssim for cfg4 / cfg5")
    ap.add_argument("--lambda-depth-normal", type=float, default=None,
                    help="weight of the RaDe-GS depth-normal regulariser (cfg5 default 0.05): the blend backward then runs its <depth, normal> instance")
    ap.add_argument("--clamp", action="store_true", help="clamp variant of the rasterizer (gradients clamped to +-15; default on for cfg5)")
    ap.add_argument("--densify", action="store_true", help="cfg4 / cfg5: densify-and-prune as configs/demo.yaml:57-62 (changes the Gaussian count)")
    ap.add_argument("--cpu-views", type=int, default=2, help="views timed on the scalar C oracle (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage HIP events")
    ap.add_argument("--no-cold-leg", action="store_true", help="cfg3: skip the cold K steps run before everything else (ms_per_step_cold)")
    ap.add_argument("--no-side-legs", action="store_true", help="cfg3: skip the cfg4 / cfg5 per-step side legs")
    ap.add_argument("--side-steps", type=int, default=100)
    ap.add_argument("--no-dropin-leg", action="store_true", help="cfg3: skip the unchanged-caller leg (tools/dropin_loop.py)")
    ap.add_argument("--dropin-steps", type=int, default=60)
    ap.add_argument("--colour-only-forward", action="store_true",
                    help="NOT the reference configuration: render colour only (require_coord = require_depth = False)")
    ap.add_argument("--viewspace-grad", action="store_true", help="also produce dL/d(screen-space mean) (the densification statistic)")
    ap.add_argument("--no-spatial-sort", action="store_true", help="keep the Gaussians in the (random) order of the synthetic scene")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the\n"
                    "N > 1 code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--exchange", default="colors", choices=["colors", "gradients"], help="N > 1 gradient exchange (DESIGN.md 6)")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="timed region: record the per-stage HIP events on every N-th step (each event costs stream time); the stage\n"
                         "means of the roofline come from a separate pass right after it with events on EVERY step")
    ap.add_argument("--profile-steps", type=int, default=48, help="steps of that separate profiled pass (0 = skip it)")
    args = ap.parse_args()
    if args.mode == "forward":
        args.config = "cfg2"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, args.backend))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE is %d: refusing to report one under the other's label" % (args.gpus, world), file=sys.stderr)
        sys.exit(3)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            if torch.cuda.device_count() < world:
                raise SystemExit("bench.py: %d ranks but %d GPUs" % (world, torch.cuda.device_count()))
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
        joined = dist.get_world_size()
    else:
        joined = 1
    dev = torch.device("cuda", local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    from igs_amd import _cabi, rasterizer
    from igs_amd.refine import GaussianParams, Refiner, render, psnr
    from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate, SEAR_STEAK_BBOX

    cfg = args.config
    stream = cfg in ("cfg4", "cfg5")
    loss = args.loss or ("l1_ssim" if stream else "l1")
    ldn = args.lambda_depth_normal if args.lambda_depth_normal is not None else (0.05 if cfg == "cfg5" else 0.0)
    clamp = args.clamp or cfg == "cfg5"
    steps = args.steps if args.steps is not None else {"cfg2": 200, "cfg3": 200, "cfg4": 2500, "cfg5": 15000}[cfg]
    if stream:
        frames = max(1, steps // ITERS_PER_FRAME)
        steps = frames * ITERS_PER_FRAME

    rasterizer.NAN_CHECKS = False          # the reference's 7 NaN asserts are host syncs; parity tests keep them on
    scale_mean = -2.7 if args.scene == "dense" else -4.0
    raw, cams_all, bg = sear_steak_like_scene(P=args.points, n_cams=args.cams, width=args.width, height=args.height,
                                              scale_mean=scale_mean, held_out=True)
    cams_all = [c.to(dev) for c in cams_all]
    cams, test_cam = cams_all[:-1], cams_all[-1]
    bg = bg.to(dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x, [x]
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        vals = [float(v.item()) for v in allv]
        return max(vals), vals

    log("config %s, scene %s, %d rank(s): scene built" % (cfg, args.scene, world))
    out_extra = {}
    stages, r_sum, calls = ({}, 0.0, 0)
    stages_timed = {}
    gauss_views = None            # Gaussians x views processed by THIS rank in the timed region (densification changes P)
    ref = None

    # ------------------------------------------------------------------------------------------------------------------
    if not stream:
        # ground truth: the same renderer on a perturbed copy (synthetic data)
        gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
        with torch.no_grad():
            gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
            gt_test = render(activate(gt_raw), test_cam, bg)["images_pred"].clone()
        params = GaussianParams(raw, dev)
        if not args.no_spatial_sort:
            params.spatial_sort()              # once per frame, outside the timed region: Morton order of the positions

        def eval_psnr():
            with torch.no_grad():
                return float(psnr(render(params.activated(), test_cam, bg)["images_pred"], gt_test))

        if cfg == "cfg3":
            ref = Refiner(params, cams, gts, bg, loss=loss, world_size=world, rank=rank, seed=0, lambda_depth_normal=ldn)
            ref.require_geometry = not args.colour_only_forward
            ref.clamp = clamp
            ref.want_viewspace_grad = args.viewspace_grad
            ref.exchange = args.exchange
            psnr_before = eval_psnr()

            def scratch_refiner(seed, loss_=loss, ldn_=ldn, clamp_=clamp):
                """The same step on a SCRATCH copy of the scene (the real parameters, optimiser state and view sequence stay untouched)."""
                p_s = GaussianParams(raw, dev)
                if not args.no_spatial_sort:
                    p_s.spatial_sort()
                r_s = Refiner(p_s, cams, gts, bg, loss=loss_, seed=seed, lambda_depth_normal=ldn_)
                r_s.require_geometry, r_s.clamp, r_s.want_viewspace_grad = ref.require_geometry, clamp_, args.viewspace_grad
                return r_s

            def timed(r, n_warm, n_steps):
                for _ in range(n_warm):
                    r.step()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n_steps):
                    r.step()
                torch.cuda.synchronize()
                return 1000.0 * (time.perf_counter() - t) / n_steps

            if world == 1 and not args.no_cold_leg:
                # FIRST GPU work of the refine loop in this process: W warm-up + K timed steps on a scratch copy, no spin-up before it --
                # what a caller sees who times K steps right after start-up (the GPU's clocks are still ramping: DESIGN.md section 5)
                r_cold = scratch_refiner(0)
                out_extra["ms_per_step_cold"] = timed(r_cold, args.warmup, steps)
                out_extra["cold_note"] = ("the same %d warm-up + %d timed steps run FIRST, on a scratch copy, without any spin-up before them; "
                                          "`ms_per_step` is measured afterwards (and after --spinup-ms of further untimed work)" % (args.warmup, steps))
                log("cold leg: %.4f ms per step" % out_extra["ms_per_step_cold"])
                del r_cold
            if args.spinup_ms > 0:
                # device spin-up on a scratch copy (no collectives, the real parameters and optimiser state stay as they are)
                r_spin = scratch_refiner(12345)
                t_spin, n_spin = time.perf_counter(), 0
                while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
                    for _ in range(20):
                        r_spin.step()
                    torch.cuda.synchronize()
                    n_spin += 20
                out_extra["spinup"] = {"ms": args.spinup_ms, "steps_on_a_scratch_copy": n_spin,
                                       "note": "untimed, before the W warm-up steps, on a separate copy of the parameters: brings the GPU's clocks to "
                                               "their steady state (the same 20 steps: 0.277 ms each on an idle GPU, 0.244 after ~50 ms of work; "
                                               "tools/experiments/spin_probe.py); --spinup-ms 0 turns it off; `ms_per_step_cold` is the figure without it"}
                del r_spin
            for _ in range(args.warmup):
                ref.step()
            torch.cuda.synchronize()
            if not args.no_profile:
                _cabi.profile_enable(True, every=max(1, args.profile_every))
                _cabi.profile_read(reset=True)
            if world > 1:
                ref.exchange_events = []
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ref.step()
            torch.cuda.synchronize()
            barrier()
            elapsed = time.perf_counter() - t0
            log("timed region: %d steps, %.4f ms per step" % (steps, 1000.0 * elapsed / steps))
            if world > 1:
                out_extra["exchange_ms_per_step"] = ref.exchange_ms_per_step(steps)
                out_extra["exchange"] = args.exchange
                out_extra["exchange_note"] = ("exchange_ms_per_step = HIP-event time of the collectives the main stream WAITS for; with "
                                              "--exchange colors the all-gather of the colour gradients starts from an event recorded right after "
                                              "the blend backward and runs on a side stream underneath geom_bwd (DESIGN.md section 6), so only its "
                                              "tail and the small-group all-reduces are inside this figure")
                ref.exchange_events = None
            if not args.no_profile:
                stages_timed, r_sum, calls = _cabi.profile_read(reset=True)
                _cabi.profile_enable(False)
            psnr_after = eval_psnr()
            out_extra["psnr"] = {"camera": "held-out view between training cameras %d and %d (never trained on)" % (args.cams // 2 - 1, args.cams // 2),
                                 "before": psnr_before, "after": psnr_after, "refine_steps_between": args.warmup + steps,
                                 "formula": "-10 log10 mean((clamp(img,0,1) - gt)^2)  (infer_batch.py:350-353); gt = the renderer's image of the perturbed target scene"}
            gauss_views = params.P * steps
            # separate profiled pass: events on EVERY step, so that each stage mean has >= 20 samples
            if not args.no_profile and args.profile_steps > 0:
                _cabi.profile_enable(True, every=1)
                _cabi.profile_read(reset=True)
                for _ in range(args.profile_steps):
                    ref.step()
                torch.cuda.synchronize()
                stages, r_sum2, calls2 = _cabi.profile_read(reset=True)
                _cabi.profile_enable(False)
                if calls2:
                    r_sum, calls = r_sum2, calls2
            else:
                stages = stages_timed
            # BASELINE configs[1] on the side (the reference's own throughput figure is a forward-only fps, infer_batch.py:125-145):
            # 60 forward renders of the refined scene through the drop-in entry point, one host wait per frame, device-synchronised
            try:
                a_f = {k: v.detach() for k, v in params.activated().items()}
                bufs_f = rasterizer.RasterBuffers()
                E_f = torch.Tensor([])

                def fwd_only(i):
                    cam = cams[(i * world + rank) % len(cams)]
                    rasterizer.rasterize_gaussians(bg, a_f["means3D"], E_f, a_f["opacities"], a_f["scales"], a_f["rotations"], 1.0, E_f,
                                                   cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0, cam.height,
                                                   cam.width, a_f["shs"], 3, cam.camera_center, False, True, True, False, buffers=bufs_f)
                for i in range(10):
                    fwd_only(i)
                torch.cuda.synchronize()
                tf0 = time.perf_counter()
                for i in range(60):
                    fwd_only(i)
                torch.cuda.synchronize()
                ms_f = 1000.0 * (time.perf_counter() - tf0) / 60
                out_extra["forward_only"] = {"ms_per_frame": ms_f, "gaussians_per_s": params.P / (ms_f * 1e-3), "frames_per_s": 1000.0 / ms_f,
                                             "note": "BASELINE configs[1] on this rank after the timed region: forward-only render (coord, depth, normal on) "
                                                     "through igs_rast_forward, 60 frames; `--config cfg2` is the full-length version"}
            except Exception as e:  # noqa: BLE001
                out_extra["forward_only"] = {"error": str(e)}
            if world == 1 and not args.no_side_legs:
                # BASELINE configs[3] / configs[4] per-step workloads on THIS scene (scratch copies, 100 steps each after 20 warm-up): the
                # reference's own loss 0.8 L1 + 0.2 (1 - SSIM), and + 0.05 depth-normal regulariser with the clamp variant's +-15 clamp
                side = {}
                for name, kw in (("cfg4", dict(loss_="l1_ssim", ldn_=0.0, clamp_=False)), ("cfg5", dict(loss_="l1_ssim", ldn_=0.05, clamp_=True))):
                    try:
                        r_s = scratch_refiner(7, **kw)
                        ms_s = timed(r_s, 20, args.side_steps)
                        side[name] = {"ms_per_step": ms_s, "steps": args.side_steps, "gaussians_per_s": params.P / (ms_s * 1e-3),
                                      "loss": "0.8*L1+0.2*(1-SSIM)" + (" + 0.05*depth_normal, clamp +-15" if name == "cfg5" else "")}
                        del r_s
                    except Exception as e:  # noqa: BLE001
                        side[name] = {"error": str(e)}
                side["note"] = ("igs_refine_step on scratch copies of this scene after the timed region, one view per step: the per-step workload "
                                "of BASELINE configs[3] / configs[4] (`--config cfg4|cfg5` runs the whole frame-by-frame streams)")
                out_extra["side_legs"] = side
                log("side legs: %s" % {k: (round(v["ms_per_step"], 4) if isinstance(v, dict) and "ms_per_step" in v else v) for k, v in side.items() if k != "note"})
            if world == 1 and not args.no_dropin_leg:
                # The UNCHANGED caller: tools/dropin_loop.py = the loop body of infer_batch.py:279-324 against the packages
                # diff_gaussian_rasterization_rade(_clamp) -- nn.Parameters, PyTorch activations, GaussianRasterizer(settings)(...), loss,
                # loss.backward(), optimizer.step(), zero_grad(set_to_none=True) -- same scene, fresh parameters, device-synchronised
                try:
                    from tools.dropin_loop import CallerModel, refine_iteration, make_losses
                    from igs_amd.refine import DEFAULT_LRS
                    lf = make_losses("igs")
                    raw_sorted = {k: v.detach().clone() for k, v in scratch_refiner(0).params.leaves.items()}      # (Morton order, as the timed leg)

                    def dropin(loss_, optimizer, nan, n_steps):
                        rasterizer.NAN_CHECKS = bool(nan)
                        gs = CallerModel(raw_sorted, dev, DEFAULT_LRS, optimizer=optimizer)
                        body = lambda v: refine_iteration(gs, cams[v], gts[v], bg, loss=loss_, losses=lf)
                        for i in range(20):
                            body(i % len(cams))
                        torch.cuda.synchronize()
                        t = time.perf_counter()
                        for i in range(n_steps):
                            body(i % len(cams))
                        torch.cuda.synchronize()
                        return 1000.0 * (time.perf_counter() - t) / n_steps
                    n_d = args.dropin_steps
                    # host-bound timing is noisy (the first variant of a process runs 10-20 % slower than the same variant later): two
                    # interleaved passes over the variants, the faster of the two is reported, both are kept
                    variants = (("l1_ms", "l1", "fused", 0, n_d), ("nan_checks_ms", "l1", "fused", 1, n_d), ("l1_ssim_ms", "l1_ssim", "fused", 0, n_d),
                                ("l1_ssim_nan_checks_ms", "l1_ssim", "fused", 1, n_d), ("l1_torch_adam_ms", "l1", "torch", 1, max(20, n_d // 3)))
                    passes = [{k: dropin(ls, opt, nan, n) for k, ls, opt, nan, n in variants} for _ in range(2)]
                    d = {k: min(passes[0][k], passes[1][k]) for k in passes[0]}
                    # The graph-replay legs (the same loop body inside igs_amd.graphs.GraphedLoop) run in a CHILD process: a capture this
                    # ROCm considers invalid ends hipStreamEndCapture in a segmentation fault, not an error code -- never seen with the
                    # loop as it is built, but this process owes the driver its JSON line.  Same scene, same Morton order, two passes.
                    try:
                        gv = ["l1:fused_capturable:1", "l1_ssim:fused_capturable:1"]
                        env = dict(os.environ, IGS_DROPIN_MORTON="0" if args.no_spatial_sort else "1")
                        env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
                        cp = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "dropin_bench.py"),
                                             str(n_d)] + gv + gv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
                        pl = [l for l in cp.stdout.splitlines() if l.startswith("PAIRS ")]
                        if cp.returncode != 0 or not pl:
                            d["graph_legs_error"] = "child exited with %s: %s" % (cp.returncode, (cp.stderr or "")[-300:])
                        else:
                            got = json.loads(pl[-1][6:])
                            for name, var in (("l1_graph_ms", gv[0]), ("l1_ssim_graph_ms", gv[1])):
                                vals = [ms for v_, ms in got if v_ == var]
                                d[name] = min(vals)
                                passes[0][name], passes[1][name] = vals[0], vals[-1]
                    except Exception as e:  # noqa: BLE001
                        d["graph_legs_error"] = "%s: %s" % (type(e).__name__, e)
                    d["steps"] = n_d
                    d["both_passes"] = passes
                    d["note"] = ("one refine iteration driven exactly as infer_batch.py:279-324 drives the reference's package (tools/dropin_loop.py), through "
                                 "the compiled `_C` module; `igs_amd.losses` for l1_loss / ssim (one import line) and, except in l1_torch_adam_ms, "
                                 "`igs_amd.optim.Adam` for torch.optim.Adam (one constructor); nan_checks = the reference's NaN asserts on "
                                 "(one word from the per-Gaussian kernel, collected at the end of the backward pass).  The caller's own ~35 small "
                                 "PyTorch kernels per iteration (activations and their backward, PSNR line, fills) bound this figure from the host side; "
                                 "*_graph_ms = the same loop body wrapped in igs_amd.graphs.GraphedLoop (tools/dropin_bench.py in a child process; captured once per view with torch.cuda.graph, "
                                 "then one graph launch per iteration; igs_amd.optim.Adam(capturable=True)), which removes that host cost")
                    out_extra["dropin"] = d
                    log("drop-in leg: %s" % {k: round(v, 4) for k, v in d.items() if isinstance(v, float)})
                except Exception as e:  # noqa: BLE001
                    out_extra["dropin"] = {"error": "%s: %s" % (type(e).__name__, e)}
                finally:
                    rasterizer.NAN_CHECKS = False
        else:
            # cfg2: forward-only render through the drop-in entry point (one host wait per frame, like rasterizer_impl.cu:354)
            a = {k: v.detach() for k, v in params.activated().items()}
            bufs = rasterizer.RasterBuffers()
            E = torch.Tensor([])
            my_cams = [cams[(i * world + rank) % len(cams)] for i in range(len(cams))]

            def fwd(i):
                cam = my_cams[i % len(my_cams)]
                return rasterizer.rasterize_gaussians(bg, a["means3D"], E, a["opacities"], a["scales"], a["rotations"], 1.0, E,
                                                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0,
                                                      cam.height, cam.width, a["shs"], 3, cam.camera_center, False,
                                                      not args.colour_only_forward, not args.colour_only_forward, False, buffers=bufs)
            if args.spinup_ms > 0:              # device spin-up (the forward has no state to disturb)
                t_spin, n_spin = time.perf_counter(), 0
                while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
                    for i in range(20):
                        fwd(i)
                    torch.cuda.synchronize()
                    n_spin += 20
                out_extra["spinup"] = {"ms": args.spinup_ms, "frames": n_spin, "note": "untimed, before the W warm-up frames (GPU clock ramp; --spinup-ms 0 = off)"}
            for i in range(args.warmup):
                fwd(i)
            torch.cuda.synchronize()
            if not args.no_profile:
                _cabi.profile_enable(True, every=max(1, args.profile_every))
                _cabi.profile_read(reset=True)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                fwd(i)
            torch.cuda.synchronize()
            barrier()
            elapsed = time.perf_counter() - t0
            log("timed region: %d forward renders, %.4f ms each" % (steps, 1000.0 * elapsed / steps))
            if not args.no_profile:
                stages_timed, r_sum, calls = _cabi.profile_read(reset=True)
                _cabi.profile_enable(True, every=1)
                for i in range(args.profile_steps):
                    fwd(i)
                torch.cuda.synchronize()
                stages, r_sum2, calls2 = _cabi.profile_read(reset=True)
                _cabi.profile_enable(False)
                if calls2:
                    r_sum, calls = r_sum2, calls2
                else:
                    stages = stages_timed
            out_extra["psnr"] = {"camera": "held-out view", "before": eval_psnr(), "after": None, "refine_steps_between": 0,
                                 "note": "forward-only configuration: nothing is refined"}
            gauss_views = params.P * steps
    else:
        # ------------------------------------------------------------------------------------------------------------------
        # cfg4 / cfg5: frame-by-frame stream (infer_batch.py:245-357 around the rasterizer).  Ground truth of every frame is
        # rendered BEFORE the timed region (the reference loads it from disk): a target copy of the scene whose dynamic-bbox
        # Gaussians drift by N(0, 0.01) per frame.
        from igs_amd.stream import SyntheticStream
        from igs_amd.densify import DensifyConfig
        src = SyntheticStream(raw, cams + [test_cam], bg, dev)
        warm_frames = 1
        all_gts = [src.next_frame() for _ in range(frames + warm_frames)]
        log("ground truth of %d frames rendered" % len(all_gts))
        dcfg = DensifyConfig(until_iter=100, from_iter=0, interval=20, grad_threshold=0.00015, max_num=int(args.points * 1.05),
                             extent=15.0) if args.densify else None
        cur = {k: v.clone() for k, v in raw.items()}
        psnr_b, psnr_a, psnr_a_sampled = [], [], []
        SORT_EVERY = 10
        gv = 0

        def one_frame(f, gts_f, timed):
            nonlocal cur, gv, ref
            params = GaussianParams(cur, dev)                 # load_fromstream: new leaves and a NEW optimiser for every frame
            # Morton order (0.58 ms of PyTorch argsort + one gather pass): the store handed on by the previous frame is still in
            # the order of its last sort and a frame moves a Gaussian by ~0.01, so every SORT_EVERY-th frame is enough
            if not args.no_spatial_sort and f % SORT_EVERY == 0:
                params.spatial_sort()
            ref = Refiner(params, cams, gts_f[:-1], bg, loss=loss, world_size=world, rank=rank, seed=f, densify=dcfg,
                          lambda_depth_normal=ldn)
            ref.clamp = clamp
            ref.exchange = args.exchange
            ref.start_frame()
            if timed and world > 1:
                ref.exchange_events = []
            # (the reference evaluates the held-out view once per frame, after the refinement: infer_batch.py:350-353; the
            #  "before" value is this benchmark's own addition and is sampled on every SORT_EVERY-th frame only)
            pb = None
            if f % SORT_EVERY == 0 or f == warm_frames:
                with torch.no_grad():
                    pb = psnr(render(params.activated(), test_cam, bg)["images_pred"], gts_f[-1])
            for _ in range(ITERS_PER_FRAME):
                ref.step()
                if timed:
                    gv += params.P
            with torch.no_grad():
                pa = psnr(render(params.activated(), test_cam, bg)["images_pred"], gts_f[-1])
            if timed:
                psnr_a.append(pa)
                if pb is not None:
                    psnr_b.append(pb); psnr_a_sampled.append(pa)
            cur = {k: v.detach().clone() for k, v in params.leaves.items()}          # convert2stream: next frame starts from here
            return ref

        for f in range(warm_frames):
            one_frame(f, all_gts[f], False)
        torch.cuda.synchronize()
        if not args.no_profile:
            _cabi.profile_enable(True, every=max(1, args.profile_every))
            _cabi.profile_read(reset=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_lists = []
        for f in range(frames):
            r = one_frame(warm_frames + f, all_gts[warm_frames + f], True)
            if world > 1:
                ev_lists.append(r.exchange_events)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        log("timed region: %d frames x %d iterations, %.4f ms per step" % (frames, ITERS_PER_FRAME, 1000.0 * elapsed / steps))
        if world > 1:
            out_extra["exchange_ms_per_step"] = sum(a.elapsed_time(b) for lst in ev_lists for a, b in lst) / max(1, steps)
            out_extra["exchange"] = args.exchange
        if not args.no_profile:
            stages, r_sum, calls = _cabi.profile_read(reset=True)
            stages_timed = stages
            _cabi.profile_enable(False)
        pb = [float(x) for x in psnr_b]; pa = [float(x) for x in psnr_a]; pas = [float(x) for x in psnr_a_sampled]
        out_extra["psnr"] = {"camera": "held-out view (never trained on), evaluated against each frame's target after its %d iterations "
                                       "(every frame) and before them (first frame and every %dth)" % (ITERS_PER_FRAME, SORT_EVERY),
                             "before": sum(pb) / len(pb), "after": sum(pa) / len(pa), "after_on_the_sampled_frames": sum(pas) / len(pas),
                             "first_frame": [pb[0], pas[0]], "last_sampled_frame": [pb[-1], pas[-1]],
                             "frames": frames, "refine_steps_between": ITERS_PER_FRAME}
        gauss_views = gv
        if ref is not None and ref.densify_log:
            out_extra["densify_log_last_frame"] = ref.densify_log

    # ------------------------------------------------------------------------------------------------------------------
    elapsed_max, per_rank = max_over_ranks(elapsed)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([float(gauss_views)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_gv = float(t.item())
    else:
        total_gv = float(gauss_views)

    if world > 1:
        import torch.distributed as dist
        bk = dist.get_backend()                        # the backend the process group actually runs on
        backend_name = "RCCL (torch.distributed backend nccl)" if bk == "nccl" else "torch.distributed backend %s (NOT RCCL: a rehearsal)" % bk
    else:
        backend_name = None
    if rank == 0:
        P = args.points
        loss_txt = "L1" if loss == "l1" else "0.8*L1+0.2*(1-SSIM)"
        work = {
            "cfg3": "BASELINE.json configs[2]: sear_steak-like frame-0 stand-in, %d Gaussians, SH degree 3, %d train cams @%dx%d; per step and rank "
                    "one view: fwd + %s loss + bwd + Adam" % (P, args.cams, args.width, args.height, loss_txt),
            "cfg2": "BASELINE.json configs[1]: sear_steak-like frame-0 stand-in, %d Gaussians, @%dx%d, forward-only render (coord, depth, normal on) "
                    "through igs_rast_forward, one host wait per frame" % (P, args.width, args.height),
            "cfg4": "BASELINE.json configs[3]: %d-frame stream x %d refine iterations, %d Gaussians, %d cams @%dx%d, fresh optimiser per frame, "
                    "loss %s%s; per step and rank one view" % (steps // ITERS_PER_FRAME if stream else 0, ITERS_PER_FRAME, P, args.cams, args.width, args.height, loss_txt,
                                                              ", densify-and-prune on" if args.densify else ""),
            "cfg5": "BASELINE.json configs[4]: %d-frame stream x %d refine iterations, %d Gaussians, SH degree 3, loss %s + %.2f * depth-normal "
                    "regulariser (<depth, normal> backward), clamp variant (+-15)%s" % (steps // ITERS_PER_FRAME if stream else 0, ITERS_PER_FRAME, P, loss_txt, ldn,
                                                                                       ", densify-and-prune on" if args.densify else ""),
        }[cfg]
        if args.scene == "dense":
            work = "DIAGNOSTIC dense scene (log-scale mean -2.7 instead of SURVEY 8d's -4) -- " + work
        out = {
            "metric": "Gaussians rendered/sec (fwd+bwd) at 1352x1014; PSNR vs ref" if cfg != "cfg2" else "Gaussians rendered/sec (forward only) at 1352x1014",
            "value": total_gv / elapsed_max, "unit": "Gaussians/s", "n_gpus": joined, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed_max / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": work, "name": cfg, "scene": args.scene, "points": P, "width": args.width, "height": args.height,
                       "views": args.cams, "loss": loss, "lambda_depth_normal": ldn, "clamp": bool(clamp),
                       "parallelism": ("one GPU: the whole step inside igs_refine_step, no exchange" if world == 1 else
                                       ("views sharded over %d ranks, one view per rank and step; %s all-gather of the per-view colour "
                                        "gradients + all-reduce of the 11 small-group gradients (DESIGN.md section 6)" % (world, backend_name)
                                        if args.exchange == "colors" else
                                        "views sharded over %d ranks, one view per rank and step; %s all-reduce of the flat 59*P-float gradient" % (world, backend_name))),
                       "backend": backend_name if world > 1 else None},
            "ms_per_step_per_rank": [1000.0 * e / steps for e in per_rank],
        }
        out.update(out_extra)
        if stages and calls:
            R_avg = r_sum / calls
            dn = ldn > 0                       # depth-normal regulariser: depth + normal gradients present, coord absent
            fused = loss == "l1" and not dn    # pure L1: the loss is evaluated inside blend_bwd (reads colour + gt instead of dL_dpix)
            geo_fwd = not args.colour_only_forward
            ab = algorithmic_bytes(R_avg, args.width, args.height, geo_fwd, geo_fwd, False, dn, dn, l1_fused=fused)
            per = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in stages.items()}
            nsamp = {k: cnt for k, (ms, cnt) in stages.items() if cnt}
            per_t = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in stages_timed.items()}
            # igs_refine_step with the L1 loss blends forward AND backward in one kernel per tile (blend_step.hip): the "blend_fwd" stage is
            # then that launch and the "blend_bwd" stage the empty interval between two event marks
            tile_fusion = cfg == "cfg3" and fused and stream is False and os.environ.get("IGS_NO_TILE_FUSION") is None and \
                per.get("blend_fwd", 0) > 0 and per.get("blend_bwd", 0) < 0.25 * per.get("blend_fwd", 0)
            T_tiles = ((args.width + 15) // 16) * ((args.height + 15) // 16)
            if tile_fusion:
                per["blend_step"] = per["blend_fwd"]
                nsamp["blend_step"] = nsamp.get("blend_fwd", 0)
                per_t["blend_step"] = per_t.get("blend_fwd")
                ab["blend_step"] = ab["blend_fwd"] + ab["blend_bwd"]
                dom, oth = "blend_step", None
            else:
                dom = "blend_bwd" if per.get("blend_bwd", 0) >= per.get("blend_fwd", 0) else "blend_fwd"
                oth = "blend_fwd" if dom == "blend_bwd" else "blend_bwd"
            # the bytes the launched instance cannot avoid (SURVEY's R*100 write term counts 25 moments, the colour-only
            # instance accumulates 10; the lean forward stores no backward state; the fused kernel re-reads nothing per pixel)
            HWp = args.width * args.height
            strict = {"blend_fwd": R_avg * (100 if geo_fwd else 40) + HWp * ((60 if geo_fwd else 16) + (8 if not fused else 4)) + 8 * T_tiles,
                      "blend_bwd": R_avg * (64 if dn else 40) + HWp * ((68 if fused else 56) if dn else (40 if fused else 28)) + R_avg * (64 if dn else 40),
                      "blend_step": R_avg * ((100 if geo_fwd else 40) + 40 + 40) + HWp * ((60 if geo_fwd else 16) + 12) + 8 * T_tiles}
            # `achieved` / `frac`: the bytes this instance CANNOT avoid (strict) over the launch time -- what VERDICT r3 asked to lead with;
            # SURVEY 8(d)'s formula (which charges per-pixel state the fused kernel never writes or re-reads, and 25 moments where the
            # colour-only instance accumulates 10) is kept beside it as *_survey_bytes
            ach_survey = ab[dom] / (per[dom] * 1e-3) / 1e9 if per[dom] > 0 else 0.0
            ach = strict[dom] / (per[dom] * 1e-3) / 1e9 if per[dom] > 0 else 0.0
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "bytes_per_launch": strict[dom],
                    "achieved_survey_bytes": ach_survey, "frac_survey_bytes": ach_survey / HBM_PEAK_GBS,
                    "traffic": None,
                    "traffic_note": "HBM bytes are not measured in this process; rocprofv3 PMC passes of the same command: profiles/%s" % PMC_FILE,
                    "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": per[dom], "avg_launch_samples": nsamp.get(dom, 0),
                    "avg_launch_ms_sampled_in_timed_region": per_t.get(dom),
                    "stage_timing": ("HIP events on the kernels' own stream, every step of a separate %d-step pass right after the timed region"
                                     % args.profile_steps) if (cfg in ("cfg2", "cfg3") and args.profile_steps > 0) else
                                    "HIP events on the kernels' own stream, every %d-th step of the timed region" % max(1, args.profile_every),
                    "instance": ("blend_step<coord,depth,normal | colour-only gradients, L1 fused>: one kernel per tile does the forward "
                                 "(R*100 + H*W*88 + 8*T) and the backward (R*40 + H*W*40 + R*100) -- SURVEY 8(d)'s bytes of the two kernels it replaces"
                                 if dom == "blend_step" else
                                 (("blend_bwd<depth, normal gradients%s>: R*64 + H*W*%d + R*100 bytes" % ((", L1 fused", 68) if fused else ("", 56)) if dn else
                                   "blend_bwd<colour-only gradients%s>: R*40 + H*W*%d + R*100 bytes" % ((", L1 fused", 40) if fused else ("", 28))) if dom == "blend_bwd"
                                  else "blend_fwd<coord,depth,normal>: R*100 + H*W*88 + 8*T bytes")),
                    "strict": {"bytes_per_launch": strict[dom], "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                               "note": "bytes this instance cannot avoid: gathers R*g, the images it writes, gt, and the accumulator rows it "
                                       "actually adds to (10 moments = 40 B for the colour-only backward, not SURVEY's 25 = 100 B); the lean forward "
                                       "stores no backward state and the fused kernel re-reads no per-pixel results"},
                    "num_rendered_avg": R_avg,
                    "other": ({oth: {"achieved": (ab[oth] / (per[oth] * 1e-3) / 1e9) if per.get(oth, 0) > 0 else 0.0,
                                     "frac": (ab[oth] / (per[oth] * 1e-3) / 1e9 / HBM_PEAK_GBS) if per.get(oth, 0) > 0 else 0.0,
                                     "algorithmic_bytes_per_launch": ab[oth], "avg_launch_ms": per.get(oth)}} if oth else
                              {"note": "forward and backward blend are one launch; IGS_NO_TILE_FUSION=1 python bench.py times them separately "
                                       "(round 3, same box: blend_fwd 55 us, blend_bwd 73 us, fused 122 us)"}),
                    "stage_ms": {({"blend_bwd": "marks_after_blend_step"}.get(k, k) if tile_fusion else k): round(v, 4)
                                 for k, v in per.items() if not (tile_fusion and k == "blend_fwd")}}
            # ALU-side figure (SURVEY.md 8d / hard part 3): pixel-Gaussian pairs = sum over pixels of the contributor count
            try:
                if ref is not None:
                    pr = ref.params
                    a = {k: v.detach() for k, v in pr.activated().items()}
                    E = torch.Tensor([])
                    tot_pairs = 0
                    ncam = min(len(cams), 4)
                    for cam in cams[:ncam]:
                        o = rasterizer.rasterize_gaussians(bg, a["means3D"], E, a["opacities"], a["scales"], a["rotations"], 1.0, E,
                                                           cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 0.0,
                                                           cam.height, cam.width, a["shs"], 3, cam.camera_center, False, True, True, False)
                        d = rasterizer.debug_dump(pr.P, o[0], cam.width, cam.height, o[9], o[10], o[11])
                        tot_pairs += int(d["n_contrib"][0].to(torch.int64).sum().item())
                    pairs = tot_pairs / ncam
                    roof["pairs_per_view"] = pairs
                    roof["pairs_per_s"] = ({"blend_step (every pair once forward, once backward)": 2.0 * pairs / (per["blend_step"] * 1e-3)} if tile_fusion else
                                           {k: (pairs / (per[k] * 1e-3) if per.get(k, 0) > 0 else None) for k in ("blend_fwd", "blend_bwd")})
                    roof["pairs_note"] = ("pixel-Gaussian pairs examined per view (sum over pixels of the reference's `contributor` count, "
                                          "forward.cu:556-573), mean over %d cameras; VALU issue fraction of the blend kernels: DESIGN.md section 5 "
                                          "(rocprofv3 SQ_INSTS_VALU x measured cycles per wave64 op, tools/ubench/valu_rate)" % ncam)
            except Exception as e:  # noqa: BLE001
                roof["pairs_per_view"] = None
                roof["pairs_note"] = "failed: %s" % e
            # issue side: wave-instruction counts per launch are a property of binary + workload and come from the committed rocprofv3 PMC
            # pass of this same workload (profiles/r03_pmc.json); the launch time is this run's
            try:
                if cfg == "cfg3" and args.scene == "bench" and P == 200000 and (args.width, args.height) == (1352, 1014) and loss == "l1" and not dn:
                    pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
                    from igs_amd import build as _hipbuild
                    same_build = pm.get("_meta", {}).get("build_fingerprint") == _hipbuild._fingerprint()
                    if not same_build:
                        roof["traffic_note"] = ("profiles/%s was counted on ANOTHER build of the kernels (fingerprint mismatch): its HBM bytes and "
                                                "instruction counts are not reported next to this run's launch time" % PMC_FILE)
                    vi = {}
                    for k in ((("blend_step",) if tile_fusion else ("blend_fwd", "blend_bwd")) if same_build else ()):
                        if k in pm and per.get(k, 0) > 0 and "SQ_INSTS_VALU" in pm[k]:
                            insts = sum(pm[k].get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
                            vi[k] = {"wave_insts_per_launch": insts, "valu": pm[k]["SQ_INSTS_VALU"], "salu": pm[k].get("SQ_INSTS_SALU"),
                                     "lds": pm[k].get("SQ_INSTS_LDS"), "issue_frac": insts * 2.5 / 1024.0 / (per[k] * 1e-3 * 2.4e9)}
                    if same_build and dom in pm and pm[dom].get("hbm_bytes_per_launch"):
                        # HBM bytes per launch of the dominant kernel: FETCH_SIZE + WRITE_SIZE passes of rocprofv3 on this same command,
                        # with the guide's unit and gfx950 corrections (tools/summarize_profile.py) -- the committed profile, not this process
                        roof["traffic"] = pm[dom]["hbm_bytes_per_launch"]
                        roof["traffic_note"] = ("HBM bytes per launch from the committed rocprofv3 PMC passes of this same command on this same build "
                                                "(separate FETCH_SIZE / WRITE_SIZE runs, corrected as MI355X_MICROARCH.md prescribes: profiles/%s, "
                                                "tools/profile_all.sh); NOT counted in this process" % PMC_FILE)
                    if vi:
                        roof["issue"] = dict(vi, note="vector + scalar + LDS wave-instructions per launch from profiles/%s (same build; NOT counted in this "
                                             "process) x 2.5 cycles per instruction and SIMD -- scalar instructions cost an issue slot like vector ones "
                                             "(tools/ubench/scalar_cost, profiles/r03_ubench_scalar_cost.txt) -- / 1024 SIMDs / (this run's launch time x "
                                             "2.4 GHz peak clock)" % PMC_FILE)
            except Exception:  # noqa: BLE001
                pass
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            log("GPU part done; timing the CPU baselines (%d usable cores)" % usable_cores())
            try:
                cb = cpu_baseline_torch()
                log("torch CPU baseline done")
            except Exception as e:  # noqa: BLE001
                cb = {"value": None, "unit": "Gaussians/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %s" % e}
            if args.cpu_views > 0 and not stream:
                try:
                    gts_cpu = [g.cpu().numpy() for g in gts[:args.cpu_views]]
                    cb["c_port"] = cpu_baseline_c_port(raw, [c.to("cpu") for c in cams], bg.cpu(), gts_cpu, args.cpu_views)
                except Exception as e:  # noqa: BLE001
                    cb["c_port"] = {"value": None, "sample": "failed: %s" % e}
            if not stream and cfg == "cfg3":
                # PSNR against an INDEPENDENT image: the held-out camera of the REFINED parameters rendered by the scalar C oracle
                # (oracle/rast_oracle.c, the checker) and by the HIP path; `psnr.before / after` above compare with this renderer's own target
                try:
                    from oracle import c_oracle as co
                    import numpy as np
                    co.set_precision("float32")
                    leaves = {k: v.detach() for k, v in params.leaves.items()}
                    with torch.no_grad():
                        hip_img = render(activate(leaves), test_cam, bg)["images_pred"].cpu().numpy()
                    a_c = {k: v.detach().cpu() for k, v in activate({k: v.cpu() for k, v in leaves.items()}).items()}
                    import copy
                    tc = copy.copy(cams_all[-1]).to("cpu")
                    t_o = time.time()
                    _, o_img, _ = co.rasterize_forward(bg.cpu(), a_c["means3D"], None, a_c["opacities"], a_c["scales"], a_c["rotations"], 1.0, None,
                                                       tc.world_view_transform, tc.full_proj_transform, tc.tanfovx, tc.tanfovy, 0.0, tc.height,
                                                       tc.width, a_c["shs"], 3, tc.camera_center)
                    mse = float(np.mean((np.clip(hip_img, 0, 1).astype(np.float64) - np.clip(o_img["color"], 0, 1).astype(np.float64)) ** 2))
                    gt_np = gt_test.cpu().numpy().astype(np.float64)
                    out["psnr"]["vs_oracle_image"] = -10.0 * math.log10(max(mse, 1e-30))
                    out["psnr"]["oracle_image_vs_target"] = -10.0 * math.log10(max(float(np.mean((np.clip(o_img["color"], 0, 1) - gt_np) ** 2)), 1e-30))
                    out["psnr"]["hip_image_vs_target_same_parameters"] = -10.0 * math.log10(max(float(np.mean((np.clip(hip_img, 0, 1) - gt_np) ** 2)), 1e-30))
                    out["psnr"]["max_abs_vs_oracle_image"] = float(np.abs(hip_img - o_img["color"]).max())
                    out["psnr"]["vs_oracle_note"] = ("held-out camera of the refined parameters: HIP image against the scalar C oracle's image of the same "
                                                     "parameters (%.1f s on one core), at the END of the run (the profiled pass and the side legs have refined "
                                                     "the parameters further since `after` was taken); oracle_image_vs_target and "
                                                     "hip_image_vs_target_same_parameters are the two renderers against the target at those same parameters" % (time.time() - t_o))
                except Exception as e:  # noqa: BLE001
                    out["psnr"]["vs_oracle_image"] = None
                    out["psnr"]["vs_oracle_note"] = "failed: %s: %s" % (type(e).__name__, e)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
