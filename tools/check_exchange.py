"""Two (or more) ranks on the visible GPU(s): one refine step with the flat-gradient all-reduce and one with the colour-gradient
exchange (Refiner.exchange = "gradients" / "colors") from the same state must leave the same gradient and the same parameters
on every rank.  Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/check_exchange.py
[--backend gloo|nccl]   (gloo rehearses the path on a one-GPU box; every rank then uses cuda:0)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist

from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, activate, perturbed_copy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--loss", default="l1_ssim")
    ap.add_argument("--clamp", action="store_true")
    ap.add_argument("--densify", action="store_true", help="instead: the refine loop with densify-and-prune on every rank (statistics reduced\n"
                    "over ranks before each decision); replicas must stay bit-identical through the rebuilds")
    ap.add_argument("--time", action="store_true", help="also time both exchanges on the bench workload (200k Gaussians @1352x1014)")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=dev)
    else:
        dist.init_process_group(backend=args.backend)
    raw, cams, bg = sear_steak_like_scene(P=30000, n_cams=4, width=400, height=300, focal=220.0)
    cams = [c.to(dev) for c in cams]
    bg = bg.to(dev)
    gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw, sigma=0.05).items()}
    with torch.no_grad():
        gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
    if args.densify:
        from igs_amd.densify import DensifyConfig
        cfg = DensifyConfig(until_iter=30, from_iter=0, interval=8, grad_threshold=2e-5, min_opacity=0.005, max_num=33000,
                            percent_dense=0.01, extent=15.0)
        p = GaussianParams(raw, dev)
        r = Refiner(p, cams, gts, bg, loss=args.loss, world_size=world, rank=rank, seed=3, densify=cfg, densify_seed=11)
        r.start_frame()
        counts = []
        for _ in range(26):
            r.step()
            counts.append(p.P)
        torch.cuda.synchronize()
        n = torch.tensor([p.P, p.step_count], device=dev)
        lo, hi = n.clone(), n.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
        if same:
            for t in (p.flat, p.exp_avg, p.exp_avg_sq):
                a, b = t.clone(), t.clone()
                dist.all_reduce(a, op=dist.ReduceOp.MIN); dist.all_reduce(b, op=dist.ReduceOp.MAX)
                same = same and bool(torch.equal(a, b)) and bool(torch.isfinite(t).all())
        ok = same and [e[0] for e in r.densify_log] == [8, 16, 24] and len(set(counts)) > 1 and p.step_count == 26 - 3
        if rank == 0:
            print("densify log (iteration, cloned, split, pruned, P):", r.densify_log, "replicas identical:", same, "steps:", p.step_count)
            print("DENSIFY_CHECK_OK" if ok else "DENSIFY_CHECK_FAILED")
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    results = {}
    # "colors": the gather starts from the event the library records right after the blend backward and runs underneath geom_bwd on a
    # side stream; "colors_serial": the same exchange behind the last kernel of the step -- the two must agree (to rounding)
    for mode in ("gradients", "colors", "colors_serial"):
        p = GaussianParams(raw, dev)
        r = Refiner(p, cams, gts, bg, loss=args.loss, world_size=world, rank=rank, seed=3)
        r.exchange, r.clamp = mode.split("_")[0], args.clamp
        r.overlap_exchange = mode == "colors"
        for _ in range(3):
            if world > 1:
                r.step()
            else:       # one rank: Refiner.step() would take the single-GPU path; drive the N > 1 code (and its collectives) by hand
                view = r._next_view()
                if mode != "gradients":
                    r._colour_exchange_step(cams[view], gts[view], r.last_picks)
                else:
                    r._fused_step(cams[view], gts[view], grads_only=True)
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    r.adam_fn()
        torch.cuda.synchronize()
        sh0 = p.spans["shs"][0]
        # (the colour exchange applies the SH update without ever writing the SH gradient: compare the first moment instead)
        results[mode] = (p.grad[:sh0].clone(), p.exp_avg.clone(), p.flat.clone())
    ok = True
    # (two runs of the same step differ in the last bits already -- the blend backward's float atomics land in another order -- so the
    #  overlapped and the serial exchange are compared like the two modes: to rounding)
    for i, what in enumerate(("small-group gradient", "first moment", "parameters")):
        A, B = results["colors"][i], results["colors_serial"][i]
        close = ((A - B).abs() <= 1e-5 * B.abs() + 1e-6 * float(B.abs().max())).float().mean().item()
        if rank == 0:
            print("%s: overlapped gather vs serial gather: within tolerance %.4f of the elements" % (what, close))
        ok = ok and close > 0.999
    for i, what in enumerate(("small-group gradient", "first moment", "parameters")):
        A, B = results["colors"][i], results["gradients"][i]
        err = float((A - B).abs().max()); scale = float(B.abs().max())
        # replicas must agree bit for bit inside a mode; across modes the SH sum is formed in a different order (ulps), which Adam
        # may turn into +-lr on near-zero gradients after three steps: compare the bulk
        close = ((A - B).abs() <= 1e-5 * B.abs() + 1e-6 * scale).float().mean().item()
        mine = A.clone(); dist.all_reduce(mine, op=dist.ReduceOp.MAX); same = bool(torch.equal(mine, A) or world == 1)
        mn = A.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN); same = same and bool(torch.equal(mn, A))
        if rank == 0:
            print("%s: max |colors - gradients| = %.3g (scale %.3g), within tolerance: %.4f of the elements, replicas identical: %s" % (what, err, scale, close, same))
        ok = ok and close > 0.999 and same
    if world == 1:
        # one rank: the exchange step (gradients to HBM, collectives over a one-rank communicator, igs_adam_exchange_step) must land where
        # the single-GPU fused step (igs_refine_step applying Adam itself) lands from the same start with the same views
        p = GaussianParams(raw, dev)
        r = Refiner(p, cams, gts, bg, loss=args.loss, seed=3)
        r.clamp = args.clamp
        for _ in range(3):
            r.step()
        torch.cuda.synchronize()
        for i, what in ((1, "first moment"), (2, "parameters")):
            A, B = results["colors"][i], (p.exp_avg, p.flat)[i - 1]
            close = ((A - B).abs() <= 1e-5 * B.abs() + 1e-6 * float(B.abs().max())).float().mean().item()
            print("%s: exchange step (backend %s, 1 rank) vs single-GPU fused step: within tolerance %.4f of the elements" % (what, args.backend, close))
            ok = ok and close > 0.999
    if rank == 0:
        print("EXCHANGE_CHECK_OK" if ok else "EXCHANGE_CHECK_FAILED")
    if args.time:
        import time
        raw, cams, bg = sear_steak_like_scene(P=200000, n_cams=10, width=1352, height=1014)
        cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
        gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}          # ground truth as in bench.py
        with torch.no_grad():
            gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
        for mode in ("gradients", "colors"):
            p = GaussianParams(raw, dev); p.spatial_sort()
            r = Refiner(p, cams, gts, bg, loss="l1", world_size=world, rank=rank, seed=3)
            r.exchange = mode

            def one():
                if world > 1:
                    r.step()
                else:
                    view = r._next_view()
                    if mode == "colors":
                        r._colour_exchange_step(cams[view], gts[view], r.last_picks)
                    else:
                        r._fused_step(cams[view], gts[view], grads_only=True)
                        dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                        r.adam_fn()
            for _ in range(20):
                one()
            torch.cuda.synchronize(); dist.barrier(); t = time.perf_counter()
            for _ in range(100):
                one()
            torch.cuda.synchronize(); dist.barrier()
            if rank == 0:
                print("exchange %-9s: %.3f ms per step (%d rank(s), backend %s)" % (mode, (time.perf_counter() - t) * 10.0, world, args.backend))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
