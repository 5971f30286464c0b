"""Host-side cost of one N > 1 refine step (colour exchange) with world size 1 on RCCL: cProfile over 200 steps.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/profile_exchange_host.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate

dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(backend="nccl", device_id=dev)
raw, cams, bg = sear_steak_like_scene(P=200000, n_cams=10, width=1352, height=1014)
cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}      # ground truth as in bench.py
with torch.no_grad():
    gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
p = GaussianParams(raw, dev); p.spatial_sort()
r = Refiner(p, cams, gts, bg, loss="l1", world_size=1, rank=0, seed=3)


def one():
    view = r._next_view()
    r._colour_exchange_step(cams[view], gts[view], r.last_picks)


for _ in range(30):
    one()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    one()
t_enq = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print("200 steps: enqueue %.3f ms/step, complete %.3f ms/step" % (t_enq * 5, t_all * 5))
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    one()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
