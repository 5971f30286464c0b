#!/bin/bash
# HBM bytes per launch of the image-space loss kernels (cfg5's per-step workload): separate FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/pmc_loss.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_loss
rm -rf $O && mkdir -p $O
cat > $O/run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from igs_amd.refine import GaussianParams, Refiner, render
from igs_amd.scenes import sear_steak_like_scene, perturbed_copy, activate
dev = torch.device("cuda:0")
raw, cams, bg = sear_steak_like_scene()
cams = [c.to(dev) for c in cams]; bg = bg.to(dev)
gt_raw = {k: v.to(dev) for k, v in perturbed_copy(raw).items()}
with torch.no_grad():
    gts = [render(activate(gt_raw), c, bg)["images_pred"].clone() for c in cams]
p = GaussianParams(raw, dev); p.spatial_sort()
r = Refiner(p, cams, gts, bg, loss="l1_ssim", lambda_depth_normal=0.05)
for _ in range(24):
    r.step()
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $O/run.py > $O/$c.log 2>&1
done
python3 - "$O" "$R/gpurun_out/pmc_loss.txt" <<'PY'
import csv, glob, sys, os, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(sys.argv[1], c, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as o:
    for k, v in sorted(agg.items()):
        if not any(x in k for x in ("ssim", "depth_normal", "blend", "geom", "preprocess")): continue
        f = sum(v["FETCH_SIZE"][-8:]) / max(1, len(v["FETCH_SIZE"][-8:])); w = sum(v["WRITE_SIZE"][-8:]) / max(1, len(v["WRITE_SIZE"][-8:]))
        line = "%-50s FETCH %.1f MB (x2 corrected %.1f)  WRITE %.1f MB  total corrected %.1f MB" % (k, f / 1024, 2 * f / 1024, w / 1024, (2 * f + w) / 1024)
        print(line); o.write(line + "\n")
PY
find $O -name "*.db" -delete
