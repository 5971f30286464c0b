#!/bin/bash
# Kernel trace of the N > 1 step's compute on ONE GPU (world size 1, RCCL): the program itself follows `--` (no launcher hop behind
# the profiler: the rank environment is exported here instead).  usage: tools/profile_exchange.sh <tag>
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_exchange_$TAG
rm -rf $O && mkdir -p $O
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/check_exchange.py --backend nccl --time > $O/run.log 2>&1
f=$(ls $O/stats/*/*_kernel_stats.csv | head -1)
python3 - "$f" "$R/profiles/${TAG}_exchange_step_world1_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:34]:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
grep "exchange " $O/run.log || true
cp $R/profiles/${TAG}_exchange_step_world1_kernel_stats.csv $R/gpurun_out/
find $O -name "*.db" -delete
