#!/bin/bash
# rocprofv3 kernel-trace stats of the DENSE diagnostic scene (bench.py --scene dense): the blend kernels where blending dominates.
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_dense_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --scene dense --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
f=$(ls $O/stats/*/*_kernel_stats.csv | head -1)
python3 - "$f" "$R/profiles/${TAG}_dense_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:20]:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
cp $R/profiles/${TAG}_dense_kernel_stats.csv $R/gpurun_out/
tail -1 $O/bench_under_rocprof.log | cut -c1-600
find $O -name "*.db" -delete
