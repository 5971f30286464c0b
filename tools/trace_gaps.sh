#!/bin/bash
# Kernel timeline of the refine step: rocprofv3 --kernel-trace (no counters) of tools/prof_run.py, then the gaps between consecutive
# kernels of a step -> gpurun_out/gaps_<tag>.txt      usage: tools/trace_gaps.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/prof_run.py 40 > $O/run.log 2>&1
python3 - "$O" "$R/gpurun_out/gaps_$TAG.txt" <<'PY'
import csv, glob, sys, os, collections
f = glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("preprocess_fwd", "tile_sort", "blend_", "geom_bwd"))]
rows = rows[len(rows) // 2:]          # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    na, nb = a["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""), b["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
    dur[na].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    gap[na + " -> " + nb].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
with open(sys.argv[2], "w") as o:
    for k, v in dur.items():
        line = "kernel %-28s mean %.1f us  (n=%d)" % (k, sum(v) / len(v), len(v)); print(line); o.write(line + "\n")
    for k, v in gap.items():
        v = sorted(v); line = "gap    %-50s median %.2f us  mean %.2f" % (k, v[len(v) // 2], sum(v) / len(v)); print(line); o.write(line + "\n")
PY
find $O -name "*.db" -delete
