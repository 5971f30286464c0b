#!/bin/bash
# rocprofv3 kernel trace of the unchanged caller loop (tools/dropin_bench.py, one variant): kernel sequence of one steady-state step
# with durations and gaps -> gpurun_out/dropin_timeline_<tag>.txt       usage: tools/trace_dropin_loop.sh <tag> <variant>
TAG=${1:-x}; VAR=${2:-l1:fused:0}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_dropin_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/dropin_bench.py 40 $VAR > $O/run.log 2>&1
python3 - "$O" "$R/gpurun_out/dropin_timeline_$TAG.txt" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "preprocess_fwd" in r["Kernel_Name"]]
a, b = idx[-6], idx[-5]
# a step = from the first kernel after the previous step's optimiser ... use preprocess to preprocess
t0 = int(rows[a]["Start_Timestamp"])
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
busy = 0.0
with open(sys.argv[2], "w") as o:
    prev_end = None
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        busy += (e - s) / 1e3
        name = r["Kernel_Name"].replace("void ", "")[:90]
        line = "%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name)
        print(line); o.write(line + "\n")
        prev_end = e
    line = "step span %.1f us, kernels busy %.1f us, %d kernels" % (span, busy, b - a)
    print(line); o.write(line + "\n")
PY
find $O -name "*.db" -delete
