#!/bin/bash
# Instruction counts per kernel of the refine step (one rocprofv3 --pmc pass, no trace domains):  tools/pmc_quick.sh <tag> [env assignments...]
# prints SQ_WAVES / SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS / SQ_INSTS_VMEM per launch of every kernel -> gpurun_out/pmc_<tag>.txt
TAG=${1:-x}; shift || true
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmcq_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O -- python3 $R/tools/prof_run.py 8 > $O/run.log 2>&1
python3 - "$O" "$R/gpurun_out/pmc_$TAG.txt" <<'PY'
import csv, glob, sys, collections, os
f = glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as o:
    for k, c in sorted(agg.items()):
        if not any(x in k for x in ("blend", "preprocess", "geom_bwd", "tile_sort")): continue
        line = "%-62s " % k + "  ".join("%s %.3gM" % (n.replace("SQ_INSTS_", "").replace("SQ_", ""), sum(v) / len(v) / 1e6) for n, v in sorted(c.items()))
        print(line); o.write(line + "\n")
PY
find $O -name "*.db" -delete
