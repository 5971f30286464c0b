#!/bin/bash
# rocprofv3 kernel trace of the unchanged-caller path (tools/profile_dropin_host.py): which kernels make up its step?
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_dropin
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/profile_dropin_host.py > $O/run.log 2>&1
f=$(ls $O/stats/*/*_kernel_stats.csv | head -1)
head -40 "$f" | cut -c1-220 > $R/gpurun_out/r02_dropin_kernels.csv
find $O -name "*.db" -delete
