#!/bin/bash
# rocprofv3 kernel stats of another bench configuration: tools/profile_cfg.sh <tag> <bench args...>  -> gpurun_out/profiles_<tag>/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-cold-leg --no-side-legs --no-dropin-leg "$@" > $O/bench_under_rocprof.log 2>&1
mkdir -p $R/gpurun_out/profiles_$TAG
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/profiles_$TAG/${TAG}_kernel_stats.csv
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
head -12 $R/gpurun_out/profiles_$TAG/${TAG}_kernel_stats.csv | cut -c1-150
