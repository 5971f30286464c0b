#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py + separate PMC passes of tools/prof_run.py (counters never share a run with the trace
# domains; FETCH_SIZE and WRITE_SIZE in passes of their own, MI355X_MICROARCH.md).  usage: tools/profile_all.sh <tag> [bench args]
# Summaries go to profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json (tools/summarize_profile.py).
set -e
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-cold-leg --no-side-legs --no-dropin-leg "$@" > $O/bench_under_rocprof.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/prof_run.py 12 > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/prof_run.py 12 > $O/write.log 2>&1
echo write done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/tools/prof_run.py 12 > $O/sq.log 2>&1
echo sq done
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d $O/misc -- python3 $R/tools/prof_run.py 12 > $O/misc.log 2>&1
echo misc done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
python3 $R/tools/summarize_profile.py $TAG $O/stats --fetch $O/fetch --write $O/write --sq $O/sq --misc $O/misc > $O/summary.log 2>&1 || true
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/ 2>/dev/null || true
du -sh $O
