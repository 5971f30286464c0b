#!/bin/bash
# usage: tools/gpu_step.sh <seconds> <logfile> <command...>   -- runs one GPU step under `timeout -k 10`, logs to gpurun_out/<logfile>;
# exits 0 unless the step was killed at its limit (so that `&&` chains stop after a hang, but not after an ordinary test failure)
lim=$1; log=$2; shift 2
mkdir -p gpurun_out
echo "== $* ==" > "gpurun_out/$log"
timeout -k 10 "$lim" "$@" >> "gpurun_out/$log" 2>&1
rc=$?
echo "rc=$rc" >> "gpurun_out/$log"
echo "[$log] rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
exit 0
