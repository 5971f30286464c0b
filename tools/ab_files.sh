#!/bin/bash
# Same-box A/B of kernel SOURCE variants: tools/ab_files.sh <dir>...  -- every <dir> holds replacement files for igs_amd/csrc/;
# each variant is copied in, built, benchmarked twice (bench.py, no CPU baseline), and the original files are restored at the end.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/ab_orig && cp $R/igs_amd/csrc/*.hip $R/igs_amd/csrc/*.h /tmp/ab_orig/
for d in "$@"; do
  cp $d/* $R/igs_amd/csrc/
  if ! python -c "import igs_amd.build as b; b.build()" > /tmp/ab_build.log 2>&1; then echo "[$d] BUILD FAILED"; tail -5 /tmp/ab_build.log; cp /tmp/ab_orig/* $R/igs_amd/csrc/; continue; fi
  for i in 1 2; do
    timeout -k 10 150 python $R/bench.py --no-cpu-baseline ${AB_BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms']; print('[%s]' % sys.argv[1], 'ms/step', round(d['ms_per_step'],4), 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'], 'geom', s['geom_bwd'], 'pre', s['preprocess'], 'psnr', round(d['psnr']['after'],2))" "$d" || echo "[$d] run failed"
  done
  cp /tmp/ab_orig/* $R/igs_amd/csrc/
done
python -c "import igs_amd.build as b; b.build()" > /dev/null 2>&1
