#!/bin/bash
# A/B on ONE box of the per-step workloads of every BASELINE config (bench.py: cfg3 headline + cfg4 / cfg5 side legs):
# tools/ab_side.sh "<flags A>" "<flags B>" ...   (each variant: rebuild with IGS_EXTRA_FLAGS, then 2 bench runs)
for f in "$@"; do
  export IGS_EXTRA_FLAGS="$f"
  python -c "import igs_amd.build as b; b.build()" || exit 1
  for i in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-dropin-leg --no-cold-leg --side-steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['side_legs']; print('[%s]' % sys.argv[1], 'cfg3', round(d['ms_per_step'],4), 'cfg4', round(s['cfg4']['ms_per_step'],4), 'cfg5', round(s['cfg5']['ms_per_step'],4))" "$f" || exit 1
  done
done
export IGS_EXTRA_FLAGS=""; python -c "import igs_amd.build as b; b.build()" > /dev/null 2>&1
