#!/bin/bash
# A/B on ONE box: the per-Gaussian backward taking Sigma^-1 from the forward's plane cache (default) against running the eigen-solver
# again (IGS_NO_PLANE_CACHE=1); BASELINE configs[4] side leg of bench.py, alternating, 3 rounds
for i in 1 2 3; do
  for v in cache nocache; do
    if [ $v = nocache ]; then export IGS_NO_PLANE_CACHE=1; else unset IGS_NO_PLANE_CACHE; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-dropin-leg --no-cold-leg --side-steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['side_legs']; print('[%s]' % sys.argv[1], 'cfg3', round(d['ms_per_step'],4), 'cfg4', round(s['cfg4']['ms_per_step'],4), 'cfg5', round(s['cfg5']['ms_per_step'],4))" "$v" || exit 1
  done
done
