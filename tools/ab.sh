#!/bin/bash
# A/B on ONE box (AB_BENCH_ARGS = extra bench.py arguments): tools/ab.sh "<flags A>" "<flags B>" ...   (each variant: rebuild with IGS_EXTRA_FLAGS, then 2 bench runs)
for f in "$@"; do
  export IGS_EXTRA_FLAGS="$f"
  python -c "import igs_amd.build as b; b.build()" || exit 1
  for i in 1 2; do
    timeout -k 10 200 python bench.py --cpu-views 0 $AB_BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms']; print('[%s]' % sys.argv[1], round(d['ms_per_step'],4), s['blend_fwd'], s['blend_bwd'], s['geom_bwd'], s['preprocess'])" "$f" || exit 1
  done
done
