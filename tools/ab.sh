#!/bin/bash
# A/B on ONE box (AB_BENCH_ARGS = extra bench.py arguments): tools/ab.sh "<flags A>" "<flags B>" ...   (each variant: rebuild with IGS_EXTRA_FLAGS, then 2 bench runs)
for f in "$@"; do
  export IGS_EXTRA_FLAGS="$f"
  python -c "import igs_amd.build as b; b.build()" || exit 1
  for i in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline ${AB_BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms']; print('[%s]' % sys.argv[1], 'ms/step', round(d['ms_per_step'],4), 'blend', s.get('blend_step', s.get('blend_fwd')), 'bwd', s.get('blend_bwd'), 'geom', s['geom_bwd'], 'pre', s['preprocess'], 'sort', s['tile_sort'])" "$f" || exit 1
  done
done
export IGS_EXTRA_FLAGS=""; python -c "import igs_amd.build as b; b.build()" > /dev/null 2>&1
