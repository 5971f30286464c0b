#!/bin/bash
for v in 0 1 0 1; do
  IGS_SCRATCH_CLEAN=$v timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_ms']; print('[clean=%s]' % sys.argv[1], 'ms/step', round(d['ms_per_step'],4), 'fwd', s['blend_fwd'], 'bwd', s['blend_bwd'], 'geom', s['geom_bwd'], 'pre', s['preprocess'], 'sort', s['tile_sort'], 'fill', s.get('memset'))" "$v" || exit 1
done
