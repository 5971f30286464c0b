#!/bin/bash
for a in "--steps 20 --warmup 5" "--steps 200 --warmup 30" "--steps 20 --warmup 5" "--steps 20 --warmup 2"; do
  timeout -k 10 200 python bench.py --gpus 1 $a --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[%s]' % sys.argv[1], 'ms/step', round(d['ms_per_step'],4), 'value %.3g' % d['value'])" "$a" || exit 1
done
