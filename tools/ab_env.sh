#!/bin/bash
# A/B of one environment knob on the headline step, same box, alternating: tools/ab_env.sh VAR "v1 v2 v1 v2" [extra bench.py flags]
var=$1; vals=$2; shift 2
for v in $vals; do
  env "$var=$v" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision --no-roofline "$@" 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value']), round(d['ms_per_step'],3))"
done
