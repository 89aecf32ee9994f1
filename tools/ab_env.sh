#!/bin/bash
# A/B of an environment switch of the library on ONE box: tools/ab_env.sh VAR a b  -> bench.py's headline line, three alternating pairs
export TMPDIR=/tmp
for r in 1 2 3; do for v in $2 $3; do
  env $1=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$1=$v', round(d['value']), round(d['ms_per_step'],4), round(k['linearize'],4), round(k['qp'],4), round(k['linesearch'],4))"
done; done
