#!/bin/bash
# bench.py's headline line three times (value, ms per step, the three kernels' durations); UPR_LIB selects another build
export TMPDIR=/tmp
for r in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print(round(d['value']), round(d['ms_per_step'],4), round(k['linearize'],4), round(k['qp'],4), round(k['linesearch'],4), k['launches'])"
done
