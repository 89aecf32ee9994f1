#!/bin/bash
# bench line of several experiment builds (UPR_LIB), three runs each
export TMPDIR=/tmp
for lib in "$@"; do
  for r in 1 2 3; do
    echo -n "$lib: "; UPR_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"qp": [0-9.]*\|"qp_iters_mean": [0-9.]*' | tr '\n' ' '; echo
  done
done
