#!/bin/bash
# the library's ahead-of-time kernels against run-time instantiations of the current header, several workloads: bash tools/r6_abs.sh "<flags>" w1 w2 ...
export TMPDIR=/tmp
FLAGS="$1"; shift
for w in ${@:-headline}; do
  for rep in 1 2; do UPR_JIT_FLAGS="$FLAGS" python tools/exp_ab.py $w ${B:-1024}; done
done
