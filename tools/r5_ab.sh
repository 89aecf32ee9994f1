#!/bin/bash
# A/B of the library's ahead-of-time headline kernel against the run-time instantiation of the same header with other flags:
#   bash tools/r5_ab.sh "<UPR_JIT_FLAGS>" [workload ...]
export TMPDIR=/tmp
FLAGS="$1"; shift
for w in ${@:-headline}; do
  UPR_JIT_FLAGS="$FLAGS" python tools/exp_ab.py $w 1024
done
