#!/bin/bash
# A/B/A/B of run-time instantiations of the current upr_qp3.h under flag sets: bash tools/r6_ab.sh <workload> <B> "<flags0>" "<flags1>" ...
export TMPDIR=/tmp
W=$1; B=$2; shift; shift
for rep in 1 2 3; do python tools/exp_flags.py $W $B "$@"; done
