#!/bin/bash
# per-phase cycle table of the headline QP kernel: run-time instantiations of the current header with the flags given
#   bash tools/r5_prof.sh "<flags A>" "<flags B>" ...
export TMPDIR=/tmp
for F in "$@"; do
  echo "=== UPR_JIT_FLAGS=$F"
  UPR_QP3_JIT=2 UPR_JIT_FLAGS="$F" python tools/dbg_profile.py 1024 2>/dev/null
done
