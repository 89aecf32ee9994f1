#!/bin/bash
# linearisation experiments: bench lines of experiment builds (UPR_LIB names as arguments), then the phase cycles of
# libupright_mi_prof.so (-DUPR_LIN_PROF) when it exists
export TMPDIR=/tmp
for lib in "$@"; do
  for r in 1 2; do
    echo -n "$lib: "; UPR_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"linearize": [0-9.]*\|"qp": [0-9.]*\|"linesearch": [0-9.]*' | tr '\n' ' '; echo
  done
done
[ -f upright_amd/libupright_mi_prof.so ] && UPR_LIB=libupright_mi_prof.so timeout 300 python tools/dbg_lin.py 2>&1 | tail -7
