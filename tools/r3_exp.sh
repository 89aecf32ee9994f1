#!/bin/bash
# perf iteration on an experiment build (tools/exp_build.sh): smoke parity, bench line, phase table; args: library names
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in "$@"; do
  echo "=== $lib"
  UPR_LIB=$lib timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  UPR_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"qp": [0-9.]*' | tr '\n' ' '; echo
  UPR_LIB=$lib timeout 300 python tools/dbg_profile.py 1024 256 2>&1 | grep -v "^   (a phase\|amdgpu.ids"
done
