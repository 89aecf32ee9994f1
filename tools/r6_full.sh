#!/bin/bash
# GPU pass: parity tests + the bench line as the driver runs it (extras and CPU baseline included)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6_tests.log
tail -15 gpurun_out/r6_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_full.json 2> gpurun_out/r6_bench_full.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_full.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['roofline'], d.get('cpu_baseline'), d['kernel_ms'])
for e in d.get('extra_workloads',[]):
    print(e['workload'][:60], {k:e.get(k) for k in ('value','ms_per_step','ms_per_tick','qp_converged_fraction','qp_iters_mean','qp_status_counts','value_converged_subset','qp_not_converged_fraction','note')}, (e.get('warm') or {}).get('value'), e.get('kernel_ms') or {k: e[k] for k in ('N20','N100') if k in e})
PY
tail -3 gpurun_out/r6_bench_full.err
