#!/bin/bash
# GPU pass: parity tests, bench line, phase table
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r3_bench.log 2>&1
timeout 300 python tools/dbg_profile.py 1024 256 > gpurun_out/r3_phase256.log 2>&1
tail -5 gpurun_out/r3_tests.log; cat gpurun_out/r3_bench.log | cut -c1-400; cat gpurun_out/r3_phase256.log
