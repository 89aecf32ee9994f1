#!/bin/bash
# Profile bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag>      e.g.  tools/profile.sh r01b
# Three separate rocprofv3 passes (kernel trace + stats, PMC FETCH_SIZE, PMC WRITE_SIZE), then
# tools/summarize_profiles.py condenses them into gpurun_out/<tag>_*.csv (copy those into profiles/).
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
# (the timing pass runs the driver's step count: its average then is the steady state, not the first launch without a dispatch order)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -o ${TAG} -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/prof_${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -o ${TAG} -- python3 $ARGS > $OUT/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -o ${TAG} -- python3 $ARGS > $OUT/prof_${TAG}_write.log 2>&1
python3 tools/summarize_profiles.py $TAG
