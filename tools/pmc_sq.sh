#!/bin/bash
# SQ / instruction-cache counters of bench.py (GPU box; run through gpurun).  usage: tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/prof_${TAG}_ic -o ${TAG} -- python3 $ARGS > $OUT/prof_${TAG}_ic.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_IFETCH SQ_WAVES --output-format csv -d $OUT/prof_${TAG}_sq -o ${TAG} -- python3 $ARGS > $OUT/prof_${TAG}_sq.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("ic","sq"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/prof_${TAG}_%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "upr_qp" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
