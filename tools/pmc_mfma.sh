#!/bin/bash
# MFMA / fp64 instruction counters of bench.py's kernels (GPU box; run through gpurun).  usage: tools/pmc_mfma.sh <tag>
set -u
TAG=${1:-mf}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU --output-format csv -d $OUT/prof_${TAG}_mf -o ${TAG} -- python3 $ARGS > $OUT/prof_${TAG}_mf.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/prof_${TAG}_mf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "upr_" in k or "feedback" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/${TAG}_pmc_mfma.csv","w") as fh:
    fh.write("# rocprofv3 --pmc (one pass), bench.py --steps 2 --warmup 1, B=1024 (tools/pmc_mfma.sh); per-dispatch means summed over the chip\n")
    fh.write('"kernel","counter","dispatches","mean"\n')
    for k,d in acc.items():
        for c,v in sorted(d.items()): fh.write('"%s","%s",%d,%.1f\n'%(k,c,len(v),sum(v)/len(v)))
print(open("$OUT/${TAG}_pmc_mfma.csv").read())
PY
