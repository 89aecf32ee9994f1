#!/bin/bash
# LDS counters of the headline QP kernel (GPU box; through gpurun): how busy the LDS pipe is and how much of that is conflicts.
# usage: tools/pmc_lds.sh <tag>
set -u
TAG=${1:-lds}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS\|SQ_WAIT_INST_LDS" | sort -u > $OUT/${TAG}_lds_counters.txt
cat $OUT/${TAG}_lds_counters.txt | tr '\n' ' '; echo
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/prof_${TAG}_$n -o ${TAG} -- python3 $ARGS > $OUT/prof_${TAG}_$n.log 2>&1 || echo "pass failed: $set"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/prof_${TAG}_SQ*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "upr_qp" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/${TAG}_pmc_lds.csv","w") as o:
    o.write('# rocprofv3 --pmc, headline QP kernel, bench.py --steps 2 --warmup 1, B=1024 (tools/pmc_lds.sh); per-dispatch means\n"counter","dispatches","mean"\n')
    for k,v in sorted(acc.items()):
        print(k, len(v), sum(v)/len(v)); o.write('"%s",%d,%.1f\n'%(k,len(v),sum(v)/len(v)))
PY
