#!/bin/bash
# Evidence set of a round on the GPU box (run through gpurun from the repo root): tools/profile_all.sh <tag>
#   1. bench.py as the driver runs it (headline + extra workloads + CPU baseline)      -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the headline workload                        -> <tag>_kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)                        -> <tag>_pmc_hbm.csv
#   4. rocprofv3 --pmc instruction mix (tools/pmc_mfma.sh)                              -> <tag>_pmc_mfma.csv
#   5. rocprofv3 --kernel-trace --stats of bench.py WITH the extra workloads            -> <tag>_kernel_stats_all.csv
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
bash tools/profile.sh $TAG > $OUT/prof_${TAG}.log 2>&1
bash tools/pmc_mfma.sh $TAG > $OUT/prof_${TAG}_mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_all -o ${TAG}all -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_all.log 2>&1
cp $(find $OUT/prof_${TAG}_all -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_all.csv
#   6. the same counters for the kernels of the extra workloads (short passes WITH the extras)   -> <tag>_pmc_hbm_all.csv, <tag>_pmc_mfma_all.csv
XARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --extra-steps 2 --closed-loop-ticks 20"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch_all -o ${TAG} -- python3 $XARGS > $OUT/prof_${TAG}_fetch_all.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write_all -o ${TAG} -- python3 $XARGS > $OUT/prof_${TAG}_write_all.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/prof_${TAG}_mf_all -o ${TAG} -- python3 $XARGS > $OUT/prof_${TAG}_mf_all.log 2>&1
python3 tools/summarize_profiles.py $TAG all
#   7. (round 4) the same counters PER WORKLOAD, one short run of each workload alone                 -> <tag>_pmc_workloads.csv
bash tools/pmc_workloads.sh $TAG > $OUT/prof_${TAG}_workloads.log 2>&1
head -c 600 $OUT/${TAG}_bench.json; echo
head -8 $OUT/${TAG}_kernel_stats_all.csv
