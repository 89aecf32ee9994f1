#!/bin/bash
# Per-workload counter passes (GPU box; run through gpurun from the repo root):  tools/pmc_workloads.sh <tag>
# For the headline and each of the other workloads: three rocprofv3 --pmc passes of a short run of THAT workload alone
# (FETCH_SIZE | WRITE_SIZE | instruction mix + matrix-pipe busy cycles; --kernel-trace only next to --pmc), condensed into
# gpurun_out/<tag>_pmc_workloads.csv: workload, kernel, counter, dispatches, per-dispatch mean, batch.  bench.py reads the newest
# profiles/r*_pmc_workloads.csv (pmc_workload) for `traffic`, `frac_issued` and roofline_linearize.mfma_util of every workload.
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
MIX="SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU"
for W in headline config3 config4 config5 config5s contract headline_r03; do
  if [ $W = headline ]; then ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"; else ARGS="bench.py --only $W --extra-steps 2 --closed-loop-ticks 20 --no-cpu-baseline"; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pw_${TAG}_${W}_fetch -o $W -- python3 $ARGS > $OUT/pw_${TAG}_${W}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pw_${TAG}_${W}_write -o $W -- python3 $ARGS > $OUT/pw_${TAG}_${W}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc $MIX --output-format csv -d $OUT/pw_${TAG}_${W}_mix -o $W -- python3 $ARGS > $OUT/pw_${TAG}_${W}_mix.log 2>&1
done
python3 - <<PY
import csv, glob, collections
batch = {"headline": 1024, "config3": 4096, "config4": 1024, "config5": 1024, "config5s": 1024, "contract": 1024, "headline_r03": 1024}
with open("$OUT/${TAG}_pmc_workloads.csv", "w") as fh:
    fh.write("# tools/pmc_workloads.sh: rocprofv3 --kernel-trace --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | instruction mix), one short run per workload\n")
    import sys as _s; _s.path.insert(0, ".")
    import bench as _b
    fh.write("# kernel sources sha256: %s\n" % _b.library_sha())   # (comments and white space stripped; the build the counters belong to: bench.py compares it with the tree's)
    fh.write("# FETCH_SIZE / WRITE_SIZE in KB per dispatch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, MI355X_MICROARCH.md); the rest: per-dispatch means summed over the chip\n")
    fh.write("# configs[2]: config3 = the converging regime (warm launches), config3_cold = the launches at the iteration cap\n")
    fh.write('"workload","kernel","counter","dispatches","mean","batch"\n')
    for w in batch:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob("$OUT/pw_${TAG}_%s_*/**/*counter_collection.csv" % w, recursive=True):
            seen = collections.defaultdict(set)
            for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
                k = r["Kernel_Name"].split("(")[0]
                if "upr_qp" in k or "upr_linearize" in k or "upr_linesearch" in k:
                    # (round 5) configs[2]: the first extra-steps + 1 = 3 solves of the pass are cold launches at the iteration cap,
                    # the twelve behind them the converging regime -- kept apart: "config3_cold" / "config3"
                    seen[k].add(r["Dispatch_Id"])
                    ww = "config3_cold" if (w == "config3" and len(seen[k]) <= 3) else w
                    acc[(ww, k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (ww, k), d in acc.items():
            for c, v in sorted(d.items()):
                fh.write('"%s","%s","%s",%d,%.3f,%d\n' % (ww, k, c, len(v), sum(v) / len(v), batch[w]))
print(open("$OUT/${TAG}_pmc_workloads.csv").read()[:3000])
PY
