#!/bin/bash
# tools/probe/mall_probe under the fabric-side counters: what FETCH_SIZE / WRITE_SIZE report for a re-streamed MALL-resident set against a
# streamed (DRAM) one -> gpurun_out/r06_mall_probe.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
P=$PWD/tools/probe/mall_probe
{
echo "# tools/probe/mall_probe (tools/r6_mall.sh): 512 workgroups re-streaming a private set, 50 passes"
$P
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum; do
  rm -rf /tmp/mp_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/mp_$C -o mp -- $P > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/mp_$C/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "restream" in r["Kernel_Name"]]
    # dispatches: warm/timed pairs of l2, resident, stream -> the timed ones are 2nd, 4th, 6th
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for name, d in zip(("l2", "resident", "stream"), ids[1::2]):
        v = sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == d and r["Counter_Name"] == "$C")
        print("counter %-24s %-9s %.4g" % ("$C", name, v))
PY
done
} 2>&1 | tee $OUT/r06_mall_probe.txt
