#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh into the two small csv files kept under profiles/:
<tag>_kernel_stats.csv (copy of rocprofv3's kernel_stats) and <tag>_pmc_hbm.csv (mean counter value per
dispatch per kernel, one row per (kernel, counter))."""
import csv
import glob
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
out = "gpurun_out"
for f in glob.glob(f"{out}/prof_{tag}_trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"{out}/{tag}_kernel_stats.csv")
rows = []
for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = defaultdict(list)
    for f in glob.glob(f"{out}/prof_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0]
                acc[name].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        rows.append((k, counter, len(v), sum(v) / len(v)))
with open(f"{out}/{tag}_pmc_hbm.csv", "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 3 --warmup 1, B=1024\n")
    fh.write("# units: KB per dispatch (mean over dispatches).  gfx950 correction (MI355X_MICROARCH.md, HBM): hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024\n")
    w = csv.writer(fh, quoting=csv.QUOTE_ALL)
    w.writerow(["kernel", "counter", "dispatches", "mean_KB"])
    for k, c, n, m in rows:
        w.writerow([k, c, n, f"{m:.3f}"])
print(open(f"{out}/{tag}_pmc_hbm.csv").read())

if len(sys.argv) > 2 and sys.argv[2] == "all":
    # the all-workloads passes (tools/profile_all.sh step 6): the QP kernels of every workload, per-dispatch means
    rows = []
    for counter, d in (("FETCH_SIZE", "fetch_all"), ("WRITE_SIZE", "write_all")):
        acc = defaultdict(list)
        for f in glob.glob(f"{out}/prof_{tag}_{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter and "upr_" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            rows.append((k, counter, len(v), sum(v) / len(v)))
    with open(f"{out}/{tag}_pmc_hbm_all.csv", "w") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1 --extra-steps 2 --closed-loop-ticks 20 (all workloads)\n")
        fh.write("# units: KB per dispatch (mean over the dispatches of the pass; configs[2] mixes cold launches at the iteration cap with warm ones).  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024\n")
        w = csv.writer(fh, quoting=csv.QUOTE_ALL)
        w.writerow(["kernel", "counter", "dispatches", "mean_KB"])
        for k, c, n, m in rows:
            w.writerow([k, c, n, f"{m:.3f}"])
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{out}/prof_{tag}_mf_all/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "upr_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"{out}/{tag}_pmc_mfma_all.csv", "w") as fh:
        fh.write("# rocprofv3 --pmc (one pass), all workloads (as above); per-dispatch means summed over the chip\n")
        fh.write('"kernel","counter","dispatches","mean"\n')
        for k, d in acc.items():
            for c, v in sorted(d.items()):
                fh.write('"%s","%s",%d,%.1f\n' % (k, c, len(v), sum(v) / len(v)))
    print(open(f"{out}/{tag}_pmc_hbm_all.csv").read())
