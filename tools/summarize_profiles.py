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
