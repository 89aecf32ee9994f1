"""tools/dbg_profile.py's table at an interior-point iteration cap of 1 and 2 (headline): cap 1 = set-up + first residuals + ONE iteration + the
residuals behind it; the difference to cap 2 is one more iteration, so what cap 1 has on top of that is set-up (the feedback gains are
written after the counters are read and are not in the table)."""
import sys, json, os
os.environ["UPR_QP3_JIT"] = "2"; os.environ["UPR_JIT_FLAGS"] = (os.environ.get("UPR_JIT_FLAGS", "") + " -DUPR_QP3_PROF").strip()   # the stamps exist in run-time instantiations only
sys.path.insert(0, '.')
import numpy as np
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
names=["residuals","prep A","prep B","prep C","prep D","prep E","mat 1","mat 2","mat 3","mat 4","vec sweep","vec/mat flat","fwd sweep","fwd tail","aff sweeps","update + init"]
tabs = {}
for cap in (1, 2):
    w = bench.headline_workload(B); w["P"].qp_iter_max = cap
    mpc = bench.make_engine(w)
    mpc.advance(); mpc.qp_profile(); mpc.reset(); mpc.advance()
    prof = mpc.qp_profile()
    tabs[cap] = prof.mean(0)[0]          # wave 0, cycles per instance
print("%-16s %10s %10s %10s" % ("cycles/instance", "cap 1", "cap 2", "cap1-(cap2-cap1)"))
for i, n in enumerate(names):
    print("%-16s %10.0f %10.0f %10.0f" % (n, tabs[1][i], tabs[2][i], 2 * tabs[1][i] - tabs[2][i]))
print("%-16s %10.0f %10.0f %10.0f" % ("total", tabs[1].sum(), tabs[2].sum(), 2 * tabs[1].sum() - tabs[2].sum()))
