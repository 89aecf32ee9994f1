#!/usr/bin/env python3
"""Workgroup size of the generic QP kernel on the multi-body BASELINE configurations (GPU box).
usage: python tools/sweep_generic_nt.py [B3 [B4]]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

B3 = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B4 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
for nt in (64, 128, 256, 512, 1024):
    os.environ["UPR_QP_GENERIC_NT"] = str(nt)
    for name, w in (("config3", bench.config3_workload(B3)), ("config4", bench.config4_workload(B4))):
        r = bench.time_extra(w, 2, 1)
        print(f"NT {nt:5d} {name} B={len(w['x0'])}: {r['ms_per_step']:9.2f} ms/step  qp {r['kernel_ms']['qp']:9.2f} ms/launch  {r['value']:9.0f} solves/s  "
              f"iters {r['qp_iters_mean']:.1f} conv {r['qp_converged_fraction']:.2f}  {r['roofline']['kernel']}", flush=True)
