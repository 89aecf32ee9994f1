"""Phase stamps of the line-search kernel (library built with -DUPR_LS_PROF, loaded through UPR_LIB): cycles of
[staging + baseline pass + first reduction | second reduction | trial evaluations | (statistics)] per instance, in the QP-residual slots.
Build:  cd upright_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUPR_LS_PROF -c upr_api.hip -o /tmp/api_lsprof.o
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../libupright_mi_lsprof.so /tmp/api_lsprof.o build/upr_qp3_part*.o -lhiprtc -ldl
Run:    gpurun -- 'UPR_LIB=libupright_mi_lsprof.so python3 tools/dbg_ls_phases.py headline'
Round 4 (headline, B = 1024): 46 k cycles staging + baseline pass, 43 k the one trial evaluation (joint sines / cosines + chain walk, a lane
per knot), 4 k the rest, of the 141 k cycles (0.059 ms) the kernel lasts; the remainder is its prologue (dispatch-order ranking, problem record to LDS)."""
import sys
sys.path.insert(0, '.')
import numpy as np, bench
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
w = {"headline": lambda: bench.headline_workload(1024), "config5": lambda: bench.config5_workload(1024)}[name]()
mpc = bench.make_engine(w)
if name == "config5": mpc.set_projectile_flag(1.0)
mpc.advance()
st = mpc.stats()
print({k: float(np.mean(st[k])) for k in ("qp_res_stat", "qp_res_eq", "qp_res_ineq", "qp_res_comp")} if "qp_res_stat" in st else list(st.keys()))
print("alpha mean", float(np.mean(st["alpha"])) if "alpha" in st else None)
