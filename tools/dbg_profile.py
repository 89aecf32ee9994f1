import sys, json, os
os.environ["UPR_QP3_JIT"] = "2"; os.environ["UPR_JIT_FLAGS"] = (os.environ.get("UPR_JIT_FLAGS", "") + " -DUPR_QP3_PROF").strip()   # the stamps exist in run-time instantiations only
sys.path.insert(0,'.')
import numpy as np
from upright_amd.engine import BatchMPC
from upright_amd.problem import thing_problem
from upright_amd.sampling import level_tray_states, waypoints_for
arr=json.load(open('tests/golden/arrangements.json'))
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
# workload: "config3" / "config4" among the arguments selects bench.py's extra workloads, default the headline
import bench
bp=None
if "config3" in sys.argv: w=bench.config3_workload(B); P,x0,way=w["P"],w["x0"],w["way"]
elif "config4" in sys.argv:
    w=bench.config4_workload(B); P,x0,way,bp=w["P"],w["x0"],w["way"],w["body_params"]; P.sqp_iters=1
    if "N100" in sys.argv: P.N=100   # (the reference's own horizon: the far-array form of the kernel, upr_qp3_cfg::KFAR)
elif "config5s" in sys.argv: w=bench.config5_workload(B, slacks=True); P,x0,way=w["P"],w["x0"],w["way"]
elif "config5" in sys.argv: w=bench.config5_workload(B); P,x0,way=w["P"],w["x0"],w["way"]
else:
    NH=[int(a[2:]) for a in sys.argv if a.startswith("NH")]   # NH<n>: the headline shape at horizon n (a target the short horizon reaches)
    P=thing_problem(arr['pink_bottle'], **({"N": NH[0]} if NH else {}))
    x0=level_tray_states(B,seed=0); way=waypoints_for(P,x0, **({"offset": (-0.004*NH[0]**2, 0.002*NH[0]**2, 0.0)} if NH else {}))
names=["residuals","prep A: box rows","prep B: contacts","prep C: eq residual, S","prep D: Schur factor","prep E: C' zt","mat: phase 1","mat: aug. Cholesky","mat: V, K store","mat: P update","vec: sweep","vec/mat: flat parts","fwd: sweep","fwd: tail + costates","aff sweeps","update + init"]
MAT = "mat" in sys.argv[2:]
if MAT:
    names=["ph1 work","ph1 wait","ph2: after last pivot -> arrival (stores | mfma preload, feedback)","ph2 wait","ph3 work (P update)","ph3 wait","wave 0: operand loads (hjj, hux)","wave 0: pivots 7, 8","wave 0: pivot 0","wave 0: pivot 1","wave 0: pivot 2","wave 0: pivot 3","wave 0: pivot 4","wave 0: pivot 5","wave 0: pivot 6","outside the matrix sweep"]
for nt in [a for a in sys.argv[2:] if a.isdigit()] or ["256"]:
    os.environ["UPR_QP_NT"]=nt
    mpc=BatchMPC(P,B,way_p=way,body_params=bp); mpc.set_observation(0.0,x0)
    if P.n_dyn: mpc.set_projectile_flag(1.0)
    mpc.advance()
    mpc.qp_profile()   # arm
    mpc.reset(); mpc.advance()
    prof=mpc.qp_profile(); st=mpc.stats()
    its=st["qp_iters_last"]
    per=prof/its[:,None,None]            # [B][wave][phase], cycles per IPM iteration as seen by lane 0 of each wave
    m=per.mean(0)
    print("NT",nt,"solve ms",mpc.last_solve_ms(),"mean iters",its.mean())
    print("   (a phase ends at a barrier: the wave with the LARGEST share arrived last; every counter read costs ~290 cycles)")
    print("   %-26s %10s %10s %10s %10s"%("cycles / IPM iteration","wave 0","wave 1","wave 2","wave 3"))
    for i,n in enumerate(names):
        if n != "-": print("   %-68s %10.0f %10.0f %10.0f %10.0f"%(n,m[0,i],m[1,i],m[2,i],m[3,i]))
    if MAT: print("   per knot (19 knots): divide by 19; wait ~ 0 marks the wave the barrier waits for")
    print("   %-26s %10.0f %10.0f %10.0f %10.0f"%("total",*m.sum(1)))
    mpc.close()
