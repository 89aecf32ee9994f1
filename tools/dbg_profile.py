import sys, json, os
sys.path.insert(0,'.')
import numpy as np
from upright_amd.engine import BatchMPC
from upright_amd.problem import thing_problem
from upright_amd.sampling import level_tray_states, waypoints_for
arr=json.load(open('tests/golden/arrangements.json'))
P=thing_problem(arr['pink_bottle'])
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
x0=level_tray_states(B,seed=0); way=waypoints_for(P,x0)
names=["residuals","prep A: box rows","prep B: contacts","prep C: eq residual, S","prep D: Schur factor","prep E: C' zt","mat: phase 1","mat: aug. Cholesky","mat: V, K store","mat: P update","vec: sweep","vec/mat: flat parts","fwd: sweep","fwd: tail + costates","aff sweeps","update + init"]
for nt in sys.argv[2:] or ["256"]:
    os.environ["UPR_QP_NT"]=nt
    mpc=BatchMPC(P,B,way_p=way); mpc.set_observation(0.0,x0)
    mpc.advance()
    mpc.qp_profile()   # arm
    mpc.reset(); mpc.advance()
    prof=mpc.qp_profile(); st=mpc.stats()
    its=st["qp_iters_last"]
    per=prof[:, :16]/its[:,None]
    print("NT",nt,"solve ms",mpc.last_solve_ms(),"mean iters",its.mean())
    for i,n in enumerate(names): print("   %-14s %10.0f cycles/iter  (%.1f%%)"%(n,per[:,i].mean(),100*per[:,i].mean()/per.sum(1).mean()))
    print("   total cycles/iter",per.sum(1).mean())
    mpc.close()
