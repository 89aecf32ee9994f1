import sys, json, os
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.oracle import Oracle
from upright_amd.engine import BatchMPC
from upright_amd.problem import thing_problem
from upright_amd.sampling import level_tray_states, stationary_guess, waypoints_for
arr=json.load(open('/root/repo/tests/golden/arrangements.json'))
B=4
for its in (1,2,3,5,8,10):
    P=thing_problem(arr['pink_bottle'],qp_tol=0.0,qp_iter_max=its)
    x0=level_tray_states(B,seed=11); way=waypoints_for(P,x0); xs0,us0=stationary_guess(x0,P.N,P.nu)
    ref=[]
    for b in range(B):
        P.way_p=way[b]; ref.append(Oracle(P).qp_step(0.0,x0[b],xs0[b],us0[b]))
    for gen,nt in (("1","64"),("0","64"),("0","128"),("0","256")):
        os.environ["UPR_QP_GENERIC"]=gen; os.environ["UPR_QP_NT"]=nt
        errs=[]
        for rep in range(2):
            mpc=BatchMPC(P,B,way_p=way); mpc.set_observation(0.0,x0); mpc.set_guess(xs0,us0)
            dxs,dus=mpc.qp_step(); st=mpc.stats(); mpc.close()
            errs.append(max(np.abs(dxs[b]-ref[b][0]).max() for b in range(B)))
            errs.append(max(np.abs(dus[b]-ref[b][1]).max() for b in range(B)))
        print("its",its,"generic" if gen=="1" else "v2","nt",nt,"dx/du err (2 reps)",["%.2e"%e for e in errs],"mu",st["qp_res_comp"][0],"ref mu",ref[0][2].qp_res[3])
