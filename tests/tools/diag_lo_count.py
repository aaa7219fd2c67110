#!/usr/bin/env python3
"""Build container only (needs the reference shim).  Which iteration makes the reference's LO count differ from ours on
a calibrated case of tests/golden/estimate_full.npz?  Replays score_models<> over the minimal models of both sides'
solvers (reference binary via refshim, ours via the oracle), scoring with the reference's own scorer.
    python tests/tools/diag_lo_count.py CASE_INDEX"""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "..", ".."))
import numpy as np, refshim as rs
from oracle import pyorc as po
g=np.load(os.path.join(HERE, "..", "golden", "estimate_full.npz"))
ci=int(sys.argv[1])
case=g['cases'][ci]; kind,es,n=int(case[1]),int(case[2]),int(case[3])
x1,x2,d1,d2=g[f'x1_{ci}'],g[f'x2_{ci}'],g[f'd1_{ci}'],g[f'd2_{ci}']
f=800.0
a=x1/f; b=x2/f
thr=(2.0/f)**2
S=po.draw_samples(0,n,10000)
def run(side):
    bc,bs=0,1e300; trig=[]
    for it,s in enumerate(S):
        x1h=np.c_[a[s],np.ones(3)]; x2h=np.c_[b[s],np.ones(3)]
        if es:
            sols = rs.solver_calib(x1h,x2h,d1[s],d2[s]) if side=='ref' else po.solver_calib_shift(x1h,x2h,d1[s],d2[s])
        else:
            if side=='ref':
                X=x1h*d1[s][:,None]; xb=x2h/np.linalg.norm(x2h,axis=1,keepdims=True)
                sols=rs.p3p(xb,X)
            else:
                sols=po.solver_calib_p3p(x1h,x2h,d1[s],d2[s])
        hit=False
        for m in sols:
            m7=np.asarray(m[:7])
            sc,c=rs.msac_pose(m7,a,b,thr)
            if c>bc or sc<bs:
                if c>bc: bc=c
                if sc<bs: bs=sc
                hit=True
        if hit: trig.append((it,bc,bs,len(sols)))
    return trig
r=run('ref'); o=run('orc')
print(len(r),len(o))
ro=set(t[0] for t in r); oo=set(t[0] for t in o)
print('only ref',sorted(ro-oo),'only ours',sorted(oo-ro))
for t in r:
    if t[0] in ro-oo: print('ref',t)
for t in o:
    if t[0] in oo-ro: print('orc',t)
