import sys,time,os; sys.path.insert(0,'.')
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, synth
from oracle import st_oracle as orc
pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
p=_capi.Params.from_settings(pkg.Settings); op=orc.OrcParams.from_dict(p.as_dict())
ego,k,ox,ov=synth.generate_states(4096,k=6,kmax=8,seed=1000)
for solver in ("layered","heap"):
    for nt in (1,8,32,64,128,256):
        n=min(4096,32*nt)
        t=time.perf_counter(); r=orc.solve_batch(op,ego[:n],k[:n],ox[:n],ov[:n],solver=solver,nthreads=nt); dt=time.perf_counter()-t
        print(solver,nt,"threads: %.1f solves/s  (%.1f per thread)"%(n/dt,n/dt/nt),flush=True)
