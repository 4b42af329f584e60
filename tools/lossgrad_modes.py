import sys, time, os
import numpy as np, torch
sys.path[:0]=['/root/repo/conditional-ude_amd','/root/repo/oracle']
import cude_oracle as o
from cude.engine import Engine
for N in (57, 64, 65, 128, 300, 1000):
    arch=(2,4,2)
    tp,G,cp,age,t2,bt,rng=o.synthetic_cpep_population(N)
    eng=Engine("cpep",arch,n_steps=32,n_state=2)
    eng.set_population_cpep(tp,G,cp,age,t2); eng.set_params(o.glorot_params(arch,1),bt)
    res=[]
    for mode in ("grad+cond","grad","grad+cond","grad","forward"):
        f = (lambda: eng.loss_grad()) if mode=="grad+cond" else ((lambda: eng.loss_grad(want_cond_grad=False)) if mode=="grad" else (lambda: eng.forward()))
        for _ in range(100): f()
        t=time.perf_counter()
        for _ in range(300): f()
        res.append(f"{mode} {(time.perf_counter()-t)/300*1e6:6.1f}")
    print(N, " | ".join(res), flush=True)
    eng.close()
