import sys, time, numpy as np, torch
sys.path[:0]=['/root/repo/conditional-ude_amd','/root/repo/tests','/root/repo/oracle','/root/repo']
from cude.engine import Engine
import cude_oracle as o, bench
for arch, tag in [((4,(10,20,30),("tanh","relu","softplus"),"softplus"),"doc example 4-10-20-30-1 (P=931)"), ((4,(3,)*5,("tanh",)*5,"softplus"),"4-3x5-1 forced onto the fallback kernel (P=67)")]:
    for N in (37, 1000, 100000):
        tp, data, theta = bench.synthetic_suppression(N, 779)
        eng = Engine("supp", (4,3,5), n_steps=30, lam=0.01)
        eng.set_option("force_fallback", 1)
        eng.set_network(list(arch[1]), list(arch[2])+[arch[3]])
        eng.set_population_supp(tp, data); eng.set_params(o.glorot_params(arch, 1), theta)
        eng.loss_grad(); eng.forward()
        t=time.perf_counter(); k = 3 if N>=100000 else 10
        for _ in range(k): eng.loss_grad(want_cond_grad=False)
        dg=(time.perf_counter()-t)/k
        t=time.perf_counter()
        for _ in range(k): eng.forward()
        df=(time.perf_counter()-t)/k
        print(f"{tag}: N={N:6d} fixed 30 steps: gradient {dg*1e3:9.3f} ms, forward {df*1e3:8.3f} ms", flush=True)
        eng.close()
