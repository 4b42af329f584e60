import sys, os, time
import numpy as np, torch
sys.path[:0]=['/root/repo/conditional-ude_amd','/root/repo']
import bench
from cude.engine import Engine
arch=(2,4,2); nn=bench.glorot(arch,1234)
N=int(sys.argv[1]); team=int(sys.argv[2])
eng0,pop=bench.cpep_engine(Engine,arch,2,N,777,0,nn); eng0.close()
eng=Engine("cpep",arch,n_steps=0,n_state=2); eng.set_option("adaptive_team",team)
eng.set_population_cpep(pop["tp"],pop["G"],pop["obs"],pop["age"],pop["t2dm"]); eng.set_params(nn,pop["beta0"])
for _ in range(50): eng.forward(); eng.loss_grad(want_cond_grad=False)
for _ in range(200): eng.forward()
for _ in range(200): eng.loss_grad(want_cond_grad=False)
cnt=[len(eng.adaptive_steps(i)[0]) for i in range(N)]
print("accepted steps", min(cnt), int(np.median(cnt)), max(cnt))
eng.close()
