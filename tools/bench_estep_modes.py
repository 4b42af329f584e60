"""SAEM E-step (1e4 subjects x 100 Metropolis steps, 2-4-4-1) on the fixed step grid and with the reference's adaptive solver
(src/saem.jl:52).  python tools/bench_estep_modes.py"""
import os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, ROOT)
import bench
if os.environ.get("CUDE_ABL"):                       # A/B runs: a library variant from tools/abl_so/
    from cude import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", os.environ["CUDE_ABL"] + ".so")
    _lib.STRICT = False
from cude.engine import Engine
N, n_mc, arch = 10000, 100, (2, 4, 2)
nn4 = bench.glorot(arch, 99)
eng0, pop = bench.cpep_engine(Engine, arch, 2, N, 780, 0, nn4); eng0.close()
for n_steps in (30, 0):
    eng = Engine("cpep", arch, n_steps=n_steps, n_state=2)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn4, pop["beta0"]); eng.set_rng(20250905)
    for _ in range(3):
        eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
    eng.set_params(nn4, pop["beta0"]); eng.set_rng(20250905)
    t0 = time.perf_counter()
    acc = eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
    dt = time.perf_counter() - t0
    print(f"n_steps={n_steps or 'adaptive'}: E-step {dt*1e3:.3f} ms ({dt/n_mc*1e6:.1f} us per Metropolis step), accepted {int(acc.sum())}")
    eng.close()
