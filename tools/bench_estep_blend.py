"""SAEM E-step in the stochastic-approximation phase (gamma < 1): one solve launch per Metropolis step (the proposal and both
possible next states as three parameter sets; option "mh_pair") against the two-launch form, fixed-step (time-split) and
adaptive (the mirrors' default), device-side draws, 100 steps, best of 3.   python tools/bench_estep_blend.py [N ...]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cude.engine import Engine  # noqa: E402

arch = (2, 4, 2)
nn = bench.glorot(arch, 99)
for N in [int(v) for v in sys.argv[1:]] or [57, 1000, 10000]:
    eng, pop = bench.cpep_engine(Engine, arch, 2, N, 780, 0, nn)
    eng.close()
    for steps, tag in ((30, "fixed 30 steps"), (0, "adaptive")):
        base = None
        for pair, depth in ((0, 0), (1, 0), (1, 2), (1, 3), (1, -1)):
            eng = Engine("cpep", arch, n_steps=steps, n_state=2)
            eng.set_option("mh_pair", pair)
            eng.set_option("mh_spec", depth)
            eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
            best = 1e9
            for rep in range(4):
                eng.set_params(nn, pop["beta0"]); eng.set_rng(20250905)
                t0 = time.perf_counter()
                acc = eng.mh_estep(None, None, 0.4, -0.6, 0.9, 0.3, 1.0, 0.25, n_mc=100)
                dt = time.perf_counter() - t0
                if rep: best = min(best, dt)
            _, state = eng.get_params()
            key = (int(acc.sum()), float(state.sum()))
            if pair == 0: base = key
            print(f"N={N:6d} {tag:15s} gamma=0.25 mh_pair={pair} mh_spec={depth:2d}: E-step {best * 1e3:8.3f} ms  {best * 1e4:7.1f} us per step  "
                  f"accepted {key[0]}  same chain: {key == base}", flush=True)
            eng.close()
