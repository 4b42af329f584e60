"""SAEM E-step in the API mirrors' DEFAULT discretisation (the reference's adaptive solve) with and without speculative
Metropolis steps (option "mh_spec": the candidates of d steps as parameter sets of one adaptive launch + a resolver
launch), device-side draws, 100 steps, best of 3.   python tools/bench_estep_adaptive.py [N ...]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cude.engine import Engine  # noqa: E402

arch = (2, 4, 2)
nn = bench.glorot(arch, 99)
for N in [int(v) for v in sys.argv[1:]] or [57, 117, 1000, 5000, 10000]:
    base = None
    for steps, tag in ((0, "adaptive"), (30, "one-lane fixed")):
        for depth in [int(v) for v in os.environ.get("DEPTHS", "0,2,3,-1").split(",")]:
            eng, pop = bench.cpep_engine(Engine, arch, 2, N, 780, 0, nn)
            eng.close()
            eng = Engine("cpep", arch, n_steps=steps, n_state=2)
            if steps: eng.set_option("cpep_path", "1")
            eng.set_option("mh_spec", depth)
            eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
            eng.set_params(nn, pop["beta0"])
            eng.set_rng(20250905)
            best = 1e9
            for rep in range(4):
                t0 = time.perf_counter()
                acc = eng.mh_estep(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=100)
                dt = time.perf_counter() - t0
                if rep: best = min(best, dt)
            eng.set_params(nn, pop["beta0"]); eng.set_rng(20250905)
            acc = eng.mh_estep(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=100)
            _, state = eng.get_params()
            key = (int(acc.sum()), float(state.sum()))
            if depth == 0: base = key
            print(f"N={N:6d} {tag:15s} depth={depth:2d}: E-step {best * 1e3:8.3f} ms  {best * 1e4:7.1f} us per step  accepted {key[0]}  "
                  f"same chain: {key == base}", flush=True)
            eng.close()
