"""SAEM E-step with speculative Metropolis steps (option "mh_spec", csrc/cude_kernels.h MhSpecArgs): time per E-step of
n_mc steps with device-side draws, for population sizes from one GPU's share of BASELINE configs[4] on 8 GPUs (1 250
subjects) to the whole of it (1e4), depth 0 (two launches per step) against 2, 3, 4 (one forward + scan + resolver per
2, 3, 4 steps).  Every depth must leave the same chain: the state checksum and the acceptance count are printed.
    python tools/bench_estep_spec.py [n_mc] [N ...]          (CUDE_SPEC_DEPTHS=0,3 restricts the depths)"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cude.engine import Engine  # noqa: E402

n_mc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sizes = [int(v) for v in sys.argv[2:]] or [625, 1250, 2500, 5000, 10000]
arch = (2, 4, 2)
nn4 = bench.glorot(arch, 99)
for N in sizes:
    base = None
    for depth in [int(v) for v in os.environ.get("CUDE_SPEC_DEPTHS", "0,2,3,4").split(",")]:
        eng, pop = bench.cpep_engine(Engine, arch, 2, N, 780, 0, nn4)
        eng.set_option("mh_spec", depth)
        eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
        eng.set_params(nn4, pop["beta0"])
        eng.set_rng(20250905)
        for _ in range(3):
            eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
        best = 1e9
        for _ in range(5):
            eng.set_params(nn4, pop["beta0"])
            eng.set_rng(20250905)
            t0 = time.perf_counter()
            acc = eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
            best = min(best, time.perf_counter() - t0)
        _, cond = eng.get_params()
        eng.close()
        key = (int(acc.sum()), float(cond.sum()))
        base = base or (best, key)
        print(f"N={N:6d} n_mc={n_mc} depth={depth}: E-step {best * 1e3:8.3f} ms  {best / n_mc * 1e6:6.1f} us per step  "
              f"x{base[0] / best:4.2f}  {N * n_mc / best:.3e} draws/s  accepted {key[0]}  same chain: {key == base[1]}", flush=True)
