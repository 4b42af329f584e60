"""SAEM E-step (BASELINE configs[4] shape) at a given population size: time per E-step of n_mc Metropolis steps with
device-side draws.   python tools/bench_estep.py [N] [n_mc]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
if os.environ.get("CUDE_ABL"):                       # A/B runs: a library variant from tools/abl_so/
    from cude import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", os.environ["CUDE_ABL"] + ".so")
    _lib.STRICT = False
from cude.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_mc = int(sys.argv[2]) if len(sys.argv) > 2 else 100
arch = (2, 4, 2)
nn4 = bench.glorot(arch, 99)
eng, pop = bench.cpep_engine(Engine, arch, 2, N, 780, 0, nn4)
eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
eng.set_params(nn4, pop["beta0"])
eng.set_rng(20250905)
for _ in range(3):
    eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
eng.set_params(nn4, pop["beta0"])
eng.set_rng(20250905)
t0 = time.perf_counter()
acc = eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
dt = time.perf_counter() - t0
_, cond = eng.get_params()
print(f"N={N} n_mc={n_mc} fuse={'off' if os.environ.get('CUDE_NO_MH_FUSE') else 'on'}: E-step {dt * 1e3:.3f} ms "
      f"({dt / n_mc * 1e6:.1f} us per Metropolis step), accepted {int(acc.sum())}, state checksum {cond.sum():.12f}")
eng.close()
