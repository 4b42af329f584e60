"""Population-size sweep on one GPU (round 3): queued training step (cude_adam_run) and forward call of the headline
instance (CPEP3, 2-6-6-1, 30 steps), library's own path selection.   python tools/size_sweep.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude.engine import Engine  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [57, 1000, 3000, 10000, 30000, 65536, 100000, 125000, 131072, 200000, 262144,
                                           500000, 1000000]
arch = (2, 6, 2)
for N in sizes:
    tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
    eng = Engine("cpep", arch, n_steps=30, n_state=3)
    eng.set_population_cpep(tp, G, cp, age, t2)
    eng.set_params(o.glorot_params(arch, 1), bt)
    eng.adam_init(1e-2)
    eng.adam_run(64)                                     # steady clock
    K = 64 if N <= 200000 else 24
    t0 = time.perf_counter()
    eng.adam_run(K)
    step = (time.perf_counter() - t0) / K
    for _ in range(20):
        eng.forward()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.forward()
    fwd = (time.perf_counter() - t0) / 50
    print(f"N={N:8d}  training step {step * 1e3:8.4f} ms  {N / step:.3e} traj/s   forward call {fwd * 1e3:8.4f} ms  "
          f"{N / fwd:.3e} traj/s", flush=True)
    eng.close()
