"""Development aid: does the gradient launch get faster as the GPU warms up?  Prints the per-launch kernel time of
consecutive optimiser steps (headline instance) in groups of 10."""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
arch = (2, 6, 2)
tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
nn = o.glorot_params(arch, 1)
eng = Engine("cpep", arch, n_steps=30, n_state=3)
eng.set_population_cpep(tp, G, cp, age, t2)
eng.set_params(nn, bt)
eng.adam_init(1e-2)
eng.set_kernel_timing(True)
t0 = time.perf_counter()
for grp in range(40):
    for _ in range(10):
        eng.adam_step(want_loss=False)
    ms, n = eng.kernel_time_ms()
    print(f"t={time.perf_counter() - t0:7.3f}s steps {grp * 10:4d}-{grp * 10 + 9:4d}: kernel {ms:.4f} ms", flush=True)
time.sleep(2.0)
for grp in range(5):
    for _ in range(10):
        eng.adam_step(want_loss=False)
    ms, n = eng.kernel_time_ms()
    print(f"after 2 s idle: kernel {ms:.4f} ms", flush=True)
