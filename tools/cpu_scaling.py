"""Thread scaling of the CPU baseline (oracle/cude_oracle_rev.c: per-subject reverse mode, OpenMP static schedule) on
the bench workload.  usage: python tools/cpu_scaling.py [n_subjects=32000] [threads ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import c_oracle as co  # noqa: E402
import cude_oracle as o  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
threads = [int(v) for v in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64, 128]
tp, G, cp, age, t2, beta, rng = o.synthetic_cpep_population(n, 20250905)
nn = o.glorot_params((2, 6, 2), 1234)
print(f"CPU baseline (reverse-mode port), CPEP3 2x6x6x1, 30 steps, {n} subjects; host threads available: {os.cpu_count()}")
base = None
for t in threads:
    if t > (os.cpu_count() or 1):
        continue
    m = max(256, n * t // max(threads)) if t < 8 else n          # keep the single-thread runs short
    args = (tp, G[:m], cp[:m], age[:m], t2[:m], (2, 6, 2), nn, beta[:m], 30, 3)
    co.cpep(*args, method="reverse", nthreads=t)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        co.cpep(*args, method="reverse", nthreads=t)
        reps += 1
    rate = m * reps / (time.perf_counter() - t0)
    base = base or rate / t
    print(f"  threads {t:4d}: {rate:10.3e} subject-trajectories/s   ({m} subjects per call)   "
          f"parallel efficiency {rate / (t * base):.2f}")
