"""Development aid: gradient-launch time of the headline instance (CPEP3, 2x6x6x1, S = 30) over population sizes and
time-split factors L (CUDE_CPEP_PATH is read at set_population).  Prints one line per (N, L) and the best L per N.

  python tools/sweep_chunks.py [N ...]
"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude.engine import Engine  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [32768, 50000, 65536, 80000, 100000, 115000, 125000, 160000, 200000]
arch = (2, 6, 2)
nn = o.glorot_params(arch, 1)
for N in sizes:
    tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
    res = {}
    for L in (0, 1, 2, 3, 5, 6, 10):
        if L == 0:
            os.environ.pop("CUDE_CPEP_PATH", None)          # the library's own selector
        elif L == 1:
            os.environ["CUDE_CPEP_PATH"] = "1"
        else:
            os.environ["CUDE_CPEP_PATH"] = f"2:{L}"
        eng = Engine("cpep", arch, n_steps=30, n_state=3)
        eng.set_population_cpep(tp, G, cp, age, t2)
        eng.set_params(nn, bt)
        eng.adam_init(1e-2)
        for _ in range(48):                          # reach the steady GPU clock (profiles/r02/clock_ramp.txt)
            eng.adam_step(want_loss=False)
        eng.set_kernel_timing(True)
        for _ in range(20):
            eng.adam_step(want_loss=False)
        ms, n = eng.kernel_time_ms()
        loss = eng.adam_step()
        eng.close()
        res[L] = ms
        print(f"N={N:7d} L={L:2d} grad launch {ms:.4f} ms  {N / ms / 1e3:.4e} traj/s  loss {loss:.10f}", flush=True)
    best = min((L for L in res if L), key=lambda L: res[L])
    print(f"N={N:7d} best L={best} ({res[best]:.4f} ms); selector gives {res[0]:.4f} ms", flush=True)
