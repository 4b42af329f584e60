"""Queued Adam iteration and forward call of the time-split path for forced chunk counts ("cpep_path" = "2:L") against the
selector's choice, small populations.   python tools/sweep_chunks_small.py [N ...]   (ARCH=2,4,2 STEPS=30 NSTATE=2)"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude.engine import Engine  # noqa: E402

ARCH = tuple(int(v) for v in os.environ.get("ARCH", "2,4,2").split(","))
STEPS = int(os.environ.get("STEPS", "30"))
NSTATE = int(os.environ.get("NSTATE", "2"))
for N in [int(v) for v in sys.argv[1:]] or [57, 1000, 4000, 10000, 20000]:
    tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
    nn = o.glorot_params(ARCH, 1)
    for L in [0] + [d for d in range(2, STEPS + 1) if STEPS % d == 0]:
        eng = Engine("cpep", ARCH, n_steps=STEPS, n_state=NSTATE)
        if L: eng.set_option("cpep_path", f"2:{L}")
        else: eng.set_option("debug_selector", 1)
        eng.set_population_cpep(tp, G, cp, age, t2)
        eng.set_params(nn, bt)
        for _ in range(100): eng.forward()
        bf = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200): eng.forward()
            bf = min(bf, (time.perf_counter() - t0) / 200)
        eng.adam_init(1e-3); eng.adam_run(64)
        bq = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); eng.adam_run(256); bq = min(bq, (time.perf_counter() - t0) / 256)
        print(f"N={N:6d} L={'selector' if L == 0 else L:>8}: forward call {bf * 1e6:7.1f} us   queued Adam iteration {bq * 1e6:7.1f} us", flush=True)
        eng.close()
