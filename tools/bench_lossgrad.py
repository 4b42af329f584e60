"""Host-visible time of one loss + gradient call (what a host-driven L-BFGS pays per evaluation) at small populations.
python tools/bench_lossgrad.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
if os.environ.get("CUDE_ABL"):                       # A/B runs: a library variant from tools/abl_so/
    from cude import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", os.environ["CUDE_ABL"] + ".so")
    _lib.STRICT = False
from cude.engine import Engine  # noqa: E402

for N in [int(v) for v in sys.argv[1:]] or [57, 1000, 10000]:
    arch = (2, 4, 2)
    tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
    eng = Engine("cpep", arch, n_steps=32, n_state=2)
    eng.set_population_cpep(tp, G, cp, age, t2)
    eng.set_params(o.glorot_params(arch, 1), bt)
    for _ in range(100):
        eng.loss_grad()
    t0 = time.perf_counter()
    for _ in range(300):
        out = eng.loss_grad()
    dt = (time.perf_counter() - t0) / 300
    t0 = time.perf_counter()
    for _ in range(300):
        eng.loss_grad(want_cond_grad=False)
    dt2 = (time.perf_counter() - t0) / 300
    print(f"N={N:6d} loss_grad call {dt * 1e6:7.1f} us   without the conditional gradient copy {dt2 * 1e6:7.1f} us   loss {out[0]:.10f}")
    eng.close()
