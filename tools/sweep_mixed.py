"""Development aid: gradient-launch time of the headline instance for forced MIXED launches (CUDE_CPEP_PATH=3:<blocks on
the one-lane kernel>:<L of the time-split remainder>) next to the library's own choice.
  python tools/sweep_mixed.py N blk0,L [blk0,L ...]        (blk0 = 0: time-split for all; L = 1 with blk0 = 0: one-lane)
  CUDE_SWEEP_ARCH=2,4,2,2 selects (inputs, width, depth, states); default 2,6,2,3"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude.engine import Engine  # noqa: E402

N = int(sys.argv[1])
_a = [int(v) for v in os.environ.get("CUDE_SWEEP_ARCH", "2,6,2,3").split(",")]
arch, n_state = tuple(_a[:3]), _a[3]
nn = o.glorot_params(arch, 1)
tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
for spec in ["auto"] + sys.argv[2:]:
    if spec == "auto":
        os.environ.pop("CUDE_CPEP_PATH", None)
    else:
        b, L = spec.split(",")
        os.environ["CUDE_CPEP_PATH"] = f"3:{b}:{L}" if int(b) > 0 else ("1" if int(L) == 1 else f"2:{L}")
    eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
    eng.set_population_cpep(tp, G, cp, age, t2)
    eng.set_params(nn, bt)
    eng.adam_init(1e-2)
    for _ in range(48):
        eng.adam_step(want_loss=False)
    eng.set_kernel_timing(True)
    for _ in range(20):
        eng.adam_step(want_loss=False)
    ms, n = eng.kernel_time_ms()
    loss = eng.adam_step()
    eng.close()
    print(f"N={N:7d} {spec:>10s} grad launch {ms:.4f} ms  {N / ms / 1e3:.4e} traj/s  loss {loss:.10f}", flush=True)
