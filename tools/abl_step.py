"""Queued optimiser iterations (cude_adam_run: one captured hipGraph replayed) of a library variant, for step anatomy under
`rocprofv3 --kernel-trace` (tools/step_gaps.py).  usage: abl_step.py variant [N] [arch]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude import _lib  # noqa: E402

variant = sys.argv[1]
_lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", variant + ".so")
_lib.STRICT = False
from cude.engine import Engine  # noqa: E402

N = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
arch = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2, 6, 2)
tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
eng = Engine("cpep", arch, n_steps=30, n_state=3)
eng.set_population_cpep(tp, G, cp, age, t2)
eng.set_params(o.glorot_params(arch, 1), bt)
eng.adam_init(1e-2)
eng.adam_run(64)
t0 = time.perf_counter()
losses = eng.adam_run(64)
dt = (time.perf_counter() - t0) / 64 * 1e3
print(f"{variant:8s} N={N} {dt:.4f} ms per queued step  loss {losses[-1]:.12f}")
