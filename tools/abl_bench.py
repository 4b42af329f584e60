"""Times library variants (tools/abl_so/<name>.so) on the bench workload (development aid for A/B runs in one
gpurun call; build variants with extra -D flags into tools/abl_so/)."""
import os
import sys

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
nn = o.glorot_params(arch, 1)
eng = Engine("cpep", arch, n_steps=30, n_state=3)
eng.set_population_cpep(tp, G, cp, age, t2)
eng.set_params(nn, bt)
eng.adam_init(1e-2)
for _ in range(64):                      # reach the steady GPU clock first (profiles/r02/clock_ramp.txt)
    eng.adam_step(want_loss=False)
eng.set_kernel_timing(True)
for _ in range(30):
    eng.adam_step(want_loss=False)
ms, n = eng.kernel_time_ms()
import time  # noqa: E402
eng.forward()
t0 = time.perf_counter()
for _ in range(20):
    eng.forward()
fwd_ms = (time.perf_counter() - t0) / 20 * 1e3
print(f"{variant:10s} arch={arch} N={N} grad kernel {ms:.4f} ms  forward call {fwd_ms:.4f} ms  loss {eng.adam_step():.12f}")
