"""Where a small scan launch spends its time: a library built with -DCUDE_SCAN_TIMING (tools/build_variant.sh scan_timing
-DCUDE_SCAN_TIMING, ONLY=cude_cpep2) prints the wall-clock distances between the scan's phases for workgroup 0.
python tools/scan_timing.py [N]"""
import os, sys
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", "scan_timing.so")
_lib.STRICT = False
from cude.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 57
arch = (2, 4, 2)
tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
eng = Engine("cpep", arch, n_steps=32, n_state=2)
eng.set_population_cpep(tp, G, cp, age, t2)
eng.set_params(o.glorot_params(arch, 1), bt)
for _ in range(6):
    eng.loss_grad()
for _ in range(4):
    eng.forward()
eng.close()
