"""A/B timing of the suppression-model gradient launch for library variants (tools/abl_so/<name>.so)."""
import os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, ROOT)
from cude import _lib
variant = sys.argv[1]
_lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", variant + ".so")
_lib.STRICT = False
from cude.engine import Engine
import bench
for N in [int(v) for v in sys.argv[2:]] or [100000]:
    tp, data, theta = bench.synthetic_suppression(N, 779)
    eng = Engine("supp", (4, 3, 5), n_steps=30, lam=0.01)
    eng.set_population_supp(tp, data); eng.set_params(bench.glorot((4, 3, 5), 1234), theta); eng.adam_init(1e-3)
    for _ in range(30): eng.adam_step(want_loss=False)
    eng.set_kernel_timing(True)
    for _ in range(20): eng.adam_step(want_loss=False)
    ms, n = eng.kernel_time_ms(); loss = eng.adam_step()
    print(f"{variant:10s} CUDE_SUPP_CKPT={os.environ.get('CUDE_SUPP_CKPT','-'):6s} N={N:7d} grad launch {ms:.4f} ms  loss {loss:.10f}", flush=True)
    eng.close()
