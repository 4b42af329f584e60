"""Forward-only ensemble (BASELINE configs[1]) at a given population size: per-call time; run under rocprofv3
--kernel-trace --stats to see the kernels of the chosen path.   python tools/bench_fwd.py [N] [calls]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
if os.environ.get("CUDE_ABL"):                       # A/B runs: a library variant from tools/abl_so/
    from cude import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", os.environ["CUDE_ABL"] + ".so")
    _lib.STRICT = False
from cude.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng, pop = bench.cpep_engine(Engine, bench.ARCH, bench.N_STATE, N, 777, 0, bench.glorot(bench.ARCH, 1234))
eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
eng.set_params(bench.glorot(bench.ARCH, 1234), pop["beta0"])
for _ in range(100):
    eng.forward()
eng.set_kernel_timing(True)
t0 = time.perf_counter()
for _ in range(calls):
    out = eng.forward()
dt = (time.perf_counter() - t0) / calls
ms, n = eng.kernel_time_ms()
print(f"N={N} forward call {dt * 1e3:.4f} ms, launch (HIP events) {ms:.4f} ms over {n} launches, loss {out['loss']:.10f}")
eng.close()
