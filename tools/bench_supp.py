"""Timing of the suppression-model (3-state nonlinear cUDE, 4->3x5->1 MLP, T=8, S=30) ensemble kernels
(development aid; numbers quoted in DESIGN.md)."""
import sys, os, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import make_supp_case
from cude.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
c = make_supp_case(N)
eng = Engine("supp", c["arch"], n_steps=30, lam=0.01)
eng.set_population_supp(c["tp"], c["data"])
eng.set_params(c["nn"], c["theta"])
eng.adam_init(1e-3)
for _ in range(40): eng.adam_step()
eng.set_kernel_timing(True)
K = 10
eng.synchronize(); t = time.perf_counter()
for _ in range(K): eng.adam_step()
eng.synchronize(); dt = (time.perf_counter() - t) / K
ms, n = eng.kernel_time_ms()
print(f"SUPP N={N} adam step {dt*1e3:.3f} ms kernel {ms:.3f} ms  traj/s {N/dt:.3e}")
t = time.perf_counter()
for _ in range(K): eng.forward()
dt = (time.perf_counter() - t) / K
print(f"SUPP forward-only {dt*1e3:.3f} ms traj/s {N/dt:.3e}")
if "--no-cpu" in sys.argv:
    sys.exit(0)
import c_oracle as co
n = min(N, 20000)
t = time.perf_counter(); co.supp(c["tp"], c["data"][:, :, :n], c["arch"], c["nn"], c["theta"][:n], 0.01, 30); dt = time.perf_counter() - t
print(f"CPU oracle (forward duals, {co.num_threads()} threads): {n/dt:.3e} traj/s")
