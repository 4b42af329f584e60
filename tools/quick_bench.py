"""Quick single-GPU timing of the fwd+adjoint kernel (development aid; bench.py is the contract)."""
import sys, os, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from cude.engine import Engine
import cude_oracle as o

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
arch = (2, 6, 2)
tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
nn = o.glorot_params(arch, 1)
eng = Engine("cpep", arch, n_steps=30, n_state=3)
eng.set_population_cpep(tp, G, cp, age, t2)
eng.set_params(nn, bt)
f = eng.forward(want_traj=True)
obs = f["traj"][0].T * (1 + 0.05 * rng.standard_normal((N, 5))); obs[:, 0] = cp[:, 0]
eng.set_population_cpep(tp, G, obs, age, t2)
eng.set_params(nn, bt + 0.3 * rng.standard_normal(N))
eng.adam_init(1e-2)
for _ in range(3): eng.adam_step()
eng.set_kernel_timing(True)
K = 20
eng.synchronize(); t = time.perf_counter()
for _ in range(K): eng.adam_step(want_loss=False)
eng.synchronize(); dt = (time.perf_counter() - t) / K
ms, n = eng.kernel_time_ms()
print(f"N={N} step {dt*1e3:.3f} ms  kernel {ms:.3f} ms ({n} launches)  traj/s {N/dt:.3e}  loss {eng.adam_step():.6f}")
t = time.perf_counter()
for _ in range(K): eng.forward()
dt = (time.perf_counter() - t) / K
print(f"forward-only {dt*1e3:.3f} ms traj/s {N/dt:.3e}")
