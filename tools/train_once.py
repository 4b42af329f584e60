"""One cude_train_restarts call on a synthetic population, for tracing: python3 tools/train_once.py N K adam_iters lbfgs_iters
(profiles/r05/train_copy_trace.txt: host <-> device copies as a function of the iteration counts)."""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cude.engine import Engine  # noqa: E402

N, K, n_adam, n_lbfgs = (int(v) for v in sys.argv[1:5])
nn0 = bench.glorot(bench.ARCH, 1234)
eng, pop = bench.cpep_engine(Engine, bench.ARCH, bench.N_STATE, N, 776, 0, nn0)
eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
rng = np.random.default_rng(11)
nn_sets = nn0[None, :] * (1.0 + 0.1 * rng.standard_normal((K, nn0.size)))
cond_sets = pop["beta0"][None, :] + 0.1 * rng.standard_normal((K, N))
if len(sys.argv) > 5:
    eng.set_option("train_host", sys.argv[5])
_, _, obj = eng.train_restarts(nn_sets, cond_sets, n_adam, 1e-3, n_lbfgs)
print("objectives", float(obj.min()), float(obj.max()))
eng.close()
