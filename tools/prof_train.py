import os, sys, time
import numpy as np
import torch
ROOT = "/root/repo" if os.path.isdir("/root/repo/conditional-ude_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cude import api
from cude.engine import Engine
from conftest import make_cpep_case
c = make_cpep_case(57, (2, 4, 2))
eng = Engine("cpep", (2, 4, 2), n_steps=32, n_state=2)
eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
rng = np.random.default_rng(0)
K = 25
nn = np.stack([api.init_params(api.chain(4, 2, "tanh"), rng) for _ in range(K)]); cond = rng.uniform(-2, 0, (K, 57))
eng.multistart_loss_grad(nn, cond)
t = time.perf_counter()
for _ in range(200): eng.multistart_loss_grad(nn, cond)
print("multistart_loss_grad K=25 N=57: %.3f ms per call" % ((time.perf_counter() - t) / 200 * 1e3))
nn25k = np.stack([api.init_params(api.chain(4, 2, "tanh"), rng) for _ in range(25000)]); c25k = rng.uniform(-2, 0, (25000, 57))
t = time.perf_counter(); eng.multistart_forward(nn25k, c25k); print("screening 25000 sets: %.1f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); api._batched_adam_then_lbfgs(eng, nn, cond, 1000, 0, 1e-2); print("Adam x1000 (25 restarts): %.2f s" % (time.perf_counter() - t))
t = time.perf_counter(); api._batched_adam_then_lbfgs(eng, nn, cond, 0, 1000, 1e-2); print("L-BFGS x1000 (25 restarts): %.2f s" % (time.perf_counter() - t))
