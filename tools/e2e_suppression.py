"""End-to-end sanity envelope: the reference's suppression experiment (suppression/suppression.jl, lambda = 0)
re-run through the GPU path -- 10 000 random initial guesses screened, the best K trained with Adam(1e-3) x 2000
then L-BFGS x 2000 on the reference's own stored `group_data` (tests/golden/suppression_lambda0.npz) -- and the
final training losses compared with the reference's stored ones (summary_lambda=0.0.csv: min/median/max
0.429/0.492/0.616 over its 25 kept runs; Spearman rho(theta, truth) median 0.873 is not checked here because
gt_sup_param is not in the fixture).

usage: python tools/e2e_suppression.py [K=5] [adam_iters=2000] [lbfgs_iters=2000] [lambda=0.0] [seed=27052023]
"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
from cude import api  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
adam_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
lbfgs_iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
lam = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 27052023
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "suppression_lambda0.npz")))
data, tp = g["group_data"], g["timepoints"]
rng = np.random.default_rng(seed)
net = api.neural_network_model(5, 3, input_dims=4)
prob = api.SuppressionProblem(net)
p_init = [api.ComponentArray(theta=rng.standard_normal(data.shape[2]), neural=api.init_params(net, rng))
          for _ in range(10000)]
t0 = time.perf_counter()
sols, traces = api.fit_suppression_model(p_init, prob, data, tp, lam, select_best_n=K, adam_iters=adam_iters,
                                         lbfgs_iters=lbfgs_iters)
dt = time.perf_counter() - t0
for s, tr in zip(sols, traces):
    print(f"  run: after Adam {tr[min(adam_iters, len(tr)) - 1]:.4f}, L-BFGS iterations {max(0, len(tr) - adam_iters)}, "
          f"final {s.objective:.4f}")
losses = np.array(sorted(s.objective for s in sols))
print(f"{len(sols)} runs in {dt:.1f} s; final training losses: {np.round(losses, 4).tolist()}")
ref = {0.0: "0.429/0.492/0.616", 0.01: "0.598/0.607/(max n.a.)"}.get(lam, "n.a.")
print(f"lambda={lam} seed={seed}: min/median/max = {losses.min():.3f}/{np.median(losses):.3f}/{losses.max():.3f}   "
      f"n>0.7: {int(np.sum(losses > 0.7))}   (reference's 25 kept runs: {ref})")
