"""End-to-end sanity envelope for the c-peptide cUDE (c-peptide/02-conditional.jl): train from scratch on the 57
Ohashi subjects the reference trained on (identified by matching its stored betas, tests/test_soft_pins.py) with
the reference's own recipe -- 25 000 screened initial guesses (network init + Latin hypercube betas in [-2,0]),
the best K trained with Adam(1e-2) x 1000 then L-BFGS x 1000 -- and compare the final objective with the
objective of the reference's STORED optimum (model k=0 weights + its betas) evaluated by the same loss.

usage: python tools/e2e_cpeptide.py [K=5] [adaptive]      adaptive: train on the reference's own objective (adaptive
Tsit5, abstol 1e-6 / reltol 1e-3, gradient = adjoint of the accepted steps) instead of the fixed-step discretisation
"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401
from scipy.optimize import linear_sum_assignment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
from cude import api  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N_STEPS = api.ADAPTIVE if len(sys.argv) > 2 and sys.argv[2] == "adaptive" else None
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ohashi_cude.npz")))
net = api.chain(4, 2, "tanh")
all_models = [api.CPeptideConditionalUDEModel(g["glucose"][i], g["timepoints"], g["ages"][i], net, g["cpeptide"][i],
                                              g["t2dm"][i]) for i in range(len(g["ages"]))]
# the reference's 57 training subjects: match its stored betas (model 0) to per-subject refits
nn0, betas0 = g["nn_2x4x4x1"][0], g["betas_train"][0]
fit = api.train(all_models, g["timepoints"], g["cpeptide"], nn0, lbfgs_lower_bound=-4.0, lbfgs_upper_bound=3.0)
beta_hat = np.array([s.u[0] for s in fit])
rows, cols = linear_sum_assignment(np.abs(betas0[:, None] - beta_hat[None, :]))
sel = cols[np.argsort(rows)]
models = [all_models[i] for i in sel]
data = g["cpeptide"][sel]
stored = api.loss(api.ComponentArray(neural=nn0, conditional=betas0[:, None]), (models, g["timepoints"], data),
                  n_steps=N_STEPS)
print(f"objective of the reference's stored optimum (model 0) on its 57 subjects"
      f"{' (adaptive solve)' if N_STEPS == api.ADAPTIVE else ''}: {stored:.4f}")

t0 = time.perf_counter()
sols = api.train(models, g["timepoints"], data, np.random.default_rng(232705), initial_guesses=25_000,
                 selected_initials=K, n_steps=N_STEPS)
dt = time.perf_counter() - t0
obj = np.array(sorted(s.objective for s in sols))
print(f"{len(sols)} runs (25 000 screened + Adam x1000 + L-BFGS x1000 each) in {dt:.1f} s; objectives {np.round(obj, 4).tolist()}")
