"""The c-peptide training envelope (round 4): what did the reference's `train` (src/parameter-estimation.jl:340-386) END
at, and where does the product's restatement of the same recipe end?

The reference stored 25 (network, betas) end points of its training run on 57 subjects
(source_data/cude_neural_parameters.jld2, written by c-peptide/02-conditional.jl:32-50), a second run
(`_sigma`), and 25 more pairs under source_data/advi/.  Everything below runs through libcude_hip.so in ADAPTIVE mode
(the reference's own objective) on the 57 subjects those betas belong to (identified by matching, as
tests/test_soft_pins.py):
  1. the objective of every stored end point: the reference's distribution (min / median / max);
  2. how stationary each stored end point is: gradient norms there, relative to a perturbed point;
  3. the product's L-BFGS continued FROM each stored end point: if the objective still drops a lot, the reference's
     second stage stopped before convergence;
  4. the product's own 25 runs of the full recipe (25 000 screened, Adam x1000, L-BFGS x1000), and of the recipe
     without the second stage.
usage: python tools/envelope_cpeptide.py > profiles/r04/envelope_cpeptide.txt   (needs a GPU; reads tests/golden only)"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401
from scipy.optimize import linear_sum_assignment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
from cude import api  # noqa: E402
from cude.engine import Engine  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ohashi_cude.npz")))
advi = dict(np.load(os.path.join(ROOT, "tests", "golden", "advi_cude.npz")))
tp = g["timepoints"]
net = api.chain(4, 2, "tanh")
ARCH = (2, 4, 2)


def summary(x):
    x = np.sort(np.asarray(x))
    return f"min {x[0]:.4f}  q25 {np.quantile(x, .25):.4f}  median {np.median(x):.4f}  q75 {np.quantile(x, .75):.4f}  max {x[-1]:.4f}"


def engine_for(rows):
    eng = Engine("cpep", ARCH, n_steps=0, n_state=2)
    eng.set_population_cpep(tp, g["glucose"][rows], g["cpeptide"][rows], g["ages"][rows], g["t2dm"][rows])
    return eng


# ---- the 57 training subjects, in the stored order: match the best model's betas to per-subject refits among the 82
# subjects of the reference's prepared training set
train82 = np.flatnonzero(np.isin(g["subject_no"], g["train_subject_numbers"]))
k_best = int(g["best_model_index"]) - 1
e82 = engine_for(train82)
e82.set_params(g["nn_2x4x4x1"][k_best], np.zeros(len(train82)))
beta_hat, _, _ = e82.fit_conditional(-4.0, 3.0, n_grid=141, n_iters=45)
e82.close()
cost = np.abs(g["betas_train"][k_best][:, None] - beta_hat[None, :])
r, c = linear_sum_assignment(cost)
rows57 = train82[c[np.argsort(r)]]
print(f"# 57 training subjects identified among the 82 of the prepared set: |beta_stored - beta_refit| median "
      f"{np.median(cost[r, c]):.2e}, max {cost[r, c].max():.2e}")
eng = engine_for(rows57)
N = 57


def objective(nn, beta):
    eng.set_params(nn, beta)
    return eng.forward()["loss"]


# ---- 1. the stored end points
stored = {"cude_neural_parameters.jld2": (g["nn_2x4x4x1"], g["betas_train"]),
          "cude_neural_parameters_sigma.jld2": (g["nn_2x4x4x1_sigma"], g["betas_train_sigma"]),
          "advi/cude_result_{1..25}.jld2": (advi["nn_2x4x4x1"], advi["betas_train"])}
print("\n# 1. objective (mean SSE over the 57 subjects, adaptive solve) of the reference's stored end points")
objs = {}
for name, (nns, betas) in stored.items():
    objs[name] = np.array([objective(nns[k], betas[k]) for k in range(len(nns))])
    print(f"{name:36s} {summary(objs[name])}")
    print("    sorted:", np.round(np.sort(objs[name]), 4).tolist())

# ---- 2. stationarity
print("\n# 2. stationarity of the stored end points of cude_neural_parameters.jld2: |g_nn|_2 and |g_beta|_inf at the "
      "stored point, relative to the same norms at a perturbed point (network x 1.1, betas + 0.2)")
nns, betas = stored["cude_neural_parameters.jld2"]
rel = []
for k in range(25):
    eng.set_params(nns[k], betas[k])
    _, gn, gb = eng.loss_grad()
    eng.set_params(nns[k] * 1.1, betas[k] + 0.2)
    _, gn1, gb1 = eng.loss_grad()
    rel.append((np.linalg.norm(gn) / np.linalg.norm(gn1), np.max(np.abs(gb)) / np.max(np.abs(gb1))))
rel = np.array(rel)
order = np.argsort(objs["cude_neural_parameters.jld2"])
print("model(1-based) objective  |g_nn| ratio  |g_beta| ratio")
for k in order:
    print(f"   {k + 1:2d}         {objs['cude_neural_parameters.jld2'][k]:.4f}     {rel[k, 0]:.3e}     {rel[k, 1]:.3e}")

# ---- 3. continue from the stored end points with the product's second stage
print("\n# 3. the product's L-BFGS (Optim's LBFGS + BackTracking restated) continued from every stored end point, 1000 iterations")
for name, (nns, betas) in stored.items():
    t0 = time.perf_counter()
    _, _, cont = eng.train_restarts(nns, betas, 0, 1e-2, 1000)
    print(f"{name:36s} before: {summary(objs[name])}")
    print(f"{'':36s} after : {summary(cont)}   ({time.perf_counter() - t0:.1f} s)")
    drop = (objs[name] - cont) / objs[name]
    print(f"{'':36s} relative drop: {summary(drop)}")

# ---- 4. the product's own runs of the recipe
print("\n# 4. the product's 25 runs of the reference's recipe on the same 57 subjects (25 000 screened candidates, seed 232705)")
models = [api.CPeptideConditionalUDEModel(g["glucose"][i], tp, g["ages"][i], net, g["cpeptide"][i], g["t2dm"][i])
          for i in rows57]
data = g["cpeptide"][rows57]
for label, lb in (("Adam x1000 + L-BFGS x1000", 1000), ("Adam x1000 only", 0)):
    t0 = time.perf_counter()
    sols = api.train(models, tp, data, np.random.default_rng(232705), initial_guesses=25_000, selected_initials=25,
                     number_of_iterations_lbfgs=lb, n_steps=api.ADAPTIVE)
    o = np.array([s.objective for s in sols])
    print(f"{label:28s} {len(sols)} runs in {time.perf_counter() - t0:.1f} s: {summary(o)}")
    print("    sorted:", np.round(np.sort(o), 4).tolist())
eng.close()
