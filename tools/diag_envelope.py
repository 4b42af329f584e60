"""Diagnostic for the training-envelope gap: where do the runs that end above 0.7 stall?"""
import os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
from cude import api
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 27052023
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "suppression_lambda0.npz")))
data, tp = g["group_data"], g["timepoints"]
rng = np.random.default_rng(seed)
net = api.neural_network_model(5, 3, input_dims=4)
prob = api.SuppressionProblem(net)
p_init = [api.ComponentArray(theta=rng.standard_normal(data.shape[2]), neural=api.init_params(net, rng)) for _ in range(10000)]
pop = api._supp_population(prob, data, tp, 0.0)
eng = pop.engine
losses0 = eng.multistart_forward(np.stack([p.neural for p in p_init]), np.stack([p.theta for p in p_init]))
best = np.argsort(losses0, kind="stable")[:25]
print("initial losses of the kept 25:", np.round(losses0[best], 3))
nn0 = np.stack([p_init[k].neural for k in best]); th0 = np.stack([p_init[k].theta for k in best])
nn, th, obj, tr = eng.train_restarts(nn0, th0, 2000, 1e-3, 2000, want_trace=True)
for k in range(25):
    t = tr[k]; lb = t[2000:]; n_lb = int(np.sum(np.isfinite(lb)))
    pick = lambda i: lb[min(i, n_lb - 1)] if n_lb else np.nan
    print(f"run {k:2d}: init {losses0[best[k]]:.3f} adam500 {t[499]:.3f} adam2000 {t[1999]:.3f} | lbfgs its {n_lb:4d}: @100 {pick(99):.4f} @500 {pick(499):.4f} @1000 {pick(999):.4f} final {obj[k]:.4f}")
tail = np.flatnonzero(obj > 0.7)
if tail.size:
    nn2, th2, obj2, tr2 = eng.train_restarts(nn[tail], th[tail], 0, 1e-3, 6000, want_trace=True)
    for j, k in enumerate(tail):
        n_lb = int(np.sum(np.isfinite(tr2[j])))
        print(f"tail run {k}: {obj[k]:.4f} -> after {n_lb} more L-BFGS iterations {obj2[j]:.4f}")
    # and with Adam at a 10x larger rate from the original start
    nn3, th3, obj3 = eng.train_restarts(nn0[tail], th0[tail], 2000, 1e-2, 2000)
    print("tail runs re-trained with Adam(1e-2):", np.round(obj3, 4))
