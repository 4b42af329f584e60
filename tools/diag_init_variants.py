"""Diagnostic for the training-envelope gap: which initialiser of the network reproduces the reference's distribution of
final suppression losses (0 of 125 runs above 0.7)?  SimpleChains.init_params is third-party and its RNG stream cannot be
reproduced; variants: a = Glorot normal, zero bias (api.init_params); b = Glorot uniform, zero bias; c = Glorot normal
over [W b] (random bias); d = a with sigma from size([W b])."""
import os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
from cude import api
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "suppression_lambda0.npz")))
data, tp = g["group_data"], g["timepoints"]
net = api.neural_network_model(5, 3, input_dims=4)
prob = api.SuppressionProblem(net)
eng = api._supp_population(prob, data, tp, 0.0).engine

def init(variant, rng):
    parts, fan, w = [], 4, 3
    dims = [(w, fan)] + [(w, w)] * 4 + [(1, w)]
    for out, fin in dims:
        if variant == "a":
            parts += [rng.standard_normal(out * fin) * np.sqrt(2.0 / (fin + out)), np.zeros(out)]
        elif variant == "b":
            lim = np.sqrt(6.0 / (fin + out)); parts += [rng.uniform(-lim, lim, out * fin), np.zeros(out)]
        elif variant == "c":
            s = np.sqrt(2.0 / (fin + 1 + out)); parts += [rng.standard_normal(out * fin) * s, rng.standard_normal(out) * s]
        elif variant == "d":
            parts += [rng.standard_normal(out * fin) * np.sqrt(2.0 / (fin + 1 + out)), np.zeros(out)]
    return np.concatenate(parts)

for variant in "abcd":
    for seed in (1, 2, 3):
        rng = np.random.default_rng(seed)
        th = rng.standard_normal((10000, data.shape[2])); nn = np.stack([init(variant, rng) for _ in range(10000)])
        idx, l0, nn_s, th_s = eng.screen_candidates(10000, 25, lambda f, n: (nn[f:f + n], th[f:f + n]))
        _, _, obj = eng.train_restarts(nn_s, th_s, 2000, 1e-3, 2000)
        obj = np.sort(obj)
        print(f"variant {variant} seed {seed}: initial {l0.min():.2f}..{l0.max():.2f}  final min/med/max {obj.min():.3f}/{np.median(obj):.3f}/{obj.max():.3f}  n>0.7: {int(np.sum(obj > 0.7))}", flush=True)
