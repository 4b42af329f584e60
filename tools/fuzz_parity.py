"""Randomised parity sweep (development aid): random shapes, population sizes, time grids, step counts and weight
scales for the three models, HIP vs the C / numpy oracle.  Prints the worst relative errors; exits non-zero on a
violation of the test-suite tolerances (loss 1e-10, gradients 1e-9 of the max-norm).

usage: python tools/fuzz_parity.py [n_cases=40] [seed=0]
"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("conditional-ude_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import c_oracle as co  # noqa: E402
import cude_oracle as o  # noqa: E402
from cude.engine import Engine  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(loss=0.0, g_nn=0.0, g_cond=0.0)
n_bad = 0
CPEP = [(2, 4, 2), (2, 6, 2), (3, 4, 2), (2, 8, 2), (2, 4, 3)]
SUPP = [(4, 3, 5), (4, 3, 2), (4, 4, 2), (4, 6, 2)]


def rel(a, b):
    # max-norm relative error with an absolute floor: a saturated network has gradients ~1e-9 whose last bits are
    # rounding noise of O(1) intermediate quantities (the two CPU oracles then disagree at 1e-7 "relative" as well)
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-5))


for case in range(n_cases):
    kind = rng.choice(["cpep", "cpep", "sym", "supp"])
    N = int(rng.choice([1, 3, 63, 64, 65, 200, 1000, 5000]))
    T = int(rng.integers(2, 12))
    S = int(rng.choice([1, 7, 30, 32, 61, 120]))
    tp = np.concatenate([[0.0], np.cumsum(rng.uniform(2.0, 40.0, T - 1))])
    S = max(S, int(np.ceil(tp[-1] / 25.0)))      # keep h k inside Tsit5's stability region (k ~ 0.06 / min)
    scale = float(rng.choice([0.3, 1.0, 3.0]))
    if kind == "supp":
        arch = SUPP[rng.integers(len(SUPP))]
        # the reference's horizon (30 time units): over ~200 units with 3x weights the gradient is so ill-conditioned
        # (|g| ~ 1e6, the two CPU oracles themselves differ by 2e-9) that no implementation pair agrees to 1e-9
        tp = tp * (30.0 / tp[-1])
        if scale > 1.0:        # 3x Glorot weights make the suppression dynamics violent (loss ~1e3, |g| ~1e4): forward-
            scale = 1.0        # and reverse-mode gradients of the SAME map then differ by ~3e-9 (CPU oracles: 1e-9)
        t = tp[None, :, None]
        data = np.stack([10 * np.exp(-0.1 * t[0]) * np.ones((T, N)), 2 + np.sin(0.05 * t[0]) * np.ones((T, N)),
                         1 + 0.02 * t[0] * np.ones((T, N))]) * (1 + 0.1 * rng.standard_normal((3, T, N)))
        data = np.abs(data) + 0.05
        nn, th, lam = o.glorot_params(arch, case) * scale, rng.standard_normal(N), float(rng.choice([0.0, 0.05]))
        ref = co.supp(tp, data, arch, nn, th, lam, S)
        eng = Engine("supp", arch, n_steps=S, lam=lam)
        eng.set_population_supp(tp, data)
        eng.set_params(nn, th)
        loss, g_nn, g_c = eng.loss_grad()
        r = (rel(loss, ref["loss"]), rel(g_nn, ref["g_nn"]), rel(g_c, ref["g_theta"]))
    else:
        age, t2 = rng.uniform(20, 79, N), rng.random(N) < 0.4
        G = 5.0 + np.cumsum(rng.standard_normal((N, T)), axis=1) * 1.5            # rises and falls: dG of both signs
        obs = 0.3 + rng.random((N, T))
        n_state = int(rng.choice([2, 3]))
        if kind == "sym":
            space = str(rng.choice(["raw", "log"]))
            k = np.exp(rng.normal(3.5, 0.7, N))
            cond = k if space == "raw" else np.log(k)
            pop = o.CPepPopulation(tp, G, obs, age, t2)
            rl, rgp, rgc, _ = o.cpep_loss_grad_torch(np.array([1.78 * scale]), cond, pop, o.SYMBOLIC, S, n_state, space)
            eng = Engine("cpep_sym", n_steps=S, n_state=n_state, cond_space=space)
            eng.set_population_cpep(tp, G, obs, age, t2)
            eng.set_params([1.78 * scale], cond)
            loss, g_nn, g_c = eng.loss_grad()
            r = (rel(loss, rl), rel(g_nn, rgp), rel(g_c, rgc))
            arch = ("sym", space)
        else:
            arch = CPEP[rng.integers(len(CPEP))]
            nn, beta = o.glorot_params(arch, case) * scale, rng.normal(-0.6, 0.6, N)
            path = str(rng.choice(["1", "auto"]))
            if path == "1":
                os.environ["CUDE_CPEP_PATH"] = "1"
            else:
                os.environ.pop("CUDE_CPEP_PATH", None)
            ref = co.cpep(tp, G, obs, age, t2, arch, nn, beta, S, n_state, covariate=(arch[0] == 3))
            eng = Engine("cpep", arch, n_steps=S, n_state=n_state)
            eng.set_population_cpep(tp, G, obs, age, t2)
            eng.set_params(nn, beta)
            loss, g_nn, g_c = eng.loss_grad()
            r = (rel(loss, ref["loss"]), rel(g_nn, ref["g_nn"]), rel(g_c, ref["g_beta"]))
            arch = arch + (path,)
    eng.close()
    for key, v in zip(worst, r):
        worst[key] = max(worst[key], v)
    flag = "" if (r[0] <= 1e-10 and r[1] <= 1e-9 and r[2] <= 1e-9) else "   <-- VIOLATION"
    print(f"{case:3d} {kind:5s} {str(arch):22s} N={N:5d} T={T:2d} S={S:3d} scale={scale}: loss {r[0]:.1e} g_nn {r[1]:.1e} "
          f"g_cond {r[2]:.1e}{flag}")
    n_bad += bool(flag)
print("worst:", {k: f"{v:.1e}" for k, v in worst.items()}, "violations:", n_bad)
sys.exit(1 if n_bad else 0)
