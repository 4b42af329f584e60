"""Exploration for tests/test_gpu_gradient_pins.py: size of the HIP gradient at the reference's stored optima."""
import os, sys
import numpy as np
import torch  # noqa
from scipy.optimize import linear_sum_assignment
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
from cude.engine import Engine
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ohashi_cude.npz")))
tp = g["timepoints"]
for tag, arch, nnk, bk in (("cude", (2, 4, 2), "nn_2x4x4x1", "betas_train"), ("cov", (3, 4, 2), "nn_3x4x4x1_cov", "betas_train_cov"),
                           ("sigma", (2, 4, 2), "nn_2x4x4x1_sigma", "betas_train_sigma")):
    eng = Engine("cpep", arch, n_steps=32, n_state=2)
    eng.set_population_cpep(tp, g["glucose"], g["cpeptide"], g["ages"], g["t2dm"])
    for k in range(min(6, g[nnk].shape[0])):
        nn, stored = g[nnk][k], g[bk][k]
        eng.set_params(nn, None)
        bh, _, _ = eng.fit_conditional(-5.0, 3.0, 81, 48)
        cost = np.abs(stored[:, None] - bh[None, :])
        r, c = linear_sum_assignment(cost)
        dev = cost[r, c]
        e57 = Engine("cpep", arch, n_steps=32, n_state=2)
        e57.set_population_cpep(tp, g["glucose"][c], g["cpeptide"][c], g["ages"][c], g["t2dm"][c])
        e57.set_params(nn, stored)
        L0, gn0, gb0 = e57.loss_grad()
        rng = np.random.default_rng(k)
        u_n = rng.standard_normal(nn.size); u_b = rng.standard_normal(57)
        nrm = np.sqrt(u_n @ u_n + u_b @ u_b); u_n /= nrm; u_b /= nrm
        eps = 1e-3
        e57.set_params(nn + eps * u_n, stored + eps * u_b); Lp, gnp, gbp = e57.loss_grad()
        e57.set_params(nn - eps * u_n, stored - eps * u_b); Lm, gnm, gbm = e57.loss_grad()
        fd1 = (Lp - Lm) / (2 * eps); an1 = gn0 @ u_n + gb0 @ u_b
        fd2 = (Lp - 2 * L0 + Lm) / eps ** 2; an2 = ((gnp - gnm) @ u_n + (gbp - gbm) @ u_b) / (2 * eps)
        e57.set_params(nn * 1.1, stored + 0.2); L1, gn1, gb1 = e57.loss_grad()
        ad = Engine("cpep", arch, n_steps=0, n_state=2)
        ad.set_population_cpep(tp, g["glucose"][c], g["cpeptide"][c], g["ages"][c], g["t2dm"][c])
        ad.set_params(nn, stored); La = ad.forward()["loss"]; ad.close()
        print(f"{tag} k={k} med|db|={np.median(dev):.1e} max={dev.max():.1e} L={L0:.6f} Ladapt={La:.6f} |g_nn|inf={np.max(np.abs(gn0)):.2e} "
              f"|g_b|inf={np.max(np.abs(gb0)):.2e} far: {np.max(np.abs(gn1)):.2e} {np.max(np.abs(gb1)):.2e}  "
              f"dir: fd {fd1:.4e} an {an1:.4e}  curv: fd {fd2:.5e} an {an2:.5e}")
        e57.close()
    eng.close()
