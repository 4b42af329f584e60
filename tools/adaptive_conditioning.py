"""How far one rounding moves the reference's solver: the oracle's adaptive Tsit5 (OrdinaryDiffEq defaults) with the
right-hand side perturbed by eps * N(0, 1), against itself unperturbed, on the synthetic c-peptide population of the
tests.  CPU only.  (Measured: eps = 1e-16 -> 1e-9 ... 2e-6 nmol/L; eps = 1e-14 -> up to 3e-4.)  This is the bound on
how closely ANY two implementations of the adaptive path can agree; tests/test_gpu_adaptive.py quotes it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "conditional-ude_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
from conftest import make_cpep_case  # noqa: E402
import cude_oracle as o  # noqa: E402

arch, N = (2, 4, 2), 131
c = make_cpep_case(N, arch)
pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
rng = np.random.default_rng(0)
tp = [float(t) for t in c["tp"]]
for eps in (1e-16, 1e-14, 1e-12):
    moved = []
    for i in range(0, N, 4):
        rhs0 = o.cpep_rhs_scalar(pop, i, c["nn"], float(np.exp(c["beta"][i])), arch)
        c0 = float(pop.c0[i])
        u0 = [c0, float(pop.k2[i] / pop.k1[i]) * c0]
        b = o.solve_adaptive(rhs0, u0, tp)
        a = o.solve_adaptive(lambda t, u: [v + eps * rng.standard_normal() for v in rhs0(t, u)], u0, tp)
        moved.append(max(abs(a[k][0] - b[k][0]) for k in range(len(tp))))
    moved = np.array(moved)
    print(f"eps = {eps:g}: trajectories move by median {np.median(moved):.2e}, max {moved.max():.2e} nmol/L")
