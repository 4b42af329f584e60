import os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("conditional-ude_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from conftest import make_cpep_case
import c_oracle as co
import cude_oracle as o
from cude.engine import Engine
arch = (2, 4, 2)
N = 131
c = make_cpep_case(N, arch)
eng = Engine("cpep", arch, n_steps=0, n_state=2)
eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
eng.set_params(c["nn"], c["beta"])
got = eng.forward(want_traj=True)["traj"][0].T
ref = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], np.exp(c["beta"]), c["tp"])
err = np.max(np.abs(got - ref), axis=1)
order = np.argsort(-err)
print("worst subjects", order[:8], err[order[:8]])
print("quantiles", np.quantile(err, [0.5, 0.9, 0.99]))
pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
for i in order[:3]:
    c0 = float(pop.c0[i]); u0 = [c0, float(pop.k2[i] / pop.k1[i]) * c0]
    sol = o.solve_adaptive(o.cpep_rhs_scalar(pop, int(i), c["nn"], float(np.exp(c["beta"][i])), arch), u0, [float(t) for t in c["tp"]])
    py = np.array([s[0] for s in sol])
    print(i, "beta", c["beta"][i], "G", c["G"][i], "\n gpu", got[i], "\n C  ", ref[i], "\n py ", py)
for tol in ((1e-8, 1e-6), (1e-5, 1e-2)):
    eng.set_tolerances(*tol)
    g2 = eng.forward(want_traj=True)["traj"][0].T
    r2 = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], np.exp(c["beta"]), c["tp"], abstol=tol[0], reltol=tol[1])
    e2 = np.max(np.abs(g2 - r2), axis=1)
    print(tol, "max", e2.max(), "median", np.median(e2), "n>1e-9", int(np.sum(e2 > 1e-9)))
