import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_supp_case
from cude.engine import Engine
arch=(4,3,5)
c = make_supp_case(500, arch)
eng = Engine("supp", arch, n_steps=30, lam=0.0)
eng.set_population_supp(c["tp"], c["data"]); print("pop ok", flush=True)
eng.set_params(c["nn"], c["theta"]); print("params ok", flush=True)
print(eng.forward(), flush=True)
print(eng.forward(want_sse=True)["sse"][:3], flush=True)
print(eng.forward(want_traj=True)["traj"].shape, flush=True)
print(eng.loss_grad()[0], flush=True)
