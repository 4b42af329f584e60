"""Resident waves per CU the runtime grants the one-lane gradient kernels (cude_grad_occupancy:
hipOccupancyMaxActiveBlocksPerMultiprocessor with the launch's LDS size); 8 = two waves per SIMD."""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("conditional-ude_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
from conftest import make_cpep_case, make_supp_case  # noqa: E402
from cude.engine import Engine  # noqa: E402

c = make_supp_case(256)
for arch in ((4, 3, 5), (4, 3, 3), (4, 4, 3), (4, 6, 2)):
    cc = make_supp_case(256, arch)
    eng = Engine("supp", arch, n_steps=30)
    eng.set_population_supp(cc["tp"], cc["data"])
    print(f"supp_kernel<{arch[1]},{arch[2]},grad>: {eng.grad_occupancy()} waves per CU")
    eng.close()
for arch, ns in (((2, 6, 2), 3), ((2, 4, 2), 2), ((3, 4, 2), 2)):
    cc = make_cpep_case(256, arch)
    eng = Engine("cpep", arch, n_steps=30, n_state=ns)
    eng.set_population_cpep(cc["tp"], cc["G"], cc["obs"], cc["age"], cc["t2dm"])
    print(f"cpep_kernel<Mlp<{arch[0]},{arch[1]},{arch[2]},1>,{ns},grad>: {eng.grad_occupancy()} waves per CU")
    eng.close()
