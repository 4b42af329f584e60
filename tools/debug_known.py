"""Development aid: time the product's per-subject fit for every stored lambda = 0 suppression network (which one is slow?)."""
import os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "conditional-ude_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from test_known_answers_runs import ARCH, load_runs
from cude.engine import Engine
tp, sets, runs = load_runs()
tag = sys.argv[1] if len(sys.argv) > 1 else "0.0"
for name, data in sets.items():
    eng = Engine("supp", ARCH, n_steps=0, lam=0.0)
    eng.set_population_supp(tp, data)
    for n, nn in enumerate(runs[tag]["nn"]):
        eng.set_params(nn, np.zeros(eng.N))
        t0 = time.perf_counter()
        sse = eng.forward(want_sse=True)["sse"]
        t1 = time.perf_counter()
        # the grid of the fit, value by value: which theta is slow?
        slow = []
        for th in np.linspace(-8.0, 6.0, 29):
            eng.set_params(nn, np.full(eng.N, th))
            ta = time.perf_counter()
            eng.forward()
            dt = time.perf_counter() - ta
            if dt > 0.02:
                slow.append((round(float(th), 2), round(dt, 3)))
        print(f"{tag} {name} net {n}: forward {1e3 * (t1 - t0):.2f} ms; slow thetas: {slow}", flush=True)
    eng.close()
