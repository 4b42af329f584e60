"""cude_fit_conditional (per-subject fits of the conditional parameter, shared parameters frozen: the reference's test-set
`train`, src/parameter-estimation.jl:388-430) with several probes per forward launch (option "fit_spec") against one probe per
launch: host-visible time of a fit with a 41-point scan and 48 golden-section steps.   python tools/bench_fit.py [N ...]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cude.engine import Engine  # noqa: E402

arch = (2, 4, 2)
nn = bench.glorot(arch, 99)
for N in [int(v) for v in sys.argv[1:]] or [57, 1000, 10000]:
    eng, pop = bench.cpep_engine(Engine, arch, 2, N, 780, 0, nn)
    eng.close()
    for steps, tag in ((30, "fixed 30 steps"), (0, "adaptive")):
        base = None
        for depth in (0, 1, 2, 3, 4, -1):
            eng = Engine("cpep", arch, n_steps=steps, n_state=2)
            eng.set_option("fit_spec", depth)
            eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
            eng.set_params(nn, pop["beta0"])
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                x, obj, sse = eng.fit_conditional(-4.0, 3.0, 41, 48)
                dt = time.perf_counter() - t0
                if rep: best = min(best, dt)
            if depth == 0: base = x
            print(f"N={N:6d} {tag:15s} fit_spec={depth:2d}: fit {best * 1e3:8.3f} ms   same minimisers: {np.array_equal(x, base)}", flush=True)
            eng.close()
