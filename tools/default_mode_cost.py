"""What the API's default discretisation (the reference's adaptive solve) costs against the explicit fast mode (fixed steps,
time-split kernels) at the reference's own population sizes: host-visible time of a loss + gradient call and of a queued
Adam iteration, c-peptide 2-4-4-1 on the Ohashi-like five-time grid, and the reference's `train` recipe (K restarts side by
side) per iteration.   python tools/default_mode_cost.py [N ...]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cude.engine import Engine  # noqa: E402

arch = (2, 4, 2)
nn = bench.glorot(arch, 1234)
for N in [int(v) for v in sys.argv[1:]] or [57, 117, 1000]:
    eng0, pop = bench.cpep_engine(Engine, arch, 2, N, 777, 0, nn)
    eng0.close()
    for n_steps, tag in ((0, "adaptive (default)"), (32, "fixed 32 steps (fast mode)")):
        eng = Engine("cpep", arch, n_steps=n_steps, n_state=2)
        eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
        eng.set_params(nn, pop["beta0"])
        for _ in range(100): eng.loss_grad(want_cond_grad=False)
        t0 = time.perf_counter()
        for _ in range(300): eng.loss_grad(want_cond_grad=False)
        lg = (time.perf_counter() - t0) / 300
        eng.adam_init(1e-3); eng.adam_run(64)
        t0 = time.perf_counter(); eng.adam_run(512); it = (time.perf_counter() - t0) / 512
        K = 25
        rng = np.random.default_rng(1)
        nn_sets = nn[None, :] * (1.0 + 0.1 * rng.standard_normal((K, nn.size)))
        b_sets = pop["beta0"][None, :] + 0.1 * rng.standard_normal((K, N))
        eng.train_restarts(nn_sets, b_sets, 20, 1e-2, 5)
        t0 = time.perf_counter(); eng.train_restarts(nn_sets, b_sets, 200, 1e-2, 0); ta = (time.perf_counter() - t0) / 200
        t0 = time.perf_counter(); _, _, obj = eng.train_restarts(nn_sets, b_sets, 0, 1e-2, 50); tl = (time.perf_counter() - t0) / 50
        print(f"N={N:5d} {tag:28s}: loss+gradient call {lg * 1e6:7.1f} us | queued Adam iteration {it * 1e6:7.1f} us | "
              f"train, 25 restarts: Adam iteration {ta * 1e6:7.1f} us, L-BFGS iteration {tl * 1e6:7.1f} us", flush=True)
        eng.close()
