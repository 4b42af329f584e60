"""Restart trainer (cude_train_restarts = the reference's `train` second phase, src/parameter-estimation.jl:340-386,
`_optimize` :170-183) at population scale: K restarts x N subjects, Adam iterations and L-BFGS iterations timed apart,
next to K x the single-set optimiser step of the same population (cude_adam_run).

usage: python3 tools/bench_train.py [N=100000] [K=25] [adam_iters=20] [lbfgs_iters=10] [model=cpep|cpep4|supp] [n_steps=30]
prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first: shared HIP runtime)

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cude.engine import Engine  # noqa: E402


def main():
    av = sys.argv[1:]
    N = int(av[0]) if len(av) > 0 else 100000
    K = int(av[1]) if len(av) > 1 else 25
    n_adam = int(av[2]) if len(av) > 2 else 20
    n_lbfgs = int(av[3]) if len(av) > 3 else 10
    model = av[4] if len(av) > 4 else "cpep"
    n_steps = int(av[5]) if len(av) > 5 else 30
    rng = np.random.default_rng(11)
    if model == "supp":
        arch = (4, 3, 5)
        tp, data, theta = bench.synthetic_suppression(N, 779)
        eng = Engine("supp", arch, n_steps=n_steps, lam=0.01)
        eng.set_population_supp(tp, data)
        nn0, cond0 = bench.glorot(arch, 1234), theta
    else:
        arch, ns = ((2, 6, 2), 3) if model == "cpep" else ((2, 4, 2), 2)
        nn0 = bench.glorot(arch, 1234)
        eng, pop = bench.cpep_engine(Engine, arch, ns, N, 776, 0, nn0)
        if n_steps != 30:
            eng.close()
            eng = Engine("cpep", arch, n_steps=n_steps, n_state=ns)
        eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
        cond0 = pop["beta0"]
    nn_sets = nn0[None, :] * (1.0 + 0.1 * rng.standard_normal((K, nn0.size)))
    cond_sets = cond0[None, :] + 0.1 * rng.standard_normal((K, N))
    # single-set step of the same population (device-resident Adam, queued iterations)
    eng.set_params(nn0, cond0)
    eng.adam_init(1e-3)
    eng.adam_run(8)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.adam_run(20)
    eng.synchronize()
    single = (time.perf_counter() - t0) / 20
    out = {"model": model, "arch": arch, "n_steps": n_steps, "subjects": N, "restarts": K,
           "single_set_step_ms": single * 1e3}
    eng.train_restarts(nn_sets, cond_sets, 2, 1e-3, 0)                     # warm: scratch allocation
    if n_adam > 0:
        t0 = time.perf_counter()
        _, _, obj_a = eng.train_restarts(nn_sets, cond_sets, n_adam, 1e-3, 0)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.train_restarts(nn_sets, cond_sets, 2 * n_adam, 1e-3, 0)
        dt2 = time.perf_counter() - t0
        per_iter = (dt2 - dt) / n_adam                                     # (difference: upload / download of the sets drops out)
        out.update(adam_iters=n_adam, adam_call_s=dt, adam_ms_per_iteration=per_iter * 1e3,
                   adam_iteration_over_K_single_steps=per_iter / (K * single),
                   adam_subject_trajectories_per_s=K * N / per_iter,
                   adam_objective_min_max=[float(np.min(obj_a)), float(np.max(obj_a))])
    if n_lbfgs > 0:
        t0 = time.perf_counter()
        _, _, obj_l, tr = eng.train_restarts(nn_sets, cond_sets, 0, 1e-3, n_lbfgs, want_trace=True)
        dt = time.perf_counter() - t0
        out.update(lbfgs_iters=n_lbfgs, lbfgs_call_s=dt, lbfgs_ms_per_iteration=dt / n_lbfgs * 1e3,
                   lbfgs_iteration_over_K_single_steps=dt / n_lbfgs / (K * single),
                   lbfgs_objective_min_max=[float(np.min(obj_l)), float(np.max(obj_l))],
                   lbfgs_accepted_iterations=int(np.sum(np.isfinite(tr))))
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
