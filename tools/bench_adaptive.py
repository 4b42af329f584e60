"""Adaptive-mode launches (cude_config.n_steps = 0) at a given population size: forward solve, gradient (tape + reverse
sweep), one Adam step, next to the fixed-step launch of the same 2-state model.   python tools/bench_adaptive.py [N] [W]"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
if os.environ.get("CUDE_ABL"):                       # A/B runs: a library variant from tools/abl_so/
    from cude import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", os.environ["CUDE_ABL"] + ".so")
    _lib.STRICT = False
from cude.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4
arch = (2, W, 2)
nn = bench.glorot(arch, 1234)
eng0, pop = bench.cpep_engine(Engine, arch, 2, N, 777, 0, nn)
eng0.close()
for n_steps in (0, 30):
    eng = Engine("cpep", arch, n_steps=n_steps, n_state=2)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn, pop["beta0"])
    for what, call in (("forward", eng.forward), ("loss+gradient", lambda: eng.loss_grad(want_cond_grad=False))):
        for _ in range(30):
            call()
        eng.set_kernel_timing(True)
        for _ in range(50):
            out = call()
        ms, n = eng.kernel_time_ms()
        eng.set_kernel_timing(False)
        print(f"N={N} 2x{W}x{W}x1 n_steps={n_steps or 'adaptive'} {what}: {ms:.4f} ms per launch ({n} launches), "
              f"{N / ms * 1e3:.3e} subject-trajectories/s")
    if n_steps == 0:
        cnt = np.array([len(eng.adaptive_steps(i)[0]) for i in range(0, N, max(1, N // 400))])
        print(f"   accepted steps per subject: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}")
        # lanes of a wave run as long as the slowest of them: order the launch by accepted-step count and time again
        before, after = eng.adaptive_regroup()
        print(f"   cude_adaptive_regroup: mean (max - min accepted steps) within a wave {before} -> {after}")
        for what, call in (("forward", eng.forward), ("loss+gradient", lambda: eng.loss_grad(want_cond_grad=False))):
            for _ in range(30):
                call()
            eng.set_kernel_timing(True)
            for _ in range(50):
                call()
            ms, n = eng.kernel_time_ms()
            eng.set_kernel_timing(False)
            print(f"N={N} 2x{W}x{W}x1 adaptive, regrouped {what}: {ms:.4f} ms per launch ({n} launches), "
                  f"{N / ms * 1e3:.3e} subject-trajectories/s")
    eng.close()
