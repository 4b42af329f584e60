"""Adaptive-mode launches of the suppression model (4-3x5-1, T = 8) next to the fixed-step ones.
python tools/bench_adaptive_supp.py [N]"""
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
arch = (4, 3, 5)
tp, data, theta = bench.synthetic_suppression(N, 779)
nn = bench.glorot(arch, 1234)
for n_steps in (0, 30):
    eng = Engine("supp", arch, n_steps=n_steps, lam=0.01)
    eng.set_population_supp(tp, data)
    eng.set_params(nn, theta)
    for what, call in (("forward", eng.forward), ("loss+gradient", lambda: eng.loss_grad(want_cond_grad=False))):
        for _ in range(10):
            call()
        eng.set_kernel_timing(True)
        for _ in range(20):
            call()
        ms, n = eng.kernel_time_ms()
        eng.set_kernel_timing(False)
        print(f"N={N} supp 4x3x5x1 n_steps={n_steps or 'adaptive'} {what}: {ms:.4f} ms per launch ({n} launches), "
              f"{N / ms * 1e3:.3e} subject-trajectories/s")
    if n_steps == 0:
        cnt = np.array([len(eng.adaptive_steps(i)[0]) for i in range(0, N, max(1, N // 400))])
        print(f"   accepted steps per subject: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}")
        before, after = eng.adaptive_regroup()
        print(f"   cude_adaptive_regroup: mean (max - min accepted steps) within a wave {before} -> {after}")
        for what, call in (("forward", eng.forward), ("loss+gradient", lambda: eng.loss_grad(want_cond_grad=False))):
            for _ in range(10):
                call()
            eng.set_kernel_timing(True)
            for _ in range(20):
                call()
            ms, n = eng.kernel_time_ms()
            eng.set_kernel_timing(False)
            print(f"N={N} supp 4x3x5x1 adaptive, regrouped {what}: {ms:.4f} ms per launch ({n} launches), "
                  f"{N / ms * 1e3:.3e} subject-trajectories/s")
    eng.close()
