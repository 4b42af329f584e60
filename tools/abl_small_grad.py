"""Small-population gradient path (time-split kernels): host-visible time of a loss + gradient call and of a queued Adam
iteration with the round-5 options on and off ("fused_tail": one tail launch instead of three; "scan_map": the scan's adjoint
recursion as a per-subject linear map).   python tools/abl_small_grad.py [N ...]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude.engine import Engine  # noqa: E402

for N in [int(v) for v in sys.argv[1:]] or [57, 1000, 10000]:
    arch = (2, 4, 2)
    tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
    for fused, smap in ((0, 0), (1, 0), (0, 1), (1, 1)):
        eng = Engine("cpep", arch, n_steps=32, n_state=2)
        eng.set_option("fused_tail", fused); eng.set_option("scan_map", smap)
        eng.set_population_cpep(tp, G, cp, age, t2)
        eng.set_params(o.glorot_params(arch, 1), bt)
        for _ in range(200): eng.loss_grad(want_cond_grad=False)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200): eng.loss_grad(want_cond_grad=False)
            best = min(best, (time.perf_counter() - t0) / 200)
        eng.adam_init(1e-3); eng.adam_run(64)
        bq = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); eng.adam_run(256); bq = min(bq, (time.perf_counter() - t0) / 256)
        print(f"N={N:6d} fused_tail={fused} scan_map={smap}: loss+gradient call {best * 1e6:7.1f} us   queued Adam iteration {bq * 1e6:7.1f} us", flush=True)
        eng.close()
