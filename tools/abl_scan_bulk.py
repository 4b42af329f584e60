"""Scan of a small launch with every row fetched through LDS up front (option "scan_bulk", cpep2_scan_bulk_kernel) against the
register-staged scan: host-visible time of a forward call, a loss + gradient call and a queued Adam iteration, and whether
the results are the same bits.   python tools/abl_scan_bulk.py [N ...]   (ARCH=2,4,2 STEPS=32 NSTATE=2)"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude.engine import Engine  # noqa: E402

ARCH = tuple(int(v) for v in os.environ.get("ARCH", "2,4,2").split(","))
STEPS = int(os.environ.get("STEPS", "32"))
NSTATE = int(os.environ.get("NSTATE", "2"))


def timed(f, n=200, rep=5):
    for _ in range(n): f()
    best = 1e9
    for _ in range(rep):
        t0 = time.perf_counter()
        for _ in range(n): f()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


for N in [int(v) for v in sys.argv[1:]] or [57, 1000, 10000, 16384]:
    tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
    nn = o.glorot_params(ARCH, 1)
    res = {}
    for bulk in (0, 1):
        eng = Engine("cpep", ARCH, n_steps=STEPS, n_state=NSTATE)
        eng.set_option("scan_bulk", bulk)
        eng.set_population_cpep(tp, G, cp, age, t2)
        eng.set_params(nn, bt)
        tf = timed(lambda: eng.forward())
        tg = timed(lambda: eng.loss_grad(want_cond_grad=False))
        lf = eng.forward(want_sse=True)
        l, g, gc = eng.loss_grad()
        eng.adam_init(1e-3); eng.adam_run(64)
        bq = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); eng.adam_run(256); bq = min(bq, (time.perf_counter() - t0) / 256)
        res[bulk] = (l, g, gc, lf)
        print(f"N={N:6d} scan_bulk={bulk}: forward call {tf:7.1f} us   loss+gradient call {tg:7.1f} us   "
              f"queued Adam iteration {bq * 1e6:7.1f} us", flush=True)
        eng.close()
    (l0, g0, c0, f0), (l1, g1, c1, f1) = res[0], res[1]
    same_f = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(f0, f1))
    print(f"          identical bits: loss {l0 == l1}, network gradient {np.array_equal(g0, g1)}, conditional gradient "
          f"{np.array_equal(c0, c1)}, forward call {same_f}", flush=True)
