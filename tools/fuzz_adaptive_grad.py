"""Randomised parity sweep of the ADAPTIVE gradient (development aid): random model / shape / population size / time
grid / solver tolerances / weight scale; the device's loss and gradient (tape + reverse sweep, csrc/cude_adaptive.hip)
against the oracle's complex-step derivative of the device's own accepted-step sequences (cude_oracle.*_replay_loss_grad).
Bars: loss 1e-10, gradients 1e-8 of the max-norm (the test suite's).  Exits non-zero on a violation.

usage: python tools/fuzz_adaptive_grad.py [n_cases=40] [seed=0]
"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("conditional-ude_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import cude_oracle as o  # noqa: E402
if os.environ.get("CUDE_ABL"):                       # A/B runs: a library variant from tools/abl_so/
    from cude import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", os.environ["CUDE_ABL"] + ".so")
    _lib.STRICT = False
from cude.engine import Engine  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
CPEP = [(2, 4, 2), (2, 6, 2), (3, 4, 2), (2, 8, 2), (2, 4, 3), (2, 3, 1), (2, 5, 2), (2, 7, 2), (3, 6, 2), (2, 6, 3), (3, 4, 3)]
SUPP = [(4, 3, 5), (4, 3, 2), (4, 4, 2), (4, 6, 2), (4, 5, 3), (4, 8, 1)]
worst = dict(loss=0.0, g_nn=0.0, g_cond=0.0)
n_bad = 0


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-5))


ONLY = os.environ.get("FUZZ_ONLY")          # "k": evaluate case k only (same random stream) and print diagnostics
ONLY = None if ONLY is None else int(ONLY)


class _Skip:
    """stand-in engine for the cases skipped under FUZZ_ONLY"""
    def __getattr__(self, name):
        return lambda *a, **k: None


for case in range(n_cases):
    kind = rng.choice(["cpep", "cpep", "sym", "supp"])
    N = int(rng.choice([1, 3, 63, 64, 65, 130]))
    T = int(rng.integers(2, 10))
    tp = np.concatenate([[0.0], np.cumsum(rng.uniform(2.0, 40.0, T - 1))])
    abstol, reltol = [(1e-6, 1e-3), (1e-8, 1e-6), (1e-4, 1e-2), (1e-10, 1e-8)][rng.integers(4)]
    scale = float(rng.choice([0.3, 1.0, 2.0]))
    if kind == "supp":
        arch = SUPP[rng.integers(len(SUPP))]
        tp = tp * (30.0 / tp[-1])
        t = tp[None, :, None]
        data = np.stack([10 * np.exp(-0.1 * t[0]) * np.ones((T, N)), 2 + np.sin(0.05 * t[0]) * np.ones((T, N)),
                         1 + 0.02 * t[0] * np.ones((T, N))]) * (1 + 0.1 * rng.standard_normal((3, T, N)))
        data = np.abs(data) + 0.05
        nn, th, lam = o.glorot_params(arch, case) * min(scale, 1.0), rng.standard_normal(N), float(rng.choice([0.0, 0.05]))
        if ONLY is not None and case != ONLY:
            continue
        eng = Engine("supp", arch, n_steps=0, lam=lam)
        eng.set_tolerances(abstol, reltol)
        eng.set_population_supp(tp, data)
        eng.set_params(nn, th)
        loss, g_nn, g_c = eng.loss_grad()
        steps = [list(zip(*eng.adaptive_steps(i))) for i in range(N)]
        rl, rg, rc, sse_all = o.supp_replay_loss_grad(nn, th, data, tp, arch, lam, steps)
    else:
        age, t2 = rng.uniform(20, 79, N), rng.random(N) < 0.4
        G = 5.0 + np.cumsum(rng.standard_normal((N, T)), axis=1) * 1.5
        obs = 0.3 + rng.random((N, T))
        if kind == "sym":
            space = str(rng.choice(["raw", "log"]))
            k = np.exp(rng.normal(3.5, 0.7, N))
            cond, arch, nn = (k if space == "raw" else np.log(k)), o.SYMBOLIC, np.array([1.78 * scale])
            eng = Engine("cpep_sym", n_steps=0, n_state=2, cond_space=space)
        else:
            space = "log"
            arch = CPEP[rng.integers(len(CPEP))]
            nn, cond = o.glorot_params(arch, case) * scale, rng.normal(-0.6, 0.6, N)
            eng = Engine("cpep", arch, n_steps=0, n_state=2)
        pop = o.CPepPopulation(tp, G, obs, age, t2, covariate=(arch[0] == 3))
        if ONLY is not None and case != ONLY:
            eng.close()
            continue
        eng.set_tolerances(abstol, reltol)
        eng.set_population_cpep(tp, G, obs, age, t2)
        eng.set_params(nn, cond)
        loss, g_nn, g_c = eng.loss_grad()
        steps = [list(zip(*eng.adaptive_steps(i))) for i in range(N)]
        rl, rg, rc, sse_all = o.cpep_replay_loss_grad(nn, cond, pop, arch, steps, space)
    if ONLY is not None:
        sse_dev = eng.forward(want_sse=True)["sse"]
        print("loss", loss, rl, "max|g_nn|", np.max(np.abs(rg)), "max|g_cond|", np.max(np.abs(rc)))
        print("g_nn abs err", np.max(np.abs(g_nn - rg)), "g_cond abs err", np.max(np.abs(g_c - rc)), "at subject",
              int(np.argmax(np.abs(g_c - rc))))
        if kind != "supp":
            ol = o.cpep_adaptive_loss_grad(nn, cond, pop, arch, abstol, reltol, space)
            print("loss: forward kernel", eng.forward()["loss"], "gradient kernel", loss, "replay of its steps", rl,
                  "oracle's own adaptive solve", ol[0])
            # conditioning of the replayed map: the same sequences with the parameters moved by one rounding error
            rl2 = o.cpep_replay_loss_grad(nn * (1.0 + 2.2e-16), cond, pop, arch, steps, space)[0]
            print(f"replayed loss with nn * (1 + eps): changes by {abs(rl2 - rl) / rl:.1e} relative "
                  f"(first-order prediction {2.2e-16 * abs(float(rg @ nn)) / rl:.1e})")
            # subject by subject: replay of the device's steps vs the device's forward kernel
            bad = np.flatnonzero(np.abs(sse_dev - sse_all) > 1e-9 * np.maximum(sse_all, 1e-30))
            print("subjects whose forward-kernel SSE differs from the replay by > 1e-9:", bad.tolist())
            for i in bad[:3]:
                rec = []
                c0 = float(pop.c0[i])
                cc = float(np.exp(cond[i])) if space == "log" else float(cond[i])
                o.solve_adaptive(o.cpep_rhs_scalar(pop, i, nn, cc, arch), [c0, float(pop.k2[i] / pop.k1[i]) * c0],
                                 [float(v) for v in tp], abstol, reltol, record=rec)
                print(" subject", i, "device steps", [(round(a, 6), round(b, 6)) for a, b in steps[i]])
                print("           oracle steps", [(round(a, 6), round(b, 6)) for a, b in rec])
        print("per-subject sse rel err (device forward vs replay):", np.max(np.abs(sse_dev - sse_all) / np.maximum(sse_all, 1e-30)),
              "subject", int(np.argmax(np.abs(sse_dev - sse_all) / np.maximum(sse_all, 1e-30))))
    eng.close()
    n_st = [len(s) for s in steps]
    r = (rel(loss, rl), rel(g_nn, rg), rel(g_c, rc))
    for key, v in zip(worst, r):
        worst[key] = max(worst[key], v)
    flag = "" if (r[0] <= 1e-10 and r[1] <= 1e-8 and r[2] <= 1e-8) else "   <-- VIOLATION"
    if flag:
        # Is the CHECKER itself stable here?  The same sequences with the shared parameters moved by one rounding error:
        # loose tolerances on a long span accept steps far outside Tsit5's stability region (dt = 100 min against a
        # 5-minute half-life), which amplify rounding by many orders of magnitude; saturated networks have gradients
        # that are differences of O(1) numbers.  An error within 10x the oracle's own movement is rounding, not a bug.
        # ... and with the DATA moved by one rounding error (the initial state is the kinetics' steady state: its
        # derivative is pure rounding residue, which unstable steps amplify independently of the parameters)
        if kind == "supp":
            q = o.supp_replay_loss_grad(nn * (1.0 + 2.2e-16), th, data, tp, arch, lam, steps)
            d2 = data.copy()
            d2[:, 0, :] *= 1.0 + 2.2e-16
            q2 = o.supp_replay_loss_grad(nn, th, d2, tp, arch, lam, steps)
        else:
            q = o.cpep_replay_loss_grad(nn * (1.0 + 2.2e-16), cond, pop, arch, steps, space)
            pop2 = o.CPepPopulation(tp, G, obs, age, t2, covariate=(arch[0] == 3))
            pop2.k2 = pop2.k2 * (1.0 + 2.2e-16)
            q2 = o.cpep_replay_loss_grad(nn, cond, pop2, arch, steps, space)
        own = tuple(max(rel(a[k], b), rel(c_[k], b)) for k, (a, c_, b) in enumerate(((q, q2, rl), (q, q2, rg), (q, q2, rc))))
        # (the device's arithmetic differs from the oracle's by many roundings, not one: two orders of magnitude of slack)
        if own[0] > 1e-10 and r[0] <= 100.0 * own[0]:
            # one ulp already moves the oracle's LOSS by more than its bar: the accepted sequence amplifies rounding
            # > 1e6-fold (steps far outside the stability region) -- nothing can be held to 1e-8 on it
            flag = (f"   (unstable step sequence: one ulp moves the oracle's loss by {own[0]:.0e}, its gradients by "
                    f"{own[1]:.0e} / {own[2]:.0e})")
        elif all(e <= max(b, 100.0 * w) for e, b, w in zip(r, (1e-10, 1e-8, 1e-8), own)):
            flag = f"   (rounding-limited: the oracle itself moves by {own[0]:.0e} / {own[1]:.0e} / {own[2]:.0e} under one ulp of nn / of the data)"
        else:
            # A saturated network: the gradient is ~0 (|g| << 1e-5, the floor of rel()) as the difference of the stage
            # terms and the baseline term, which the device accumulates separately (- sum(w) * grad NN([0; e^beta]) once
            # at the end) -- absolute error eps * sum|w| * |grad NN|.  Negligible against the loss: <= 1e-10 |loss|.
            abs_err = (float(np.max(np.abs(np.asarray(g_nn) - rg))), float(np.max(np.abs(np.asarray(g_c) - rc))))
            if r[0] <= 1e-10 and max(abs_err) <= 1e-10 * abs(rl) and max(np.max(np.abs(rg)), np.max(np.abs(rc))) < 1e-5:
                flag = f"   (vanishing gradient, |g| = {np.max(np.abs(rg)):.0e}: absolute error {max(abs_err):.0e} <= 1e-10 |loss|)"
            else:
                flag += f"   [oracle's own movement: {own[0]:.0e} / {own[1]:.0e} / {own[2]:.0e}]"
    print(f"{case:3d} {kind:5s} {str(arch):12s} N={N:4d} T={T:2d} tol=({abstol:.0e},{reltol:.0e}) scale={scale} steps "
          f"{min(n_st)}..{max(n_st)}: loss {r[0]:.1e} g_nn {r[1]:.1e} g_cond {r[2]:.1e}{flag}", flush=True)
    n_bad += "VIOLATION" in flag
print("worst:", {k: f"{v:.1e}" for k, v in worst.items()}, "violations:", n_bad)
sys.exit(1 if n_bad else 0)
