"""Second training stage: the product's L-BFGS + BackTracking -- the C++ state machine of
csrc/cude_optim.h through cude_lbfgs_minimize, and its Python statement cude/lbfgs.py -- against the independent
restatement oracle/lbfgs_oracle.py (dense inverse Hessian from the stored pairs, interpolation steps by a linear
solve), ITERATE BY ITERATE, on the oracle's own objectives at the reference's sizes: the c-peptide cUDE on the 57
training subjects it was trained on (src/parameter-estimation.jl:126-140, 179-180) and the suppression cUDE on its 37
subjects (suppression/src/suppression_model.jl:117-130, 166-168).  CPU only (the objectives are the C oracle's)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cpep_objective(n_steps=40):
    import c_oracle as co
    d = np.load(os.path.join(GOLD, "ohashi_cude.npz"))
    tp = d["timepoints"]
    train = np.isin(d["subject_no"], d["train_subject_numbers"])
    idx = np.nonzero(train)[0][:57]
    G, cp, age, t2 = d["glucose"][idx], d["cpeptide"][idx], d["ages"][idx], d["t2dm"][idx]
    arch, P = (2, 4, 2), 37

    def fg(x):
        r = co.cpep(tp, G, cp, age, t2, arch, x[:P], x[P:], n_steps, 2, method="reverse")
        if r["n_failed"] or not np.isfinite(r["loss"]):
            return np.inf, np.zeros_like(x)
        return r["loss"], np.concatenate([r["g_nn"], r["g_beta"]])
    rng = np.random.default_rng(4)
    nn0 = d["nn_2x4x4x1"][3] * (1.0 + 0.2 * rng.standard_normal(P))        # a stored optimum, knocked off it
    x0 = np.concatenate([nn0, d["betas_train"][3] + 0.3 * rng.standard_normal(57)])
    return fg, x0


def _supp_objective(lam=0.0, n_steps=30):
    import c_oracle as co
    d = np.load(os.path.join(GOLD, "suppression_lambda0.npz"))
    tp, data = d["timepoints"], d["group_data"]
    arch, P, N = (4, 3, 5), 67, 37

    def fg(x):
        r = co.supp(tp, data, arch, x[:P], x[P:], lam, n_steps, method="reverse")
        if r["n_failed"] or not np.isfinite(r["loss"]):
            return np.inf, np.zeros_like(x)
        return r["loss"], np.concatenate([r["g_nn"], r["g_theta"]])
    rng = np.random.default_rng(1)
    x0 = np.concatenate([d["nn_4x3x5x1"][0] * (1.0 + 0.05 * rng.standard_normal(P)), 0.5 * rng.standard_normal(N)])
    return fg, x0


def _own_sensitivity(fg, x0, ks, n_probe=4):
    """How far the ORACLE's iterates move when its start moves by one unit in the last place of one coordinate: the
    resolution at which iterate k of this problem is defined at all.  L-BFGS on these objectives amplifies rounding
    exponentially (c-peptide: 1e-17 at iteration 1, 1e-12 at 21, 1e-6 at 50; suppression: O(1) after ~10 iterations --
    an Armijo decision flips), so a fixed tolerance over 50 iterations cannot be met by ANY two correct statements;
    the two product statements differ from each other by the same amounts."""
    from lbfgs_oracle import lbfgs_oracle
    base = lbfgs_oracle(fg, x0, maxiters=max(ks))
    sens = {k: 0.0 for k in ks}
    rng = np.random.default_rng(99)
    for _ in range(n_probe):
        xp = x0.copy()
        j = int(rng.integers(x0.size))
        xp[j] = np.nextafter(xp[j], np.inf)
        r = lbfgs_oracle(fg, xp, maxiters=max(ks))
        for k in ks:
            if k < len(r["trace"]) and k < len(base["trace"]):
                sens[k] = max(sens[k], float(np.max(np.abs(r["trace"][k][0] - base["trace"][k][0]))))
    return base, sens


# iterations up to which the problem itself is well enough conditioned for a FIXED bar (measured sensitivities above)
_FIXED_BAR = {"cpep": (21, 1e-10), "supp": (5, 1e-10)}


@pytest.mark.parametrize("problem", ["cpep", "supp"])
def test_native_lbfgs_follows_the_oracle_iterate_by_iterate(problem):
    from cude.engine import lbfgs_minimize
    fg, x0 = _cpep_objective() if problem == "cpep" else _supp_objective()
    ks = [1, 2, 3, 5, 8, 13, 21, 34, 50]
    ref, sens = _own_sensitivity(fg, x0, ks)
    assert ref["iterations"] == 50 and not ref["ls_failed"]
    assert ref["f"] < 0.7 * ref["trace"][0][1]                             # it is optimising, not idling
    k_fixed, bar = _FIXED_BAR[problem]
    from lbfgs_oracle import lbfgs_oracle
    for k in ks:
        r = lbfgs_minimize(fg, x0, k)                 # the product hands out only its final point: stop it at k
        xk, fk = ref["trace"][k]
        dx = float(np.max(np.abs(r["x"] - xk)))
        assert r["iterations"] == k
        assert dx <= 1e-12 + 100.0 * sens[k], (k, dx, sens[k])             # as close as the problem defines iterate k
        if k <= k_fixed:
            assert dx <= bar, (k, dx)
            assert abs(r["f"] - fk) <= 1e-9 * max(1.0, abs(fk))
            assert r["f_calls"] == lbfgs_oracle(fg, x0, maxiters=k, keep_trace=False)["f_calls"]
    # at the end of the 50 iterations: the same quality of optimum (the paths may have parted, the basin has not)
    r = lbfgs_minimize(fg, x0, 50)
    assert abs(r["f"] - ref["f"]) <= 0.05 * ref["f"]


@pytest.mark.parametrize("problem", ["cpep", "supp"])
def test_python_statement_follows_the_oracle(problem):
    """cude/lbfgs.py (the generator the batched Python drivers use) against the same oracle, every iterate through
    its callback."""
    from cude.lbfgs import lbfgs
    fg, x0 = _cpep_objective() if problem == "cpep" else _supp_objective(lam=0.01)
    K = 40
    ks = list(range(1, K + 1))
    ref, sens = _own_sensitivity(fg, x0, ks, n_probe=3)
    seen = []
    r = lbfgs(fg, x0, maxiters=K, callback=lambda x, f: seen.append((x.copy(), f)) and False)
    assert r["iterations"] == ref["iterations"] == K
    k_fixed, bar = _FIXED_BAR[problem]
    for k, (x, f) in enumerate(seen, start=1):
        xk, fk = ref["trace"][k]
        dx = float(np.max(np.abs(x - xk)))
        assert dx <= 1e-12 + 100.0 * sens[k], (k, dx, sens[k])
        if k <= k_fixed:
            assert dx <= bar and abs(f - fk) <= 1e-9 * max(1.0, abs(fk)), (k, dx)
    if problem == "cpep":
        assert r["f_calls"] == ref["f_calls"]


def test_oracle_direction_equals_the_two_loop_recursion():
    """The dense inverse-Hessian product of the oracle against a two-loop recursion written out here (Nocedal & Wright
    algorithm 7.4 with Optim's ring addressing), including a wrapped ring (more than m pairs) and the unscaled first
    step."""
    from lbfgs_oracle import _Pairs, direction_dense
    rng = np.random.default_rng(0)
    n, m = 9, 4
    pairs = _Pairs(m)
    for pseudo in range(1, 9):
        g = rng.standard_normal(n)
        d = direction_dense(g, pairs, pseudo, m)
        q = g.copy()
        ks = [k for k in range(pseudo - m, pseudo) if k >= 1]
        al = {}
        for k in reversed(ks):
            s, y, rho = pairs.get(k)
            al[k] = rho * (s @ q)
            q -= al[k] * y
        if pseudo > 1:
            s, y, _ = pairs.get(pseudo - 1)
            q *= (s @ y) / (y @ y)
        for k in ks:
            s, y, rho = pairs.get(k)
            q += (al[k] - rho * (y @ q)) * s
        assert np.allclose(d, -q, rtol=1e-11, atol=1e-13)
        s = rng.standard_normal(n)
        y = s * (0.5 + rng.random(n)) + 0.1 * rng.standard_normal(n)
        pairs.store(pseudo, s, y, 1.0 / (s @ y))


def test_oracle_line_search_and_stopping_rules():
    from lbfgs_oracle import backtracking, lbfgs_oracle
    # Armijo holds at the first step: one evaluation
    a, f, n, ok = backtracking(lambda a: (a - 1.0) ** 2, 1.0, -2.0)
    assert ok and a == 1.0 and n == 1 and f == 0.0
    # quadratic interpolation lands on the minimiser of a parabola in one shrink (within [rho_lo, rho_hi] of the step)
    phi = lambda a: (4.0 * a - 1.0) ** 2
    a, f, n, ok = backtracking(phi, 1.0, -8.0)
    assert ok and n == 2 and abs(a - 0.25) < 1e-15
    # a non-finite region is backed out of by halving, without touching the interpolation state
    phi = lambda a: np.inf if a > 0.3 else (a - 0.2) ** 2
    a, f, n, ok = backtracking(phi, 0.04, -0.4)
    assert ok and a == 0.25 and n == 3
    # running out of shrinks: the exception's step is the last one tried.  (In floating point the Armijo test passes
    # by rounding once the step is tiny unless phi stays ABOVE phi(0) = 0 at every representable step.)
    a, f, n, ok = backtracking(lambda a: 1e-300 if a > 0 else 0.0, 0.0, -1.0)
    assert not ok and n == 1001 and a > 0 and f == 1e-300
    # stopping rules: already converged; gradient tolerance; identical objective twice in a row
    quad = lambda x: (0.5 * float(x @ x), x.copy())
    r = lbfgs_oracle(quad, np.zeros(3), 10)
    assert r["converged"] and r["iterations"] == 0 and r["f_calls"] == 1
    r = lbfgs_oracle(quad, np.ones(3), 10)
    assert r["converged"] and r["iterations"] == 1 and np.all(r["x"] == 0.0)
    # LineSearchException: the step IS taken (Optim's perform_linesearch!), then the run stops
    up = lambda x: ((1e-300 if np.any(x != 0) else 0.0), -np.ones_like(x))  # no step along -g is ever accepted
    r = lbfgs_oracle(up, np.zeros(2), 5)
    assert r["ls_failed"] and r["iterations"] == 1 and r["f_calls"] == 1002 and np.all(r["x"] > 0) and r["f"] == 1e-300


def test_product_takes_the_last_trial_when_the_line_search_runs_out():
    """ADVICE (round 2): on a LineSearchException Optim moves to the last step tried before stopping; both product
    statements do the same and agree with the oracle on point, value and call count."""
    from cude.engine import lbfgs_minimize
    from cude.lbfgs import lbfgs
    from lbfgs_oracle import lbfgs_oracle
    up = lambda x: ((1e-300 if np.any(x != 0) else 0.0), -np.ones_like(x))
    o = lbfgs_oracle(up, np.zeros(2), 5)
    with np.errstate(all="ignore"):
        a, b = lbfgs_minimize(up, np.zeros(2), 5), lbfgs(up, np.zeros(2), maxiters=5)
    for r in (a, b):
        assert r["iterations"] == 1 and r["f_calls"] == o["f_calls"] == 1002 and not r["converged"]
        # the point moved to the last trial (its size, ~1e-312, is interpolation arithmetic in the subnormal range)
        assert np.all(r["x"] > 0) and np.all(r["x"] < 1e-290) and r["f"] == o["f"] == 1e-300
    assert np.array_equal(a["x"], b["x"])
