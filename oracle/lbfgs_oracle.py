"""Oracle for the SECOND TRAINING STAGE of the reference -- TEST INFRASTRUCTURE ONLY (tests/ import it; the product never
does: conditional-ude_amd/csrc/cude_optim.h and cude/lbfgs.py are the product's statements).

What it restates.  Every training driver of the reference finishes with

    Optimization.solve(optprob, LBFGS(linesearch = LineSearches.BackTracking()), maxiters = ...)

(src/parameter-estimation.jl:179-180 `_optimize`; suppression/src/suppression_model.jl:166-168; src/saem.jl:128 uses
the same optimiser for the M-step of the symbolic model).  The algorithm lives in third-party packages that are NOT in
the reference tree: Optim.jl (no compat bound in the reference's Project.toml: the 1.x series that Optimization 4 /
OptimizationOptimJL resolve to), LineSearches.jl (Project.toml: "7.3"), NLSolversBase.jl.  Restated here from their
published algorithm with the defaults the reference's call reaches them with:

  Optim.LBFGS            m = 10, alphaguess = InitialStatic(alpha = 1.0, scaled = false), scaleinvH0 = true, no
                         preconditioner, Flat manifold
  LineSearches.BackTracking
                         c_1 = 1e-4, rho_hi = 0.5, rho_lo = 0.1, iterations = 1000, order = 3, maxstep = Inf,
                         at most -log2(eps) = 52 halvings to reach a finite value
  Optim.Options          through OptimizationOptimJL: iterations = maxiters, x_abstol = x_reltol = f_abstol = f_reltol
                         = 0, g_abstol = 1e-8 (max-norm), successive_f_tol = 1, allow_f_increases = true

Route.  Deliberately NOT the arithmetic of the product (which follows Optim's own code: two-loop recursion over a ring
of (dx, dg) pairs, closed-form interpolation coefficients): the search direction is -H g with the inverse-Hessian
approximation H built as a DENSE matrix from the stored pairs by the BFGS product formula (Nocedal & Wright (7.16),
(7.19), initial H0 = gamma I with (7.20)), and the cubic / quadratic trial steps come from fitting the interpolating
polynomial with a linear solve (Nocedal & Wright section 3.5) instead of the expanded coefficients.  In exact arithmetic
both routes are the same map; in floating point they differ by rounding, so agreement of iterates to ~1e-8 over tens of
iterations is a check of the algorithm, not of shared code.

Pin.  PARITY UNPINNED at the level of iterates: the reference stores no optimiser trace.  What it does hold is the
distribution of FINAL objectives of whole training runs (suppression/results/*.csv, source_data/*.jld2), which the
product is held to end to end (tools/e2e_suppression.py) -- a pin of the whole recipe, not of this function.
Two details are taken from the packages' source as published and are flagged where they occur: the quadratic fallback
of the cubic step (LineSearches writes dphi_0 / (2 b), sign included) and the history reset only on an INFINITE rho.
"""
import numpy as np

C1, RHO_HI, RHO_LO = 1e-4, 0.5, 0.1
LS_ITERATIONS, LS_FINITE = 1000, 52
EPS = np.finfo(np.float64).eps


class _Pairs:
    """Optim's ring of (dx, dg, rho) triples: the pair stored at pseudo-iteration k lives in slot mod1(k, m)."""

    def __init__(self, m):
        self.m = m
        self.slot = {}

    def store(self, k, dx, dg, rho):
        self.slot[(k - 1) % self.m] = (dx.copy(), dg.copy(), rho)

    def get(self, k):
        return self.slot[(k - 1) % self.m]


def direction_dense(g, pairs, pseudo_iteration, m):
    """-H g with H assembled densely from the pairs `pseudo_iteration - m .. pseudo_iteration - 1` (those >= 1).
    Equals Optim's twoloop! (scaleinvH0: gamma = dx'dg / dg'dg of the NEWEST pair, not applied when
    pseudo_iteration == 1, i.e. on the first step and on the first step after a reset)."""
    n = g.size
    ks = [k for k in range(pseudo_iteration - m, pseudo_iteration) if k >= 1]
    H = np.eye(n)
    if pseudo_iteration > 1:
        dx, dg, _ = pairs.get(pseudo_iteration - 1)
        H *= float(dx @ dg) / float(dg @ dg)
    eye = np.eye(n)
    for k in ks:                                   # oldest first: H <- V' H V + rho s s'
        s, y, rho = pairs.get(k)
        V = eye - rho * np.outer(y, s)
        H = V.T @ H @ V + rho * np.outer(s, s)
    return -(H @ g)


def _cubic_step(a1, phi_a1, a2, phi_a2, phi0, dphi0):
    """Minimiser of the cubic through phi(0), phi'(0), phi(a1), phi(a2) (Nocedal & Wright (3.59)), coefficients by a
    linear solve.  a1 = previous step, a2 = current (smaller) step."""
    M = np.array([[a1 ** 3, a1 ** 2], [a2 ** 3, a2 ** 2]])
    r = np.array([phi_a1 - phi0 - dphi0 * a1, phi_a2 - phi0 - dphi0 * a2])
    with np.errstate(all="ignore"):
        try:
            a, b = np.linalg.solve(M, r)
        except np.linalg.LinAlgError:
            return np.nan
        if abs(a) <= EPS:                          # isapprox(a, 0, atol = eps): the cubic is a parabola
            return dphi0 / (2.0 * b)               # (as published by LineSearches, sign included)
        disc = max(b * b - 3.0 * a * dphi0, 0.0)
        return (-b + np.sqrt(disc)) / (3.0 * a)


def backtracking(phi, phi0, dphi0, alpha0=1.0):
    """LineSearches.BackTracking(order = 3).  phi(alpha) -> value.  Returns (alpha, phi(alpha), n_evaluations, ok);
    ok = False is the LineSearchException (iterations exhausted) with the last step tried."""
    a1 = a2 = alpha0
    phi_1 = phi(a2)
    phi_prev = phi0
    n_eval = 1
    k = 0
    while not np.isfinite(phi_1) and k < LS_FINITE:
        k += 1
        a1, a2 = a2, a2 / 2.0
        phi_1 = phi(a2)
        n_eval += 1
    it = 0
    while phi_1 > phi0 + C1 * a2 * dphi0:
        it += 1
        if it > LS_ITERATIONS:
            return a2, phi_1, n_eval, False
        with np.errstate(all="ignore"):
            if it == 1:                            # parabola through phi(0), phi'(0), phi(a2): its minimiser
                a_tmp = -(dphi0 * a2 ** 2) / (2.0 * (phi_1 - phi0 - dphi0 * a2))
            else:
                a_tmp = _cubic_step(a1, phi_prev, a2, phi_1, phi0, dphi0)
        a1 = a2
        a_tmp = a2 * RHO_HI if np.isnan(a_tmp) else min(a_tmp, a2 * RHO_HI)     # NaNMath.min / NaNMath.max
        a2 = max(a_tmp, a2 * RHO_LO)
        phi_prev, phi_1 = phi_1, phi(a2)
        n_eval += 1
    return a2, phi_1, n_eval, True


def lbfgs_oracle(fg, x0, maxiters=1000, m=10, g_tol=1e-8, keep_trace=True):
    """Minimise fg(x) -> (f, g) as the reference's second stage does.  Returns dict(x, f, iterations, f_calls,
    converged, ls_failed, trace) with trace = [(x_k, f_k)] after every iteration (k = 0: the start)."""
    x = np.array(x0, dtype=np.float64).copy()
    f, g = fg(x)
    f, g = float(f), np.array(g, dtype=np.float64)
    f_calls = 1
    trace = [(x.copy(), f)] if keep_trace else None
    pairs = _Pairs(m)
    pseudo = 0
    iteration = 0
    counter_f_tol = 0
    converged = bool(np.isfinite(f) and np.max(np.abs(g)) <= g_tol)       # initial_convergence
    ls_failed = False
    cache = {}

    def value_at(xt):
        ft, gt = fg(xt)
        cache["x"], cache["f"], cache["g"] = xt, float(ft), np.array(gt, dtype=np.float64)
        return float(ft)

    while not converged and iteration < maxiters:
        if not np.isfinite(f):
            break                                  # (Optim would walk on through NaNs to maxiters: same answer)
        iteration += 1
        # ---- update_state!
        pseudo += 1
        s = direction_dense(g, pairs, pseudo, m)
        dphi0 = float(g @ s)
        if not (dphi0 < 0.0):                      # dphi_0 >= 0 (or NaN): reset_search_direction!
            pseudo = 1
            s = -g
            dphi0 = float(g @ s)
            if not (dphi0 < 0.0):
                iteration -= 1
                break
        x_prev, f_prev, g_prev = x, f, g
        alpha, f_new, n_eval, ok = backtracking(lambda a: value_at(x_prev + a * s), f_prev, dphi0)
        f_calls += n_eval
        dx = alpha * s
        x = x_prev + dx
        if not ok:                                 # LineSearchException: the step is taken, the loop breaks
            f, g = cache["f"], cache["g"]
            ls_failed = True
            if keep_trace:
                trace.append((x.copy(), f))
            break
        # ---- update_g!  (the last line-search evaluation was at this x: value and gradient are those)
        f, g = cache["f"], cache["g"]
        if keep_trace:
            trace.append((x.copy(), f))
        # ---- assess_convergence (all x / f tolerances are 0)
        x_converged = float(np.max(np.abs(x - x_prev))) <= 0.0
        f_converged = abs(f - f_prev) <= 0.0
        g_converged = bool(np.max(np.abs(g)) <= g_tol)
        counter_f_tol = counter_f_tol + 1 if f_converged else 0
        converged = x_converged or g_converged or counter_f_tol > 1
        # ---- update_h!
        if not converged:
            dg = g - g_prev
            with np.errstate(all="ignore"):
                rho = float(np.float64(1.0) / np.float64(dx @ dg))     # (inf on 0, as Julia's 1 / 0.0)
            if np.isinf(rho):
                pseudo = 0
            else:
                pairs.store(pseudo, dx, dg, rho)
    return dict(x=x, f=f, iterations=iteration, f_calls=f_calls, converged=bool(converged), ls_failed=ls_failed,
                trace=trace)
