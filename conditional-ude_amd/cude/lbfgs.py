"""Host L-BFGS with backtracking line search driving the device loss+gradient.

Second training stage of the reference: `Optimization.solve(prob, LBFGS(linesearch=BackTracking()), maxiters)`
(src/parameter-estimation.jl:179-180; suppression/src/suppression_model.jl:168).  Optim.jl defaults are
restated: memory m = 10, scaleinvH0, InitialStatic (step 1 in every iteration), BackTracking(c_1 = 1e-4,
rho_hi = 0.5, rho_lo = 0.1, quadratic/cubic interpolation order 3, 1000 shrinks), gradient-norm stop g_tol = 1e-8.  Only vector algebra happens
here; every loss/gradient value comes from the HIP engine.

The algorithm is written once, as a generator that YIELDS the points it wants evaluated and is SENT (f, g):
  * `lbfgs(fg, x0)` drives one such generator with a callback -- the reference's serial use;
  * `lbfgs_batched(fg_batch, X0)` drives K of them in lock step, handing all pending points to ONE batched
    evaluation per round (cude_multistart_loss_grad: the restarts of `train` side by side).  Every problem sees
    exactly the sequence of values it would see alone, so the batched run reproduces K serial runs.
"""
import numpy as np


class LineSearchFailed(Exception):
    """LineSearches.LineSearchException: BackTracking ran out of shrinks.  Carries (alpha, f, g, n_eval) of the last
    step tried: Optim's perform_linesearch! takes that step (state.alpha = ex.alpha; state.x += alpha s) and the main
    loop then stops."""


def _backtracking(x, f0, dphi0, d, alpha0=1.0, c1=1e-4, rho_hi=0.5, rho_lo=0.1, max_iter=1000, max_finite=52):
    """LineSearches.BackTracking (order 3) as a generator.  Returns (alpha, f, g, n_eval)."""
    a1, a2 = alpha0, alpha0
    phi1 = f0
    n_eval = 0
    f, g = yield x + a2 * d
    n_eval += 1
    # halve until finite (the reference's loss returns Inf on a failed solve)
    it = 0
    while not np.isfinite(f) and it < max_finite:
        a1, a2 = a2, a2 * 0.5
        f, g = yield x + a2 * d
        n_eval += 1
        it += 1
    it = 0
    while f > f0 + c1 * a2 * dphi0:
        it += 1
        if it > max_iter:
            raise LineSearchFailed(a2, f, g, n_eval)
        if it == 1:
            a_tmp = -(dphi0 * a2 ** 2) / (2.0 * (f - f0 - dphi0 * a2))
        else:
            div = 1.0 / (a1 ** 2 * a2 ** 2 * (a2 - a1))
            a = (a1 ** 2 * (f - f0 - dphi0 * a2) - a2 ** 2 * (phi1 - f0 - dphi0 * a1)) * div
            b = (-a1 ** 3 * (f - f0 - dphi0 * a2) + a2 ** 3 * (phi1 - f0 - dphi0 * a1)) * div
            if abs(a) <= 2.220446049250313e-16:
                a_tmp = dphi0 / (2.0 * b)
            else:
                disc = max(b * b - 3.0 * a * dphi0, 0.0)
                a_tmp = (-b + np.sqrt(disc)) / (3.0 * a)
        a1 = a2
        a_new = a2 * rho_hi if np.isnan(a_tmp) else min(a_tmp, a2 * rho_hi)      # NaNMath.min / NaNMath.max
        a2 = max(a_new, a2 * rho_lo)
        phi1 = f
        f, g = yield x + a2 * d
        n_eval += 1
    return a2, f, g, n_eval


def lbfgs_steps(x0, maxiters=1000, m=10, g_tol=1e-8, callback=None, n_shared=None, reduce=None):
    """The L-BFGS iteration as a generator: yields points x, expects (f, g) to be sent back, returns
    dict(x, f, g, iterations, f_calls, converged).  Optim.jl's algorithm with its defaults (see cude_optim.h):
    InitialStatic step 1 in every iteration, (dx, dg) pairs kept whatever the sign of dx'dg, a non-descent direction
    replaced by -g in the same iteration.

    Sharded vectors: x = [shared (n_shared entries, replicated); local]; reduce(value, op) sums (op 0) or maximises
    (op 1) the local part of an inner product / max-norm over the ranks."""
    x = np.array(x0, dtype=np.float64)
    ns = x.size if reduce is None or n_shared is None else int(n_shared)

    def dot(a, b):
        if reduce is None:
            return float(a @ b)
        return float(a[:ns] @ b[:ns]) + float(reduce(float(a[ns:] @ b[ns:]), 0))

    def amax(a):
        loc = float(np.max(np.abs(a[ns:]))) if a.size > ns else 0.0
        sh = float(np.max(np.abs(a[:ns]))) if ns > 0 else 0.0
        return sh if reduce is None else max(sh, float(reduce(loc, 1)))

    f, g = yield x
    calls = 1
    S, Y, RHO = {}, {}, {}                    # slot -> vector, addressed like Optim's ring: pair k in slot (k-1) % m
    pseudo = 0
    it = 0
    flat = 0
    converged = bool(amax(g) <= g_tol) if np.isfinite(f) else False
    while it < maxiters and not converged and np.isfinite(f):
        pseudo += 1
        upper, lower = pseudo - 1, max(1, pseudo - m)
        q = g.copy()
        alphas = {}
        for k in range(upper, lower - 1, -1):
            sl = (k - 1) % m
            alphas[k] = RHO[sl] * dot(S[sl], q)
            q -= alphas[k] * Y[sl]
        if pseudo > 1:
            sl = (upper - 1) % m
            q *= dot(S[sl], Y[sl]) / dot(Y[sl], Y[sl])
        for k in range(lower, upper + 1):
            sl = (k - 1) % m
            b = RHO[sl] * dot(Y[sl], q)
            q += (alphas[k] - b) * S[sl]
        d = -q
        dphi0 = dot(g, d)
        if not (dphi0 < 0):
            pseudo = 1
            d = -g
            dphi0 = dot(g, d)
            if not (dphi0 < 0):
                break
        try:
            alpha, f_new, g_new, n_eval = yield from _backtracking(x, f, dphi0, d)
        except LineSearchFailed as e:
            alpha, f, g, n_eval = e.args
            calls += n_eval
            x = x + alpha * d
            it += 1
            break
        calls += n_eval
        s = alpha * d
        y = g_new - g
        x_new = x + s
        moved = np.abs(x_new - x)
        x_same = (float(np.max(moved[:ns])) if ns > 0 else 0.0) == 0.0 if reduce is None else \
            max(float(np.max(moved[:ns])) if ns > 0 else 0.0,
                float(reduce(float(np.max(moved[ns:])) if moved.size > ns else 0.0, 1))) == 0.0
        x = x_new
        f_prev, f, g = f, f_new, g_new
        it += 1
        stop = callback is not None and callback(x, f)
        flat = flat + 1 if abs(f_prev - f) == 0.0 else 0
        converged = bool(x_same) or bool(amax(g) <= g_tol) or flat > 1
        if stop:
            break
        if not converged:
            sy = dot(s, y)
            rho = np.inf if sy == 0.0 else 1.0 / sy
            if np.isinf(rho):
                pseudo = 0
            else:
                sl = (pseudo - 1) % m
                S[sl], Y[sl], RHO[sl] = s, y, rho
    return dict(x=x, f=f, g=g, iterations=it, f_calls=calls, converged=converged)


def lbfgs(fg, x0, maxiters=1000, m=10, g_tol=1e-8, callback=None, n_shared=None, reduce=None):
    """Minimise with L-BFGS.  fg(x) -> (f, g).  Returns dict(x, f, g, iterations, f_calls, converged)."""
    gen = lbfgs_steps(x0, maxiters, m, g_tol, callback, n_shared, reduce)
    x = next(gen)
    try:
        while True:
            f, g = fg(x)
            x = gen.send((float(f), np.array(g, dtype=np.float64)))
    except StopIteration as done:
        return done.value


def lbfgs_batched(fg_batch, X0, maxiters=1000, m=10, g_tol=1e-8, callbacks=None):
    """K independent L-BFGS runs in lock step.  fg_batch(X[K, n]) -> (f[K], g[K, n]) evaluates one point per
    problem; a problem that has finished keeps re-submitting its final point (its values are ignored), so the
    batch shape is fixed.  callbacks: optional list of K per-problem callbacks (x, f) -> stop.  Returns the list of
    the K result dicts of `lbfgs`."""
    X0 = np.asarray(X0, dtype=np.float64)
    K = X0.shape[0]
    gens = [lbfgs_steps(X0[k], maxiters, m, g_tol, None if callbacks is None else callbacks[k]) for k in range(K)]
    pending = np.stack([next(gn) for gn in gens])
    results = [None] * K
    while any(r is None for r in results):
        f, g = fg_batch(pending)
        for k in range(K):
            if results[k] is not None:
                continue
            try:
                pending[k] = gens[k].send((float(f[k]), np.array(g[k], dtype=np.float64)))
            except StopIteration as done:
                results[k] = done.value
                pending[k] = done.value["x"]
    return results
