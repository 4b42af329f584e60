"""Host L-BFGS with backtracking line search driving the device loss+gradient.

Second training stage of the reference: `Optimization.solve(prob, LBFGS(linesearch=BackTracking()), maxiters)`
(src/parameter-estimation.jl:179-180; suppression/src/suppression_model.jl:168).  Optim.jl defaults are
restated: memory m = 10, BackTracking(c_1 = 1e-4, rho_hi = 0.5, rho_lo = 0.1, quadratic/cubic
interpolation order 3, initial step 1), gradient-norm stop g_tol = 1e-8.  Only vector algebra happens
here; every loss/gradient value comes from the HIP engine.

The algorithm is written once, as a generator that YIELDS the points it wants evaluated and is SENT (f, g):
  * `lbfgs(fg, x0)` drives one such generator with a callback -- the reference's serial use;
  * `lbfgs_batched(fg_batch, X0)` drives K of them in lock step, handing all pending points to ONE batched
    evaluation per round (cude_multistart_loss_grad: the restarts of `train` side by side).  Every problem sees
    exactly the sequence of values it would see alone, so the batched run reproduces K serial runs.
"""
import numpy as np


def _backtracking(x, f0, g0, d, alpha0=1.0, c1=1e-4, rho_hi=0.5, rho_lo=0.1, max_iter=50):
    """LineSearches.BackTracking (order 3) as a generator.  Returns (alpha, f, g, n_eval) or None when no finite
    decrease is found."""
    dphi0 = float(g0 @ d)
    if not (dphi0 < 0):
        return None
    a1, a2 = alpha0, alpha0
    phi1 = phi2 = f0
    n_eval = 0
    f, g = yield x + a2 * d
    n_eval += 1
    # shrink until finite (the reference's solver returns Inf on failure)
    it = 0
    while not np.isfinite(f) and it < max_iter:
        a1, a2 = a2, a2 * 0.5
        f, g = yield x + a2 * d
        n_eval += 1
        it += 1
    phi1, phi2 = phi2, f
    it = 0
    while f > f0 + c1 * a2 * dphi0:
        it += 1
        if it > max_iter:
            return None
        if it == 1 or not np.isfinite(phi1):
            a_tmp = -(dphi0 * a2 ** 2) / (2.0 * (f - f0 - dphi0 * a2))
        else:
            div = 1.0 / (a1 ** 2 * a2 ** 2 * (a2 - a1))
            a = (a1 ** 2 * (f - f0 - dphi0 * a2) - a2 ** 2 * (phi1 - f0 - dphi0 * a1)) * div
            b = (-a1 ** 3 * (f - f0 - dphi0 * a2) + a2 ** 3 * (phi1 - f0 - dphi0 * a1)) * div
            if abs(a) < 1e-300:
                a_tmp = dphi0 / (2.0 * b)
            else:
                disc = max(b * b - 3.0 * a * dphi0, 0.0)
                a_tmp = (-b + np.sqrt(disc)) / (3.0 * a)
        a1 = a2
        if not np.isfinite(a_tmp):
            a_tmp = a2 * rho_hi
        a2 = min(max(a_tmp, a2 * rho_lo), a2 * rho_hi)
        phi1 = f
        f, g = yield x + a2 * d
        n_eval += 1
    return a2, f, g, n_eval


def lbfgs_steps(x0, maxiters=1000, m=10, g_tol=1e-8, callback=None):
    """The L-BFGS iteration as a generator: yields points x, expects (f, g) to be sent back, returns
    dict(x, f, g, iterations, f_calls, converged)."""
    x = np.array(x0, dtype=np.float64)
    f, g = yield x
    calls = 1
    S, Y, RHO = [], [], []
    it = 0
    converged = bool(np.max(np.abs(g)) <= g_tol) if np.isfinite(f) else False
    while it < maxiters and not converged and np.isfinite(f):
        q = g.copy()
        alphas = []
        for s, y, rho in zip(reversed(S), reversed(Y), reversed(RHO)):
            a = rho * (s @ q)
            alphas.append(a)
            q -= a * y
        if S:
            q *= (S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])
        for (s, y, rho), a in zip(zip(S, Y, RHO), reversed(alphas)):
            b = rho * (y @ q)
            q += (a - b) * s
        d = -q
        res = yield from _backtracking(x, f, g, d,
                                       alpha0=1.0 if S else min(1.0, 1.0 / max(np.linalg.norm(g), 1e-300)))
        if res is None:
            if not S:
                break
            S, Y, RHO = [], [], []          # reset to steepest descent once, as Optim does on a failed search
            it += 1
            continue
        alpha, f_new, g_new, n_eval = res
        calls += n_eval
        s = alpha * d
        y = g_new - g
        x = x + s
        sy = float(s @ y)
        if sy > 1e-300:
            S.append(s); Y.append(y); RHO.append(1.0 / sy)
            if len(S) > m:
                S.pop(0); Y.pop(0); RHO.pop(0)
        f_prev, f, g = f, f_new, g_new
        it += 1
        if callback is not None and callback(x, f):
            break
        converged = bool(np.max(np.abs(g)) <= g_tol) or abs(f_prev - f) == 0.0
    return dict(x=x, f=f, g=g, iterations=it, f_calls=calls, converged=converged)


def lbfgs(fg, x0, maxiters=1000, m=10, g_tol=1e-8, callback=None):
    """Minimise with L-BFGS.  fg(x) -> (f, g).  Returns dict(x, f, g, iterations, f_calls, converged)."""
    gen = lbfgs_steps(x0, maxiters, m, g_tol, callback)
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
