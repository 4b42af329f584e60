"""cude_fit_conditional: all subjects' 1-D fits of the conditional parameter on the device, against the same search
driven from the host (one forward call per probe, numpy bookkeeping) and against the defining property of its result."""
import math

import numpy as np
import pytest
import torch  # noqa: F401

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu


def _host_search(f, lo, hi, n_grid, iters):
    """The algorithm of cude_fit_conditional in numpy (coarse scan, bracket, golden section, midpoint)."""
    step = (hi - lo) / (n_grid - 1)
    grid = np.array([hi if k == n_grid - 1 else math.fma(k, step, lo) if hasattr(math, "fma") else lo + k * step
                     for k in range(n_grid)])
    vals = np.stack([f(np.full_like(f.template, g)) for g in grid])
    vals = np.where(np.isfinite(vals), vals, np.inf)
    k = np.clip(np.argmin(vals, axis=0), 1, n_grid - 2)
    a, b = lo + (k - 1) * step, lo + (k + 1) * step
    gr = (math.sqrt(5) - 1) / 2
    c, d = b - gr * (b - a), a + gr * (b - a)
    for _ in range(iters):
        fc, fd = f(c), f(d)
        fc, fd = np.where(np.isfinite(fc), fc, np.inf), np.where(np.isfinite(fd), fd, np.inf)
        left = fc < fd
        b = np.where(left, d, b)
        a = np.where(left, a, c)
        c, d = b - gr * (b - a), a + gr * (b - a)
    x = 0.5 * (a + b)
    return x, f(x)


class _Objective:
    def __init__(self, eng, N, w=0.0, mu=0.0):
        self.eng, self.w, self.mu, self.template = eng, w, mu, np.zeros(N)

    def __call__(self, x):
        self.eng.set_params(None, x)
        return self.eng.forward(want_sse=True)["sse"] + self.w * (x - self.mu) ** 2


@pytest.mark.parametrize("model,w", [("cpep", 0.0), ("cpep", 0.35), ("supp", 0.0)])
def test_device_search_equals_host_search(model, w):
    from cude.engine import Engine
    if model == "cpep":
        c = make_cpep_case(150, (2, 4, 2))
        eng = Engine("cpep", (2, 4, 2), n_steps=30, n_state=2)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        nn, N, box = c["nn"], 150, (-4.0, 3.0)
    else:
        c = make_supp_case(90)
        eng = Engine("supp", c["arch"], n_steps=30)
        eng.set_population_supp(c["tp"], c["data"])
        nn, N, box = c["nn"], 90, (-6.0, 4.0)
    eng.set_params(nn, np.zeros(N))
    x, obj, sse = eng.fit_conditional(box[0], box[1], 41, 40, w, -0.6)
    f = _Objective(eng, N, w, -0.6)
    hx, hf = _host_search(f, box[0], box[1], 41, 40)
    # same probes and decisions until the two golden probes differ by less than rounding (the SSE is flat to 1e-16
    # within ~1e-7 of the minimiser), hence the same minimiser to that width and the same objective value
    assert np.max(np.abs(x - hx)) < 1e-6
    assert np.allclose(obj, hf, rtol=1e-10, atol=1e-14)
    assert np.allclose(sse + w * (x - (-0.6)) ** 2, obj, rtol=1e-12, atol=1e-14)
    # the result is a local minimiser of each subject's objective
    for d in (1e-4, -1e-4):
        inside = (x + d > box[0]) & (x + d < box[1])
        assert np.all(f(x + d)[inside] >= obj[inside] - 1e-10)
    # the context's conditional parameters are untouched by the search itself (the host search set them last)
    eng.set_params(None, np.full(N, 0.25))
    eng.fit_conditional(box[0], box[1], 11, 3)
    assert np.array_equal(eng.get_params()[1], np.full(N, 0.25))
    eng.close()


def test_failed_solves_and_argument_errors():
    from cude.engine import Engine
    from cude._lib import CudeError
    c = make_cpep_case(70, (2, 4, 2))
    G = c["G"].copy()
    G[9, 2] = np.nan                                           # this subject fails at every probe
    eng = Engine("cpep", (2, 4, 2), n_steps=30, n_state=2)
    eng.set_population_cpep(c["tp"], G, c["obs"], c["age"], c["t2dm"])
    with pytest.raises(CudeError):
        eng.fit_conditional(-4.0, 3.0)                         # shared parameters not set
    eng.set_params(c["nn"], None)
    x, obj, sse = eng.fit_conditional(-4.0, 3.0, 21, 20)
    ok = np.ones(70, bool)
    ok[9] = False
    assert np.all(np.isfinite(obj[ok])) and np.isinf(obj[9]) and np.all((x >= -4.0) & (x <= 3.0))
    for bad in ((1.0, 1.0, 21, 20), (-4.0, 3.0, 2, 20), (-4.0, 3.0, 21, 0)):
        with pytest.raises(CudeError):
            eng.fit_conditional(*bad)
    eng.close()
