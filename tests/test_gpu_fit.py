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


@pytest.mark.parametrize("model", ["cpep", "supp"])
def test_fit_and_profile_against_the_oracle(model):
    """The per-subject fits and likelihood profiles against the CPU ORACLE (not the device's own forward kernel):
    cude_profile_conditional rows equal the oracle's per-subject SSE at every scan value (1e-10), and the fitted
    conditional parameter of every subject is the minimiser of the oracle's own per-subject objective -- found
    independently with scipy's bounded Brent search on the C oracle -- wherever that minimum is interior and unique."""
    import c_oracle as co
    from scipy.optimize import minimize_scalar
    from cude.engine import Engine
    if model == "cpep":
        arch, N = (2, 4, 2), 24
        c = make_cpep_case(N, arch)
        eng = Engine("cpep", arch, n_steps=30)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])

        def sse_all(x):
            return co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], x, 30, 2, want_grad=False)["sse"]
        lo, hi = -3.0, 2.0
    else:
        arch, N = (4, 3, 5), 16
        c = make_supp_case(N, arch)
        eng = Engine("supp", arch, n_steps=30)
        eng.set_population_supp(c["tp"], c["data"])

        def sse_all(x):          # the oracle's per-subject SSE is already divided by scale^2 (suppression_model.jl:126-128)
            return co.supp(c["tp"], c["data"], arch, c["nn"], x, 0.0, 30, want_grad=False)["sse"]
        lo, hi = -2.0, 2.0
    eng.set_params(c["nn"], np.zeros(N))
    values = np.linspace(lo, hi, 41)
    prof = eng.profile_conditional(values)
    ref = np.stack([sse_all(np.full(N, v)) for v in values])
    assert prof.shape == ref.shape and np.max(np.abs(prof - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref)))
    x_dev, f_dev, sse_dev = eng.fit_conditional(lo, hi, 41, 48)
    eng.close()
    assert np.max(np.abs(sse_dev - sse_all(x_dev))) <= 1e-10 * max(1.0, np.max(sse_dev))
    checked = 0
    for i in range(N):
        k = int(np.argmin(ref[:, i]))
        others = np.delete(ref[:, i], [max(k - 1, 0), k, min(k + 1, 40)])
        if k in (0, 40) or others.min() < ref[k, i] * (1 + 1e-3) + 1e-9:
            continue                                    # boundary minimum or a second basin as deep: nothing unique to compare
        one = lambda v, i=i: sse_all(np.where(np.arange(N) == i, v, 0.0))[i]       # noqa: E731
        r = minimize_scalar(one, bounds=(values[k - 1], values[k + 1]), method="bounded", options=dict(xatol=1e-10))
        assert abs(x_dev[i] - r.x) < 2e-6 and f_dev[i] <= r.fun * (1 + 1e-10) + 1e-14, (i, x_dev[i], r.x)
        checked += 1
    assert checked >= N // 2


@pytest.mark.parametrize("case", ["cpep-time-split", "cpep-adaptive", "cpep-one-lane", "supp", "supp-adaptive"])
def test_several_probes_per_launch_give_the_same_search(case):
    """Option "fit_spec": the grid values of the coarse scan as parameter sets of one launch, and -- a golden-section step
    has two outcomes -- the probes of the next d steps as a heap of 2^d - 1 brackets evaluated together.  The same
    expressions on the same values as the one-probe-per-launch form (fit_spec = 0): minimisers, objectives and SSEs bit for
    bit, for every depth, when the step count is not a multiple of the depth, with and without a penalty."""
    from cude.engine import Engine
    if case.startswith("cpep"):
        N, arch = 150, (2, 4, 2)
        c = make_cpep_case(N, arch)
        box = (-4.0, 3.0)
    else:
        c = make_supp_case(90)
        N, arch, box = 90, c["arch"], (-6.0, 4.0)
    out = []
    for depth in (0, 1, 2, 3, 4, -1):
        if case.startswith("cpep"):
            eng = Engine("cpep", arch, n_steps=0 if case == "cpep-adaptive" else 30, n_state=2)
            if case == "cpep-one-lane":
                eng.set_option("cpep_path", "1")
            eng.set_option("fit_spec", depth)
            eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        else:
            eng = Engine("supp", arch, n_steps=0 if case == "supp-adaptive" else 30)
            eng.set_option("fit_spec", depth)
            eng.set_population_supp(c["tp"], c["data"])
        eng.set_params(c["nn"], np.zeros(N))
        r = list(eng.fit_conditional(box[0], box[1], 21, 13)) + list(eng.fit_conditional(box[0], box[1], 41, 24, 0.35, -0.6))
        out.append(r)
        eng.close()
    assert np.all(np.isfinite(out[0][1]))
    for r in out[1:]:
        for a, b in zip(out[0], r):
            assert np.array_equal(a, b)
