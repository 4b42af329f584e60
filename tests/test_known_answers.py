"""Known answers produced by the reference itself.

`suppression/results/lambda=1.0.jld2` holds the 25 final objectives of the reference's most regularised run
(`losses[n] = res.objective`, suppression/suppression.jl:57) together with the 25 trained networks.  At lambda = 1 the
L2 term has driven every weight to ~1e-7: the network output is the constant softplus(b_out), the loss no longer
depends on the conditional parameters (which the reference did not save), and

    suppression_loss(p, (prob, group_data, timepoints, 1.0))        (suppression/src/suppression_model.jl:117-130)

is a function of STORED quantities only.  It exercises the mechanistic right-hand side (:88-95), u0 = data[:, 1, :],
`scale = mean(maximum(data, dims=2), dims=3)` (:126), the normalisation by the number of subjects, the L2 term
`lambda * sum(abs2, neural)` (:128), softplus, the data layout and the time grid.  The reference integrated with
adaptive Tsit5 (reltol 1e-3, abstol 1e-6), so its own number carries an error of ~1e-6: the converged fixed-step
value differs from it by 1.26e-6 (3e-7 relative), identically for all 25 models.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARCH = (4, 3, 5)


def _load():
    g1 = dict(np.load(os.path.join(GOLD, "suppression_lambda1.npz")))
    g0 = np.load(os.path.join(GOLD, "suppression_lambda0.npz"))
    return g1["nn_4x3x5x1"], g1["losses"], g0["group_data"], g0["timepoints"]


def test_oracle_reproduces_the_reference_objectives():
    import c_oracle as co
    nns, stored, data, tp = _load()
    assert nns.shape == (25, 67) and np.max(np.abs(nns[:, :60])) < 1e-5          # collapsed hidden layers
    rng = np.random.default_rng(0)
    for n in range(25):
        theta = rng.uniform(-3.0, 3.0, data.shape[2])                           # irrelevant, as claimed
        got = co.supp(tp, data, ARCH, nns[n], theta, 1.0, 240, want_grad=False)["loss"]
        assert abs(got - stored[n]) < 2e-6, (n, got, stored[n])
        assert abs((got - stored[n]) - 1.2609e-6) < 2e-9                          # the reference solver's own error
    # and the value is converged in the step count: S = 240 vs S = 960
    a = co.supp(tp, data, ARCH, nns[0], np.zeros(37), 1.0, 240, want_grad=False)["loss"]
    b = co.supp(tp, data, ARCH, nns[0], np.zeros(37), 1.0, 960, want_grad=False)["loss"]
    assert abs(a - b) < 1e-10


def test_adaptive_restatement_reproduces_the_reference_to_1e_minus_10():
    """The oracle's ADAPTIVE mode restates what the reference actually ran: OrdinaryDiffEq's Tsit5 with its default
    tolerances (abstol 1e-6, reltol 1e-3), PI step-size control, initial-step heuristic and `saveat` interpolation
    (cude_oracle.solve_adaptive).  With it the stored objective is reproduced to 4e-10 -- i.e. the same sequence of
    accepted steps -- which pins the solver restatement itself, and identifies the 1.26e-6 of the fixed-step path as
    the reference solver's own discretisation error."""
    import cude_oracle as o
    nns, stored, data, tp = _load()
    scale = o.supp_scale(data)
    N = data.shape[2]
    for n in (0, 13):
        nn = nns[n]
        total = 0.0
        for i in range(N):
            et = float(np.exp(0.1 * i - 2.0))                                   # any value: the loss does not see it
            rhs = lambda t, u: [float(v) for v in o.supp_rhs(np, nn, et, ARCH, t, [np.float64(x) for x in u])]
            sol = o.solve_adaptive(rhs, list(data[:, 0, i]), list(tp), abstol=1e-6, reltol=1e-3)
            for ti in range(len(tp)):
                for s in range(3):
                    r = (sol[ti][s] - data[s, ti, i]) / scale[s]
                    total += r * r
        loss = total / N + 1.0 * float(np.sum(nn * nn))
        assert abs(loss - stored[n]) < 2e-9, (n, loss, stored[n])


def _load_validation():
    g1 = np.load(os.path.join(GOLD, "suppression_lambda1.npz"))
    g0 = np.load(os.path.join(GOLD, "suppression_lambda0.npz"))
    sets = [(g0["validation_data"], g1["losses_valid"]), (g0["validation_data_nonoise"], g1["losses_valid_nonoise"])]
    return g1["nn_4x3x5x1"], sets, g0["timepoints"]


def test_oracle_reproduces_the_reference_validation_objectives():
    """The same file stores, per network, the objective `validate_suppression_model` returned on the two validation
    sets (suppression/suppression.jl:59-65): `suppression_loss` with lambda = 0 and the network held fixed
    (suppression_model.jl:179-187).  With the collapsed networks it is again independent of the fitted conditional
    parameters: 2 x 25 further known answers on two further data sets (30 subjects each, their own `scale`)."""
    import c_oracle as co
    nns, sets, tp = _load_validation()
    rng = np.random.default_rng(2)
    offsets = []
    for data, stored in sets:
        assert data.shape == (3, 8, 30) and stored.shape == (25,)
        for n in range(25):
            theta = rng.uniform(-3.0, 3.0, data.shape[2])
            got = co.supp(tp, data, ARCH, nns[n], theta, 0.0, 240, want_grad=False)["loss"]
            assert abs(got - stored[n]) < 2e-6, (n, got, stored[n])
            offsets.append(got - stored[n])
    # the offsets are the reference solver's own discretisation error: one value per data set, not per network
    offsets = np.array(offsets).reshape(2, 25)
    assert np.all(np.ptp(offsets, axis=1) < 5e-9) and np.all(offsets > 5e-7)


def test_adaptive_restatement_reproduces_the_validation_objectives():
    import cude_oracle as o
    nns, sets, tp = _load_validation()
    for data, stored in sets:
        scale = o.supp_scale(data)
        N = data.shape[2]
        for n in (3, 21):
            nn = nns[n]
            total = 0.0
            for i in range(N):
                rhs = lambda t, u: [float(v) for v in o.supp_rhs(np, nn, 1.0, ARCH, t, [np.float64(x) for x in u])]
                sol = o.solve_adaptive(rhs, list(data[:, 0, i]), list(tp), abstol=1e-6, reltol=1e-3)
                for ti in range(len(tp)):
                    for s in range(3):
                        r = (sol[ti][s] - data[s, ti, i]) / scale[s]
                        total += r * r
            assert abs(total / N - stored[n]) < 2e-9, (n, total / N, stored[n])


def test_adaptive_restatement_reproduces_the_reference_noise_free_data():
    """A third kind of stored output: `validation_data_nonoise` (suppression/suppression.jl:31-34) is what the reference's
    OWN solver returned -- `Array(solve(ODEProblem(lsup!, [10, 0, 0], (0, 30), p), Tsit5(), saveat = timepoints))` for the
    ground-truth model `lsup!` (suppression_model.jl:16-20,39-63), no noise added -- for 30 subjects whose fourth
    parameter is stored (`gt_validation_param_nonoise`) and whose first three (0.4 / 0.9 / 0.3 +- 0.1) are not.  24 stored
    values per subject against 3 unknowns: the oracle's adaptive Tsit5 (OrdinaryDiffEq defaults restated) reproduces all
    of them to ~1e-8 once the three rates are fitted, four orders below the solver's own truncation error (~1e-4: the
    fixed-step solution of the same problem differs by that much) -- i.e. it walks the reference's accepted steps on a
    nonlinear three-state problem as well.  (This model is the data generator, not part of the product path: the pin is
    on the solver restatement that the product's adaptive kernel is held to.)"""
    import cude_oracle as o
    from scipy.optimize import least_squares
    g = np.load(os.path.join(GOLD, "suppression_lambda0.npz"))
    data, p4, tp = g["validation_data_nonoise"], g["gt_validation_param_nonoise"], g["timepoints"]
    assert data.shape == (3, 8, 30) and np.all(data[:, 0, :] == np.array([10.0, 0.0, 0.0])[:, None])

    def rhs_of(p):
        def rhs(t, u):
            a = p[1] * u[1] / (1.0 + p[3] * u[2])
            return np.array([-p[0] * u[0], p[0] * u[0] - a, a - p[2] * u[2]])
        return rhs
    worst, worst_fixed = 0.0, 0.0
    for i in range(0, 30, 3):                                     # two subjects of each of the five + one groups
        y = data[:, :, i]
        p1 = float(np.mean(-np.log(y[0, 1:] / 10.0) / tp[1:]))    # u1 = 10 exp(-p1 t) up to the solver's error

        def resid(q):
            sol = np.asarray(o.solve_adaptive(rhs_of([q[0], q[1], q[2], p4[i]]), np.array([10.0, 0.0, 0.0]), tp))
            return ((sol if sol.shape == (3, 8) else sol.T) - y).ravel()
        fit = least_squares(resid, [p1, 0.9, 0.3], xtol=1e-15, ftol=1e-15, gtol=1e-15)
        assert np.all(np.abs(fit.x - np.array([0.4, 0.9, 0.3])) < 0.45) and fit.x.min() >= 0.05
        worst = max(worst, float(np.max(np.abs(fit.fun))))
        fixed = np.asarray(o.solve_fixed(rhs_of([*fit.x, p4[i]]), np.array([10.0, 0.0, 0.0]), tp, 960))
        worst_fixed = max(worst_fixed, float(np.max(np.abs((fixed if fixed.shape == (3, 8) else fixed.T) - y))))
    assert worst < 1e-7, worst
    assert 1e-6 < worst_fixed < 1e-2, worst_fixed                # the converged solution is NOT what was stored


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_validation_objectives():
    """... and through the product path: the loss itself, and `validate_suppression_model` end to end (the objective
    is flat in the conditional parameters -- the reference's stored validation correlations are accordingly one and
    the same number for all 25 networks -- so whatever the per-subject search returns, its objective is the known
    answer)."""
    import torch  # noqa: F401
    from cude import api
    nns, sets, tp = _load_validation()
    prob = api.SuppressionProblem(api.neural_network_model(5, 3, input_dims=4))
    rng = np.random.default_rng(3)
    for data, stored in sets:
        for n in (0, 12):
            p = api.ComponentArray(theta=rng.uniform(-3.0, 3.0, 30), neural=nns[n])
            got = api.suppression_loss(p, (prob, data, tp, 0.0), n_steps=240)
            assert abs(got - stored[n]) < 2e-6, (n, got, stored[n])
        p_init = [rng.random(30) for _ in range(4)]
        res, objective = api.validate_suppression_model(p_init, prob, data, tp, nns[5], n_steps=240)
        assert abs(objective - stored[5]) < 2e-6
        assert res.shape == (30,) and np.all(np.isfinite(res))
    api.clear_cache()


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_objectives():
    import torch  # noqa: F401
    from cude import api
    nns, stored, data, tp = _load()
    prob = api.SuppressionProblem(api.neural_network_model(5, 3, input_dims=4))
    rng = np.random.default_rng(1)
    for n in (0, 7, 24):
        p = api.ComponentArray(theta=rng.uniform(-3.0, 3.0, 37), neural=nns[n])
        got = api.suppression_loss(p, (prob, data, tp, 1.0), n_steps=240)
        assert abs(got - stored[n]) < 2e-6, (n, got, stored[n])
    # all 25 at once through the multi-start entry point
    pop = api._supp_population(prob, data, tp, 1.0, 240)
    losses = pop.engine.multistart_forward(nns, rng.uniform(-3.0, 3.0, (25, 37)))
    assert np.max(np.abs(losses - stored)) < 2e-6
    api.clear_cache()
