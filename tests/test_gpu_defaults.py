"""The default discretisation of the reference-facing layer (cude/api.py, julia/CUDEHip.jl): the reference's own solver
mode.  `solve(prob, Tsit5())` with OrdinaryDiffEq's default tolerances is what every loss of the reference evaluates
(src/parameter-estimation.jl:59, suppression/src/suppression_model.jl:113,123, src/saem.jl:52); a call that passes no
`n_steps` therefore runs the adaptive kernels and reproduces the reference's stored objectives, and the fixed-step fast
mode is what has to be asked for.  (The two mirrors' defaults are compared in tests/test_julia_shim.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_suppression_loss_without_n_steps_reproduces_the_reference_stored_objectives():
    import torch  # noqa: F401
    from cude import api
    assert api.default_steps() == api.ADAPTIVE == 0
    g1 = np.load(os.path.join(GOLD, "suppression_lambda1.npz"))
    g0 = np.load(os.path.join(GOLD, "suppression_lambda0.npz"))
    nns, stored, data, tp = g1["nn_4x3x5x1"], g1["losses"], g0["group_data"], g0["timepoints"]
    prob = api.SuppressionProblem(api.neural_network_model(5, 3, input_dims=4))
    rng = np.random.default_rng(1)
    for n in (0, 11, 24):
        p = api.ComponentArray(theta=rng.uniform(-3.0, 3.0, 37), neural=nns[n])
        got = api.suppression_loss(p, (prob, data, tp, 1.0))                 # no n_steps: the reference's mode
        assert abs(got - stored[n]) < 2e-9
        fast = api.suppression_loss(p, (prob, data, tp, 1.0), n_steps=api.DEFAULT_STEPS)
        assert 1e-9 < abs(fast - stored[n]) < 1e-2 * stored[n]               # the fast mode is close, not equal
    api.clear_cache()


def test_cpeptide_loss_without_n_steps_is_the_adaptive_solve():
    import torch  # noqa: F401
    import cude_oracle as o
    from cude import api
    g = np.load(os.path.join(GOLD, "ohashi_cude.npz"))
    tp = g["timepoints"]
    net = api.chain(4, 2, "tanh")
    n = 16
    models = [api.CPeptideConditionalUDEModel(g["glucose"][i], tp, g["ages"][i], net, g["cpeptide"][i], bool(g["t2dm"][i]))
              for i in range(n)]
    rng = np.random.default_rng(3)
    theta = api.ComponentArray(neural=g["nn_2x4x4x1"][0], conditional=rng.uniform(-2, 0, (n, 1)))
    args = (models, tp, g["cpeptide"][:n])
    assert api.default_steps(tp) == api.ADAPTIVE
    val = api.loss(theta, args)
    assert val == api.loss(theta, args, n_steps=api.ADAPTIVE)
    # the oracle's adaptive solve (OrdinaryDiffEq's controller restated, oracle/cude_oracle.py) of the same subjects
    pop = o.CPepPopulation(tp, g["glucose"][:n], g["cpeptide"][:n], g["ages"][:n], g["t2dm"][:n])
    ref, g_nn_ref, g_b_ref, _ = o.cpep_adaptive_loss_grad(theta.neural, theta.conditional[:, 0], pop, (2, 4, 2))
    assert abs(val - ref) <= 1e-4 * ref      # (solver-tolerance level where a step sequence differs, tests/test_gpu_adaptive_grad.py)
    fast = api.loss(theta, args, n_steps=api.fixed_steps(tp))
    assert api.fixed_steps(tp) == 32 and 0 < abs(fast - val) < 1e-2 * val
    # the module-wide switch: what the fixed-step test modules select (tests/conftest.py fixed_step_default)
    prev = api.set_default_steps("fixed")
    try:
        assert prev == api.ADAPTIVE and api.default_steps(tp) == 32 and api.default_steps() == 30
        assert api.loss(theta, args) == fast
        assert api.default_steps(np.array([0.0, 10.0, 30.0])) == 30         # not equidistant: the 30-step grid
    finally:
        api.set_default_steps(prev)
    assert api.default_steps(tp) == api.ADAPTIVE
    val2, grad = api.loss_and_gradient(theta, args)                         # gradients in the default mode as well
    assert abs(val2 - val) <= 1e-14 * val                                   # (the gradient launch sums in another order)
    assert np.max(np.abs(grad.neural - g_nn_ref)) <= 5e-3 * np.max(np.abs(g_nn_ref))
    assert np.median(np.abs(grad.conditional[:, 0] - g_b_ref)) <= 1e-7 * np.max(np.abs(g_b_ref))
    api.clear_cache()
