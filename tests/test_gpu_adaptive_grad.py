"""Gradient of the ADAPTIVE solve on the device (cude_config.n_steps = 0 + the gradient entry points,
csrc/cude_adaptive.hip) -- what the reference trains with: `Optimization.AutoForwardDiff()` through
`solve(model.problem, p = theta, saveat = timepoints)` (src/parameter-estimation.jl:59,165;
suppression/src/suppression_model.jl:123,155).  Under ForwardDiff only the parameters carry partials (tspan and dt stay
Float64), so the gradient is that of the accepted step sequence taken as fixed arithmetic.

Checker: the oracle replays a step sequence (cude_oracle.replay_steps) and differentiates it by the complex-step method
-- forward-mode, per perturbation, machine-precise -- against the device's hand-written reverse sweep over its tape.
The sequence replayed is the DEVICE's own (cude_adaptive_steps), because at OrdinaryDiffEq's default tolerances two
correct solvers do not always accept the same steps (tests/test_gpu_adaptive.py header); that the device's sequence IS
the oracle's own adaptive sequence (same accept / reject decisions, step sizes to the controller's rounding sensitivity)
is asserted separately, and the oracle's end-to-end adaptive gradient (cude_oracle.*_adaptive_loss_grad) is compared at
the solver's own tolerance.
Tolerances: loss / SSE 1e-10 relative, gradients 1e-8 of the largest entry."""
import os

import numpy as np
import pytest

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("arch,N", [((2, 4, 2), 70), ((2, 6, 2), 131), ((3, 4, 2), 64), ((2, 8, 2), 30), ((2, 4, 3), 30),
                                    ((2, 5, 1), 20), ((2, 8, 3), 20)])
def test_cpep_adaptive_gradient(arch, N):
    import cude_oracle as o
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], covariate=(arch[0] == 3))
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    fwd = eng.forward(want_sse=True)
    loss, g_nn, g_b = eng.loss_grad()
    assert abs(loss - fwd["loss"]) <= 1e-14 * loss              # the forward half of the gradient launch is the forward launch
    steps = [eng.adaptive_steps(i) for i in range(N)]
    assert all(8 <= len(t) <= 200 for t, _ in steps)
    for t, dt in steps:                                          # a partition of the time span
        assert t[0] == c["tp"][0] and np.all(t[1:] == t[:-1] + dt[:-1]) and abs(t[-1] + dt[-1] - c["tp"][-1]) < 1e-12
    # (1) the adjoint of the device's own sequence
    rl, rg, rb, rsse = o.cpep_replay_loss_grad(c["nn"], c["beta"], pop, arch, [list(zip(t, dt)) for t, dt in steps])
    assert abs(loss - rl) <= 1e-10 * rl
    assert np.max(np.abs(g_nn - rg)) <= 1e-8 * np.max(np.abs(rg))
    assert np.max(np.abs(g_b - rb)) <= 1e-8 * np.max(np.abs(rb))
    # (2) that sequence is the oracle's adaptive one wherever no error estimate sat on the acceptance threshold
    same = 0
    for i in range(N):
        rec = []
        c0 = float(pop.c0[i])
        o.solve_adaptive(o.cpep_rhs_scalar(pop, i, c["nn"], float(np.exp(c["beta"][i])), arch),
                         [c0, float(pop.k2[i] / pop.k1[i]) * c0], pop.timepoints, record=rec)
        t, dt = steps[i]
        # (same accept / reject decisions; the step sizes themselves carry the rounding sensitivity of the controller
        # -- est^(7/50) of a cancelling sum -- measured: median 7e-5, max 2e-2 of a 120-minute span)
        if len(rec) == len(t) and np.max(np.abs(np.array(rec)[:, 1] - dt)) <= 1e-3 * (c["tp"][-1] - c["tp"][0]):
            same += 1
    assert same >= 0.9 * N, same
    # (3) end to end against the oracle's own adaptive gradient: solver-error level where sequences differ, so the
    #     population gradient is held to the solver's tolerance and most per-subject entries to rounding
    ol, og, ob, _ = o.cpep_adaptive_loss_grad(c["nn"], c["beta"], pop, arch)
    assert abs(loss - ol) <= 1e-4 * ol
    assert np.max(np.abs(g_nn - og)) <= 5e-3 * np.max(np.abs(og))
    assert np.median(np.abs(g_b - ob)) <= 1e-7 * np.max(np.abs(ob))
    eng.close()


@pytest.mark.parametrize("tp", [[0.0, 120.0], [0.0, 60.0, 120.0], [0.0, 10.0, 45.0, 50.0, 120.0],
                                [0.0, 15.0, 30.0, 45.0, 60.0, 75.0, 90.0, 120.0]], ids=["2", "3", "5-irregular", "8"])
@pytest.mark.parametrize("arch", [(2, 4, 2), (2, 6, 2)])
def test_cpep_adaptive_other_sampling_grids(tp, arch):
    """Sampling grids other than the reference's five times: up to five knots run the kernel whose stages are unrolled
    (interior knots in scalar registers, absent ones at +inf), longer grids the phase-machine kernel.  Steps straddle
    knots, several observations fall into one step, T = 2 has no interior knot at all."""
    import cude_oracle as o
    from cude.engine import Engine
    N = 67
    rng = np.random.default_rng(11)
    tp = np.array(tp)
    age, t2 = rng.uniform(20, 79, N), rng.random(N) < 0.4
    G = 5.0 + np.abs(rng.standard_normal((N, tp.size))).cumsum(1)
    obs = 0.5 + rng.random((N, tp.size))
    nn, beta = o.glorot_params(arch, 9), rng.normal(-0.6, 0.5, N)
    pop = o.CPepPopulation(tp, G, obs, age, t2)
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(tp, G, obs, age, t2)
    eng.set_params(nn, beta)
    fwd = eng.forward(want_sse=True, want_traj=True)
    loss, g_nn, g_b = eng.loss_grad()
    assert abs(loss - fwd["loss"]) <= 1e-14 * loss
    steps = [eng.adaptive_steps(i) for i in range(N)]
    rl, rg, rb, rsse = o.cpep_replay_loss_grad(nn, beta, pop, arch, [list(zip(t, dt)) for t, dt in steps])
    assert abs(loss - rl) <= 1e-10 * rl
    assert np.max(np.abs(fwd["sse"] - rsse)) <= 1e-10 * np.max(rsse)
    assert np.max(np.abs(g_nn - rg)) <= 1e-8 * np.max(np.abs(rg))
    assert np.max(np.abs(g_b - rb)) <= 1e-8 * np.max(np.abs(rb))
    # the oracle's own adaptive solve: same trajectories at the solver's tolerance
    ol, og, ob, _ = o.cpep_adaptive_loss_grad(nn, beta, pop, arch)
    assert abs(loss - ol) <= 1e-4 * ol
    assert np.max(np.abs(g_nn - og)) <= 5e-3 * np.max(np.abs(og))
    eng.close()


def test_supp_adaptive_gradient():
    import cude_oracle as o
    from cude.engine import Engine
    for arch, N in (((4, 3, 5), 37), ((4, 6, 2), 9), ((4, 3, 1), 9)):
        s = make_supp_case(N, arch)
        eng = Engine("supp", arch, n_steps=0, lam=0.01)
        eng.set_population_supp(s["tp"], s["data"])
        eng.set_params(s["nn"], s["theta"])
        loss, g_nn, g_t = eng.loss_grad()
        # (1) the adjoint of the device's own sequences
        rl, rg, rt, _ = o.supp_replay_loss_grad(s["nn"], s["theta"], s["data"], s["tp"], arch, 0.01,
                                                [list(zip(*eng.adaptive_steps(i))) for i in range(N)])
        assert abs(loss - rl) <= 1e-10 * rl
        assert np.max(np.abs(g_nn - rg)) <= 1e-8 * np.max(np.abs(rg))
        assert np.max(np.abs(g_t - rt)) <= 1e-8 * np.max(np.abs(rt))
        # (2) end to end against the oracle's own adaptive solve + gradient (its own accepted steps)
        ol, og, ot, _ = o.supp_adaptive_loss_grad(s["nn"], s["theta"], s["data"], s["tp"], arch, 0.01)
        assert abs(loss - ol) <= 1e-8 * ol
        assert np.max(np.abs(g_nn - og)) <= 1e-6 * np.max(np.abs(og))
        assert np.max(np.abs(g_t - ot)) <= 1e-6 * np.max(np.abs(ot))
        t, dt = eng.adaptive_steps(N - 1)
        rec = []
        et = float(np.exp(s["theta"][N - 1]))
        o.solve_adaptive(lambda tt, u: o.supp_rhs(__import__("math"), [float(v) for v in s["nn"]], et, arch, tt, u),
                         [float(v) for v in s["data"][:, 0, N - 1]], [float(v) for v in s["tp"]], record=rec)
        assert len(rec) == len(t) and np.max(np.abs(np.array(rec)[:, 1] - dt)) <= 1e-6 * s["tp"][-1]
        eng.close()


def test_adaptive_gradient_everywhere_the_fixed_step_one_goes():
    """Adam steps, parameter sets side by side, the symbolic model, a failing subject, a short tape."""
    import os
    import cude_oracle as o
    from cude.engine import Engine
    arch = (2, 4, 2)
    N = 57
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    # multi-start: every set equals its own loss_grad call bit for bit
    rng = np.random.default_rng(3)
    K = 5
    nn_sets = c["nn"][None, :] * (1.0 + 0.2 * rng.standard_normal((K, c["nn"].size)))
    cond_sets = c["beta"][None, :] + 0.3 * rng.standard_normal((K, N))
    cond_sets[2, 11] = np.nan
    L, Gn, Gc = eng.multistart_loss_grad(nn_sets, cond_sets)
    assert np.isinf(L[2]) and np.all(np.isfinite(np.delete(L, 2)))
    for k in (0, 1, 3, 4):
        eng.set_params(nn_sets[k], cond_sets[k])
        l1, gn1, gc1 = eng.loss_grad()
        assert l1 == L[k] and np.array_equal(gn1, Gn[k]) and np.array_equal(gc1, Gc[k])
    # Adam on the adaptive objective descends it
    eng.set_params(c["nn"], c["beta"])
    eng.adam_init(1e-2)
    l0 = eng.forward()["loss"]
    trace = eng.adam_run(60)
    assert np.all(np.isfinite(trace)) and eng.forward()["loss"] < 0.7 * l0
    eng.close()
    # a tape shorter than the solve: the evaluation fails (+Inf), nothing else does
    os.environ["CUDE_TAPE_STEPS"] = "4"
    try:
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        assert np.isinf(eng.loss_grad()[0]) and np.isfinite(eng.forward()["loss"])
        eng.close()
    finally:
        del os.environ["CUDE_TAPE_STEPS"]
    # symbolic model
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng = Engine("cpep_sym", o.SYMBOLIC, n_steps=0, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    k = np.log(5.0 + 3.0 * rng.random(N))
    eng.set_params(np.array([1.78]), k)
    loss, g_p, g_k = eng.loss_grad()
    ol, og, ok, _ = o.cpep_adaptive_loss_grad(np.array([1.78]), k, pop, o.SYMBOLIC)
    assert abs(loss - ol) <= 1e-3 * ol and abs(g_p[0] - og[0]) <= 2e-2 * abs(og[0])
    assert np.median(np.abs(g_k - ok)) <= 1e-3 * np.max(np.abs(ok))
    eng.close()


def test_training_on_the_reference_objective_through_the_api():
    """train / fit_suppression_model with n_steps = ADAPTIVE: screening, Adam and L-BFGS (inside the library) on the
    adaptive-solve loss -- the function the reference's own optimisers see (parameter-estimation.jl:340-386,
    suppression_model.jl:140-177)."""
    from cude import api
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    net = api.chain(4, 2, "tanh")
    n = 40
    models = [api.CPeptideCUDEModel(g["glucose"][i], g["timepoints"], g["ages"][i], net, g["cpeptide"][i], bool(g["t2dm"][i]))
              for i in range(n)]
    sols = api.train(models, g["timepoints"], g["cpeptide"][:n], np.random.default_rng(5), initial_guesses=200,
                     selected_initials=2, number_of_iterations_adam=150, number_of_iterations_lbfgs=40, n_steps=api.ADAPTIVE)
    for s in sols:
        assert np.isfinite(s.objective) and s.objective < 1.5
        assert abs(api.loss(s.u, (models, g["timepoints"], g["cpeptide"][:n]), n_steps=api.ADAPTIVE) - s.objective) < 1e-9
        # the fixed-step loss at the same point differs by the solver's own error, no more
        assert abs(api.loss(s.u, (models, g["timepoints"], g["cpeptide"][:n])) - s.objective) < 2e-2 * s.objective
    api.clear_cache()
    gs = dict(np.load(os.path.join(GOLD, "suppression_lambda0.npz")))
    prob = api.SuppressionProblem(api.neural_network_model(5, 3, input_dims=4))
    rng = np.random.default_rng(2)
    p0 = [api.ComponentArray(theta=rng.uniform(-1.0, 1.0, gs["group_data"].shape[2]), neural=api.init_params(prob.network, rng))
          for _ in range(3)]
    fits, _ = api.fit_suppression_model(p0, prob, gs["group_data"], gs["timepoints"], 0.0, select_best_n=3, adam_iters=150,
                                        lbfgs_iters=60, n_steps=api.ADAPTIVE)
    start = [api.suppression_loss(p, (prob, gs["group_data"], gs["timepoints"], 0.0), n_steps=api.ADAPTIVE) for p in p0]
    assert len(fits) == 3
    for f, l0 in zip(fits, sorted(start)):                 # best initial guess first
        assert np.isfinite(f.objective) and f.objective < 0.95 * l0
        assert abs(api.suppression_loss(f.u, (prob, gs["group_data"], gs["timepoints"], 0.0), n_steps=api.ADAPTIVE) - f.objective) < 1e-9
    api.clear_cache()
