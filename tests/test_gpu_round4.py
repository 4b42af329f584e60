"""Round-4 behaviour of the host side of libcude_hip.so: the run-time options that replaced the library's environment
switches, and cude_adam_run's re-ordering schedule for large adaptive populations (the launch order is the summation
order of the shared gradient -- what the reference's serial loop fixes by construction, src/parameter-estimation.jl:126-140)."""
import numpy as np
import pytest

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu


def test_regroup_schedule_does_not_depend_on_how_a_run_is_cut():
    """cude_adam_run re-orders an adaptive population of >= 8192 subjects by accepted-step count after its 1st, 200th,
    400th ... iteration on that population, counted over calls: one run of 410 iterations and the same iterations cut
    into 150 + 250 + 10 give the same losses and parameters bit for bit; the tape stays readable afterwards
    (cude_adaptive_steps) and the order in place after iteration 400 is still sorted (advisor, round 3: the second
    automatic re-ordering used to be the last, and left the tape marked unreadable)."""
    from cude.engine import Engine
    arch, N = (2, 4, 2), 8300
    c = make_cpep_case(N, arch)

    def run(cuts, auto=True):
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        if not auto:
            eng.set_option("auto_regroup", 0)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        eng.adam_init(1e-3)
        tr = np.concatenate([eng.adam_run(k) for k in cuts])
        nn, cond = eng.get_params()
        t, dt = eng.adaptive_steps(17)                   # the tape of the last iteration, whatever order it is in
        assert len(t) >= 5 and np.all(dt > 0)
        spread = eng.adaptive_regroup()
        eng.close()
        return tr, nn, cond, spread
    a, b = run([410]), run([150, 250, 10])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    plain = run([410], auto=False)
    assert a[3][0] <= 2 and a[3][0] < plain[3][0]        # mean step spread within a wave: (still nearly) sorted vs the caller's order
    assert np.allclose(a[0], plain[0], rtol=1e-4)        # same training to the solver's own sensitivity
    assert a[0][0] == plain[0][0]                        # iteration 1 runs in the caller's order in both


def test_options_replace_the_environment_switches(monkeypatch):
    """cude_set_option takes what the CUDE_* variables took (they are read once, at cude_create); unknown names and
    malformed values are errors; a launch-path option applies from the next population upload."""
    from cude.engine import Engine, CudeError
    arch = (2, 6, 2)
    c = make_cpep_case(500, arch)

    def grad(setup):
        eng = Engine("cpep", arch, n_steps=30, n_state=3)
        setup(eng)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        out = eng.loss_grad()
        eng.close()
        return out
    auto = grad(lambda e: None)
    one = grad(lambda e: e.set_option("cpep_path", "1"))
    split = grad(lambda e: e.set_option("cpep_path", "2:5"))
    monkeypatch.setenv("CUDE_CPEP_PATH", "2:5")
    split_env = grad(lambda e: None)
    monkeypatch.delenv("CUDE_CPEP_PATH")
    assert np.array_equal(split[1], split_env[1]) and split[0] == split_env[0]
    assert not np.array_equal(one[1], split[1])          # different kernels, different summation order ...
    for other in (one, split):                           # ... same numbers
        assert abs(other[0] - auto[0]) <= 1e-12 * abs(auto[0])
        assert np.max(np.abs(other[1] - auto[1])) <= 1e-11 * np.max(np.abs(auto[1]))
    eng = Engine("cpep", arch, n_steps=30, n_state=3)
    with pytest.raises(CudeError):
        eng.set_option("no_such_option", 1)
    with pytest.raises(CudeError):
        eng.set_option("cpep_path", "sideways")
    with pytest.raises(CudeError):
        eng.set_option("tape_steps", "many")
    eng.close()


def test_forward_launches_tell_the_step_counts_and_the_other_entry_points_regroup_too():
    """Every adaptive launch -- forward-only ones included -- leaves the accepted-step counts behind, so the launch can be
    ordered before any gradient has been taken; cude_fit_conditional (after its first probe) and
    cude_multistart_loss_grad / cude_train_restarts (from their second evaluation on) do it by themselves for >= 8192
    subjects.  Per-subject results do not depend on the order (bit for bit); the shared gradient is a sum in another
    order (rounding)."""
    from cude.engine import Engine
    arch, N = (2, 4, 2), 8400
    c = make_cpep_case(N, arch)

    def make(auto):
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_option("auto_regroup", 1 if auto else 0)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        return eng
    # ---- explicit call behind a forward launch
    eng = make(False)
    sse0 = eng.forward(want_sse=True)["sse"]
    before, after = eng.adaptive_regroup()
    assert after <= 1 < before
    assert np.array_equal(eng.forward(want_sse=True)["sse"], sse0)
    eng.close()
    # ---- per-subject fits: same optimum, same objective, whatever the order
    fits, left = {}, {}
    for auto in (False, True):
        eng = make(auto)
        fits[auto] = eng.fit_conditional(-3.0, 2.0, n_grid=7, n_iters=6)
        left[auto] = eng.adaptive_regroup()[0]           # spread of the last launch's counts in the order the fit left behind
        eng.close()
    for x, y in zip(fits[False], fits[True]):
        assert np.array_equal(x, y)
    assert left[True] < left[False]                      # (ordered by the first probe's counts: closer, not sorted, at the optimum)
    # ---- restarts side by side: the second evaluation runs in the sorted order
    rng = np.random.default_rng(3)
    nn_sets = c["nn"][None, :] * (1.0 + 0.05 * rng.standard_normal((2, c["nn"].size)))
    cond_sets = np.stack([c["beta"], c["beta"] + 0.1])
    out = {}
    for auto in (False, True):
        eng = make(auto)
        eng.multistart_loss_grad(nn_sets, cond_sets)
        out[auto] = eng.multistart_loss_grad(nn_sets, cond_sets)
        spread = eng.adaptive_regroup()
        assert (spread[0] <= 2) == auto
        eng.close()
    assert np.array_equal(out[False][2], out[True][2])                                   # dL/dbeta: per subject
    assert np.allclose(out[False][0], out[True][0], rtol=1e-13)
    assert np.allclose(out[False][1], out[True][1], rtol=0, atol=1e-12 * np.max(np.abs(out[False][1])))


@pytest.mark.parametrize("model", ["cpep", "supp"])
def test_a_tape_too_short_fails_the_gradient_not_the_process(model):
    """The adaptive gradient's tape holds a fixed number of accepted steps per subject (option tape_steps; by default
    what 4 GB hold, at least 64).  A subject that takes more fails ITS evaluation the way a failed solve does -- the loss
    is +Inf with status 0 (reference: parameter-estimation.jl:61-64) -- while forward launches, which keep no tape, are
    not affected; with room for the steps the same engine data give a finite gradient."""
    from cude.engine import Engine
    if model == "cpep":
        arch = (2, 4, 2)
        c = make_cpep_case(200, arch)

        def make(steps):
            eng = Engine("cpep", arch, n_steps=0, n_state=2)
            if steps:
                eng.set_option("tape_steps", steps)
            eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
            eng.set_params(c["nn"], c["beta"])
            return eng
    else:
        arch = (4, 3, 5)
        c = make_supp_case(90, arch)

        def make(steps):
            eng = Engine("supp", arch, n_steps=0, lam=0.01)
            if steps:
                eng.set_option("tape_steps", steps)
            eng.set_population_supp(c["tp"], c["data"])
            eng.set_params(c["nn"], c["theta"])
            return eng
    eng = make(0)
    loss, g_nn, g_c = eng.loss_grad()
    n_steps = max(len(eng.adaptive_steps(i)[0]) for i in range(0, 90, 7))
    assert np.isfinite(loss) and np.all(np.isfinite(g_nn)) and n_steps > 6
    eng.close()
    short = make(n_steps - 3)
    fwd = short.forward()["loss"]
    assert abs(fwd - loss) <= 1e-14 * loss                 # no tape in a forward launch
    loss_short = short.loss_grad()[0]
    assert np.isinf(loss_short) and loss_short > 0
    short.close()
    enough = make(n_steps + 8)
    loss2, g2, _ = enough.loss_grad()
    assert loss2 == loss and np.array_equal(g2, g_nn)
    enough.close()
