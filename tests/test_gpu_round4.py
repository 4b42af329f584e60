"""Round-4 behaviour of the host side of libcude_hip.so: the run-time options that replaced the library's environment
switches, and cude_adam_run's re-ordering schedule for large adaptive populations (the launch order is the summation
order of the shared gradient -- what the reference's serial loop fixes by construction, src/parameter-estimation.jl:126-140)."""
import numpy as np
import pytest

from conftest import make_cpep_case

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
