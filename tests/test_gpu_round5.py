"""Round-5 changes that must not change a bit: the tail of a time-split gradient evaluation in one launch (option
"fused_tail": network-gradient columns, loss / failure columns with the optimiser's state advance, and the chunks' shares of
the conditional gradient; three launches before), on the single-set path, under captured graphs and on the side-by-side
restarts of the reference's `train` (src/parameter-estimation.jl:340-386)."""
import numpy as np
import pytest

from conftest import make_cpep_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,arch,n_state", [(57, (2, 4, 2), 2), (1000, (2, 6, 2), 3), (117, (3, 4, 2), 2)])
def test_one_launch_tail_of_the_time_split_gradient_is_bit_identical(N, arch, n_state):
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(2)
    nn_sets = c["nn"][None, :] * (1.0 + 0.05 * rng.standard_normal((5, c["nn"].size)))
    b_sets = c["beta"][None, :] + 0.1 * rng.standard_normal((5, N))
    mask = np.ones(c["nn"].size)
    mask[3] = 0.0
    out = []
    for fused in (0, 1):
        eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
        eng.set_option("fused_tail", fused)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        eng.set_param_mask(mask)
        r = list(eng.loss_grad())
        eng.adam_init(1e-2)
        r.append(np.array([eng.adam_step() for _ in range(3)] + list(eng.adam_run(11))))
        r += list(eng.get_params())
        r += list(eng.multistart_loss_grad(nn_sets, b_sets))
        r += list(eng.train_restarts(nn_sets, b_sets, 20, 1e-2, 5))
        out.append(r)
        eng.close()
    assert out[0][1][3] == 0.0 and np.all(np.isfinite(out[0][3]))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("N,arch,n_state,tp", [(57, (2, 4, 2), 2, None), (700, (2, 6, 2), 3, None),
                                               (90, (2, 4, 2), 2, [0.0, 0.0001, 30.0, 31.0, 31.5, 120.0])],
                         ids=["57", "700-three-states", "several-observations-in-one-step"])
def test_scan_adjoint_recursion_as_a_linear_map(N, arch, n_state, tp):
    """The time-split path's per-subject scan runs the stage-adjoint recursion (no network in it: J_f = A) as the subject's
    linear map, precomputed from the same algebra on unit inputs (Cpep2Args::adj_map; option "scan_map" = 0: the
    stage-by-stage form).  Same numbers to rounding -- also when several observations fall into one step and when one sits
    a hair behind the initial time --, and the same as the oracle's reverse mode."""
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    if tp is not None:                         # a grid whose observation times crowd into two of the 30 steps
        t_old = c["tp"]
        G = np.stack([np.interp(tp, t_old, c["G"][i]) for i in range(N)])
        obs = np.stack([np.interp(tp, t_old, c["obs"][i]) for i in range(N)])
        c = dict(c, tp=np.array(tp), G=G, obs=obs)
    out = []
    for use_map in (1, 0):
        eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
        eng.set_option("cpep_path", "2:6")
        eng.set_option("scan_map", use_map)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        out.append(eng.loss_grad())
        eng.close()
    (l1, g1, b1), (l0, g0, b0) = out
    assert l1 == l0                                                     # (the forward half is untouched)
    assert np.max(np.abs(g1 - g0)) <= 1e-13 * np.max(np.abs(g0)) and np.max(np.abs(b1 - b0)) <= 1e-13 * np.max(np.abs(b0))
    ref = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], 30, n_state)
    assert abs(l1 - ref["loss"]) <= 1e-10 * ref["loss"]
    assert np.max(np.abs(g1 - ref["g_nn"])) <= 1e-9 * np.max(np.abs(ref["g_nn"]))
    assert np.max(np.abs(b1 - ref["g_beta"])) <= 1e-9 * np.max(np.abs(ref["g_beta"]))


@pytest.mark.parametrize("N,arch", [(57, (2, 4, 2)), (117, (3, 4, 2)), (200, (2, 6, 2)), (64, (2, 8, 2))])
def test_adaptive_solve_of_a_small_population_on_a_team_of_waves(N, arch):
    """csrc/cude_adaptive_team.hip: the reference's own population sizes in the reference's solver mode (the mirrors'
    default since round 5) -- a step's five network evaluations on five waves, every wave carrying the integrator
    redundantly.  Against the one-wave kernel (option "adaptive_team" = 0): loss, per-subject SSE, trajectories and accepted
    steps bit for bit (the same arithmetic), gradients to 1e-13 (five partial sums in another association); the same for
    side-by-side parameter sets; and the gradient against the oracle's replay of the device's own steps."""
    import cude_oracle as o
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(4)
    nn_sets = c["nn"][None, :] * (1.0 + 0.05 * rng.standard_normal((3, c["nn"].size)))
    b_sets = c["beta"][None, :] + 0.1 * rng.standard_normal((3, N))
    out = []
    for team in (1, 0):
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_option("adaptive_team", team)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        fwd = eng.forward(want_sse=True, want_traj=True)
        loss, g_nn, g_b = eng.loss_grad()
        steps = [eng.adaptive_steps(i) for i in range(N)]
        ms = eng.multistart_loss_grad(nn_sets, b_sets)
        msf = eng.multistart_forward(nn_sets, b_sets)
        eng.adam_init(1e-2)
        tr = np.array([eng.adam_step() for _ in range(2)] + list(eng.adam_run(9)))
        out.append((fwd, loss, g_nn, g_b, steps, ms, msf, tr))
        eng.close()
    (f1, l1, g1, b1, s1, m1, mf1, t1), (f0, l0, g0, b0, s0, m0, mf0, t0) = out
    assert f1["loss"] == f0["loss"] and np.array_equal(f1["sse"], f0["sse"]) and np.array_equal(f1["traj"], f0["traj"])
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(s1, s0))
    assert abs(l1 - l0) <= 1e-14 * abs(l0)                       # (the workgroup's loss sum is formed by the same tree)
    assert np.max(np.abs(g1 - g0)) <= 1e-13 * np.max(np.abs(g0)) and np.max(np.abs(b1 - b0)) <= 1e-13 * np.max(np.abs(b0))
    assert np.allclose(m1[0], m0[0], rtol=1e-14) and np.max(np.abs(m1[1] - m0[1])) <= 1e-13 * np.max(np.abs(m0[1]))
    assert np.max(np.abs(m1[2] - m0[2])) <= 1e-13 * np.max(np.abs(m0[2])) and np.allclose(mf1, mf0, rtol=1e-14)
    # (along a training run step-size control amplifies last-place differences: the first iterations agree to rounding,
    #  the later ones to the solver's tolerance)
    assert np.allclose(t1[:1], t0[:1], rtol=1e-13) and np.allclose(t1, t0, rtol=2e-3) and np.all(np.isfinite(t1))
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], covariate=(arch[0] == 3))
    rl, rg, rb, _ = o.cpep_replay_loss_grad(c["nn"], c["beta"], pop, arch, [list(zip(t, dt)) for t, dt in s1])
    assert abs(l1 - rl) <= 1e-10 * rl
    assert np.max(np.abs(g1 - rg)) <= 1e-8 * np.max(np.abs(rg)) and np.max(np.abs(b1 - rb)) <= 1e-8 * np.max(np.abs(rb))


@pytest.mark.parametrize("N,arch,n_state,steps,tp", [(57, (2, 4, 2), 2, 32, None), (700, (2, 6, 2), 3, 30, None),
                                                     (90, (2, 4, 2), 2, 30, [0.0, 0.0001, 30.0, 31.0, 31.5, 120.0]),
                                                     (130, (3, 4, 2), 2, 20, None)],
                         ids=["57", "700-three-states", "several-observations-in-one-step", "covariate-20-steps"])
def test_scan_of_a_small_launch_on_eight_waves_is_bit_identical(N, arch, n_state, steps, tp):
    """csrc/cude_cpep2.hip cpep2_scan_bulk_kernel (launches of at most one scan workgroup per compute unit; option
    "scan_bulk" = 0: the one-wave scan): every row through LDS up front, the adjoint recursion's state in wave 0 and the
    weights of a segment of steps in every wave.  The same operations on the same values: loss, gradients, per-subject
    SSE, queued Adam iterations, side-by-side parameter sets and a Metropolis chain (plain and speculative) bit for bit,
    also when several observations fall into one step and when the step count is not a multiple of eight."""
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    if tp is not None:
        t_old = c["tp"]
        G = np.stack([np.interp(tp, t_old, c["G"][i]) for i in range(N)])
        obs = np.stack([np.interp(tp, t_old, c["obs"][i]) for i in range(N)])
        c = dict(c, tp=np.array(tp), G=G, obs=obs)
    rng = np.random.default_rng(7)
    nn_sets = c["nn"][None, :] * (1.0 + 0.05 * rng.standard_normal((3, c["nn"].size)))
    b_sets = c["beta"][None, :] + 0.1 * rng.standard_normal((3, N))
    normals, uniforms = rng.standard_normal((7, N)), rng.random((7, N))
    out = []
    for bulk in (1, 0):
        r = []
        for spec in (0, 3):
            eng = Engine("cpep", arch, n_steps=steps, n_state=n_state)
            eng.set_option("scan_bulk", bulk)
            eng.set_option("mh_spec", spec)
            eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
            eng.set_params(c["nn"], c["beta"])
            if spec == 0:
                f = eng.forward(want_sse=True)
                r += [np.float64(f["loss"]), f["sse"]] + list(eng.loss_grad())
                r += list(eng.multistart_loss_grad(nn_sets, b_sets)) + [eng.multistart_forward(nn_sets, b_sets)]
                eng.adam_init(1e-2)
                r.append(np.array(eng.adam_run(9)))
                r += list(eng.get_params())
                eng.set_params(c["nn"], c["beta"])
            acc, samples = eng.mh_chain(normals, uniforms, 0.4, -0.6, 0.9, 0.3)
            r += [acc, samples]
            eng.close()
        out.append(r)
    assert np.isfinite(out[0][0]) and np.all(np.isfinite(out[0][3]))
    for a, b in zip(*out):
        assert np.array_equal(np.asarray(a), np.asarray(b))
