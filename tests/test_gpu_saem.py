"""SAEM E-step (per-subject Metropolis-Hastings, src/saem.jl:86-108,177-186) on the GPU vs the oracle chain with
the SAME host-supplied draws -- both the fused device call (cude_mh_estep) and the host loop over forward
launches (api.mcmc_steps) -- and the config-5 size (1e4 subjects x 100 MC steps) through invariants."""
import time

import numpy as np
import pytest
import torch  # noqa: F401

from conftest import make_cpep_case

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fixed_step_default")]


def _engine(c, arch):
    from cude.engine import Engine
    eng = Engine("cpep", arch, n_steps=30)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])

    def sse_of(beta):
        eng.set_params(None, beta)
        return eng.forward(want_sse=True)["sse"]
    return eng, sse_of


@pytest.mark.parametrize("gamma,temperature", [(1.0, 1.0), (0.4, 2.5)])
def test_mh_chain_matches_oracle_with_same_draws(gamma, temperature):
    import cude_oracle as o
    from cude import api
    arch, N, steps = (2, 4, 2), 48, 12
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(7)
    normals, uniforms = rng.standard_normal((steps, N)), rng.random((steps, N))
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    p_ref, acc_ref = o.mh_chain(c["nn"], c["beta"], pop, arch, 30, 0.4, -0.6, 0.9, 0.3, temperature, gamma, normals,
                                uniforms)
    eng, sse_of = _engine(c, arch)
    p_host, acc_host = api.mcmc_steps(sse_of, c["beta"], 5, 0.4, 0.9, 0.3, -0.6, temperature, gamma, normals, uniforms)
    assert np.array_equal(acc_host, acc_ref) and np.max(np.abs(p_host - p_ref)) < 1e-12
    eng.set_params(c["nn"], c["beta"])
    acc_dev = eng.mh_estep(normals, uniforms, 0.4, -0.6, 0.9, 0.3, temperature, gamma)
    _, p_dev = eng.get_params()
    eng.close()
    assert np.array_equal(acc_dev, acc_ref)          # identical accept/reject decisions
    assert np.max(np.abs(p_dev - p_ref)) < 1e-12


def test_chain_samples_and_individual_effects():
    """cude_mh_chain keeps every state of every subject's chain (c-peptide/06-saem.jl:107-112) -- equal to the
    oracle chain step by step -- and api.individual_effects returns samples, MAP modes and MLE estimates whose
    defining properties hold (the mode minimises the negative log-posterior, the MLE the SSE)."""
    import cude_oracle as o
    from cude import api
    arch, N, steps = (2, 4, 2), 40, 25
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(11)
    normals, uniforms = rng.standard_normal((steps, N)), rng.random((steps, N))
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    want = []
    start = np.full(N, -0.6)
    _, acc_ref = o.mh_chain(c["nn"], start, pop, arch, 30, 0.4, -0.6, 0.9, 0.3, 1.0, 1.0, normals, uniforms,
                            samples=want)
    eng, _ = _engine(c, arch)
    eng.set_params(c["nn"], start)
    acc, samples = eng.mh_chain(normals, uniforms, 0.4, -0.6, 0.9, 0.3)
    eng.close()
    assert np.array_equal(acc, acc_ref) and samples.shape == (steps, N)
    assert np.max(np.abs(samples - np.stack(want))) < 1e-12
    # API mirror on the same subjects
    net = api.chain(4, 2, "tanh")
    models = [api.CPeptideConditionalUDEModel(c["G"][i], c["tp"], c["age"][i], net, c["obs"][i], bool(c["t2dm"][i]))
              for i in range(N)]
    saem = api.SimpleNamespace(p_neural=c["nn"], eta=-0.6, Omega=0.9, sigma=0.4)
    eff = api.individual_effects(models, c["tp"], c["obs"], saem, n_samples=60, rng=np.random.default_rng(3), n_steps=30)
    assert eff.samples.shape == (60, N) and 0.0 < eff.acceptance_rate < 1.0

    def neg_log_post(b):
        sse = o.cpep_loss(np, c["nn"], b, pop, arch, 30)[1]
        return sse / (2 * 0.4 ** 2) + 0.5 * ((b - (-0.6)) / 0.9) ** 2
    at = neg_log_post(eff.modes)
    assert np.all(neg_log_post(eff.modes + 1e-3) >= at - 1e-9) and np.all(neg_log_post(eff.modes - 1e-3) >= at - 1e-9)
    sse_at = o.cpep_loss(np, c["nn"], eff.mle, pop, arch, 30)[1]
    for d in (1e-3, -1e-3):
        assert np.all(o.cpep_loss(np, c["nn"], eff.mle + d, pop, arch, 30)[1] >= sse_at - 1e-9)
    assert np.allclose(eff.mse, o.cpep_loss(np, c["nn"], eff.modes, pop, arch, 30)[1], rtol=1e-9)
    api.clear_cache()


def test_failed_solves_are_rejected():
    """A proposal whose solve is non-finite has log-likelihood -Inf and is never accepted (saem.jl:59-62)."""
    arch, N = (2, 4, 2), 70
    c = make_cpep_case(N, arch)
    eng, _ = _engine(c, arch)
    normals = np.zeros((3, N))
    normals[:, 5] = np.inf                           # proposal beta = +Inf for subject 5
    uniforms = np.full((3, N), 0.999999)
    acc = eng.mh_estep(normals, uniforms, 0.5, -0.6, 1.0, 0.3)
    _, p = eng.get_params()
    eng.close()
    assert acc[5] == 0 and p[5] == c["beta"][5]
    assert np.all(acc[np.arange(N) != 5] == 3)       # zero-step proposals: ratio 0 > log(0.999999)


def test_config5_size_estep_invariants():
    """1e4 subjects x 100 Metropolis steps (BASELINE configs[4] shape, one GPU's share): chain states stay finite,
    acceptance is in (0,1), the chain moves towards higher likelihood on average, and re-running with the same
    draws is bitwise reproducible."""
    arch, N, steps = (2, 4, 2), 10000, 100
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(3)
    normals, uniforms = rng.standard_normal((steps, N)), rng.random((steps, N))
    eng, sse_of = _engine(c, arch)
    start = np.full(N, -0.6)
    sse0 = sse_of(start).copy()
    t0 = time.perf_counter()
    acc1 = eng.mh_estep(normals, uniforms, 0.5, -0.6, 1.0, 0.3)
    dt = time.perf_counter() - t0
    _, p1 = eng.get_params()
    sse1 = sse_of(p1).copy()
    eng.set_params(None, start)
    acc2 = eng.mh_estep(normals, uniforms, 0.5, -0.6, 1.0, 0.3)
    _, p2 = eng.get_params()
    eng.close()
    print(f"E-step 1e4 x 100: {dt*1e3:.1f} ms = {2 * steps * N / dt:.3e} forward solves/s")
    assert np.array_equal(p1, p2) and np.array_equal(acc1, acc2)
    assert np.all(np.isfinite(p1)) and 0.02 < acc1.mean() / steps < 0.98
    assert np.median(sse1) < np.median(sse0)


def test_device_side_draws():
    """normals = uniforms = NULL: Philox4x32-10 draws generated on the device (cude_set_rng).  (i) The draws are the
    ones the published algorithm gives for (seed, global subject index, step) -- uniforms bit for bit, normals to the
    rounding of log / cos; (ii) an E-step with device draws IS the E-step with those draws supplied by the host (same
    accept counts, same states), so everything proved about that path carries over; (iii) successive calls continue the
    stream, cude_set_rng rewinds it; (iv) a shard with subject_offset draws what the full population draws for its
    subjects; (v) first moments and range of 2e5 draws."""
    from conftest import device_draw
    from cude.engine import Engine
    arch, N = (2, 4, 2), 96
    c = make_cpep_case(N, arch)

    def engine(lo=0, hi=N):
        eng = Engine("cpep", arch, n_steps=30)
        eng.set_population_cpep(c["tp"], c["G"][lo:hi], c["obs"][lo:hi], c["age"][lo:hi], c["t2dm"][lo:hi])
        eng.set_params(c["nn"], c["beta"][lo:hi])
        return eng
    seed = 0x9E3779B97F4A7C15
    eng = engine()
    eng.set_rng(seed)
    z, u = eng.rng_draws(0, 9)
    for step, i in ((0, 0), (0, 95), (3, 17), (8, 64)):
        zn, un = device_draw(seed, i, step)
        assert u[step, i] == un and abs(z[step, i] - zn) <= 4e-15 * max(1.0, abs(zn))
    # (ii) + (iii)
    acc_a = eng.mh_estep(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=5)
    acc_b, samples_b = eng.mh_chain(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=4)          # steps 5 ... 8 of the stream
    _, p_dev = eng.get_params()
    ref = engine()
    acc_ra = ref.mh_estep(z[:5], u[:5], 0.4, -0.6, 0.9, 0.3)
    acc_rb, samples_rb = ref.mh_chain(z[5:9], u[5:9], 0.4, -0.6, 0.9, 0.3)
    _, p_ref = ref.get_params()
    assert np.array_equal(acc_a, acc_ra) and np.array_equal(acc_b, acc_rb)
    assert np.array_equal(samples_b, samples_rb) and np.array_equal(p_dev, p_ref)
    assert 0 < acc_a.sum() < 5 * N
    eng.set_rng(seed)                                                                   # rewind: the same chain again
    eng.set_params(c["nn"], c["beta"])
    assert np.array_equal(eng.mh_estep(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=5), acc_a)
    # (iv)
    part = engine(40, N)
    part.set_rng(seed, subject_offset=40)
    zp, up = part.rng_draws(2, 3)
    assert np.array_equal(zp, z[2:5, 40:]) and np.array_equal(up, u[2:5, 40:])
    acc_p = part.mh_estep(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=5)
    assert np.array_equal(acc_p, acc_a[40:])
    for e in (eng, ref, part):
        e.close()
    # (v)
    big = Engine("cpep", arch, n_steps=30)
    cb = make_cpep_case(2000, arch)
    big.set_population_cpep(cb["tp"], cb["G"], cb["obs"], cb["age"], cb["t2dm"])
    zz, uu = big.rng_draws(0, 100)
    big.close()
    n = zz.size
    assert abs(zz.mean()) < 4 / np.sqrt(n) and abs(zz.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs(uu.mean() - 0.5) < 4 / np.sqrt(12 * n) and 0 < uu.min() and uu.max() < 1
    assert abs(np.corrcoef(zz[:-1].ravel(), zz[1:].ravel())[0, 1]) < 4 / np.sqrt(n)       # step-to-step
    assert abs(np.corrcoef(zz[:, :-1].ravel(), zz[:, 1:].ravel())[0, 1]) < 4 / np.sqrt(n)  # subject-to-subject
    assert np.abs(zz).max() < 6.5


def test_saem_loop_with_device_draws_equals_the_same_loop_fed_those_draws():
    """cude.parallel.saem_loop(device_seed=...) -- the whole SAEM iteration with the Metropolis draws generated on the
    device -- against the same loop fed, through its `draws` hook, the draws cude_rng_draws reports for that stream."""
    from cude.engine import Engine
    from cude.parallel import saem_loop
    arch, N = (2, 4, 2), 70
    c = make_cpep_case(N, arch)
    kw = dict(iterations=6, n_burnin_iterations=3, n_mcmc_steps=2, initial_mcmc_steps=3, proposal_std=0.2, sigma=0.5,
              prior_eta=-0.6, prior_omega=0.8, m_step_iters=2)

    def engine():
        eng = Engine("cpep", arch, n_steps=30)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        return eng
    seed = 424242
    dev = engine()
    a = saem_loop(dev, len(c["tp"]), c["nn"], device_seed=seed, **kw)
    dev.close()
    src, host = engine(), engine()
    src.set_rng(seed)
    pos = [0]

    def draws(it, steps):
        z, u = src.rng_draws(pos[0], steps)
        pos[0] += steps
        return z, u
    b = saem_loop(host, len(c["tp"]), c["nn"], draws=draws, **kw)
    src.close()
    host.close()
    assert pos[0] == 3 * 3 + 3 * 2
    assert np.array_equal(a.acceptance_rates, b.acceptance_rates) and np.array_equal(a.p_individuals, b.p_individuals)
    assert np.array_equal(a.p_neural, b.p_neural) and a.sigma == b.sigma and a.Omega == b.Omega
    assert 0.0 < a.acceptance_rates[-1] < 1.0


@pytest.mark.parametrize("mode", ["time-split", "adaptive", "one-lane"])
@pytest.mark.parametrize("N", [48, 1300])
def test_speculative_metropolis_is_the_same_chain_bit_for_bit(N, mode):
    """Option "mh_spec" (csrc/cude_kernels.h MhSpecArgs): d steps of every subject's chain per dependent launch chain --
    the 2^d - 1 states the d steps can propose from are evaluated as parameter sets of one launch and the decisions
    resolved afterwards.  Same draws (the caller's rows, or the counter-based device stream: a draw depends on (seed,
    subject, step) only), same arithmetic per decision: states, acceptance counts and every intermediate sample equal
    the step-by-step path's (src/saem.jl:86-108,177-186) bit for bit, for every depth, when the step count is not a
    multiple of the depth, and with a subject whose solve fails.  Modes: the time-split fixed-step path (candidates in the
    forward chunks' grid, resolved inside the scan launch), and the contexts whose solve is one launch over the subjects --
    the adaptive solve (the mirrors' default; at these sizes the team kernel) and the one-lane fixed-step kernel --
    where the candidates are grid rows of that launch and a resolver launch follows (round 5)."""
    from cude.engine import Engine
    arch, steps = (2, 4, 2), 11
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(5)
    normals, uniforms = rng.standard_normal((steps, N)), rng.random((steps, N))
    start = c["beta"].copy()
    start[N // 3] = 800.0                      # exp overflows: this subject's likelihood is -Inf throughout

    def run(depth, device_draws):
        eng = Engine("cpep", arch, n_steps=0 if mode == "adaptive" else 30)
        if mode == "one-lane":
            eng.set_option("cpep_path", "1")
        eng.set_option("mh_spec", depth)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], start)
        eng.set_rng(1234, 77)
        if device_draws:
            acc, samples = eng.mh_chain(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=steps)
            acc2 = eng.mh_estep(None, None, 0.4, -0.6, 0.9, 0.3, n_mc=5)        # the stream of draws continues
        else:
            acc, samples = eng.mh_chain(normals, uniforms, 0.4, -0.6, 0.9, 0.3)
            acc2 = eng.mh_estep(normals[:5], uniforms[:5], 0.4, -0.6, 0.9, 0.3)
        _, p = eng.get_params()
        eng.close()
        return acc, samples, acc2, p
    for device_draws in (False, True):
        ref = run(0, device_draws)
        assert 0 < ref[0].sum() < steps * N and ref[0][N // 3] == 0
        for depth in (2, 3, 4, -1):
            got = run(depth, device_draws)
            for a, b in zip(ref, got):
                assert np.array_equal(a, b), (depth, device_draws)


@pytest.mark.parametrize("mode", ["time-split", "adaptive", "one-lane"])
def test_blend_phase_solves_proposal_and_both_next_states_in_one_launch(mode):
    """gamma < 1 (the stochastic-approximation phase, src/saem.jl:177-186): the next state is (1 - gamma) p + gamma x with x
    the accepted proposal or p itself -- either way a point whose likelihood is unknown, which cost a second solve per
    step.  Both possible next states depend on (p, q, gamma) alone, so they ride in the proposal's launch as parameter
    sets and the decision picks state and SSE (option "mh_pair" = 0: the two-launch form).  The same chain bit for bit --
    states, acceptance counts, every intermediate sample --, with the caller's draws and the device's, with a subject
    whose solve fails."""
    from cude.engine import Engine
    arch, N, steps = (2, 4, 2), 150, 9
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(6)
    normals, uniforms = rng.standard_normal((steps, N)), rng.random((steps, N))
    start = c["beta"].copy()
    start[N // 3] = 800.0

    def run(pair, device_draws, depth=0):
        eng = Engine("cpep", arch, n_steps=0 if mode == "adaptive" else 30)
        if mode == "one-lane":
            eng.set_option("cpep_path", "1")
        eng.set_option("mh_pair", pair)
        eng.set_option("mh_spec", depth)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], start)
        eng.set_rng(99, 3)
        if device_draws:
            acc, samples = eng.mh_chain(None, None, 0.4, -0.6, 0.9, 0.3, 2.0, 0.35, n_mc=steps)
        else:
            acc, samples = eng.mh_chain(normals, uniforms, 0.4, -0.6, 0.9, 0.3, 2.0, 0.35)
        _, p = eng.get_params()
        eng.close()
        return acc, samples, p
    for device_draws in (False, True):
        ref = run(0, device_draws)
        assert 0 < ref[0].sum() < steps * N
        # ... and d such steps per launch by speculation (every node of the candidate heap = its state and its proposal)
        for depth in (0, 2, 3, -1):
            got = run(1, device_draws, depth)
            for a, b in zip(ref, got):
                assert np.array_equal(a, b), (device_draws, depth)
