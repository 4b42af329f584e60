"""SAEM E-step (per-subject Metropolis-Hastings, src/saem.jl:86-108,177-186) on the GPU vs the oracle chain with
the SAME host-supplied draws -- both the fused device call (cude_mh_estep) and the host loop over forward
launches (api.mcmc_steps) -- and the config-5 size (1e4 subjects x 100 MC steps) through invariants."""
import time

import numpy as np
import pytest
import torch  # noqa: F401

from conftest import make_cpep_case

pytestmark = pytest.mark.gpu


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
