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
