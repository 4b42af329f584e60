"""cude_adam_run (captured hipGraph of one optimiser iteration, replayed) vs the step-by-step cude_adam_step,
for both c-peptide paths and the suppression model; also times the small-population case it exists for."""
import os
import time

import numpy as np
import pytest
import torch  # noqa: F401

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu


def _cpep_engine(c, arch, n_state=3):
    from cude.engine import Engine
    eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    return eng


@pytest.mark.parametrize("N", [57, 5000])
def test_adam_run_equals_stepwise(N):
    arch = (2, 6, 2)
    c = make_cpep_case(N, arch)
    a = _cpep_engine(c, arch)
    a.adam_init(1e-2)
    ref = [a.adam_step() for _ in range(12)]
    nn_a, cond_a = a.get_params()
    a.close()
    b = _cpep_engine(c, arch)
    b.adam_init(1e-2)
    got = np.concatenate([b.adam_run(5), b.adam_run(7)])          # two replays of the same captured graph
    nn_b, cond_b = b.get_params()
    # changing the parameters / hyper-parameters must not be baked into the graph
    b.set_params(c["nn"], c["beta"])
    b.adam_init(1e-2)
    again = b.adam_run(12)
    b.close()
    assert np.array_equal(got, ref) and np.array_equal(again, ref)
    assert np.array_equal(nn_a, nn_b) and np.array_equal(cond_a, cond_b)


def test_adam_run_suppression_and_failure():
    from cude.engine import Engine
    c = make_supp_case(300)
    e1 = Engine("supp", c["arch"], n_steps=30, lam=0.01)
    e1.set_population_supp(c["tp"], c["data"])
    e1.set_params(c["nn"], c["theta"])
    e1.adam_init(1e-3)
    ref = [e1.adam_step() for _ in range(6)]
    e1.set_params(c["nn"], c["theta"])
    e1.adam_init(1e-3)
    got = e1.adam_run(6)
    assert np.array_equal(got, ref)
    bad = c["theta"].copy()
    bad[3] = np.nan
    e1.set_params(c["nn"], bad)
    e1.adam_init(1e-3)
    tr = e1.adam_run(3)
    assert np.all(np.isinf(tr)) and e1.n_failed() == 1           # update skipped every time
    nn_after, _ = e1.get_params()
    assert np.array_equal(nn_after, c["nn"])
    e1.close()
    # graph capture as the FIRST gradient evaluation of a fresh context (nothing may be allocated under capture)
    e2 = Engine("supp", c["arch"], n_steps=30, lam=0.01)
    e2.set_population_supp(c["tp"], c["data"])
    e2.set_params(c["nn"], c["theta"])
    e2.adam_init(1e-3)
    assert np.array_equal(e2.adam_run(6), ref)
    e2.close()


def test_small_population_step_latency():
    """The reference's own training size (57 subjects): per-iteration device time with graph replay."""
    arch = (2, 4, 2)
    c = make_cpep_case(57, arch)
    eng = _cpep_engine(c, arch, n_state=2)
    eng.adam_init(1e-2)
    eng.adam_run(50)
    t0 = time.perf_counter()
    eng.adam_run(1000)
    dt_graph = (time.perf_counter() - t0) / 1000
    t0 = time.perf_counter()
    for _ in range(200):
        eng.adam_step()
    dt_step = (time.perf_counter() - t0) / 200
    eng.close()
    print(f"57 subjects: {dt_graph*1e6:.1f} us/iteration (graph replay) vs {dt_step*1e6:.1f} us (stepwise with loss read-back)")
    assert dt_graph > 0 and dt_step > 0          # timings are informational (box-dependent), not asserted
