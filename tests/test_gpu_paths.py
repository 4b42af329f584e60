"""Both c-peptide gradient paths against the oracle and against each other: the one-lane-per-subject kernel
(cude_cpep.hip, CUDE_CPEP_PATH=1) and the time-split kernels (cude_cpep2.hip) for several chunk counts,
including uneven chunks and chunks without observations."""
import os

import numpy as np
import pytest
import torch  # noqa: F401

from conftest import make_cpep_case

pytestmark = pytest.mark.gpu


def _run(path, c, arch, n_state, n_steps):
    from cude.engine import Engine
    old = os.environ.get("CUDE_CPEP_PATH")
    os.environ["CUDE_CPEP_PATH"] = path
    try:
        eng = Engine("cpep", arch, n_steps=n_steps, n_state=n_state)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    finally:
        if old is None:
            del os.environ["CUDE_CPEP_PATH"]
        else:
            os.environ["CUDE_CPEP_PATH"] = old
    eng.set_params(c["nn"], c["beta"])
    f = eng.forward(want_sse=True)
    loss, g_nn, g_cond = eng.loss_grad()
    eng.adam_init(1e-2)
    steps = [eng.adam_step() for _ in range(3)]
    nn, cond = eng.get_params()
    eng.close()
    return dict(fwd=f["loss"], sse=f["sse"], loss=loss, g_nn=g_nn, g_cond=g_cond, steps=steps, nn=nn, cond=cond)


@pytest.mark.parametrize("arch,n_state,n_steps,N", [((2, 6, 2), 3, 30, 700), ((2, 4, 2), 2, 14, 130), ((3, 4, 2), 2, 30, 65),
                                                     ((2, 7, 2), 3, 30, 70), ((2, 5, 2), 2, 12, 64), ((2, 6, 1), 2, 30, 65),
                                                     ((2, 6, 3), 3, 30, 64), ((3, 6, 2), 2, 30, 66)])
def test_paths_agree_with_oracle_and_each_other(arch, n_state, n_steps, N):
    import c_oracle as co
    c = make_cpep_case(N, arch, n_steps=n_steps)
    ref = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], n_steps, n_state,
                  covariate=(arch[0] == 3))
    nb = (N + 63) // 64
    mixed = [f"3:{nb - 1}:2", f"3:{max(nb // 2, 1)}:{n_steps}"] if nb > 1 else []     # one-lane blocks + time-split remainder
    results = {p: _run(p, c, arch, n_state, n_steps) for p in ["1", "2:2", "2:3", "2:7", f"2:{n_steps}"] + mixed}
    for p, r in results.items():
        assert abs(r["fwd"] - ref["loss"]) < 1e-10 * ref["loss"], p
        assert abs(r["loss"] - ref["loss"]) < 1e-10 * ref["loss"], p
        assert np.max(np.abs(r["sse"] - ref["sse"])) < 1e-10, p
        assert np.max(np.abs(r["g_nn"] - ref["g_nn"])) < 1e-9 * np.max(np.abs(ref["g_nn"])), p
        assert np.max(np.abs(r["g_cond"] - ref["g_beta"])) < 1e-9 * np.max(np.abs(ref["g_beta"])), p
    base = results["1"]
    for p, r in results.items():
        assert np.allclose(r["steps"], base["steps"], rtol=1e-11), p
        assert np.max(np.abs(r["nn"] - base["nn"])) < 1e-10 and np.max(np.abs(r["cond"] - base["cond"])) < 1e-10, p


def test_chunked_path_failure_convention():
    from cude.engine import Engine
    arch, N = (2, 4, 2), 200
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch)                       # small N -> time-split path
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    beta = c["beta"].copy()
    beta[17] = np.inf
    eng.set_params(c["nn"], beta)
    assert eng.forward()["loss"] == np.inf and eng.n_failed() == 1
    assert eng.loss_grad()[0] == np.inf
    nn_bad = c["nn"].copy()
    nn_bad[20] = np.nan
    eng.set_params(nn_bad, c["beta"])
    assert eng.forward()["loss"] == np.inf and eng.n_failed() == N
    eng.adam_init(1e-2)
    assert eng.adam_step() == np.inf                 # update skipped: parameters unchanged
    nn_after, _ = eng.get_params()
    assert np.array_equal(np.isnan(nn_after), np.isnan(nn_bad))
    eng.close()


@pytest.mark.parametrize("arch", [(4, 3, 5), (4, 6, 2)])
def test_suppression_kept_activations_equal_recomputation(arch, monkeypatch):
    """The suppression gradient kernel either keeps the network activations of the forward sweep in HBM (small
    populations: latency-bound) or recomputes them in the reverse sweep (large ones).  Same operations in the same
    order on the same values: the two variants must agree bit for bit, for one set and for side-by-side sets."""
    from conftest import make_supp_case
    from cude.engine import Engine
    c = make_supp_case(130, arch)
    rng = np.random.default_rng(1)
    nn_sets = c["nn"][None, :] * (1.0 + 0.1 * rng.standard_normal((3, c["nn"].size)))
    th_sets = c["theta"][None, :] + 0.2 * rng.standard_normal((3, 130))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CUDE_SUPP_STORE", mode)
        eng = Engine("supp", arch, n_steps=30, lam=0.01)
        eng.set_population_supp(c["tp"], c["data"])
        eng.set_params(c["nn"], c["theta"])
        out[mode] = eng.loss_grad() + eng.multistart_loss_grad(nn_sets, th_sets)
        eng.close()
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)
