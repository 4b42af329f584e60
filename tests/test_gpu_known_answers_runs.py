"""The soft known answers of tests/test_known_answers_runs.py THROUGH THE PRODUCT: for every one of the reference's 110
stored suppression networks with live hidden layers, libcude_hip.so's adaptive mode (the reference's solver settings)
and its on-device per-subject fit (cude_fit_conditional: `validate_suppression_model`'s inner problem,
suppression/src/suppression_model.jl:179-222, for all subjects at once) must (i) find the same per-subject minima as the
oracle's search and (ii) land at or below the objectives the reference stored, as close to them as the oracle does."""
import numpy as np
import pytest

from test_known_answers_runs import ARCH, BOUNDS, RUNS, load_runs, oracle_minimum

pytestmark = pytest.mark.gpu


def _product_minimum(eng, nn, lo=-8.0, hi=6.0, n_grid=141, n_iters=45):
    eng.set_params(nn, np.zeros(eng.N))
    x, _, sse = eng.fit_conditional(lo, hi, n_grid=n_grid, n_iters=n_iters)
    return x, sse


@pytest.mark.parametrize("tag", RUNS)
def test_product_reaches_every_stored_objective(tag):
    from cude.engine import Engine
    tp, sets, runs = load_runs()
    run = runs[tag]
    engines = {}
    for name, data in sets.items():
        engines[name] = Engine("supp", ARCH, n_steps=0, lam=0.0)
        engines[name].set_population_supp(tp, data)
    lo, hi = BOUNDS[tag]
    ratios = {k: [] for k in sets}
    for n, nn in enumerate(run["nn"]):
        for name, data in sets.items():
            stored = run[name][n]
            if not np.isfinite(stored):
                continue
            _, sse = _product_minimum(engines[name], nn)
            value = sse.sum() / data.shape[2] + (run["lam"] * float(nn @ nn) if name == "train" else 0.0)
            ratios[name].append(value / stored)
            if n % 6 == 0:          # the oracle's search on the same problem (a quarter of the networks: CPU time)
                _, best = oracle_minimum(tp, data, nn)
                assert np.max(np.abs(sse - best)) <= 2e-6 * np.max(best), (tag, n, name)
    for e in engines.values():
        e.close()
    r = np.array(ratios["train"])
    assert lo <= r.min() and r.max() <= hi and np.median(r) >= 0.97, (tag, r.min(), np.median(r), r.max())
    for name in ("valid", "valid_nonoise"):
        r = np.array(ratios[name])
        assert r.max() <= 1.03, (tag, name, r.max())
        if tag in ("0.0", "0.001", "0.01"):
            assert np.median(r) >= 0.90, (tag, name, np.median(r))
