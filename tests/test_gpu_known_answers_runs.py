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
            if n % 6 == 0:          # the oracle's search on the same problem (a sixth of the networks: CPU time)
                # The adaptive objective is rippled (every theta takes its own steps) and a subject's profile can hold
                # several shallow minima: two golden-section searches that round differently may settle in neighbouring
                # dips (measured: 4 of 30 subjects apart by up to 1e-3, either way).  Most subjects agree to rounding,
                # and the population sums to a fraction of the distance to the stored value.
                _, best = oracle_minimum(tp, data, nn)
                diff = np.abs(sse - best)
                assert np.median(diff) <= 1e-7 * np.max(best) and np.mean(diff <= 2e-6 * np.max(best)) >= 0.75, (tag, n, name)
                assert abs(sse.sum() - best.sum()) <= 3e-3 * best.sum(), (tag, n, name)
    for e in engines.values():
        e.close()
    r = np.array(ratios["train"])
    assert lo <= r.min() and r.max() <= hi and np.median(r) >= 0.97, (tag, r.min(), np.median(r), r.max())
    for name in ("valid", "valid_nonoise"):
        r = np.array(ratios[name])
        assert r.max() <= 1.03, (tag, name, r.max())
        if tag in ("0.0", "0.001", "0.01"):
            assert np.median(r) >= 0.90, (tag, name, np.median(r))


def test_stored_cpeptide_end_points_are_unconverged_and_continue_into_the_products_envelope():
    """The reference's `train` (src/parameter-estimation.jl:340-386) left 25 end points (network, 57 betas) in
    source_data/cude_neural_parameters.jld2 and 25 more in the `_sigma` file.  Through the product, on the reference's
    own (adaptive) objective and the 57 subjects the betas belong to (identified by matching per-subject refits, as
    tests/test_soft_pins.py does on the CPU):
      * their objectives span 0.33 ... 1.02 (median 0.43) -- the product's own 25 runs of the same recipe end at
        0.29 ... 0.44 (median 0.34; tools/envelope_cpeptide.py, profiles/r04/envelope_cpeptide.txt);
      * they are NOT stationary: the gradient with respect to the betas is 1 ... 55 % of what it is at a clearly
        perturbed point (a converged fit: < 1e-6, tests/test_soft_pins.py::test_cpeptide_stationarity_of_stored_optimum
        after re-fitting the betas);
      * the product's second stage (Optim's L-BFGS + BackTracking restated, held to an independent oracle in
        tests/test_lbfgs_oracle.py) started AT them lowers the objective by a median 6-8 %, up to 65 %, and ends at
        0.32 ... 0.48: inside the product's envelope.
    So the gap between the two distributions is where the reference's second stage STOPPED, not a different objective,
    initialiser or data reading: every stored end point is a valid iterate of the same problem."""
    from scipy.optimize import linear_sum_assignment
    from cude.engine import Engine
    import os
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ohashi_cude.npz")))
    tp, arch = g["timepoints"], (2, 4, 2)

    def engine_for(rows):
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_population_cpep(tp, g["glucose"][rows], g["cpeptide"][rows], g["ages"][rows], g["t2dm"][rows])
        return eng
    train82 = np.flatnonzero(np.isin(g["subject_no"], g["train_subject_numbers"]))
    k_best = int(g["best_model_index"]) - 1
    e82 = engine_for(train82)
    e82.set_params(g["nn_2x4x4x1"][k_best], np.zeros(len(train82)))
    beta_hat, _, _ = e82.fit_conditional(-4.0, 3.0, n_grid=141, n_iters=45)
    e82.close()
    cost = np.abs(g["betas_train"][k_best][:, None] - beta_hat[None, :])
    r, c = linear_sum_assignment(cost)
    assert np.median(cost[r, c]) < 5e-3
    eng = engine_for(train82[c[np.argsort(r)]])
    for sfx in ("", "_sigma"):
        nns, betas = g["nn_2x4x4x1" + sfx], g["betas_train" + sfx]
        before, gb_ratio = [], []
        for k in range(25):
            eng.set_params(nns[k], betas[k])
            loss, _, gb = eng.loss_grad()
            eng.set_params(nns[k] * 1.1, betas[k] + 0.2)
            _, _, gb1 = eng.loss_grad()
            before.append(loss)
            gb_ratio.append(np.max(np.abs(gb)) / np.max(np.abs(gb1)))
        before, gb_ratio = np.array(before), np.array(gb_ratio)
        assert 0.30 < before.min() < 0.36 and 0.40 < np.median(before) < 0.46 and before.max() > 0.9
        assert np.median(gb_ratio) > 0.05 and gb_ratio.min() > 5e-3           # nowhere near a stationary point
        _, _, after = eng.train_restarts(nns, betas, 0, 1e-2, 1000)
        drop = (before - after) / before
        assert np.all(after <= before + 1e-12) and np.median(drop) > 0.04 and drop.max() > 0.5
        assert 0.29 < after.min() and after.max() < 0.50 and np.median(after) < 0.41
    eng.close()
