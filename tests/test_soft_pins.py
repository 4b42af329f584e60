"""Soft pins of the CPU oracle against the reference's STORED results (tests/golden/*.npz).

The reference has no tests and cannot run here (Julia).  Next to the known answers of tests/test_known_answers.py
(suppression path, stored objectives) and tests/test_figure_pins.py (c-peptide path, the reference's vector figures),
these checks tie the restated equations, unit conversions, van Cauter constants, MLP parameter layout and loss
definitions to numbers the reference itself produced.  Tolerances are the adaptive-solver level
(reference: reltol 1e-3), not the 1e-6 kernel-parity level.
"""
import os

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _argmin_1d(f, lo, hi, n_grid=71, iters=45):
    """Vectorised bounded 1-D minimisation (grid + golden section) of f: (N,) -> (N,)."""
    grid = np.linspace(lo, hi, n_grid)
    vals = np.stack([f(np.full_like(lo_, g)) for g, lo_ in ((g, np.zeros(f.n)) for g in grid)])
    k = np.clip(np.argmin(vals, axis=0), 1, n_grid - 2)
    a, b = grid[k - 1], grid[k + 1]
    gr = (np.sqrt(5) - 1) / 2
    c, d = b - gr * (b - a), a + gr * (b - a)
    fc, fd = f(c), f(d)
    for _ in range(iters):
        left = fc < fd
        b = np.where(left, d, b)
        a = np.where(left, a, c)
        c2, d2 = b - gr * (b - a), a + gr * (b - a)
        fc, fd = f(c2), f(d2)
        c, d = c2, d2
    x = 0.5 * (a + b)
    return x, f(x)


class _Sse:
    def __init__(self, fn, n):
        self.fn, self.n = fn, n

    def __call__(self, x):
        return self.fn(x)


def _cpep_refit(nn, g):
    import c_oracle as co
    N = g["glucose"].shape[0]

    def sse(beta):
        return co.cpep(g["timepoints"], g["glucose"], g["cpeptide"], g["ages"], g["t2dm"], (2, 4, 2), nn, beta, 30,
                       2, want_grad=False)["sse"]
    return _argmin_1d(_Sse(sse, N), -4.0, 3.0)


def test_cpeptide_stored_betas_are_recovered():
    """With stored model k=0 (37 weights, SimpleChains column-major layout) the per-subject refit of beta on
    the 117 Ohashi subjects reproduces each of the 57 stored training betas."""
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    nn, stored = g["nn_2x4x4x1"][0], g["betas_train"][0]
    beta_hat, sse = _cpep_refit(nn, g)
    cost = np.abs(stored[:, None] - beta_hat[None, :])
    r, c = linear_sum_assignment(cost)
    d = cost[r, c]
    assert np.median(d) < 5e-3, np.median(d)
    assert np.quantile(d, 0.9) < 3e-2, np.quantile(d, 0.9)
    assert np.mean(sse[c]) < 0.5           # nmol^2/L^2; SURVEY probe: mean 0.31, median 0.16
    assert np.median(sse[c]) < 0.3
    # the 57 training subjects are a stratified 70 % draw from the 82 subjects of the reference's prepared
    # `train` set (data/ohashi.jld2, c-peptide/02-conditional.jl:19).  The blind match over all 117 subjects
    # lands in that set for 53 of 57 (expected by chance: 40 +- 3.5), and the match restricted to it is as tight.
    in_train = np.isin(g["subject_no"], g["train_subject_numbers"])
    assert in_train.sum() == 82 and in_train[c].sum() >= 50
    idx = np.flatnonzero(in_train)
    r2, c2 = linear_sum_assignment(cost[:, idx])
    d2 = cost[:, idx][r2, c2]
    assert np.median(d2) < 5e-3 and np.quantile(d2, 0.9) < 3e-2
    assert np.mean(sse[idx[c2]]) < 0.4


def _match_in_train(g, stored, beta_hat):
    idx = np.flatnonzero(np.isin(g["subject_no"], g["train_subject_numbers"]))
    cost = np.abs(stored[:, None] - beta_hat[None, idx])
    r, c = linear_sum_assignment(cost)
    return cost[r, c], idx[c]


def test_covariate_model_stored_betas_are_recovered():
    """Covariate cUDE (3 -> 4 -> 4 -> 1, network input [dG, exp(beta), age] with the age in years,
    src/c-peptide-models.jl:96-104; stored run of c-peptide/07-covariate-inclusion.jl:59-65): refitting beta per
    subject with a stored network reproduces the stored training betas.  Negative control: feeding the age and
    exp(beta) in the other order does not."""
    import c_oracle as co
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    k = int(g["best_model_index_cov"]) - 1
    nn, stored = g["nn_3x4x4x1_cov"][k], g["betas_train_cov"][k]
    N = g["glucose"].shape[0]

    def refit(weights):
        def sse(beta):
            return co.cpep(g["timepoints"], g["glucose"], g["cpeptide"], g["ages"], g["t2dm"], (3, 4, 2), weights,
                           beta, 30, 2, want_grad=False, covariate=True)["sse"]
        return _argmin_1d(_Sse(sse, N), -4.0, 3.0)
    beta_hat, sse = refit(nn)
    d, who = _match_in_train(g, stored, beta_hat)
    assert np.median(d) < 5e-3 and np.quantile(d, 0.9) < 0.1, (np.median(d), np.quantile(d, 0.9))
    assert np.mean(sse[who]) < 0.5
    swapped = nn.copy()                                  # columns 1 and 2 of W1 (4 x 3, column-major) exchanged
    swapped[4:8], swapped[8:12] = nn[8:12], nn[4:8]
    beta_bad, sse_bad = refit(swapped)
    d_bad, who_bad = _match_in_train(g, stored, beta_bad)
    assert np.median(d_bad) > 0.05 or np.mean(sse_bad[who_bad]) > 1.0


def test_sigma_run_stored_betas_are_recovered():
    """A second stored training run of the 2 -> 4 -> 4 -> 1 model (source_data/cude_neural_parameters_sigma.jld2)."""
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    k = int(g["best_model_index_sigma"]) - 1
    beta_hat, sse = _cpep_refit(g["nn_2x4x4x1_sigma"][k], g)
    d, who = _match_in_train(g, g["betas_train_sigma"][k], beta_hat)
    assert np.median(d) < 5e-3 and np.quantile(d, 0.9) < 3e-2
    assert np.mean(sse[who]) < 0.5


def test_cpeptide_row_major_layout_is_rejected():
    """Negative control: reading the weight matrices row-major destroys the match (pins the layout)."""
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    nn = g["nn_2x4x4x1"][0].copy()
    w1 = nn[0:8].reshape(2, 4).T.copy()     # stored column-major 4x2 -> reinterpret as row-major
    w2 = nn[12:28].reshape(4, 4).T.copy()
    nn[0:8], nn[12:28] = w1.reshape(-1), w2.reshape(-1)
    beta_hat, sse = _cpep_refit(nn, g)
    cost = np.abs(g["betas_train"][0][:, None] - beta_hat[None, :])
    r, c = linear_sum_assignment(cost)
    assert np.median(cost[r, c]) > 0.05 or np.mean(sse[c]) > 1.0


def test_cpeptide_stationarity_of_stored_optimum():
    """At (stored nn, refit betas of the matched subjects) the population gradient w.r.t. beta vanishes and
    the gradient w.r.t. the network is small relative to its size at a perturbed point."""
    import c_oracle as co
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    nn, stored = g["nn_2x4x4x1"][0], g["betas_train"][0]
    beta_hat, _ = _cpep_refit(nn, g)
    cost = np.abs(stored[:, None] - beta_hat[None, :])
    _, c = linear_sum_assignment(cost)
    sel = np.sort(c)
    args = (g["timepoints"], g["glucose"][sel], g["cpeptide"][sel], g["ages"][sel], g["t2dm"][sel], (2, 4, 2))
    at = co.cpep(*args, nn, beta_hat[sel], 30, 2)
    off = co.cpep(*args, nn * 1.1, beta_hat[sel] + 0.2, 30, 2)
    assert np.max(np.abs(at["g_beta"])) < 1e-6
    assert np.linalg.norm(at["g_nn"]) < 0.05 * np.linalg.norm(off["g_nn"])


@pytest.mark.parametrize("noise", ["", "_nonoise"])
@pytest.mark.parametrize("model", [0, 2, 7])
def test_suppression_stored_validation_results(model, noise):
    """validate_suppression_model (suppression_model.jl:179-222) fits theta on 30 validation subjects with the
    network FROZEN -- nothing unsaved enters, so the stored `losses_valid*[n]` / `correlations_valid*[n]`
    (suppression.jl:58-64) are functions of stored quantities only.  The frozen-network loss separates per subject;
    the per-subject global minimum must be <= the reference's L-BFGS result from its single start and close to it
    (measured 0.955 ... 0.993 of the stored value), and where the reference's fit converged everywhere (models 0
    and 7) the rank correlation with the ground truth agrees to 0.01."""
    import c_oracle as co
    from scipy.stats import spearmanr
    g = dict(np.load(os.path.join(GOLD, "suppression_lambda0.npz")))
    nn, data, tp = g["nn_4x3x5x1"][model], g["validation_data" + noise], g["timepoints"]
    N = data.shape[2]

    def sse(theta):
        return co.supp(tp, data, (4, 3, 5), nn, theta, 0.0, 60, want_grad=False)["sse"]
    theta_hat, best = _argmin_1d(_Sse(sse, N), -8.0, 5.0, n_grid=161)
    loss, stored = best.sum() / N, g["losses_valid" + noise][model]
    assert 0.93 * stored <= loss <= 1.005 * stored
    if model in (0, 7):
        rho = spearmanr(theta_hat, g["gt_validation_param" + noise])[0]
        assert abs(rho - g["correlations_valid" + noise][model]) < 0.01


@pytest.mark.parametrize("model", [0, 2, 3, 7, 12])
def test_suppression_stored_losses_and_correlations(model):
    """min_theta dataterm(theta, stored nn_n) must be <= and close to the stored final loss
    (lambda = 0; theta was not saved by the reference: suppression/suppression.jl:76-91), and the re-estimated
    thetas must rank the subjects like the reference's trained ones did: the stored
    `correlations[n] = corspearman(gt_sup_param, res.u.theta)` (:56) is reproduced within 0.03, sign included
    (the orientation of the latent parameter differs between trained models; model 12 is anti-correlated)."""
    import c_oracle as co
    from scipy.stats import spearmanr
    g = dict(np.load(os.path.join(GOLD, "suppression_lambda0.npz")))
    nn, data, tp = g["nn_4x3x5x1"][model], g["group_data"], g["timepoints"]
    N = data.shape[2]

    def sse(theta):
        return co.supp(tp, data, (4, 3, 5), nn, theta, 0.0, 60, want_grad=False)["sse"]
    theta_hat, best = _argmin_1d(_Sse(sse, N), -6.0, 4.0, n_grid=101)
    loss = best.sum() / N
    stored = g["losses"][model]
    assert loss <= stored * 1.02
    assert loss >= stored * 0.80
    assert abs(spearmanr(theta_hat, g["gt_sup_param"])[0] - g["correlations"][model]) < 0.03
