"""Every trained network the reference stored for the suppression experiment, as a soft known answer (round 4).

`suppression/results/lambda={0.0, 0.001, 0.01, 0.010000000000000002, 0.1}.jld2` hold 110 trained 4->3x5->1 networks
with live hidden layers, their final training objectives and the objectives `validate_suppression_model` returned on
the two validation sets (suppression/suppression.jl:53-91; fixtures: tools/make_golden.py).  The conditional parameters
were not saved, but with the network frozen the loss separates per subject (suppression_model.jl:117-130), so

    min_theta dataterm(theta, nn_n) + lambda |nn_n|^2

is a function of STORED quantities only, must not exceed what the reference's joint optimisation reached, and must be
close to it.  Measured over all 110 networks with the oracle's adaptive solve (the reference's own solver settings;
oracle/cude_oracle.c cude_oracle_supp_adaptive), ratio = recomputed / stored:

    lambda      training objective (min / median / max)     validation sets (median; every ratio <= 1.03)
    0           0.881 / 0.979 / 1.000                        0.964 / 0.978
    0.001       0.895 / 0.976 / 1.030                        0.920 / 0.914
    0.01 (x2)   0.968 / 0.995 / 1.000                        0.910 / 0.920   (one stored validation fit failed: Inf)
    0.1         0.975 / 0.995 / 0.999                        0.533 / 0.517

(validation: the reference starts ONE L-BFGS run from a random point and often ends in a worse local minimum than the
per-subject global search, the more so the flatter the regularised network's dependence on theta is: there the stored
value is an upper bound only).  With lambda = 1 the networks collapse and the same quantity is a hard known answer to
2e-9: tests/test_known_answers.py.

NOT usable, for the record (tools/scan_reference_runs.py, profiles/r04/reference_runs_scan.txt): the directories
suppression/results/init_run and test_run (12 x 50 networks 4->3x3->1 and 8 x 10 networks on 60 subjects).  Their
collapsed networks (lambda >= 10) make the objective independent of theta, yet suppression_loss as committed gives 11.4
where 71.93 is stored (all 50 runs of lambda = 100 agree on 71.93291897 to 1e-11): those files were written by an earlier
revision of the experiment (60 training subjects, other numbers of kept runs) whose objective is not part of the
reference tree.  No per-state weighting, initial condition or output scaling of the committed model reproduces the
stored numbers."""
import os

import numpy as np
import pytest

from test_soft_pins import _Sse, _argmin_1d

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARCH = (4, 3, 5)
RUNS = ["0.0", "0.001", "0.01", "0.01b", "0.1"]


def load_runs():
    g0 = dict(np.load(os.path.join(GOLD, "suppression_lambda0.npz")))
    gm = dict(np.load(os.path.join(GOLD, "suppression_lambda_mid.npz")))
    sets = {"train": g0["group_data"], "valid": g0["validation_data"], "valid_nonoise": g0["validation_data_nonoise"]}
    runs = {"0.0": dict(lam=0.0, nn=g0["nn_4x3x5x1"], train=g0["losses"], valid=g0["losses_valid"],
                        valid_nonoise=g0["losses_valid_nonoise"])}
    for tag in RUNS[1:]:
        runs[tag] = dict(lam=float(gm["lam_" + tag]), nn=gm["nn_" + tag], train=gm["losses_" + tag],
                         valid=gm["losses_valid_" + tag], valid_nonoise=gm["losses_valid_nonoise_" + tag])
    return g0["timepoints"], sets, runs


def oracle_minimum(tp, data, nn, lo=-8.0, hi=6.0, n_grid=141):
    """(theta_hat[N], scaled SSE at it [N]): per-subject global minimum of the frozen-network data term, adaptive solve"""
    import c_oracle as co
    scale = data.max(axis=1).mean(axis=1)

    def sse(theta):
        r = (co.supp_adaptive(tp, data, ARCH, nn, theta) - data) / scale[:, None, None]
        return np.sum(r * r, axis=(0, 1))
    return _argmin_1d(_Sse(sse, data.shape[2]), lo, hi, n_grid=n_grid)


BOUNDS = {"0.0": (0.85, 1.005), "0.001": (0.85, 1.035), "0.01": (0.96, 1.001), "0.01b": (0.96, 1.001), "0.1": (0.96, 1.001)}


@pytest.mark.parametrize("tag", RUNS)
def test_every_stored_training_objective_is_reached_from_stored_quantities(tag):
    tp, sets, runs = load_runs()
    run = runs[tag]
    data = sets["train"]
    lo, hi = BOUNDS[tag]
    ratios = []
    for n, nn in enumerate(run["nn"]):
        _, best = oracle_minimum(tp, data, nn)
        value = best.sum() / data.shape[2] + run["lam"] * float(nn @ nn)
        ratios.append(value / run["train"][n])
    ratios = np.array(ratios)
    assert lo <= ratios.min() and ratios.max() <= hi, (tag, ratios.min(), ratios.max())
    assert np.median(ratios) >= 0.97, (tag, np.median(ratios))


@pytest.mark.parametrize("which", ["valid", "valid_nonoise"])
@pytest.mark.parametrize("tag", RUNS)
def test_every_stored_validation_objective_bounds_the_per_subject_minimum(tag, which):
    tp, sets, runs = load_runs()
    run = runs[tag]
    data = sets[which]
    ratios = []
    for n, nn in enumerate(run["nn"]):
        stored = run[which][n]
        if not np.isfinite(stored):                      # (the reference's own fit failed: `return p_init_best, Inf`)
            continue
        _, best = oracle_minimum(tp, data, nn)
        ratios.append(best.sum() / data.shape[2] / stored)
    ratios = np.array(ratios)
    assert len(ratios) >= len(run["nn"]) - 1
    assert ratios.max() <= 1.03, (tag, which, ratios.max())
    if tag in ("0.0", "0.001", "0.01"):
        assert np.median(ratios) >= 0.90, (tag, which, np.median(ratios))


def test_c_adaptive_solver_follows_the_python_statement():
    """cude_oracle_supp_adaptive is cude_oracle.solve_adaptive + supp_rhs operation for operation (the known answers of
    test_known_answers.py pin the Python statement).  The two round the network's dot products in a different order,
    and step-size control amplifies a last-place difference of the right-hand side to 1e-9 ... 5e-7 in the trajectory
    (DESIGN.md 2): measured 1e-14 ... 3e-8 on these subjects."""
    import c_oracle as co
    import cude_oracle as o
    tp, sets, runs = load_runs()
    data = sets["train"][:, :, :6]
    nn = runs["0.0"]["nn"][3]
    theta = np.linspace(-1.5, 1.0, 6)
    got = co.supp_adaptive(tp, data, ARCH, nn, theta)
    for i in range(6):
        et = float(np.exp(theta[i]))
        rhs = lambda t, u: [float(v) for v in o.supp_rhs(np, nn, et, ARCH, t, [np.float64(x) for x in u])]
        sol = np.array(o.solve_adaptive(rhs, list(data[:, 0, i]), list(tp), abstol=1e-6, reltol=1e-3))
        assert np.max(np.abs(sol.T - got[:, :, i])) < 5e-7 * np.max(np.abs(sol))
