"""Adaptive Tsit5 on the device (cude_config.n_steps = 0, csrc/cude_adaptive.hip): what the reference actually runs
-- `solve(model.problem, p = theta, saveat = timepoints)` (src/parameter-estimation.jl:59), `solve(ensemble, Tsit5(),
EnsembleThreads(); saveat...)` (suppression/src/suppression_model.jl:113,123), src/saem.jl:52 -- against the oracle's
adaptive restatement (the one the reference's stored objectives pin to 4e-10, tests/test_known_answers.py) and against
those stored objectives themselves, THROUGH libcude_hip.so.

How close two correct implementations of this solver can be: at OrdinaryDiffEq's default tolerances the c-peptide
solve is ill-conditioned with respect to rounding -- the piecewise-linear glucose forcing has kinks the step controller
does not know about, and the error estimate is a cancelling sum -- so that perturbing the right-hand side of the ORACLE
by 1e-16 (one rounding) moves its own trajectories by 1e-9 ... 2e-6 (measured, tools/adaptive_conditioning.py).  The
device's activations differ from libm by ~3e-16, hence the c-peptide comparisons below are held to 2e-7 (median over
subjects) / 2e-5 (98 % of them) / 1e-3 (all: a subject whose error estimate sits within rounding of the acceptance
threshold takes a different step) -- against the solver's own error of 2e-4 median, 5e-3 max at these tolerances,
and the 1.2e-4 resolution at which the reference's figures pin the path.  The suppression model is smooth and is
held to 1e-7 relative; the reference's stored objectives are reproduced to 2e-9."""


def _close(err_per_subject):
    e = np.asarray(err_per_subject)
    return np.median(e) <= 2e-7 and np.quantile(e, 0.98) <= 2e-5 and e.max() <= 1e-3
import os

import numpy as np
import pytest

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("arch", [(2, 4, 2), (2, 6, 2), (3, 4, 2), (2, 8, 2), (2, 4, 3), (2, 5, 2), (2, 7, 2), (2, 6, 1),
                                  (3, 6, 2)])
def test_cpep_adaptive_matches_the_oracle(arch):
    import torch  # noqa: F401
    import c_oracle as co
    from cude.engine import Engine
    N = 131
    c = make_cpep_case(N, arch)
    cov = arch[0] == 3
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    out = eng.forward(want_sse=True, want_traj=True)
    ref = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], np.exp(c["beta"]), c["tp"],
                           covariate=cov)
    assert np.all(np.isfinite(ref))
    err = np.abs(out["traj"][0].T - ref)                              # plasma c-peptide at the observation times
    assert _close(err.max(axis=1)), (err.max(), np.median(err.max(axis=1)))
    sse = np.sum((ref - c["obs"]) ** 2, axis=1)
    assert np.max(np.abs(out["sse"] - sse)) <= 1e-3 * np.max(sse)
    assert abs(out["loss"] - sse.mean()) <= 1e-4 * sse.mean()
    assert np.array_equal(out["sse"], np.sum((out["traj"][0].T - c["obs"]) ** 2, axis=1)) or \
        np.allclose(out["sse"], np.sum((out["traj"][0].T - c["obs"]) ** 2, axis=1), rtol=1e-13)
    # dense output on the grid of the reference's model-fit figures (saveat = 0:0.1:120): same accepted steps
    times = np.arange(0.0, 120.0001, 0.1)
    dense = eng.simulate(times)[0].T
    ref_d = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], np.exp(c["beta"]), times,
                             covariate=cov)
    assert _close(np.max(np.abs(dense - ref_d), axis=1))
    assert np.max(np.abs(dense[:, ::300] - out["traj"][0].T)) <= 1e-12      # saveat does not change the steps taken
    # the fixed-step solve of the same model differs by the reference solver's own error (~1e-3), not by 1e-9
    fix = Engine("cpep", arch, n_steps=240, n_state=2)
    fix.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    fix.set_params(c["nn"], c["beta"])
    gap = np.max(np.abs(fix.forward(want_traj=True)["traj"][0].T - ref))
    assert 1e-7 < gap < 5e-2
    fix.close()
    eng.close()


def test_cpep_adaptive_profiles_screening_and_fits_use_the_same_solver():
    import torch  # noqa: F401
    import c_oracle as co
    from cude.engine import Engine
    arch, N = (2, 4, 2), 70
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    values = np.linspace(-2.5, 1.0, 9)
    prof = eng.profile_conditional(values)                           # (9, N) SSEs
    for k, v in enumerate(values):
        ref = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], np.full(N, np.exp(v)),
                               c["tp"])
        sse = np.sum((ref - c["obs"]) ** 2, axis=1)
        assert np.max(np.abs(prof[k] - sse)) <= 1e-3 * max(1.0, np.max(sse))
    rng = np.random.default_rng(0)
    nn_sets = c["nn"][None, :] * (1 + 0.1 * rng.standard_normal((3, c["nn"].size)))
    cond_sets = c["beta"][None, :] + 0.2 * rng.standard_normal((3, N))
    losses = eng.multistart_forward(nn_sets, cond_sets)
    for k in range(3):
        ref = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, nn_sets[k], np.exp(cond_sets[k]),
                               c["tp"])
        assert abs(losses[k] - np.sum((ref - c["obs"]) ** 2) / N) <= 1e-4 * losses[k]
    x, obj, sse = eng.fit_conditional(-4.0, 1.0, 21, 30)
    ref = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], np.exp(x), c["tp"])
    assert np.max(np.abs(sse - np.sum((ref - c["obs"]) ** 2, axis=1))) <= 1e-3 * max(1.0, np.max(sse))
    eng.close()


def test_symbolic_model_adaptive_matches_the_oracle():
    import torch  # noqa: F401
    import c_oracle as co
    from cude.engine import Engine
    N = 64
    c = make_cpep_case(N, (2, 4, 2))
    k = np.exp(np.random.default_rng(3).normal(1.0, 0.7, N))
    eng = Engine("cpep_sym", n_steps=0, n_state=2, cond_space="raw")
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params([1.78], k)
    got = eng.forward(want_traj=True)["traj"][0].T
    ref = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], (1, 0, 0), np.array([1.78]), k, c["tp"])
    err = np.max(np.abs(got - ref), axis=1)        # the production has a kink at dG = 0 on top of the forcing's
    assert _close(err), (err.max(), np.median(err))
    eng.close()


def test_failures_follow_the_reference_convention():
    """(gradients in this mode: tests/test_gpu_adaptive_grad.py)"""
    import torch  # noqa: F401
    from cude.engine import CudeError, Engine
    arch = (2, 4, 2)
    c = make_cpep_case(40, arch)
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    with pytest.raises(CudeError) as e:
        eng.adaptive_steps(0)                                         # no gradient evaluation yet
    assert e.value.status == -3
    beta = c["beta"].copy()
    beta[7] = np.nan
    eng.set_params(c["nn"], beta)
    out = eng.forward(want_sse=True)
    assert np.isinf(out["loss"]) and eng.n_failed() == 1 and not np.isfinite(out["sse"][7])
    assert np.all(np.isfinite(np.delete(out["sse"], 7)))
    with pytest.raises(CudeError):
        Engine("cpep", arch, n_steps=0, n_state=3)                    # the reference's c-peptide model has 2 states
    # tighter tolerances converge to the fine fixed-step solution
    eng.set_params(c["nn"], c["beta"])
    eng.set_tolerances(1e-12, 1e-10)
    tight = eng.forward(want_traj=True)["traj"]
    fix = Engine("cpep", arch, n_steps=960, n_state=2)
    fix.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    fix.set_params(c["nn"], c["beta"])
    assert np.max(np.abs(tight - fix.forward(want_traj=True)["traj"])) < 1e-6     # limited by the glucose kinks
    fix.close()
    eng.close()


def test_supp_adaptive_matches_the_oracle():
    import torch  # noqa: F401
    import cude_oracle as o
    from cude.engine import Engine
    s = make_supp_case(9)
    arch = s["arch"]
    eng = Engine("supp", arch, n_steps=0, lam=0.01)
    eng.set_population_supp(s["tp"], s["data"])
    eng.set_params(s["nn"], s["theta"])
    out = eng.forward(want_sse=True, want_traj=True)
    scale = o.supp_scale(s["data"])
    tot = 0.0
    for i in range(9):
        et = float(np.exp(s["theta"][i]))
        rhs = lambda t, u: [float(v) for v in o.supp_rhs(np, s["nn"], et, arch, t, [np.float64(x) for x in u])]
        sol = np.array(o.solve_adaptive(rhs, list(s["data"][:, 0, i]), list(s["tp"])))      # (T, 3)
        assert np.max(np.abs(out["traj"][:, :, i].T - sol)) <= 1e-7 * max(1.0, np.max(np.abs(sol)))
        sse_i = float(np.sum(((sol.T - s["data"][:, :, i]) / scale[:, None]) ** 2))
        assert abs(out["sse"][i] - sse_i) <= 1e-7 * max(1.0, sse_i)
        tot += sse_i
    ref_loss = tot / 9 + 0.01 * float(np.sum(s["nn"] ** 2))
    assert abs(out["loss"] - ref_loss) <= 1e-8 * ref_loss
    eng.close()


def test_product_reproduces_the_reference_stored_objectives_to_1e_minus_9():
    """The 25 + 50 objectives the reference stored for its lambda = 1 suppression run (suppression/results/
    lambda=1.0.jld2; see tests/test_known_answers.py for why they depend on stored quantities only) -- through the
    product in adaptive mode: 2e-9, i.e. the reference's own sequence of accepted steps, where the fixed-step path
    stops at the reference solver's discretisation error (1.26e-6)."""
    import torch  # noqa: F401
    from cude import api
    g1 = np.load(os.path.join(GOLD, "suppression_lambda1.npz"))
    g0 = np.load(os.path.join(GOLD, "suppression_lambda0.npz"))
    nns, stored, data, tp = g1["nn_4x3x5x1"], g1["losses"], g0["group_data"], g0["timepoints"]
    prob = api.SuppressionProblem(api.neural_network_model(5, 3, input_dims=4))
    rng = np.random.default_rng(1)
    pop = api._supp_population(prob, data, tp, 1.0, api.ADAPTIVE)
    losses = pop.engine.multistart_forward(nns, rng.uniform(-3.0, 3.0, (25, 37)))
    assert np.max(np.abs(losses - stored)) < 2e-9
    p = api.ComponentArray(theta=rng.uniform(-3.0, 3.0, 37), neural=nns[11])
    assert abs(api.suppression_loss(p, (prob, data, tp, 1.0), n_steps=api.ADAPTIVE) - stored[11]) < 2e-9
    for dkey, skey in (("validation_data", "losses_valid"), ("validation_data_nonoise", "losses_valid_nonoise")):
        vdata, vstored = g0[dkey], g1[skey]
        vpop = api._supp_population(prob, vdata, tp, 0.0, api.ADAPTIVE)
        got = vpop.engine.multistart_forward(nns, rng.uniform(-3.0, 3.0, (25, 30)))
        assert np.max(np.abs(got - vstored)) < 2e-9
    api.clear_cache()


# ------------------------------------------------------------------ the reference's figures through the product
class _GpuSubject:
    """tests/test_figure_pins._Subject with BOTH integrators taken from libcude_hip.so: `fixed` = the fixed-step
    kernels, `adaptive*` = the adaptive kernel (n_steps = 0); a vector of betas is a population of copies of the
    subject, so a 16 001-point scan is one launch."""

    def __new__(cls, base):
        import test_figure_pins as F

        class G(F._Subject):
            def _engine(self, n, n_steps):
                from cude.engine import Engine
                eng = Engine("cpep", self.arch, n_steps=n_steps, n_state=2)
                rows = [np.repeat(a, n, axis=0) for a in self.row]
                eng.set_population_cpep(self.tp, *rows)
                return eng

            def _run(self, betas, times, n_steps):
                betas = np.atleast_1d(np.asarray(betas, dtype=np.float64))
                eng = self._engine(betas.size, n_steps)
                eng.set_params(self.nn, betas)
                out = eng.simulate(np.asarray(times, dtype=np.float64))[0].T
                eng.close()
                return out

            def adaptive(self, beta, times):
                return self._run([beta], times, 0)[0]

            def adaptive_many(self, betas, times):
                return self._run(betas, times, 0)

            def fixed(self, betas, times, n_steps=240):
                return self._run(betas, times, n_steps)
        g = G.__new__(G)
        g.__dict__.update(base.__dict__)
        return g


def test_product_reproduces_the_plotted_trajectories_at_figure_resolution():
    """The c-peptide trajectories the reference plotted (model_fit_train_median.svg, model_fit_test_all.svg,
    model_fit_test_covariate_median.svg: computed by its scripts with the STORED best networks, quantised by Cairo to
    1.2e-4 nmol/L) through the product's adaptive kernel, with the whole machinery of tests/test_figure_pins.py (panel
    self-calibration, recovery of the one unknown scalar -- the beta a curve was run at -- by a 16 001-point scan,
    which here is ONE launch over a population of copies): every curve to < 4e-4 nmol/L, median <= 3e-4, where the
    fixed-step product path stops at the reference solver's own error (median 5e-3)."""
    import torch  # noqa: F401
    import test_figure_pins as F
    d = F._Data()
    errs = []
    p = d.part["train"]
    for t in F.TYPES:
        i = F._identify(d.fig[f"train_{t}_markers"], d.tp, p["C"], np.flatnonzero(p["types"] == t))
        out, _ = F._check_panel(_GpuSubject(F._Subject(d, "train", i, covariate=False)), d.fig, f"train_{t}")
        errs += [v[1] for v in out.values()]
    for i in (0, 5, 11, 17, 23, 34):
        out, _ = F._check_panel(_GpuSubject(F._Subject(d, "test", i, covariate=False)), d.fig, f"testall_{i}")
        errs += [v[1] for v in out.values()]
    p = d.part["test"]
    i = F._identify(d.fig["covariate_NGT_markers"], d.tp, p["C"], np.flatnonzero(p["types"] == "NGT"))
    out, _ = F._check_panel(_GpuSubject(F._Subject(d, "test", i, covariate=True)), d.fig, "covariate_NGT",
                            profile=(3.0, 5.0), threshold=3.8415)
    errs += [v[1] for v in out.values()]
    errs = np.array(errs)
    assert errs.size >= 25 and np.median(errs) <= 3e-4 and errs.max() < F.TOL, (np.median(errs), errs.max())


def test_product_reproduces_the_likelihood_profiles_of_all_subjects():
    """figures/revision/supplementary/likelihood_curves.svg: the reference's likelihood profile of EVERY subject
    (c-peptide/02-conditional.jl:361-423 -> src/likelihood-profiles.jl:4-17: `loss(beta, (model, timepoints, data, nn))`
    at 1000 values of beta per subject with the stored best network) through the product: cude_profile_conditional in
    adaptive mode, one launch of 1000 scan values per subject, compared with the plotted vertices (1.9e-4 quantisation)
    after the curve's one unknown -- the fitted beta_i it is centred on -- has been recovered by scans that are
    themselves populations of copies of the subject (one launch each).  At least 104 of the 116 curves with visible
    vertices must match at the figure's resolution (median deviation < 1.5e-4 = 0.8 quantisation steps) and 115 to 1e-3;
    the curves that do not are a fixed, known set of near misses (1 ... 3 steps: the ripple of the adaptive objective,
    where one rounding flips a step acceptance; profiles/r04/profiles_missed.txt)."""
    import torch  # noqa: F401
    import test_figure_pins as F
    from cude.engine import Engine
    d = F._Data()
    nn, arch, _ = d.network(False)
    stored_betas = d.g["betas_train"][int(d.g["best_model_index"]) - 1]
    good, near, bad, n_vertices = 0, 0, [], 0
    records, n_fallback = [], 0
    for part, off, n in (("train", 0, 82), ("test", 82, 35)):
        for i in range(n):
            k, y = F._profile_vertices(d.fig, off + i)
            if k.size < 3:
                continue
            sub = _GpuSubject(F._Subject(d, part, i, covariate=False))
            beta, _, res0 = F._recover_profile(sub, k, y, n_fine=4001, half_width=2e-2, edge_width=4e-2, coarse_width=0.6)
            beta_curve, fallback = beta, False
            if part == "train" and np.median(res0) >= 1.5e-4:
                # (round 3) the curve alone did not pin its centre: try the stored fitted betas of the best model too
                beta, _, _ = F._recover_profile(sub, k, y, n_fine=4001, half_width=2e-2, edge_width=4e-2,
                                                coarse_width=0.6, stored=stored_betas)
                fallback = abs(beta - beta_curve) > 1e-9      # a stored beta won over the curve-derived centre
            # the reference's own call sequence for this subject: likelihood_profile(beta_i, ...; steps = 1000)
            eng = Engine("cpep", arch, n_steps=0, n_state=2)
            eng.set_population_cpep(d.tp, *(a for a in sub.row))
            eng.set_params(nn, [beta])
            sse_min = eng.forward(want_sse=True)["sse"][0]
            prof = eng.profile_conditional(np.linspace(beta - 10.0, beta + 10.0, 1000))[:, 0]
            eng.close()
            dd = prof[k] - sse_min
            scale = (dd @ y) / (dd @ dd)
            res = np.abs(y - scale * dd)
            ok = np.median(res) < 1.5e-4 and np.quantile(res, 0.9) < F.TOLP and abs(scale * 2 * sse_min / 5 - 1) < 0.05
            good += ok
            n_fallback += bool(ok and fallback)
            near += bool(np.median(res) < 1e-3 and abs(scale * 2 * sse_min / 5 - 1) < 0.05)
            n_vertices += k.size if ok else 0
            b_star, _ = sub.argmin_sse()
            records.append((off + i, part, int(d.fig["profiles_class"][off + i]), bool(ok), bool(fallback), float(beta),
                            float(b_star), float(np.median(res)), float(np.quantile(res, 0.9)), float(res.max()),
                            float(scale * 2 * sse_min / 5), int(k.size), int(k.min()), int(k.max()),
                            int(np.max(np.diff(np.sort(k)))) if k.size > 1 else 0, float(np.finfo(float).eps + sse_min)))
            if not ok:
                bad.append((off + i, int(d.fig["profiles_class"][off + i]), round(float(beta), 5),
                            float(np.median(res)), float(np.quantile(res, 0.9)), float(scale * 2 * sse_min / 5)))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):           # the per-curve record behind profiles/r04/profiles_missed.txt
        with open(os.path.join(out_dir, "profiles_r04.txt"), "w") as fh:
            fh.write("# curve part class ok stored_beta_used beta_centre beta_argmin median|res| q90|res| max|res| "
                     "scale*2sse/5 n_vertices k_min k_max largest_gap_in_k sse_min   (res in plotted units: 1 px = 1.9e-4 x ...)\n")
            for r in records:
                fh.write(" ".join(str(v) for v in r) + "\n")
    print("profiles reproduced:", good, "of them through a stored beta:", n_fallback, "vertices:", n_vertices,
          "not reproduced:", bad)
    assert good >= 104 and near >= 115 and n_vertices > 5000, (good, near, bad)
    # (advisor, round 3) the stored fitted betas are a help for the curves whose shape does not identify their centre --
    # they must stay the exception (measured: ONE of the 108 reproduced curves owes its centre to a stored beta)
    assert n_fallback <= 3 and good - n_fallback >= 104, (good, n_fallback)
    # the curves that miss the figure's resolution are known, and all of them miss it narrowly: median deviation 0.8 ... 3.3
    # quantisation steps of the plot where 0.8 is the bar (profiles/r04/profiles_missed.txt has them one by one)
    KNOWN_NEAR_MISSES = {14, 27, 53, 61, 72, 90, 111, 116}
    assert {b[0] for b in bad} <= KNOWN_NEAR_MISSES, bad
    assert all(b[3] < 1e-3 for b in bad), bad
