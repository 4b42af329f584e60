"""Symbolic (analytic-production) c-peptide model on the GPU: CUDE_MODEL_CPEP_SYM through the C ABI against the
oracle's restatement of c-peptide/03-symreg.jl:37-40 + src/c-peptide-models.jl:68-75,118-142 and
src/saem-symreg.jl.  Tolerances: loss / SSE / trajectories rtol 1e-10, gradients 1e-9 of the max-norm (fp64,
different summation order and a reciprocal in the device gradient)."""
import math

import numpy as np
import pytest
import torch  # noqa: F401  (one HIP runtime for the process)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fixed_step_default")]


def _case(N, seed=11, n_steps=30, noise=0.05):
    import cude_oracle as o
    tp, G, cp, age, t2, _, rng = o.synthetic_cpep_population(N, seed)
    pop0 = o.CPepPopulation(tp, G, cp, age, t2)
    k_true = np.exp(rng.normal(math.log(40.0), 0.6, N))
    traj = o.cpep_forward(np, np.array([1.78]), k_true, pop0, o.SYMBOLIC, n_steps, cond_space="raw")
    obs = np.stack([traj[t][0] for t in range(len(tp))], axis=1) * (1.0 + noise * rng.standard_normal((N, len(tp))))
    obs[:, 0] = cp[:, 0]
    k = k_true * np.exp(0.3 * rng.standard_normal(N))
    return dict(tp=np.asarray(tp), G=G, obs=obs, age=age, t2dm=t2, k=k, k_true=k_true,
                pop=o.CPepPopulation(tp, G, obs, age, t2), n_steps=n_steps)


def _engine(c, cond_space, n_state=2):
    from cude.engine import Engine
    eng = Engine("cpep_sym", n_steps=c["n_steps"], n_state=n_state, cond_space=cond_space)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    return eng


@pytest.mark.parametrize("cond_space,n_state,N", [("raw", 2, 130), ("log", 2, 64), ("log", 3, 257), ("raw", 3, 1)])
def test_symbolic_loss_and_gradient(cond_space, n_state, N):
    import cude_oracle as o
    c = _case(N)
    p0 = np.array([1.6])
    cond = c["k"] if cond_space == "raw" else np.log(c["k"])
    ref_loss, ref_gp, ref_gc, ref_sse = o.cpep_loss_grad_torch(p0, cond, c["pop"], o.SYMBOLIC, c["n_steps"], n_state,
                                                              cond_space)
    ref_traj = o.cpep_forward(np, p0, cond, c["pop"], o.SYMBOLIC, c["n_steps"], n_state, cond_space)
    eng = _engine(c, cond_space, n_state)
    assert eng.P == 1
    eng.set_params(p0, cond)
    out = eng.forward(want_sse=True, want_traj=True)
    loss, g_p, g_c = eng.loss_grad()
    eng.close()
    assert abs(out["loss"] - ref_loss) <= 1e-10 * abs(ref_loss)
    assert abs(loss - ref_loss) <= 1e-10 * abs(ref_loss)
    assert np.allclose(out["sse"], ref_sse, rtol=1e-10, atol=1e-14)
    for t in range(len(c["tp"])):
        for s in range(n_state):
            assert np.allclose(out["traj"][s, t], ref_traj[t][s], rtol=1e-10, atol=1e-13)
    assert np.max(np.abs(g_p - ref_gp)) <= 1e-9 * np.max(np.abs(ref_gp))
    assert np.max(np.abs(g_c - ref_gc)) <= 1e-9 * np.max(np.abs(ref_gc))


def test_log_and_raw_parameterisations_agree():
    c = _case(200)
    raw, log = _engine(c, "raw"), _engine(c, "log")
    raw.set_params([1.78], c["k"])
    log.set_params([1.78], np.log(c["k"]))
    l1, gp1, gk = raw.loss_grad()
    l2, gp2, gl = log.loss_grad()
    raw.close()
    log.close()
    assert abs(l1 - l2) <= 1e-13 * abs(l1)
    assert np.allclose(gp1, gp2, rtol=1e-12)
    assert np.allclose(gk * c["k"], gl, rtol=1e-12, atol=1e-18)       # d/dlog k = k d/dk


def test_symbolic_failure_convention():
    """k = 0: production(0, 0) = 0/0 at t0 (03-symreg.jl:38) -> the reference's solve fails -> loss = Inf."""
    c = _case(70)
    eng = _engine(c, "raw")
    k = c["k"].copy()
    k[5] = 0.0
    k[40] = np.nan
    eng.set_params([1.78], k)
    out = eng.forward(want_sse=True)
    assert out["loss"] == np.inf and eng.n_failed() == 2
    good = np.ones(70, bool)
    good[[5, 40]] = False
    assert np.all(np.isfinite(out["sse"][good])) and not np.any(np.isfinite(out["sse"][~good]))
    eng.close()


def test_symbolic_multistart_and_estep_match_oracle():
    import cude_oracle as o
    c = _case(90)
    rng = np.random.default_rng(5)
    eng = _engine(c, "log")
    K = 7
    p_sets = 1.78 * np.exp(0.1 * rng.standard_normal((K, 1)))
    cond_sets = np.log(c["k"])[None, :] + 0.2 * rng.standard_normal((K, 90))
    got = eng.multistart_forward(p_sets, cond_sets)
    want = [o.cpep_loss(np, p_sets[j], cond_sets[j], c["pop"], o.SYMBOLIC, c["n_steps"])[0] for j in range(K)]
    assert np.allclose(got, want, rtol=1e-10)
    # Metropolis E-step of saem-symreg.jl:86-108 on log k = log km + eta
    km, omega, sigma, steps = 35.0, 0.7, 0.3, 6
    z, u = rng.standard_normal((steps, 90)), rng.random((steps, 90))
    start = math.log(km) + 0.1 * rng.standard_normal(90)
    want_state, want_acc = o.mh_chain(np.array([1.78]), start, c["pop"], o.SYMBOLIC, c["n_steps"], sigma,
                                      math.log(km), omega, 0.25, 2.0, 0.8, z, u)
    eng.set_params([1.78], start)
    acc = eng.mh_estep(z, u, sigma, math.log(km), omega, 0.25, 2.0, 0.8)
    state = eng.get_params()[1]
    eng.close()
    assert np.array_equal(acc, want_acc)
    assert np.allclose(state, want_state, rtol=1e-12, atol=1e-13)


def test_real_data_symbolic_fit_matches_the_network_model():
    """Soft pin on the 117 complete Ohashi subjects: the symbolic production was regressed from the trained
    network (03-symreg.jl), so (a) its per-subject fit must explain the data as well as the reference's stored
    network does and (b) the fitted k must be a monotone function of that network's conditional parameter
    (orientation arbitrary per training run).  Measured: mean SSE 0.320 vs 0.311, Spearman -0.977."""
    import os
    from scipy.stats import spearmanr
    from cude import api
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ohashi_cude.npz")))
    tp, N = g["timepoints"], len(g["ages"])
    ode = [api.CPeptideODEModel(g["glucose"][i], tp, g["ages"][i], api.production, g["cpeptide"][i], g["t2dm"][i])
           for i in range(N)]
    sols = api.train_symbolic(ode, tp, g["cpeptide"])
    k = np.array([s.u.ode[0] for s in sols])
    sse = np.array([s.u.sigma for s in sols]) ** 2 * len(tp)
    net = api.chain(4, 2, "tanh")
    nnm = [api.CPeptideConditionalUDEModel(g["glucose"][i], tp, g["ages"][i], net, g["cpeptide"][i], g["t2dm"][i])
           for i in range(N)]
    beta, sse_nn = api.estimate_conditional(nnm, tp, g["cpeptide"], g["nn_2x4x4x1"][0], lower=-4.0, upper=3.0)
    api.clear_cache()
    assert np.all(np.isfinite(k)) and np.all(k > 0) and np.all(k <= 1000.0)
    assert sse.mean() < 1.1 * sse_nn.mean() and np.median(sse) < 1.1 * np.median(sse_nn)
    assert abs(spearmanr(k, beta)[0]) > 0.95


def test_external_data_set_of_the_reference():
    """c-peptide/04-symreg-external.jl: the symbolic model on the 20 subjects of the Fujita data (14 irregularly
    spaced observations from -10 to 240 min, glucose falling below its basal value at the end -> the dG < 0
    branch).  Parity of loss / gradient / trajectory with the oracle on that grid, and the per-subject fit."""
    import os
    import cude_oracle as o
    from cude import api
    from cude.engine import Engine
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fujita.npz")))
    tp, N = g["timepoints"], g["glucose"].shape[0]
    assert tp[0] == -10.0 and tp.size == 14 and np.any(g["glucose"] < g["glucose"][:, :1])
    t2 = np.zeros(N, bool)
    pop = o.CPepPopulation(tp, g["glucose"], g["cpeptide"], g["ages"], t2)
    k = np.linspace(15.0, 120.0, N)
    n_steps = 100                                  # h = 2.5 min: every observation time is a step boundary
    ref_loss, ref_gp, ref_gk, ref_sse = o.cpep_loss_grad_torch(np.array([1.78]), k, pop, o.SYMBOLIC, n_steps, 2, "raw")
    ref_traj = o.cpep_forward(np, np.array([1.78]), k, pop, o.SYMBOLIC, n_steps, 2, "raw")
    eng = Engine("cpep_sym", n_steps=n_steps, n_state=2, cond_space="raw")
    eng.set_population_cpep(tp, g["glucose"], g["cpeptide"], g["ages"], t2)
    eng.set_params([1.78], k)
    out = eng.forward(want_sse=True, want_traj=True)
    loss, g_p, g_k = eng.loss_grad()
    eng.close()
    assert abs(loss - ref_loss) <= 1e-10 * ref_loss and np.allclose(out["sse"], ref_sse, rtol=1e-10)
    for t in range(tp.size):
        assert np.allclose(out["traj"][0, t], ref_traj[t][0], rtol=1e-10, atol=1e-13)
    assert np.max(np.abs(g_p - ref_gp)) <= 1e-9 * np.max(np.abs(ref_gp))
    assert np.max(np.abs(g_k - ref_gk)) <= 1e-9 * np.max(np.abs(ref_gk))
    # the script's fit: CPeptideODEModel(glucose, timepoints, 29.0, production, cpeptide, false), k in [0, 1000]
    models = [api.CPeptideODEModel(g["glucose"][i], tp, 29.0, api.production, g["cpeptide"][i], False) for i in range(N)]
    sols = api.train_symbolic(models, tp, g["cpeptide"], n_steps=n_steps)
    k_fit = np.array([s.u.ode[0] for s in sols])
    sse_fit = np.array([s.u.sigma for s in sols]) ** 2 * tp.size
    assert np.all((k_fit > 0) & (k_fit <= 1000.0)) and np.all(np.isfinite(sse_fit))
    # each fitted k is a (local) minimiser of that subject's SSE: the oracle's SSE is not lower 2 % to either side
    for sign in (0.98, 1.02):
        nearby = o.cpep_loss(np, np.array([1.78]), k_fit * sign, pop, o.SYMBOLIC, n_steps, 2, "raw")[1]
        inside = (k_fit * sign < 1000.0) & (k_fit < 999.0)
        assert np.all(nearby[inside] >= sse_fit[inside] * (1 - 1e-9))
    api.clear_cache()


def test_api_mirror_fit_recovers_k_and_saem_runs():
    import cude_oracle as o
    from cude import api
    c = _case(48, noise=0.0)
    models = [api.CPeptideODEModel(c["G"][i], c["tp"], c["age"][i], api.production, c["obs"][i], bool(c["t2dm"][i]))
              for i in range(48)]
    # loss(theta, (model, timepoints, data)) with theta = ComponentArray(ode=[k], sigma) reads theta[1]
    th = api.ComponentArray(ode=np.array([c["k"][3]]), sigma=1.0)
    got = api.loss(th, (models[3], c["tp"], c["obs"][3]), n_steps=30)
    one = o.CPepPopulation(c["tp"], c["G"][3:4], c["obs"][3:4], c["age"][3:4], c["t2dm"][3:4])
    want = o.cpep_loss(np, np.array([1.78]), c["k"][3:4], one, o.SYMBOLIC, 30, cond_space="raw")[0]
    assert abs(got - want) <= 1e-10 * want
    assert abs(api.loss_sigma(th, (models[3], c["tp"], c["obs"][3]), n_steps=30)
               - (len(c["tp"]) / 2 * math.log(1.0) + want / 2)) <= 1e-10 * want
    # noise-free data generated at k_true: the per-subject fit must find it
    sols = api.train_symbolic(models, c["tp"], c["obs"], n_steps=30)
    k_fit = np.array([s.u.ode[0] for s in sols])
    ident = c["k_true"] < 900                      # the box is [0, 1000]
    assert np.max(np.abs(np.log(k_fit[ident] / c["k_true"][ident]))) < 1e-3
    # SAEM on the symbolic model: a short run must lower the total negative log-likelihood
    c2 = _case(64, seed=3, noise=0.05)
    models2 = [api.CPeptideODEModel(c2["G"][i], c2["tp"], c2["age"][i], api.production, c2["obs"][i],
                                    bool(c2["t2dm"][i])) for i in range(64)]
    res = api.SAEM_symbolic(models2, c2["tp"], c2["obs"], 20.0, iterations=30, n_burnin_iterations=10, n_mcmc_steps=2,
                            rng=np.random.default_rng(0), n_steps=30)
    assert np.all(np.isfinite(res.total_nll_values)) and res.total_nll_values[-1] < res.total_nll_values[0]
    assert 10.0 < res.km_pop < 120.0 and res.sigma > 0
    api.clear_cache()
