"""GPU tests of the reference-API mirror (cude.api) and of the sharded training step, all through the C ABI.
They read like the reference's call sites (c-peptide/02-conditional.jl, suppression/suppression.jl, src/saem.jl)."""
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first: shared HIP runtime)

from conftest import make_cpep_case, make_supp_case, free_port

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fixed_step_default")]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ohashi_models(api, net, n=None):
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    n = g["glucose"].shape[0] if n is None else n
    models = [api.CPeptideConditionalUDEModel(g["glucose"][i], g["timepoints"], g["ages"][i], net, g["cpeptide"][i],
                                              g["t2dm"][i]) for i in range(n)]
    return g, models


def test_population_loss_and_gradient_on_ohashi_data():
    """loss(theta, (models, timepoints, cpeptide)) and its gradient on the real 117-subject data with the
    reference's stored trained weights, vs the oracle."""
    import c_oracle as co
    from cude import api
    net = api.chain(4, 2, "tanh")
    g, models = _ohashi_models(api, net)
    N = len(models)
    rng = np.random.default_rng(0)
    theta = api.ComponentArray(neural=g["nn_2x4x4x1"][0], conditional=rng.uniform(-2, 0, (N, 1)))
    args = (models, g["timepoints"], g["cpeptide"])
    val = api.loss(theta, args)
    ref = co.cpep(g["timepoints"], g["glucose"], g["cpeptide"], g["ages"], g["t2dm"], (2, 4, 2), theta.neural,
                  theta.conditional[:, 0], api.default_steps(g["timepoints"]), 2)
    assert api.default_steps(g["timepoints"]) == api.fixed_steps(g["timepoints"]) == 32   # 8 steps per 30-min interval
    assert abs(val - ref["loss"]) < 1e-10 * ref["loss"]
    val2, grad = api.loss_and_gradient(theta, args)
    assert abs(val2 - ref["loss"]) < 1e-10 * ref["loss"]
    assert np.max(np.abs(grad.neural - ref["g_nn"])) < 1e-9 * np.max(np.abs(ref["g_nn"]))
    assert np.max(np.abs(grad.conditional[:, 0] - ref["g_beta"])) < 1e-9 * np.max(np.abs(ref["g_beta"]))
    # single-subject methods
    single = api.loss(api.ComponentArray(neural=theta.neural, conditional=theta.conditional[3]),
                      (models[3], g["timepoints"], g["cpeptide"][3]))
    fixed = api.loss(theta.conditional[3], (models[3], g["timepoints"], g["cpeptide"][3], theta.neural))
    assert abs(single - ref["sse"][3]) < 1e-10 and abs(fixed - ref["sse"][3]) < 1e-10
    th = api.ComponentArray(ode=theta.conditional[3], sigma=0.7)
    nll = api.loss_sigma(th, (models[3], g["timepoints"], g["cpeptide"][3], theta.neural))
    assert abs(nll - (2.5 * np.log(0.49) + ref["sse"][3] / 0.98)) < 1e-9
    api.clear_cache()


def test_fixed_network_beta_estimation_recovers_stored_betas():
    """train(models, timepoints, data, nn) -- the per-subject conditional-parameter fit with frozen network
    (parameter-estimation.jl:272-288) on the GPU reproduces the reference's stored training betas."""
    from scipy.optimize import linear_sum_assignment
    from cude import api
    net = api.chain(4, 2, "tanh")
    g, models = _ohashi_models(api, net)
    sols = api.train(models, g["timepoints"], g["cpeptide"], g["nn_2x4x4x1"][0], lbfgs_lower_bound=-4.0,
                     lbfgs_upper_bound=3.0)
    beta_hat = np.array([s.u[0] for s in sols])
    cost = np.abs(g["betas_train"][0][:, None] - beta_hat[None, :])
    r, c = linear_sum_assignment(cost)
    assert np.median(cost[r, c]) < 5e-3 and np.quantile(cost[r, c], 0.9) < 3e-2
    ws = api.train_with_sigma(models[:10], g["timepoints"], g["cpeptide"][:10], g["nn_2x4x4x1"][0],
                              lbfgs_upper_bound=3.0)
    assert abs(ws[0].u.sigma - np.sqrt(sols[0].objective / 5)) < 1e-6
    api.clear_cache()


def test_covariate_model_beta_estimation_recovers_stored_betas():
    """CPeptideConditionalCovariateUDEModel (src/c-peptide-models.jl:196-220, network input [dG, exp(beta), age])
    through the GPU path: with a stored covariate network of c-peptide/07-covariate-inclusion.jl the per-subject
    refit reproduces that run's stored training betas (subjects of the reference's prepared train set)."""
    from scipy.optimize import linear_sum_assignment
    from cude import api
    g = dict(np.load(os.path.join(GOLD, "ohashi_cude.npz")))
    net = api.chain(4, 2, "tanh", input_dims=3)
    models = [api.CPeptideConditionalCovariateUDEModel(g["glucose"][i], g["timepoints"], g["ages"][i], net,
                                                       g["cpeptide"][i], g["t2dm"][i]) for i in range(len(g["ages"]))]
    k = int(g["best_model_index_cov"]) - 1
    beta_hat, sse = api.estimate_conditional(models, g["timepoints"], g["cpeptide"], g["nn_3x4x4x1_cov"][k],
                                             lower=-4.0, upper=3.0, n_steps=30)
    idx = np.flatnonzero(np.isin(g["subject_no"], g["train_subject_numbers"]))
    cost = np.abs(g["betas_train_cov"][k][:, None] - beta_hat[None, idx])
    r, c = linear_sum_assignment(cost)
    assert np.median(cost[r, c]) < 5e-3 and np.quantile(cost[r, c], 0.9) < 0.1
    assert np.mean(sse[idx[c]]) < 0.5
    api.clear_cache()


def test_population_training_decreases_loss():
    """train(models, timepoints, data, rng): screening -> Adam -> L-BFGS (parameter-estimation.jl:340-386)."""
    from cude import api
    net = api.chain(4, 2, "tanh")
    g, models = _ohashi_models(api, net, n=40)
    rng = np.random.default_rng(5)
    sols = api.train(models, g["timepoints"], g["cpeptide"][:40], rng, initial_guesses=200, selected_initials=2,
                     number_of_iterations_adam=150, number_of_iterations_lbfgs=40)
    assert len(sols) == 2
    for s in sols:
        assert np.isfinite(s.objective) and s.objective < 1.5          # untrained screening losses are ~5-50
        assert s.u.neural.shape == (37,) and s.u.conditional.shape == (40, 1)
        assert abs(api.loss(s.u, (models, g["timepoints"], g["cpeptide"][:40])) - s.objective) < 1e-9
    api.clear_cache()


def test_suppression_api_matches_oracle_and_fits():
    import c_oracle as co
    from cude import api
    g = dict(np.load(os.path.join(GOLD, "suppression_lambda0.npz")))
    net = api.neural_network_model(5, 3, input_dims=4)
    prob = api.SuppressionProblem(net)
    data, tp = g["group_data"], g["timepoints"]
    rng = np.random.default_rng(1)
    p = api.ComponentArray(theta=rng.standard_normal(37), neural=g["nn_4x3x5x1"][0])
    for lam in (0.0, 0.01):
        ref = co.supp(tp, data, (4, 3, 5), p.neural, p.theta, lam, 30, want_traj=True)
        assert abs(api.suppression_loss(p, (prob, data, tp, lam)) - ref["loss"]) < 1e-10 * ref["loss"]
        val, grad = api.suppression_loss_and_gradient(p, (prob, data, tp, lam))
        assert np.max(np.abs(grad.neural - ref["g_nn"])) < 1e-9 * np.max(np.abs(ref["g_nn"]))
        assert np.max(np.abs(grad.theta - ref["g_theta"])) < 1e-9 * np.max(np.abs(ref["g_theta"]))
    sims = api.simul(p, prob, data, tp)
    assert sims.shape == (3, 8, 37) and np.max(np.abs(sims - ref["traj"])) < 1e-10
    # short fit from the stored network: theta only needs to move; loss must approach the stored final loss
    inits = [api.ComponentArray(theta=rng.standard_normal(37), neural=g["nn_4x3x5x1"][0]) for _ in range(4)]
    sols, _ = api.fit_suppression_model(inits, prob, data, tp, 0.0, select_best_n=1, adam_iters=300, lbfgs_iters=60)
    assert sols and sols[0].objective < 0.75 * min(api.suppression_loss(q, (prob, data, tp, 0.0)) for q in inits)
    api.clear_cache()


@pytest.mark.parametrize("model,noise", [(0, ""), (7, ""), (0, "_nonoise"), (7, "_nonoise")])
def test_validate_suppression_model_against_stored_results(model, noise):
    """validate_suppression_model on the reference's stored validation sets with its stored networks: the stored
    `losses_valid*` / `correlations_valid*` depend on stored quantities only (the network is frozen)."""
    from scipy.stats import spearmanr
    from cude import api
    g = dict(np.load(os.path.join(GOLD, "suppression_lambda0.npz")))
    prob = api.SuppressionProblem(api.neural_network_model(5, 3, input_dims=4))
    theta, obj = api.validate_suppression_model(None, prob, g["validation_data" + noise], g["timepoints"],
                                                g["nn_4x3x5x1"][model], n_steps=60)
    stored = g["losses_valid" + noise][model]
    assert 0.93 * stored <= obj <= 1.005 * stored
    rho = spearmanr(theta, g["gt_validation_param" + noise])[0]
    assert abs(rho - g["correlations_valid" + noise][model]) < 0.01
    api.clear_cache()


def test_saem_runs_and_improves_likelihood():
    from cude import api
    net = api.chain(4, 2, "tanh")
    g, models = _ohashi_models(api, net, n=30)
    res = api.SAEM(models, g["timepoints"], g["cpeptide"][:30], g["nn_2x4x4x1"][0], sigma=0.5, prior_eta=-0.6,
                   prior_omega=1.0, iterations=6, n_burnin_iterations=3, proposal_std=0.5, n_mcmc_steps=5,
                   rng=np.random.default_rng(2))
    assert len(res.total_nll_values) == 6 and np.all(np.isfinite(res.total_nll_values))
    assert res.total_nll_values[-1] < res.total_nll_values[0]
    assert 0.0 < res.acceptance_rates[-1] <= 1.0 and res.p_individuals.shape == (30,)
    api.clear_cache()


def test_likelihood_profile_shape_and_minimum():
    from cude import api
    net = api.chain(4, 2, "tanh")
    g, models = _ohashi_models(api, net, n=1)
    nn = g["nn_2x4x4x1"][0]
    sol = api.train(models, g["timepoints"], g["cpeptide"][:1], nn, lbfgs_upper_bound=3.0)[0]
    nll, nll_min, values = api.likelihood_profile(sol.u[0], nn, models[0], g["timepoints"], g["cpeptide"][0], -4.0, 3.0,
                                                  0.3, steps=500)
    assert nll.shape == (500,) and nll.min() >= nll_min - 1e-9 and abs(values[np.argmin(nll)] - sol.u[0]) < 0.02
    api.clear_cache()


def test_likelihood_profiles_of_all_subjects_in_one_launch():
    """cude_profile_conditional: the scan value is the grid's second dimension.  Every row must equal the
    single-subject likelihood_profile (same kernel, same lanes: bit-identical SSEs), for the network model and for the
    suppression model (profile_conditional through the engine)."""
    from cude import api
    net = api.chain(4, 2, "tanh")
    g, models = _ohashi_models(api, net, n=9)
    nn = g["nn_2x4x4x1"][0]
    betas = np.linspace(-1.5, 0.5, 9)
    nll, nll_min, values = api.likelihood_profiles(betas, nn, models, g["timepoints"], g["cpeptide"][:9], -4.0, 3.0,
                                                   0.3, steps=200)
    assert nll.shape == (9, 200) and values.shape == (200,)
    for i in (0, 4, 8):
        one, one_min, _ = api.likelihood_profile(betas[i], nn, models[i], g["timepoints"], g["cpeptide"][i], -4.0, 3.0,
                                                 0.3, steps=200)
        assert np.allclose(nll[i], one, rtol=1e-12, atol=0) and abs(nll_min[i] - one_min) <= 1e-12 * one_min
    api.clear_cache()
    from cude.engine import Engine
    c = make_supp_case(40)
    eng = Engine("supp", c["arch"], n_steps=30)
    eng.set_population_supp(c["tp"], c["data"])
    eng.set_params(c["nn"], c["theta"])
    scan = eng.profile_conditional([-1.0, 0.0, 0.7])
    for k, v in enumerate([-1.0, 0.0, 0.7]):
        eng.set_params(None, np.full(40, v))
        assert np.array_equal(scan[k], eng.forward(want_sse=True)["sse"])
    eng.close()


# ------------------------------------------------------------------ two ranks sharing the one GPU of the test box
def _gpu_rank(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "conditional-ude_amd"), os.path.join(here, "..", "oracle"), here):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from cude.engine import Engine
    from cude.parallel import ShardedTrainer, TorchCollective, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = make_cpep_case(n_total, (2, 6, 2))
    lo, hi = shard_bounds(n_total, world, rank)
    eng = Engine("cpep", (2, 6, 2), n_steps=30, n_state=3, device=0)
    eng.set_population_cpep(c["tp"], c["G"][lo:hi], c["obs"][lo:hi], c["age"][lo:hi], c["t2dm"][lo:hi])
    eng.set_params(c["nn"], c["beta"][lo:hi])
    tr = ShardedTrainer(eng, TorchCollective(dist), transport="host")
    tr.sync_population_statistics()
    tr.adam_init(1e-2)
    losses = [tr.adam_step() for _ in range(5)]
    cond = tr.gather_conditional(n_total)
    nn, _ = eng.get_params()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=losses, cond=cond, nn=nn)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_two_rank_sharded_training_matches_single_engine(tmp_path):
    import torch.multiprocessing as mp
    from cude.engine import Engine
    n_total, world = 333, 2
    port = free_port()
    mp.spawn(_gpu_rank, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["losses"], r1["losses"]) and np.array_equal(r0["nn"], r1["nn"])
    c = make_cpep_case(n_total, (2, 6, 2))
    eng = Engine("cpep", (2, 6, 2), n_steps=30, n_state=3)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    eng.adam_init(1e-2)
    ref = [eng.adam_step() for _ in range(5)]
    nn, cond = eng.get_params()
    eng.close()
    assert np.allclose(r0["losses"], ref, rtol=1e-10)          # loss trajectory over 5 Adam steps
    assert np.allclose(r0["nn"], nn, rtol=0, atol=1e-10) and np.allclose(r0["cond"], cond, rtol=0, atol=1e-10)


def test_builtin_rccl_path_single_rank():
    """The built-in communicator (dlopen'ed RCCL: unique id, ncclCommInitRank by-value id, in-place all-reduce on
    the context's stream) exercised with a 1-rank communicator: results must equal the no-communicator path."""
    from cude.engine import Engine
    c = make_cpep_case(300, (2, 6, 2))

    def run(with_comm):
        eng = Engine("cpep", (2, 6, 2), n_steps=30, n_state=3)
        if with_comm:
            uid = Engine.comm_unique_id()
            assert len(uid) == 128 and any(uid)
            eng.comm_init(1, 0, uid)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        fwd = eng.forward()["loss"]
        eng.adam_init(1e-2)
        losses = [eng.adam_step() for _ in range(3)]
        nn, cond = eng.get_params()
        red = eng.allreduce_host([1.5, 2.5])
        eng.close()
        return fwd, losses, nn, cond, red
    a, b = run(False), run(True)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert np.array_equal(b[4], [1.5, 2.5])


def test_builtin_rccl_path_reports_itself_and_runs_the_lbfgs_stage():
    """cude_comm_info returns what RCCL itself says about the communicator; cude_train_restarts with an L-BFGS stage
    on a context WITH a communicator (every inner product's conditional part goes through ncclAllReduce, sum and
    max) must reproduce the no-communicator run when there is one rank -- up to the summation order of the inner
    products, whose network and conditional parts are added separately on the sharded path."""
    from cude.engine import Engine
    c = make_cpep_case(150, (2, 4, 2))

    def run(with_comm):
        eng = Engine("cpep", (2, 4, 2), n_steps=30, n_state=2)
        info = eng.comm_info()
        assert info == (1, 0, 0)
        if with_comm:
            eng.comm_init(1, 0, Engine.comm_unique_id())
            n, r, v = eng.comm_info()
            assert (n, r) == (1, 0) and v > 20000          # RCCL tracks NCCL's version numbering (2.x.y -> 2xxyy)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        out = eng.train_restarts(c["nn"][None, :], c["beta"][None, :], 3, 1e-2, 8, want_trace=True)
        eng.close()
        return out
    a, b = run(False), run(True)
    assert np.array_equal(a[3][:, :3], b[3][:, :3])          # the Adam stage is bit-identical
    assert np.array_equal(np.isfinite(a[3]), np.isfinite(b[3]))    # same number of L-BFGS iterations
    for x, y in zip(a, b):
        assert np.allclose(x, y, rtol=1e-9, atol=1e-9, equal_nan=True)
    assert np.isfinite(a[2][0]) and a[2][0] < a[3][0, 0]      # the objective went down


def test_queued_adam_run_with_a_communicator_equals_the_plain_run():
    """bench.py times K optimiser steps queued by cude_adam_run; with more than one rank every step carries the RCCL
    all-reduce on the context's stream (no hipGraph then).  With a one-rank communicator that path must reproduce the
    plain (graph-replayed) run bit for bit: same K losses, same parameters, with and without kernel timing."""
    from cude.engine import Engine
    c = make_cpep_case(300, (2, 6, 2))

    def run(with_comm, timing):
        eng = Engine("cpep", (2, 6, 2), n_steps=30, n_state=3)
        if with_comm:
            eng.comm_init(1, 0, Engine.comm_unique_id())
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        eng.adam_init(1e-2)
        eng.set_kernel_timing(timing)
        losses = np.concatenate([eng.adam_run(7), eng.adam_run(5)])
        ms, n = eng.kernel_time_ms()
        assert n == (12 if timing else 0)
        nn, cond = eng.get_params()
        eng.close()
        return losses, nn, cond
    ref = run(False, False)
    for with_comm, timing in ((True, False), (True, True), (False, True)):
        got = run(with_comm, timing)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), (with_comm, timing)
    assert ref[0][-1] < ref[0][0]


def test_unequal_width_network_through_zero_padding():
    """chain([6, 3], tanh) on the 2-6-6-1 kernels (api.pad_network): the gradient entries of the padded units vanish
    (exactly for everything flowing INTO a padded unit; at the rounding of the device's tanh(0) ~ 1e-16 for the output
    weights LEAVING one), the others equal central differences of the loss in the UNPADDED parameters; optimising in the
    unpadded parameters from the host keeps the padding exact, the library's own Adam does not (documented)."""
    from cude import api
    from cude.engine import Engine
    widths, N = [6, 3], 80
    c = make_cpep_case(N, (2, 6, 2))
    rng = np.random.default_rng(9)
    n = sum(w * f + w for w, f in zip(widths + [1], [2] + widths))
    p = 0.7 * rng.standard_normal(n)
    net, P = api.pad_network(widths, p)
    pad = P == 0.0
    eng = Engine("cpep", net.arch, n_steps=30, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(P, c["beta"])
    loss, g_nn, _ = eng.loss_grad()
    assert np.max(np.abs(g_nn[pad])) <= 1e-14 * np.max(np.abs(g_nn)) and np.count_nonzero(g_nn[~pad]) == n
    g_small = api.unpad_network(widths, g_nn)
    for k in rng.choice(n, 6, replace=False):
        e = np.zeros(n)
        e[k] = 1e-5
        vals = []
        for sgn in (1.0, -1.0):
            eng.set_params(api.pad_network(widths, p + sgn * e)[1], None)
            vals.append(eng.forward()["loss"])
        assert abs((vals[0] - vals[1]) / 2e-5 - g_small[k]) <= 1e-6 * max(1.0, abs(g_small[k]))
    # optimising in the UNPADDED parameters (host-side steps on unpad(gradient)) keeps the padding exact by construction
    q, cond = p.copy(), c["beta"].copy()
    for _ in range(25):
        eng.set_params(api.pad_network(widths, q)[1], cond)
        l, g, gc = eng.loss_grad()
        q -= 2e-3 * api.unpad_network(widths, g)
        cond -= 2e-3 * N * gc
    eng.set_params(api.pad_network(widths, q)[1], cond)
    assert eng.forward()["loss"] < 0.9 * loss
    # the library's own optimisers work on the padded vector: without a mask Adam's scale invariance amplifies the
    # ~1e-17 gradients of the padding until it comes alive; with the mask (cude_set_param_mask) it stays exactly zero
    # (round 3: the table tanh returns tanh(0) = 0 exactly, so in a hidden layer >= 2 -- this case -- the padding now
    # survives even without the mask; a padded FIRST-layer unit on the exponent-table path still sees ~1e-17, so the
    # mask remains what guarantees it)
    nn_t, _, obj, _ = eng.train_restarts(P[None, :], c["beta"][None, :], 30, 1e-2, 0, want_trace=True)
    assert obj[0] < loss
    eng.set_param_mask((~pad).astype(float))
    eng.set_params(P, c["beta"])
    _, g_m, _ = eng.loss_grad()
    assert np.all(g_m[pad] == 0.0) and np.array_equal(g_m[~pad], g_nn[~pad])
    nn_m, cond_m, obj_m, _ = eng.train_restarts(P[None, :], c["beta"][None, :], 30, 1e-2, 20, want_trace=True)
    assert obj_m[0] < loss and np.all(nn_m[0][pad] == 0.0)
    eng.adam_init(1e-2)
    eng.set_params(P, c["beta"])
    losses = eng.adam_run(25)
    nn_a, _ = eng.get_params()
    assert losses[-1] < losses[0] and np.all(nn_a[pad] == 0.0) and np.max(np.abs(nn_a[~pad] - P[~pad])) > 1e-3
    eng.set_param_mask(None)
    eng.close()


def test_train_with_unequal_widths_through_the_api():
    """api.chain([6, 3], tanh) end to end: models built on it, `loss`, `train` (screening + Adam + L-BFGS inside the
    library) -- the trained parameter vectors keep exact zeros at the padding and unpad to a 43-parameter network whose
    loss, evaluated as a padded network again, is the reported objective."""
    from cude import api
    c = make_cpep_case(40, (2, 6, 2))
    net = api.chain([6, 3], "tanh")
    models = [api.CPeptideConditionalUDEModel(c["G"][i], c["tp"], c["age"][i], net, c["obs"][i], bool(c["t2dm"][i]))
              for i in range(40)]
    sols = api.train(models, c["tp"], c["obs"], np.random.default_rng(3), initial_guesses=200, selected_initials=3,
                     number_of_iterations_adam=40, number_of_iterations_lbfgs=30, n_steps=30)
    assert len(sols) == 3
    for sol in sols:
        nn = np.asarray(sol.u.neural)
        assert nn.size == 67 and np.all(nn[net.mask == 0.0] == 0.0) and np.count_nonzero(nn) == 43
        small = api.unpad_network([6, 3], nn)
        theta = api.ComponentArray(neural=api.pad_network([6, 3], small)[1], conditional=sol.u.conditional)
        assert abs(api.loss(theta, (models, c["tp"], c["obs"]), n_steps=30) - sol.objective) <= 1e-12 * sol.objective
    api.clear_cache()


def _gpu_lbfgs_rank(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "conditional-ude_amd"), os.path.join(here, "..", "oracle"), here):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from cude.engine import Engine
    from cude.parallel import ShardedTrainer, TorchCollective, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = make_cpep_case(n_total, (2, 4, 2))
    lo, hi = shard_bounds(n_total, world, rank)
    eng = Engine("cpep", (2, 4, 2), n_steps=30, n_state=2, device=0)
    eng.set_population_cpep(c["tp"], c["G"][lo:hi], c["obs"][lo:hi], c["age"][lo:hi], c["t2dm"][lo:hi])
    eng.set_params(c["nn"], c["beta"][lo:hi])
    tr = ShardedTrainer(eng, TorchCollective(dist), transport="host")
    tr.sync_population_statistics()
    tr.adam_init(1e-2)
    for _ in range(3):
        tr.adam_step()
    res = tr.lbfgs(10)
    cond = tr.gather_conditional(n_total)
    nn, _ = eng.get_params()
    np.savez(os.path.join(out_dir, f"lb{rank}.npz"), f=res["f"], it=res["iterations"], calls=res["f_calls"], cond=cond,
             nn=nn)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_two_rank_sharded_lbfgs_matches_single_engine(tmp_path):
    """`_optimize` (src/parameter-estimation.jl:170-183: Adam, then L-BFGS + BackTracking) with the subjects sharded over
    two processes on one GPU: Adam x3 + L-BFGS x10 must follow the single-engine run (cude_train_restarts) -- same
    iteration / evaluation counts, parameters equal up to the summation order of the inner products."""
    import torch.multiprocessing as mp
    from cude.engine import Engine
    n_total, world = 157, 2
    port = free_port()
    mp.spawn(_gpu_lbfgs_rank, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "lb0.npz"), np.load(tmp_path / "lb1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k
    c = make_cpep_case(n_total, (2, 4, 2))
    eng = Engine("cpep", (2, 4, 2), n_steps=30, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    nn, cond, obj, trace = eng.train_restarts(c["nn"][None, :], c["beta"][None, :], 3, 1e-2, 10, want_trace=True)
    eng.close()
    assert int(r0["it"]) == int(np.sum(np.isfinite(trace[0, 3:]))) == 10
    assert abs(float(r0["f"]) - obj[0]) <= 1e-9 * abs(obj[0])
    assert np.allclose(r0["nn"], nn[0], rtol=0, atol=1e-8) and np.allclose(r0["cond"], cond[0], rtol=0, atol=1e-8)


# ------------------------------------------------------------------ sharded SAEM (BASELINE configs[4]), 2 ranks, 1 GPU
_SAEM_KW = dict(sigma=0.4, prior_eta=-0.6, prior_omega=0.8, iterations=6, n_burnin_iterations=2, n_mcmc_steps=3,
                initial_mcmc_steps=4, proposal_std=0.3)


def _saem_draws(n_total, lo, hi):
    def draws(it, steps):
        rng = np.random.default_rng(1000 + it)             # one global stream, sliced per shard
        return rng.standard_normal((steps, n_total))[:, lo:hi], rng.random((steps, n_total))[:, lo:hi]
    return draws


def _saem_engine(c, lo, hi):
    from cude.engine import Engine
    eng = Engine("cpep", (2, 4, 2), n_steps=30, n_state=2, device=0)
    eng.set_population_cpep(c["tp"], c["G"][lo:hi], c["obs"][lo:hi], c["age"][lo:hi], c["t2dm"][lo:hi])
    return eng


def _gpu_saem_rank(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "conditional-ude_amd"), os.path.join(here, "..", "oracle"), here):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from cude.parallel import TorchCollective, saem_loop, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = make_cpep_case(n_total, (2, 4, 2))
    lo, hi = shard_bounds(n_total, world, rank)
    eng = _saem_engine(c, lo, hi)
    coll = TorchCollective(dist)
    res = saem_loop(eng, len(c["tp"]), c["nn"], collective=coll, draws=_saem_draws(n_total, lo, hi), **_SAEM_KW)
    p_all = np.zeros(n_total)
    p_all[lo:hi] = res.p_individuals
    np.savez(os.path.join(out_dir, f"saem{rank}.npz"), nn=res.p_neural, p=coll.allreduce_sum(p_all), omega=res.Omega,
             sigma=res.sigma, eta=res.eta, nll=res.total_nll_values, acc=res.acceptance_rates)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_two_rank_sharded_saem_matches_single_engine(tmp_path):
    """The E-step needs no communication; per iteration the ranks exchange 4 + 5 x (P+2) doubles.  With the same
    global draws the sharded run must make the same accept / reject decisions and reach the same parameters."""
    import torch.multiprocessing as mp
    from cude.parallel import saem_loop
    n_total, world = 201, 2
    port = free_port()
    mp.spawn(_gpu_saem_rank, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "saem0.npz"), np.load(tmp_path / "saem1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k
    c = make_cpep_case(n_total, (2, 4, 2))
    eng = _saem_engine(c, 0, n_total)
    one = saem_loop(eng, len(c["tp"]), c["nn"], collective=None, draws=_saem_draws(n_total, 0, n_total), **_SAEM_KW)
    eng.close()
    assert np.array_equal(r0["acc"], one.acceptance_rates)
    assert np.allclose(r0["p"], one.p_individuals, rtol=0, atol=1e-10)
    assert np.allclose(r0["nn"], one.p_neural, rtol=0, atol=1e-9)
    assert np.allclose(r0["nll"], one.total_nll_values, rtol=1e-10)
    assert abs(r0["sigma"] - one.sigma) < 1e-10 and abs(r0["omega"] - one.Omega) < 1e-11


def test_validate_suppression_model_sigma_and_individual_maps():
    """validate_suppression_model_sigma (suppression_model.jl:224-275; suppression/figures.jl:46) and
    compute_individual_maps (src/saem.jl:68-84) through the GPU path, against scalar minimisations of the oracle's
    own objectives (scipy Brent on a bracket from a grid)."""
    import cude_oracle as o
    from scipy.optimize import minimize_scalar
    from cude import api
    g = dict(np.load(os.path.join(GOLD, "suppression_lambda0.npz")))
    prob = api.SuppressionProblem(api.neural_network_model(5, 3, input_dims=4))
    nn, tp, arch = g["nn_4x3x5x1"][0], g["timepoints"], (4, 3, 5)
    data = g["validation_data"][:, :, :6]
    res, obj = api.validate_suppression_model_sigma([0.0, 1.0], prob, data, tp, nn)
    assert res.ode.shape == (6,) and res.sigma.shape == (6, 3) and obj.shape == (6,)
    n = len(tp)
    for i in range(6):
        def nll(th, i=i):
            traj = o.supp_forward(np, nn, np.array([th]), data[:, :, i:i + 1], tp, arch, 30)
            sse = np.array([sum((traj[t][s][0] - data[s, t, i]) ** 2 for t in range(n)) for s in range(3)])
            return float(np.sum(0.5 * n * (np.log(sse / n) + 1.0))), sse
        grid = np.linspace(-8.0, 5.0, 53)
        k = int(np.argmin([nll(t)[0] for t in grid]))
        ref = minimize_scalar(lambda t: nll(t)[0], bounds=(grid[max(k - 1, 0)], grid[min(k + 1, 52)]), method="bounded",
                              options={"xatol": 1e-10})
        assert obj[i] <= ref.fun + 1e-7 * max(1.0, abs(ref.fun))             # the global minimum over the bracket
        assert abs(nll(res.ode[i])[0] - obj[i]) <= 1e-8 * max(1.0, abs(obj[i]))   # ... of the same function
        assert np.allclose(res.sigma[i], np.sqrt(nll(res.ode[i])[1] / n), rtol=1e-8)
    one, obj1 = api.validate_suppression_model_sigma([0.0], prob, data[:, :, 2], tp, nn)     # the reference's call shape
    assert abs(one.ode - res.ode[2]) < 1e-6 and one.sigma.shape == (3,) and abs(obj1 - obj[2]) < 1e-9
    api.clear_cache()
    # MAP estimates of the conditional parameters, c-peptide model
    net = api.chain(4, 2, "tanh")
    gc, models = _ohashi_models(api, net, n=12)
    nn4, sigma, omega, prior = gc["nn_2x4x4x1"][0], 0.35, 0.8, -0.7
    maps = api.compute_individual_maps(np.zeros(12), nn4, models, gc["timepoints"], gc["cpeptide"][:12], sigma, omega,
                                       prior_individual=prior)
    import c_oracle as co
    for i in range(12):
        def objective(x, i=i):
            r = co.cpep(gc["timepoints"], gc["glucose"][i:i + 1], gc["cpeptide"][i:i + 1], gc["ages"][i:i + 1],
                        gc["t2dm"][i:i + 1], (2, 4, 2), nn4, np.array([x]), api.default_steps(gc["timepoints"]), 2,
                        want_grad=False)
            return api.map_objective(x, r["sse"][0], 5, sigma, omega, prior_individual=prior)
        grid = np.linspace(-6.0, 4.0, 41)
        k = int(np.argmin([objective(t) for t in grid]))
        ref = minimize_scalar(objective, bounds=(grid[max(k - 1, 0)], grid[min(k + 1, 40)]), method="bounded",
                              options={"xatol": 1e-10})
        assert objective(maps[i]) <= ref.fun + 1e-8 * max(1.0, abs(ref.fun))
        assert abs(maps[i] - ref.x) < 1e-4
    api.clear_cache()
