"""cude_multistart_loss_grad: loss and gradient of K parameter sets in one launch, and the restarts of `train` /
`fit_suppression_model` trained side by side on top of it."""
import numpy as np
import pytest
import torch  # noqa: F401

from conftest import make_cpep_case, make_supp_case

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fixed_step_default")]


@pytest.mark.parametrize("model", ["cpep", "cpep4", "supp", "sym"])
def test_every_set_equals_its_own_loss_grad_call(model):
    """Bitwise: the per-set work is the same lanes, the same blocks and the same reduction tree as cude_loss_grad
    of the one-lane kernel (CUDE_CPEP_PATH=1 pins that kernel for the single-set reference evaluation)."""
    import os
    from cude.engine import Engine
    os.environ["CUDE_CPEP_PATH"] = "1"
    try:
        rng = np.random.default_rng(8)
        K = 7
        if model == "supp":
            c = make_supp_case(150)
            eng = Engine("supp", c["arch"], n_steps=30, lam=0.02)
            eng.set_population_supp(c["tp"], c["data"])
            nn0, cond0, N = c["nn"], c["theta"], 150
        elif model == "sym":
            c = make_cpep_case(150, (2, 6, 2))
            eng = Engine("cpep_sym", n_steps=30, n_state=2, cond_space="log")
            eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
            nn0, cond0, N = np.array([1.78]), np.log(40.0) + c["beta"], 150
        else:
            arch = (2, 6, 2) if model == "cpep" else (2, 4, 2)
            c = make_cpep_case(150, arch)
            eng = Engine("cpep", arch, n_steps=30, n_state=3)
            eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
            nn0, cond0, N = c["nn"], c["beta"], 150
        nn_sets = nn0[None, :] * (1.0 + 0.2 * rng.standard_normal((K, nn0.size)))
        cond_sets = cond0[None, :] + 0.3 * rng.standard_normal((K, N))
        cond_sets[3, 17] = np.nan                                   # one failing subject in set 3
        eng.set_params(nn0, cond0)
        losses, g_nn, g_cond = eng.multistart_loss_grad(nn_sets, cond_sets)
        nn_now, cond_now = eng.get_params()                         # the context's own parameters are untouched
        assert np.array_equal(nn_now, nn0) and np.array_equal(cond_now, cond0)
        assert losses[3] == np.inf
        for k in range(K):
            eng.set_params(nn_sets[k], cond_sets[k])
            l, gn, gc = eng.loss_grad()
            assert (l == losses[k]) or (np.isinf(l) and np.isinf(losses[k]))
            if k != 3:
                assert np.array_equal(gn, g_nn[k]) and np.array_equal(gc, g_cond[k])
        eng.close()
    finally:
        del os.environ["CUDE_CPEP_PATH"]


@pytest.mark.parametrize("arch,n_state,N", [((2, 6, 2), 3, 150), ((2, 4, 2), 2, 57), ((3, 4, 2), 2, 117)])
def test_sets_on_the_time_split_path(arch, n_state, N, monkeypatch):
    """Small populations run time-split, and so do their restarts: the set index is a third grid dimension of the
    chunk kernels (K x L short waves instead of K single-wave chains).  Every set must be bit-identical to its own
    cude_loss_grad on the same context (same kernels, same chunking), agree with the one-lane multi-start to rounding,
    and a failing subject must fail its set only."""
    from cude.engine import Engine
    rng = np.random.default_rng(18)
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    K = 9
    nn_sets = c["nn"][None, :] * (1.0 + 0.2 * rng.standard_normal((K, c["nn"].size)))
    cond_sets = c["beta"][None, :] + 0.3 * rng.standard_normal((K, N))
    cond_sets[4, N // 3] = np.nan
    loss, g_nn, g_cond = eng.multistart_loss_grad(nn_sets, cond_sets)
    assert np.isinf(loss[4]) and np.all(np.isfinite(np.delete(loss, 4)))
    for k in range(K):
        if k == 4:
            continue
        eng.set_params(nn_sets[k], cond_sets[k])
        l1, gn1, gc1 = eng.loss_grad()
        assert l1 == loss[k] and np.array_equal(gn1, g_nn[k]) and np.array_equal(gc1, g_cond[k])
    eng.set_option("ms_split", 0)                                # the one-lane kernel with the sets in grid y
    loss1, g_nn1, g_cond1 = eng.multistart_loss_grad(nn_sets, cond_sets)
    eng.close()
    ok = np.arange(K) != 4
    assert np.isinf(loss1[4]) and np.max(np.abs(loss1[ok] / loss[ok] - 1)) < 1e-12
    assert np.max(np.abs(g_nn1[ok] - g_nn[ok])) <= 1e-10 * np.max(np.abs(g_nn[ok]))
    assert np.max(np.abs(g_cond1[ok] - g_cond[ok])) <= 1e-10 * np.max(np.abs(g_cond[ok]))
    assert not np.array_equal(g_nn1[ok], g_nn[ok])              # ... and it IS a different path


def test_side_by_side_training_follows_the_serial_restarts():
    """_batched_adam_then_lbfgs vs the one-after-the-other loop: the first Adam iterations agree to rounding (host
    Adam vs the device Adam kernel), and a full short training ends at comparable objectives for every restart."""
    from cude import api
    from cude.engine import Engine
    c = make_cpep_case(57, (2, 4, 2))
    eng = Engine("cpep", (2, 4, 2), n_steps=32, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    rng = np.random.default_rng(2)
    K = 6
    nn_inits = np.stack([api.init_params(api.chain(4, 2, "tanh"), rng) for _ in range(K)])
    cond_inits = rng.uniform(-2.0, 0.0, (K, 57))
    few = api._batched_adam_then_lbfgs(eng, nn_inits, cond_inits, 5, 0, 1e-2, native=False)
    few_native = api._batched_adam_then_lbfgs(eng, nn_inits, cond_inits, 5, 0, 1e-2)     # cude_train_restarts
    for k in range(K):
        nn, cond, _ = api._adam_then_lbfgs(eng, nn_inits[k], cond_inits[k], 5, 0, 1e-2)
        assert np.allclose(few[k][0], nn, rtol=0, atol=1e-12) and np.allclose(few[k][1], cond, rtol=0, atol=1e-12)
        assert np.allclose(few_native[k][0], nn, rtol=0, atol=1e-12)
        assert np.allclose(few_native[k][1], cond, rtol=0, atol=1e-12)
        assert abs(few_native[k][2] - few[k][2]) <= 1e-12 * few[k][2]
    # a short L-BFGS stage: the native state machines and the Python generators take the same steps
    py = api._batched_adam_then_lbfgs(eng, nn_inits, cond_inits, 5, 6, 1e-2, native=False)
    nat = api._batched_adam_then_lbfgs(eng, nn_inits, cond_inits, 5, 6, 1e-2)
    for k in range(K):
        assert abs(nat[k][2] - py[k][2]) <= 1e-9 * py[k][2] and np.allclose(nat[k][0], py[k][0], rtol=0, atol=1e-8)
    full = api._batched_adam_then_lbfgs(eng, nn_inits, cond_inits, 150, 60, 1e-2)
    start = eng.multistart_forward(nn_inits, cond_inits)
    for k in range(K):
        _, _, obj = api._adam_then_lbfgs(eng, nn_inits[k], cond_inits[k], 150, 60, 1e-2)
        assert full[k][2] < 0.5 * start[k]
        assert abs(full[k][2] - obj) < 0.05 * obj + 1e-3
        eng.set_params(full[k][0], full[k][1])
        assert abs(eng.forward()["loss"] - full[k][2]) <= 1e-12 * full[k][2]       # objective is the loss at the result
    eng.close()


def test_suppression_restarts_side_by_side():
    from cude import api
    c = make_supp_case(37)
    rng = np.random.default_rng(5)
    net = api.neural_network_model(5, 3, input_dims=4)
    prob = api.SuppressionProblem(net)
    p_init = [api.ComponentArray(theta=rng.uniform(-1, 1, 37), neural=api.init_params(net, rng)) for _ in range(40)]
    sols, traces = api.fit_suppression_model(p_init, prob, c["data"], c["tp"], 0.0, select_best_n=5, adam_iters=60,
                                             lbfgs_iters=25)
    assert len(sols) == 5 and len(traces) == 5
    for s, tr in zip(sols, traces):
        assert len(tr) >= 60 and tr[-1] <= tr[0] and np.isfinite(s.objective)
        assert abs(api.suppression_loss(s.u, (prob, c["data"], c["tp"], 0.0)) - s.objective) <= 1e-10 * s.objective
    api.clear_cache()


def test_device_screening_keeps_the_best_candidates_across_chunks():
    """cude_screen_candidates = `losses_initial = [loss(p, ...) for p in initials]` + `partialsortperm(losses_initial,
    1:selected_initials)` (src/parameter-estimation.jl:359-372) with candidates streamed chunk by chunk and the running
    top-k kept on the device: must equal the losses of cude_multistart_forward sorted stably -- across chunk boundaries
    (70 000 candidates = 3 chunks), with ties (duplicated candidates: the lower index wins) and failed candidates."""
    import torch  # noqa: F401
    from cude.engine import Engine
    arch = (2, 4, 2)
    c = make_cpep_case(19, arch)
    K, keep = 70_000, 25
    eng = Engine("cpep", arch, n_steps=12, n_state=2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])

    def block(first, count):
        rng = np.random.default_rng(first)                      # deterministic per block, whatever the chunking
        nn = c["nn"][None, :] * (1 + 0.5 * rng.standard_normal((count, c["nn"].size)))
        cond = c["beta"][None, :] + rng.standard_normal((count, 19))
        return nn, cond
    B = 10_000
    table = [block(f, B) for f in range(0, K, B)]
    nn_all, cond_all = np.concatenate([t[0] for t in table]), np.concatenate([t[1] for t in table])
    nn_all[40_001], cond_all[40_001] = nn_all[7], cond_all[7]    # a tie across chunks
    nn_all[123, 3] = np.nan                                      # a failed candidate
    calls = []

    def gen(first, count):
        calls.append((first, count))
        return nn_all[first:first + count], cond_all[first:first + count]
    idx, loss, nn_sel, cond_sel = eng.screen_candidates(K, keep, gen)
    assert len(calls) >= 3 and sum(n for _, n in calls) == K and max(n for _, n in calls) < K
    ref = np.concatenate([eng.multistart_forward(nn_all[f:f + B], cond_all[f:f + B]) for f in range(0, K, B)])
    assert np.isinf(ref[123])
    order = np.argsort(ref, kind="stable")[:keep]
    assert np.array_equal(idx, order) and np.array_equal(loss, ref[order])
    assert np.array_equal(nn_sel, nn_all[order]) and np.array_equal(cond_sel, cond_all[order])
    if 7 in order:
        assert list(order).index(7) + 1 == list(order).index(40_001)
    # fewer candidates than requested
    idx2, loss2, _, _ = eng.screen_candidates(10, 25, lambda f, n: (nn_all[f:f + n], cond_all[f:f + n]))
    assert np.array_equal(idx2, np.argsort(ref[:10], kind="stable")) and loss2.size == 10
    eng.close()
