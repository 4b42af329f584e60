"""Round-3 additions to the C ABI and fixes of the round-2 advisor findings, on the device."""
import os

import numpy as np
import pytest
import torch

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu


def _engine(c, arch, n_steps=30, n_state=2, lam=0.0):
    from cude.engine import Engine
    eng = Engine("cpep", arch, n_steps=n_steps, n_state=n_state, lam=lam)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    return eng


def test_device_resident_partial_and_apply_match_the_host_round_trip():
    """cude_partial_buffer / cude_loss_grad_partial_device / cude_adam_apply_device: the bring-your-own-collective step
    with the P+2 doubles left on the device.  A torch tensor aliasing the buffer (what bench.py hands to
    torch.distributed.all_reduce when the built-in communicator is unavailable) shows what cude_loss_grad_partial copies
    to the host, an in-place change of it is what cude_adam_apply_device consumes, and the step equals the host one."""
    arch, N = (2, 6, 2), 300
    c = make_cpep_case(N, arch)
    a, b = _engine(c, arch, lam=0.01), _engine(c, arch, lam=0.01)
    for e in (a, b):
        e.set_global_subjects(2 * N)             # as if a second rank held as many subjects again
        e.adam_init(1e-2)
    t = a.partial_tensor(torch, torch.device("cuda", 0))
    assert t.dtype == torch.float64 and t.numel() == a.P + 2 and t.data_ptr() == a.partial_buffer()[0]
    for _ in range(3):
        part, _ = b.loss_grad_partial()
        a.loss_grad_partial_device()
        assert np.array_equal(t.cpu().numpy(), part)
        t.mul_(2.0)                              # "all-reduce" over two identical ranks, in place on the device
        torch.cuda.current_stream().synchronize()
        la, lb = a.adam_apply_device(), b.adam_apply(2.0 * part)
        assert la == lb
    (nn_a, cond_a), (nn_b, cond_b) = a.get_params(), b.get_params()
    assert np.array_equal(nn_a, nn_b) and np.array_equal(cond_a, cond_b)
    a.close()
    b.close()


def test_set_tolerances_invalidates_a_captured_optimiser_iteration():
    """in adaptive mode cude_adam_run replays a captured launch whose arguments hold the tolerances
    by value; cude_set_tolerances must drop that graph, or training silently keeps the old tolerances."""
    arch, N = (2, 4, 2), 200
    c = make_cpep_case(N, arch)
    runs = {}
    for mode in ("run", "step"):
        eng = _engine(c, arch, n_steps=0)
        eng.adam_init(1e-2)
        first = eng.adam_run(2) if mode == "run" else np.array([eng.adam_step() for _ in range(2)])
        eng.set_tolerances(1e-9, 1e-7)
        second = eng.adam_run(2) if mode == "run" else np.array([eng.adam_step() for _ in range(2)])
        runs[mode] = (first, second, eng.get_params())
        eng.close()
    assert np.array_equal(runs["run"][0], runs["step"][0])
    assert np.array_equal(runs["run"][1], runs["step"][1])                     # same losses AFTER the change ...
    assert np.array_equal(runs["run"][2][0], runs["step"][2][0])               # ... and the same parameters
    # and the change of tolerances is visible at all (the tight solve gives a different loss at the same parameters)
    eng = _engine(c, arch, n_steps=0)
    l_loose = eng.forward()["loss"]
    eng.set_tolerances(1e-9, 1e-7)
    assert eng.forward()["loss"] != l_loose
    eng.close()


def test_param_mask_set_after_adam_steps_freezes_entries_at_once():
    """a mask set AFTER Adam has gathered moments must also silence those moments, otherwise a frozen
    entry keeps drifting by lr * m_hat / (sqrt(v_hat) + eps) while m decays."""
    arch, N = (2, 4, 2), 150
    c = make_cpep_case(N, arch)
    eng = _engine(c, arch)
    eng.adam_init(1e-2)
    for _ in range(5):
        eng.adam_step()
    nn_before, _ = eng.get_params()
    mask = np.ones(eng.P)
    mask[[0, 7, 20, eng.P - 1]] = 0.0
    eng.set_param_mask(mask)
    for _ in range(5):
        eng.adam_step()
    nn_after, _ = eng.get_params()
    frozen = mask == 0.0
    assert np.array_equal(nn_after[frozen], nn_before[frozen])
    assert np.all(nn_after[~frozen] != nn_before[~frozen])
    eng.close()


def test_kept_activation_variant_of_the_gradient_kernel(monkeypatch):
    """CUDE_CPEP_KEEP=1 (cude_cpep.hip: KEEP): upper-layer activations kept in HBM between the sweeps instead of
    recomputed (measured slower at the benchmark sizes, hence off by default).  Same forward sweep, so the same loss
    bit for bit; the reverse sweep runs Mlp::backward on the kept values instead of the fused evaluation, whose
    multiply-adds the compiler contracts differently: gradients equal to rounding."""
    arch, N = (2, 6, 2), 700
    c = make_cpep_case(N, arch)
    out = {}
    for keep in ("0", "1"):
        monkeypatch.setenv("CUDE_CPEP_KEEP", keep)
        monkeypatch.setenv("CUDE_CPEP_PATH", "1")             # the one-lane kernel (a small population would time-split)
        eng = _engine(c, arch, n_state=3)
        out[keep] = eng.loss_grad()
        eng.close()
    (l0, g0, c0), (l1, g1, c1) = out["0"], out["1"]
    assert l0 == l1
    assert np.max(np.abs(g0 - g1)) <= 1e-13 * np.max(np.abs(g0)) and np.max(np.abs(c0 - c1)) <= 1e-13 * np.max(np.abs(c0))


@pytest.mark.parametrize("model", ["cpep", "supp"])
def test_adaptive_regroup_changes_the_launch_order_not_the_results(model):
    """cude_adaptive_regroup: subjects ordered by the accepted-step counts of the last gradient evaluation, so that the
    lanes of a wave finish together.  Every per-subject quantity (SSE, dL/dcond, the accepted steps themselves) is the
    same bit for bit afterwards; the loss and the shared gradient -- sums over the subjects in a new order -- to rounding;
    training continues; a new population starts in its own order again."""
    from conftest import make_supp_case
    from cude.engine import Engine
    if model == "cpep":
        arch, N = (2, 4, 2), 3000
        c = make_cpep_case(N, arch)
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
    else:
        N = 1500
        c = make_supp_case(N)
        eng = Engine("supp", c["arch"], n_steps=0, lam=0.01)
        eng.set_population_supp(c["tp"], c["data"])
        eng.set_params(c["nn"], c["theta"])
    l0, g0, c0 = eng.loss_grad()
    sse0 = eng.forward(want_sse=True)["sse"]
    eng.loss_grad()
    probe = [0, 1, 63, 64, N // 2, N - 1]
    steps0 = [eng.adaptive_steps(s) for s in probe]
    before, after = eng.adaptive_regroup()
    assert after <= 1 and after <= before                                # sorted: a wave holds (almost) one step count
    if model == "cpep":
        assert before >= 2                                                # (this population does vary; the synthetic
                                                                          # suppression one takes the same steps everywhere)
    with pytest.raises(Exception):
        eng.adaptive_steps(0)                                             # the old tape is in the old order: invalidated
    l1, g1, c1 = eng.loss_grad()
    assert np.array_equal(c0, c1)
    assert np.array_equal(sse0, eng.forward(want_sse=True)["sse"])
    assert abs(l1 - l0) <= 1e-13 * abs(l0) and np.max(np.abs(g1 - g0)) <= 1e-12 * np.max(np.abs(g0))
    eng.loss_grad()
    for s, (t_a, dt_a) in zip(probe, steps0):
        t_b, dt_b = eng.adaptive_steps(s)
        assert np.array_equal(t_a, t_b) and np.array_equal(dt_a, dt_b)
    eng.adam_init(1e-3)
    losses = eng.adam_run(4)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
    # a second regrouping (the counts have moved with the parameters) and a new population both work
    eng.loss_grad()
    eng.adaptive_regroup()
    assert np.isfinite(eng.loss_grad()[0])
    if model == "cpep":
        eng.set_population_cpep(c["tp"], c["G"][:200], c["obs"][:200], c["age"][:200], c["t2dm"][:200])
        eng.set_params(c["nn"], c["beta"][:200])
    else:
        eng.set_population_supp(c["tp"], c["data"][:, :, :200])
        eng.set_params(c["nn"], c["theta"][:200])
    l2, _, c2 = eng.loss_grad()
    assert np.isfinite(l2) and c2.size == 200
    eng.close()


def test_adam_run_regroups_a_large_adaptive_population_by_itself(monkeypatch):
    """cude_adam_run on an adaptive population of >= 8192 subjects orders the launch by accepted-step count once the first
    evaluation has told the counts (CUDE_NO_AUTO_REGROUP=1 leaves it to the caller): same losses to rounding, the order
    is in place afterwards (an explicit regroup finds nothing left to gain)."""
    from cude.engine import Engine
    arch, N = (2, 4, 2), 9000
    c = make_cpep_case(N, arch)
    traces, spread = {}, {}
    for auto in (False, True):
        if auto:
            monkeypatch.delenv("CUDE_NO_AUTO_REGROUP", raising=False)
        else:
            monkeypatch.setenv("CUDE_NO_AUTO_REGROUP", "1")
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        eng.adam_init(1e-3)
        eng.adam_step()                                   # tells the counts
        traces[auto] = eng.adam_run(5)
        eng.loss_grad()
        spread[auto], _ = eng.adaptive_regroup()
        eng.close()
    # the shared gradient is summed in another order: parameters move by rounding, and the adaptive solve turns that
    # into accept / reject flips for a few subjects (the map is discontinuous in its inputs): 1e-7 here
    assert np.allclose(traces[True], traces[False], rtol=1e-5, atol=0)
    # the first step sees identical parameters: the same per-subject residuals, summed over workgroups in the new order
    assert abs(traces[True][0] - traces[False][0]) <= 1e-13 * abs(traces[False][0])
    # the launch order adam_run put in place is still (nearly) sorted six small steps later; the caller's order never was
    assert spread[True] < spread[False] and spread[False] >= 2


@pytest.mark.parametrize("model", ["cpep", "supp"])
def test_queued_iterations_equal_single_steps_for_every_graph_layout(model):
    """cude_adam_run captures eight iterations per graph and single ones for the remainder, and the Adam state advance
    rides in the tail reduction (c-peptide, lambda = 0) or the L2-term kernel (suppression, lambda != 0): for iteration
    counts that use only the single graph (7), only the eightfold one (8, 16), both (9, 17), the loss trace and the
    parameters equal those of single cude_adam_step calls bit for bit -- also across a skipped step (a subject whose
    conditional parameter is NaN for a while: no update, no advance of the bias-correction powers)."""
    from cude.engine import Engine
    if model == "cpep":
        arch = (2, 4, 2)
        c = make_cpep_case(300, arch)
        mk = lambda: Engine("cpep", arch, n_steps=30, n_state=2)
        pop = lambda e: e.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        nn0, cond0 = c["nn"], c["beta"]
    else:
        c = make_supp_case(200)
        mk = lambda: Engine("supp", c["arch"], n_steps=30, lam=0.01)
        pop = lambda e: e.set_population_supp(c["tp"], c["data"])
        nn0, cond0 = c["nn"], c["theta"]
    for n_iters in (7, 8, 9, 16, 17):
        a, b = mk(), mk()
        for e in (a, b):
            pop(e)
            e.set_params(nn0, cond0)
            e.adam_init(1e-2)
        la = a.adam_run(n_iters)
        lb = np.array([b.adam_step() for _ in range(n_iters)])
        assert np.array_equal(la, lb), (model, n_iters)
        pa, pb = a.get_params(), b.get_params()
        assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
        # a failing subject: the step is skipped (loss +Inf), nothing moves, and the run continues as if it had not been
        bad = pa[1].copy()
        bad[3] = np.nan
        for e in (a, b):
            e.set_params(None, bad)
        la = a.adam_run(3)
        lb = np.array([b.adam_step() for _ in range(3)])
        assert np.all(np.isinf(la)) and np.all(np.isinf(lb))
        qa, qb = a.get_params(), b.get_params()
        assert np.array_equal(qa[0], pa[0]) and np.array_equal(qb[0], pa[0])
        for e in (a, b):
            e.set_params(None, pa[1])
        la = a.adam_run(9)
        lb = np.array([b.adam_step() for _ in range(9)])
        assert np.array_equal(la, lb) and np.all(np.isfinite(la))
        assert np.array_equal(a.get_params()[0], b.get_params()[0])
        a.close()
        b.close()


def test_watched_result_slots_never_return_an_earlier_launch():
    """The reduction writes [sum loss, failures] into page-locked memory and the host watches the slots instead of waiting
    for the stream -- only for calls whose single pending output is that pair.  Steps whose loss nobody fetches must not
    write there (a later watch would take their pair for its own), and a forward call that also copies per-subject
    results still waits for the stream."""
    from cude.engine import Engine
    arch = (2, 6, 2)
    c = make_cpep_case(5000, arch)
    a = Engine("cpep", arch, n_steps=30, n_state=3)
    b = Engine("cpep", arch, n_steps=30, n_state=3)
    for e in (a, b):
        e.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        e.set_params(c["nn"], c["beta"])
        e.adam_init(1e-2)
    for k in range(12):
        want = k % 4 == 3
        la = a.adam_step(want_loss=want)                  # three unfetched steps queued in front of every fetched one
        lb = b.adam_step()
        if want:
            assert la == lb, (k, la, lb)
    out = a.forward(want_sse=True)
    ref = b.forward(want_sse=True)
    assert out["loss"] == ref["loss"] and np.array_equal(out["sse"], ref["sse"]) and np.all(out["sse"] > 0)
    assert a.forward()["loss"] == out["loss"]
    a.close()
    b.close()
