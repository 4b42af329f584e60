"""Non-conditional UDE (CPeptideUDEModel, src/c-peptide-models.jl:76-84,144-168; train src/parameter-estimation.jl:205-247):
the 1-input network rides on the conditional model's kernels as a 2-input network with a frozen zero column.  CPU part:
the embedding, and the oracle's own statement of the single-input production; GPU part: the product against it."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _mean_subject():
    d = np.load(os.path.join(GOLD, "ohashi_cude.npz"))
    train = np.isin(d["subject_no"], d["train_subject_numbers"])
    return (d["timepoints"], d["glucose"][train].mean(0), d["cpeptide"][train].mean(0), float(d["ages"][train].mean()), d)


def test_embedding_layout_and_mask():
    from cude import api
    net = api.chain(4, 2, "tanh", input_dims=1)
    assert net.n_params == 4 + 4 + 20 + 5 == 33
    p = np.arange(1.0, 34.0)
    q = api.embed_single_input(4, p)
    assert q.size == api.chain(4, 2, "tanh").n_params == 37
    assert np.array_equal(q[:4], p[:4]) and np.all(q[4:8] == 0.0) and np.array_equal(q[8:], p[4:])
    assert np.array_equal(api.extract_single_input(4, q), p)
    tp, G, cp, age, _ = _mean_subject()
    m = api.CPeptideUDEModel(G, tp, age, net, cp, False)
    assert m._carrier.arch == (2, 4, 2)
    assert np.array_equal(m._carrier.mask, api.embed_single_input(4, np.ones(33)))
    with pytest.raises(ValueError):
        api.CPeptideUDEModel(G, tp, age, api.chain(4, 2, "tanh"), cp, False)          # a 2-input network
    # unequal widths ride on the same mask mechanism: the padding of chain([4, 3]) and the zero column together
    net2 = api.chain([4, 3], "tanh", input_dims=1)
    m2 = api.CPeptideUDEModel(G, tp, age, net2, cp, False)
    assert m2._carrier.mask.size == 37 and np.all(m2._carrier.mask[4:8] == 0.0)
    assert int(m2._carrier.mask.sum()) == 4 + 4 + 12 + 3 + 3 + 1


def test_oracle_single_input_production_equals_the_carrier_with_a_zero_column():
    """The oracle states neural_network_production directly (arch (1, W, D)); the same trajectory comes out of its
    conditional statement with the embedded parameters, whatever the conditional parameter."""
    import cude_oracle as o
    from cude import api
    tp, G, cp, age, _ = _mean_subject()
    pop = o.CPepPopulation(tp, G[None, :], cp[None, :], [age], [False])
    rng = np.random.default_rng(3)
    p = 0.6 * rng.standard_normal(33)
    l1, _ = o.cpep_loss(np, p, np.zeros(1), pop, (1, 4, 2), 32)
    for beta in (0.0, -1.3):
        l2, _ = o.cpep_loss(np, api.embed_single_input(4, p), np.array([beta]), pop, (2, 4, 2), 32)
        assert abs(l1 - l2) <= 1e-14 * abs(l1)
    _, g1, _, _ = o.cpep_loss_grad_torch(p, np.zeros(1), pop, (1, 4, 2), 32)
    _, g2, gb, _ = o.cpep_loss_grad_torch(api.embed_single_input(4, p), np.zeros(1), pop, (2, 4, 2), 32)
    assert np.allclose(api.extract_single_input(4, g2), g1, rtol=1e-12, atol=1e-15) and gb[0] == 0.0


@pytest.mark.gpu
@pytest.mark.usefixtures("fixed_step_default")
def test_ude_loss_gradient_and_simulation_against_the_oracle():
    import torch  # noqa: F401
    import cude_oracle as o
    from cude import api
    tp, G, cp, age, d = _mean_subject()
    net = api.chain(4, 2, "tanh", input_dims=1)
    model = api.CPeptideUDEModel(G, tp, age, net, cp, False)
    pop = o.CPepPopulation(tp, G[None, :], cp[None, :], [age], [False])
    rng = np.random.default_rng(5)
    S = api.default_steps(tp)
    for _ in range(3):
        p = 0.6 * rng.standard_normal(33)
        ref, g_ref, _, _ = o.cpep_loss_grad_torch(p, np.zeros(1), pop, (1, 4, 2), S)
        val = api.loss(p, (model, tp, cp))
        val2, g = api.loss_and_gradient(p, (model, tp, cp))
        assert abs(val - ref) <= 1e-10 * abs(ref) and abs(val2 - val) <= 1e-13 * val    # the SUM of squares of one subject
        assert g.shape == (33,) and np.max(np.abs(g - g_ref)) <= 1e-9 * np.max(np.abs(g_ref))
    # `solve(model.problem, p = neural_network_parameters, saveat = timepoints, save_idxs = 1)` for every subject
    # (c-peptide/01-non-conditional.jl:60-65): 20 subjects in one launch against the oracle's trajectories
    idx = np.arange(20)
    models = [api.CPeptideUDEModel(d["glucose"][i], tp, d["ages"][i], net, d["cpeptide"][i], d["t2dm"][i]) for i in idx]
    sol = api.simulate(p, None, models, tp, d["cpeptide"][idx])
    popn = o.CPepPopulation(tp, d["glucose"][idx], d["cpeptide"][idx], d["ages"][idx], d["t2dm"][idx])
    traj = o.cpep_forward(np, p, np.zeros(20), popn, (1, 4, 2), S, 2, "log")
    ref_traj = np.stack([traj[t][0] for t in range(len(tp))], axis=1)
    assert sol.shape == (20, len(tp)) and np.max(np.abs(sol - ref_traj)) <= 1e-10 * np.max(np.abs(ref_traj))


@pytest.mark.gpu
@pytest.mark.usefixtures("fixed_step_default")
def test_train_ude_model_as_the_reference_script_does():
    """`optsols = train(model_train, timepoints, mean_c_peptide, rng)` (c-peptide/01-non-conditional.jl:25-29) at reduced
    counts: solutions carry the 1-input parameter vector, objectives are the SSE at it, the best one fits the mean
    curve, and the frozen column never moved (the objective re-evaluated through `loss` is the stored one)."""
    import torch  # noqa: F401
    from cude import api
    tp, G, cp, age, _ = _mean_subject()
    net = api.chain(4, 2, "tanh", input_dims=1)
    model = api.CPeptideUDEModel(G, tp, age, net, cp, False)
    rng = np.random.default_rng(11)
    sols = api.train(model, tp, cp, rng, initial_guesses=400, selected_initials=4, number_of_iterations_adam=300,
                     number_of_iterations_lbfgs=150)
    assert 1 <= len(sols) <= 4 and all(s.u.shape == (33,) for s in sols)
    best = min(sols, key=lambda s: s.objective)
    assert abs(api.loss(best.u, (model, tp, cp)) - best.objective) <= 1e-9 * max(best.objective, 1e-12)
    flat = float(np.sum((cp - cp[0]) ** 2))                        # no production at all: the curve stays at c0
    assert best.objective < 0.05 * flat
    one = api.train(model, tp, cp, np.random.default_rng(11), initial_guesses=50, selected_initials=1,
                    number_of_iterations_adam=50, number_of_iterations_lbfgs=20, side_by_side=False)
    assert len(one) == 1 and np.isfinite(one[0].objective)
