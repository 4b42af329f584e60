"""The general network `chain(widths, activation_functions; input_dims, output_activation)` of the reference
(src/neural-network.jl:42-58; its docstring's example is chain([10, 20, 30], [tanh, relu, softplus]; input_dims = 4)) on
the library's fallback kernel (cude_set_network, csrc/cude_generic.hip): run-time widths, one activation function per
layer, weights staged in LDS, one lane per subject, forward and adjoint, fixed-step and adaptive, both network models.
Checked against the oracle's restatement of the same network (oracle/cude_oracle.py `mlp` with a general `arch`; torch
autograd for the fixed-step gradient, complex-step replay for the adaptive one) at loss 1e-10 / gradient 1e-9, against
the TUNED kernels on shapes both can run, and through the entry points built on top of a loss evaluation."""
import numpy as np
import pytest

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu

DOC_EXAMPLE = (4, (10, 20, 30), ("tanh", "relu", "softplus"), "softplus")      # src/neural-network.jl:36


def _supp_engine(arch, c, n_steps, lam=0.01, options=()):
    from cude.engine import Engine
    eng = Engine("supp", arch, n_steps=n_steps, lam=lam)
    for k, v in options:
        eng.set_option(k, v)
    return eng


def test_docstring_example_through_the_c_abi_matches_the_oracle():
    import cude_oracle as o
    N, S = 20, 12
    c = make_supp_case(N, (4, 3, 5))
    nn = o.glorot_params(DOC_EXAMPLE, 3)
    assert nn.size == 931
    eng = _supp_engine(DOC_EXAMPLE, c, S)
    assert eng.fallback_kernel and eng.P == 931
    eng.set_population_supp(c["tp"], c["data"])
    eng.set_params(nn, c["theta"])
    fwd = eng.forward(want_sse=True, want_traj=True)
    loss, g_nn, g_th = eng.loss_grad()
    ref, rg_nn, rg_th, _ = o.supp_loss_grad_torch(nn, c["theta"], c["data"], c["tp"], DOC_EXAMPLE, S, 0.01)
    assert abs(fwd["loss"] - ref) <= 1e-10 * abs(ref) and abs(loss - ref) <= 1e-10 * abs(ref)
    assert np.max(np.abs(g_nn - rg_nn)) <= 1e-9 * np.max(np.abs(rg_nn))
    assert np.max(np.abs(g_th - rg_th)) <= 1e-9 * np.max(np.abs(rg_th))
    assert np.count_nonzero(g_nn) > 800                       # (relu kills some units; every layer receives a gradient)
    traj = o.supp_forward(np, nn, c["theta"], c["data"], c["tp"], DOC_EXAMPLE, S)
    ref_traj = np.stack([np.stack(traj[t]) for t in range(len(c["tp"]))], axis=1)      # (3, T, N)
    assert np.max(np.abs(fwd["traj"] - ref_traj)) <= 1e-11 * np.max(np.abs(ref_traj))
    # the optimiser entry points run on it: Adam steps one by one and queued (captured graphs) give the same bits
    eng.adam_init(1e-3)
    a = [eng.adam_step() for _ in range(3)]
    eng.set_params(nn, c["theta"])
    eng.adam_init(1e-3)
    b = eng.adam_run(3)
    assert np.array_equal(a, b) and a[2] < a[0]
    eng.close()


@pytest.mark.parametrize("arch,n_state", [((2, (5, 3), ("sigmoid", "identity"), "softplus"), 2),
                                          ((2, (7,), ("softplus",), "identity"), 3),
                                          ((3, (4, 6, 2), ("relu", "tanh", "sigmoid"), "softplus"), 2)],
                         ids=["2-5-3-1", "2-7-1-three-states", "3-4-6-2-1-covariate"])
def test_cpeptide_general_networks_fixed_step(arch, n_state):
    import cude_oracle as o
    from cude.engine import Engine
    N, S = 70, 16                                         # (two workgroups, the second part-filled)
    base = make_cpep_case(N, (arch[0], 4, 2))
    nn = o.glorot_params(arch, 5)
    pop = o.CPepPopulation(base["tp"], base["G"], base["obs"], base["age"], base["t2dm"], covariate=(arch[0] == 3))
    eng = Engine("cpep", arch, n_steps=S, n_state=n_state)
    assert eng.fallback_kernel and eng.P == o.n_params(arch)
    eng.set_population_cpep(base["tp"], base["G"], base["obs"], base["age"], base["t2dm"])
    eng.set_params(nn, base["beta"])
    loss, g_nn, g_b = eng.loss_grad()
    ref, rg_nn, rg_b, _ = o.cpep_loss_grad_torch(nn, base["beta"], pop, arch, S, n_state)
    assert abs(loss - ref) <= 1e-10 * abs(ref)
    assert np.max(np.abs(g_nn - rg_nn)) <= 1e-9 * np.max(np.abs(rg_nn))
    assert np.max(np.abs(g_b - rg_b)) <= 1e-9 * np.max(np.abs(rg_b))
    assert abs(eng.forward()["loss"] - ref) <= 1e-10 * abs(ref)
    # one failing subject fails the evaluation (+Inf) and nothing else
    beta = base["beta"].copy()
    beta[11] = np.nan
    eng.set_params(nn, beta)
    out = eng.forward(want_sse=True)
    assert np.isinf(out["loss"]) and eng.n_failed() == 1 and np.isfinite(np.delete(out["sse"], 11)).all()
    eng.close()


def test_adaptive_mode_on_general_networks():
    """n_steps = 0, the reference's solver mode: the fallback kernel carries OrdinaryDiffEq's controller as the tuned
    adaptive kernels do; the gradient is the adjoint of the accepted steps."""
    import cude_oracle as o
    from cude.engine import Engine
    arch = (4, (6, 5), ("relu", "tanh"), "softplus")
    N = 9
    c = make_supp_case(N, (4, 3, 5))
    nn = o.glorot_params(arch, 9)
    eng = Engine("supp", arch, n_steps=0, lam=0.01)
    eng.set_population_supp(c["tp"], c["data"])
    eng.set_params(nn, c["theta"])
    loss, g_nn, g_th = eng.loss_grad()
    steps = [eng.adaptive_steps(i) for i in range(N)]
    assert all(5 <= len(t) <= 200 for t, _ in steps)
    rl, rg, rt, _ = o.supp_replay_loss_grad(nn, c["theta"], c["data"], c["tp"], arch, 0.01,
                                            [list(zip(t, dt)) for t, dt in steps])
    assert abs(loss - rl) <= 1e-10 * rl                       # the adjoint of the device's own step sequence
    assert np.max(np.abs(g_nn - rg)) <= 1e-8 * np.max(np.abs(rg))
    assert np.max(np.abs(g_th - rt)) <= 1e-8 * np.max(np.abs(rt))
    ol, _, _, _ = o.supp_adaptive_loss_grad(nn, c["theta"], c["data"], c["tp"], arch, 0.01)
    assert abs(loss - ol) <= 1e-4 * ol                         # ... which is the oracle's adaptive solve to solver tolerance
    eng.close()
    arch = (2, (5, 4), ("tanh", "softplus"), "softplus")
    cp = make_cpep_case(12, (2, 4, 2))
    nn = o.glorot_params(arch, 2)
    pop = o.CPepPopulation(cp["tp"], cp["G"], cp["obs"], cp["age"], cp["t2dm"])
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(cp["tp"], cp["G"], cp["obs"], cp["age"], cp["t2dm"])
    eng.set_params(nn, cp["beta"])
    loss, g_nn, g_b = eng.loss_grad()
    steps = [eng.adaptive_steps(i) for i in range(12)]
    rl, rg, rb, _ = o.cpep_replay_loss_grad(nn, cp["beta"], pop, arch, [list(zip(t, dt)) for t, dt in steps])
    assert abs(loss - rl) <= 1e-10 * rl
    assert np.max(np.abs(g_nn - rg)) <= 1e-8 * np.max(np.abs(rg))
    assert np.max(np.abs(g_b - rb)) <= 1e-8 * np.max(np.abs(rb))
    assert abs(eng.forward()["loss"] - loss) <= 1e-14 * loss
    eng.close()


@pytest.mark.parametrize("model,arch,n_steps", [("cpep", (2, 6, 2), 30), ("cpep", (2, 6, 2), 0), ("supp", (4, 3, 5), 30),
                                                ("supp", (4, 3, 5), 0)], ids=["cpep", "cpep-adaptive", "supp", "supp-adaptive"])
def test_fallback_kernel_agrees_with_the_tuned_kernels(model, arch, n_steps):
    """Two implementations of the same mathematics (registers + scalar weights + stage checkpoints vs LDS columns + a
    tape + re-run stages) on the reference's own network shapes; option "force_fallback" sends a tuned shape to the
    fallback kernel."""
    from cude.engine import Engine
    N = 100
    c = make_cpep_case(N, arch) if model == "cpep" else make_supp_case(N, arch)
    out = []
    for force in (0, 1):
        eng = Engine(model, arch, n_steps=n_steps) if model == "cpep" else Engine("supp", arch, n_steps=n_steps, lam=0.01)
        assert not eng.fallback_kernel
        eng.set_option("force_fallback", force)
        eng.set_network([arch[1]] * arch[2], ["tanh"] * arch[2] + ["softplus"])
        assert eng.fallback_kernel == bool(force) and eng.P == len(c["nn"])
        if model == "cpep":
            eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
            eng.set_params(c["nn"], c["beta"])
        else:
            eng.set_population_supp(c["tp"], c["data"])
            eng.set_params(c["nn"], c["theta"])
        out.append(eng.loss_grad() + (eng.forward(want_sse=True)["sse"],))
        eng.close()
    (l0, g0, c0, s0), (l1, g1, c1, s1) = out
    tol = 1e-11 if n_steps else 1e-6          # (adaptive: step sizes carry the controller's rounding sensitivity)
    assert abs(l0 - l1) <= tol * abs(l0)
    assert np.max(np.abs(g0 - g1)) <= 100 * tol * np.max(np.abs(g0))
    assert np.max(np.abs(c0 - c1)) <= 100 * tol * np.max(np.abs(c0))
    assert np.max(np.abs(s0 - s1)) <= 10 * tol * np.max(np.abs(s0))


def test_wide_equal_width_network_falls_back_by_itself_and_trains():
    """cude_create with a width no tuned kernel is compiled for (chain(10, 2, tanh), the other docstring form): the
    fallback kernel by itself; multi-set evaluation and the restart trainer run on it."""
    import cude_oracle as o
    from cude.engine import Engine
    N, S = 30, 10
    c = make_supp_case(N, (4, 3, 5))
    eng = Engine("supp", (4, 10, 2), n_steps=S, lam=0.01)
    assert eng.fallback_kernel and eng.P == o.n_params((4, 10, 2))
    eng.set_population_supp(c["tp"], c["data"])
    nn = o.glorot_params((4, 10, 2), 4)
    eng.set_params(nn, c["theta"])
    loss, g_nn, g_th = eng.loss_grad()
    ref, rg, rt, _ = o.supp_loss_grad_torch(nn, c["theta"], c["data"], c["tp"], (4, 10, 2), S, 0.01)
    assert abs(loss - ref) <= 1e-10 * ref and np.max(np.abs(g_nn - rg)) <= 1e-9 * np.max(np.abs(rg))
    rng = np.random.default_rng(1)
    nn_sets = nn[None, :] * (1.0 + 0.05 * rng.standard_normal((3, nn.size)))
    th_sets = c["theta"][None, :] + 0.1 * rng.standard_normal((3, N))
    losses, g_nn_sets, g_th_sets = eng.multistart_loss_grad(nn_sets, th_sets)
    for k in range(3):
        eng.set_params(nn_sets[k], th_sets[k])
        lk, gk, tk = eng.loss_grad()
        assert losses[k] == lk and np.array_equal(g_nn_sets[k], gk) and np.array_equal(g_th_sets[k], tk)
    assert np.allclose(eng.multistart_forward(nn_sets, th_sets), losses, rtol=1e-13)
    nn_o, th_o, obj = eng.train_restarts(nn_sets, th_sets, 15, 1e-2, 5)
    assert np.all(obj < losses) and nn_o.shape == nn_sets.shape
    eng.close()


def test_api_chain_in_its_general_form():
    """`chain([10, 20, 30], [tanh, relu, softplus]; input_dims = 4)` through the reference-facing layer, in its default
    (adaptive) mode and in the fixed-step one."""
    import cude_oracle as o
    from cude import api
    net = api.chain([10, 20, 30], ["tanh", "relu", "softplus"], input_dims=4)
    assert net.general and net.n_params == 931 and net.mask is None
    assert api.chain([10, 20, 30], "relu", input_dims=4).n_params == 931          # (second method, :85-87)
    assert api.chain(10, 2, "tanh", input_dims=4).n_params == o.n_params((4, 10, 2))
    assert not api.chain(3, 5, "tanh", input_dims=4).general                       # tuned shapes stay tuned
    with pytest.raises(ValueError):
        api.chain([10, 20], ["tanh"], input_dims=4)
    c = make_supp_case(12, (4, 3, 5))
    prob = api.SuppressionProblem(net)
    rng = np.random.default_rng(0)
    p = api.ComponentArray(theta=c["theta"], neural=api.init_params(net, rng))
    val = api.suppression_loss(p, (prob, c["data"], c["tp"], 0.01))
    fixed = api.suppression_loss(p, (prob, c["data"], c["tp"], 0.01), n_steps=30)
    ref = o.supp_loss(np, p.neural, p.theta, c["data"], c["tp"], DOC_EXAMPLE, 30, 0.01)[0]
    assert abs(fixed - ref) <= 1e-10 * ref and abs(val - fixed) <= 1e-2 * fixed and val != fixed
    v2, g = api.suppression_loss_and_gradient(p, (prob, c["data"], c["tp"], 0.01))
    assert abs(v2 - val) <= 1e-13 * val and g.neural.shape == (931,) and np.all(np.isfinite(g.neural))
    api.clear_cache()


def test_limits_are_reported_not_crashed_into():
    from cude.engine import CudeError, Engine
    with pytest.raises(CudeError) as e:
        Engine("supp", (4, (200, 200), ("tanh", "tanh"), "softplus"))
    assert e.value.status == -4 and "LDS" in str(e.value)
    with pytest.raises(CudeError) as e:
        Engine("supp", (4, (3,) * 9, ("tanh",) * 9, "softplus"))
    assert e.value.status == -4
    eng = Engine("supp", (4, 3, 5))
    with pytest.raises(ValueError):
        eng.set_network([3, 3], ["tanh", "tanh"])             # one activation per hidden layer AND the output layer's
    c = make_supp_case(8, (4, 3, 5))
    eng.set_population_supp(c["tp"], c["data"])
    with pytest.raises(CudeError) as e:
        eng.set_network([3, 3], ["tanh", "tanh", "softplus"])  # the network is part of the population's buffers
    assert e.value.status == -3
    eng.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_general_networks_against_the_oracle(seed):
    """Shapes nobody wrote a case for: 1 ... 8 hidden layers of width 1 ... 12 (a layer of ONE unit, layers wider and
    narrower than the inputs), a random activation function per layer and at the output, either model."""
    import cude_oracle as o
    from cude.engine import Engine
    rng = np.random.default_rng(100 + seed)
    depth = int(rng.integers(1, 9)) if seed else 8
    widths = tuple(int(w) for w in rng.integers(1, 13, depth))
    names = ("tanh", "relu", "sigmoid", "softplus", "identity")
    acts = tuple(names[int(k)] for k in rng.integers(0, 5, depth))
    out_act = names[int(rng.integers(0, 5))] if seed % 2 else "softplus"
    supp = seed % 2 == 0
    S, N = 8, 66
    if supp:
        arch = (4, widths, acts, out_act)
        c = make_supp_case(N, (4, 3, 5))
        nn = 0.7 * o.glorot_params(arch, seed)
        eng = Engine("supp", arch, n_steps=S, lam=0.003)
        eng.set_population_supp(c["tp"], c["data"])
        eng.set_params(nn, c["theta"])
        ref, rg, rc, _ = o.supp_loss_grad_torch(nn, c["theta"], c["data"], c["tp"], arch, S, 0.003)
    else:
        arch = (2, widths, acts, out_act)
        c = make_cpep_case(N, (2, 4, 2))
        nn = 0.7 * o.glorot_params(arch, seed)
        pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng = Engine("cpep", arch, n_steps=S, n_state=2)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(nn, c["beta"])
        ref, rg, rc, _ = o.cpep_loss_grad_torch(nn, c["beta"], pop, arch, S, 2)
    assert eng.fallback_kernel and eng.P == nn.size
    loss, g_nn, g_c = eng.loss_grad()
    eng.close()
    assert np.isfinite(ref) and abs(loss - ref) <= 1e-10 * abs(ref), (arch, loss, ref)
    assert np.max(np.abs(g_nn - rg)) <= 1e-9 * max(np.max(np.abs(rg)), 1e-300), arch
    assert np.max(np.abs(g_c - rc)) <= 1e-9 * max(np.max(np.abs(rc)), 1e-300), arch
