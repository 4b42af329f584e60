"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (fp64): loss rtol 1e-10, gradients 1e-9 relative to the gradient's max-norm.  The
north-star bar is rtol <= 1e-6; the discrete adjoint and the oracle's forward-mode duals are two
different algorithms for the same derivative, so agreement is expected near round-off.
"""
import numpy as np
import pytest
import torch  # noqa: F401  (imported first so PyTorch and libcude_hip share one HIP runtime)

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-10
GRAD_RTOL = 1e-9


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("arch,n_state,N", [((2, 6, 2), 3, 1000), ((2, 6, 2), 2, 333), ((2, 4, 2), 2, 64),
                                             ((2, 4, 2), 3, 65), ((3, 4, 2), 2, 200), ((2, 8, 2), 2, 130),
                                             ((2, 4, 3), 2, 77), ((2, 6, 2), 3, 1),
                                             # further chain(width, depth, tanh) shapes compiled since round 2
                                             ((2, 3, 2), 2, 70), ((2, 5, 2), 3, 129), ((2, 7, 2), 2, 66),
                                             ((3, 6, 2), 3, 90), ((2, 4, 1), 2, 100), ((2, 6, 1), 3, 64),
                                             ((2, 6, 3), 2, 65)])
def test_cpep_loss_and_gradient(arch, n_state, N):
    import c_oracle as co
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    ref = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], c["n_steps"], n_state,
                  want_grad=True, want_traj=True, covariate=(arch[0] == 3))
    eng = Engine("cpep", arch, n_steps=c["n_steps"], n_state=n_state)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    f = eng.forward(want_sse=True, want_traj=True)
    assert abs(f["loss"] - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"])
    assert _rel(f["sse"], ref["sse"]) < 1e-10
    assert _rel(f["traj"], ref["traj"].transpose(2, 1, 0)) < 1e-11
    loss, g_nn, g_cond = eng.loss_grad()
    assert abs(loss - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"])
    assert _rel(g_nn, ref["g_nn"]) < GRAD_RTOL
    assert _rel(g_cond, ref["g_beta"]) < GRAD_RTOL
    assert eng.n_failed() == 0
    eng.close()


@pytest.mark.parametrize("arch,N,lam", [((4, 3, 5), 500, 0.01), ((4, 3, 5), 37, 0.0), ((4, 3, 2), 64, 0.1),
                                         ((4, 6, 2), 129, 0.0), ((4, 5, 2), 70, 0.01), ((4, 3, 3), 65, 0.0),
                                         ((4, 8, 2), 64, 0.1)])
def test_supp_loss_and_gradient(arch, N, lam):
    import c_oracle as co
    from cude.engine import Engine
    c = make_supp_case(N, arch)
    ref = co.supp(c["tp"], c["data"], arch, c["nn"], c["theta"], lam, c["n_steps"], want_grad=True, want_traj=True)
    eng = Engine("supp", arch, n_steps=c["n_steps"], lam=lam)
    eng.set_population_supp(c["tp"], c["data"])
    eng.set_params(c["nn"], c["theta"])
    f = eng.forward(want_sse=True, want_traj=True)
    assert abs(f["loss"] - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"])
    assert _rel(f["sse"], ref["sse"]) < 1e-10
    assert _rel(f["traj"], ref["traj"]) < 1e-11
    loss, g_nn, g_cond = eng.loss_grad()
    assert abs(loss - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"])
    assert _rel(g_nn, ref["g_nn"]) < GRAD_RTOL
    assert _rel(g_cond, ref["g_theta"]) < GRAD_RTOL
    eng.close()


CPEP_SHAPES = [(2, 4, 2), (2, 6, 2), (3, 4, 2), (2, 8, 2), (2, 4, 3), (2, 3, 2), (2, 5, 2), (2, 7, 2), (3, 6, 2), (2, 4, 1),
               (2, 6, 1), (2, 6, 3), (2, 8, 1), (2, 8, 3), (3, 8, 2), (2, 3, 1), (2, 5, 1), (2, 7, 1), (2, 3, 3), (2, 5, 3),
               (2, 7, 3), (3, 4, 1), (3, 6, 1), (3, 4, 3)]
SUPP_SHAPES = [(3, 5), (3, 2), (4, 2), (6, 2), (5, 2), (3, 3), (8, 2), (3, 4), (4, 3), (4, 4), (5, 3), (6, 3), (3, 1), (4, 1),
               (6, 1), (8, 1)]


def test_every_compiled_shape_against_the_oracle(monkeypatch):
    """`chain(width, depth, tanh; input_dims)` (src/neural-network.jl:105-107) for every (inputs, width, depth) the
    library is compiled for (the CUDE_*_SHAPES lists of csrc/): loss and both gradients against the CPU oracle on the
    one-lane path and on the time-split path, forward loss in adaptive mode against the oracle's adaptive solve."""
    import c_oracle as co
    from cude.engine import Engine
    for k, arch in enumerate(CPEP_SHAPES):
        N, n_state = 66 + k, 2 + (k % 2)
        c = make_cpep_case(N, arch)
        # the forward-mode oracle carries at most 128 partials: the two largest shapes go to the reverse-mode one
        method = "forward" if c["nn"].size + 1 <= 128 else "reverse"
        ref = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], 30, n_state,
                      covariate=(arch[0] == 3), method=method)
        for path in ("1", "2:3"):
            monkeypatch.setenv("CUDE_CPEP_PATH", path)
            eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
            eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
            eng.set_params(c["nn"], c["beta"])
            loss, g_nn, g_cond = eng.loss_grad()
            eng.close()
            assert abs(loss - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"]), (arch, path)
            assert _rel(g_nn, ref["g_nn"]) < GRAD_RTOL and _rel(g_cond, ref["g_beta"]) < GRAD_RTOL, (arch, path)
        monkeypatch.delenv("CUDE_CPEP_PATH")
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        out = eng.forward(want_sse=True)
        eng.close()
        ad = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], np.exp(c["beta"]), c["tp"],
                              covariate=(arch[0] == 3))
        sse = np.sum((ad - c["obs"]) ** 2, axis=1)
        assert abs(out["loss"] - sse.mean()) <= 1e-4 * sse.mean(), arch     # adaptive solves are compared loosely
    for k, (w, d) in enumerate(SUPP_SHAPES):
        arch = (4, w, d)
        c = make_supp_case(40 + k, arch)
        ref = co.supp(c["tp"], c["data"], arch, c["nn"], c["theta"], 0.01, 30)
        eng = Engine("supp", arch, n_steps=30, lam=0.01)
        eng.set_population_supp(c["tp"], c["data"])
        eng.set_params(c["nn"], c["theta"])
        loss, g_nn, g_cond = eng.loss_grad()
        eng.close()
        assert abs(loss - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"]), arch
        assert _rel(g_nn, ref["g_nn"]) < GRAD_RTOL and _rel(g_cond, ref["g_theta"]) < GRAD_RTOL, arch


def test_baseline_config0_suppression_20_subjects():
    """BASELINE configs[0]: the reference's own CPU-sized case -- suppression/suppression.jl's synthetic cUDE with 20
    subjects from its data generator (`generate_data`, restated in the oracle module: six groups of suppression strength,
    lsup! from u0 = (10, 0, 0), 8 observations on [0, 30], 10 % multiplicative noise), 4 -> 3x5 -> 1 network, lambda from
    its regularisation grid -- loss, per-subject SSE, trajectories and both gradients against the CPU oracle."""
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    tp = np.linspace(0.0, 30.0, 8)
    data, sup = o.generate_suppression_data([0.5, 2.5, 5.0, 7.5, 10.0, 12.5], [4, 4, 3, 3, 3, 3], tp)
    assert data.shape == (3, 8, 20) and np.all(np.diff(sup[[0, 4, 8, 11, 14, 17]]) > 0)
    arch = (4, 3, 5)
    nn = o.glorot_params(arch, 27052023)
    theta = np.random.default_rng(27052023).standard_normal(20)
    for lam in (0.0, 0.01):
        ref = co.supp(tp, data, arch, nn, theta, lam, 30, want_grad=True, want_traj=True)
        eng = Engine("supp", arch, n_steps=30, lam=lam)
        eng.set_population_supp(tp, data)
        eng.set_params(nn, theta)
        f = eng.forward(want_sse=True, want_traj=True)
        loss, g_nn, g_cond = eng.loss_grad()
        eng.close()
        assert abs(f["loss"] - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"]) and loss == f["loss"]
        assert _rel(f["sse"], ref["sse"]) < 1e-10 and _rel(f["traj"], ref["traj"]) < 1e-11
        assert _rel(g_nn, ref["g_nn"]) < GRAD_RTOL and _rel(g_cond, ref["g_theta"]) < GRAD_RTOL


def test_adam_steps_match_oracle():
    """10 fused device Adam steps vs oracle gradient + restated Optimisers.Adam on the host."""
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    arch, N = (2, 6, 2), 300
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch, n_steps=30, n_state=3)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    eng.adam_init(1e-2)
    nn, beta = c["nn"].copy(), c["beta"].copy()
    m_n, v_n, m_b, v_b = np.zeros_like(nn), np.zeros_like(nn), np.zeros_like(beta), np.zeros_like(beta)
    for t in range(1, 11):
        ref = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, nn, beta, 30, 3)
        loss = eng.adam_step()
        assert abs(loss - ref["loss"]) <= 1e-9 * abs(ref["loss"])
        nn, m_n, v_n = o.adam_update(nn, ref["g_nn"], m_n, v_n, t, 1e-2)
        beta, m_b, v_b = o.adam_update(beta, ref["g_beta"], m_b, v_b, t, 1e-2)
    nn_d, beta_d = eng.get_params()
    assert np.max(np.abs(nn_d - nn)) < 1e-8
    assert np.max(np.abs(beta_d - beta)) < 1e-8
    eng.close()


def test_failure_convention():
    """A non-finite trajectory gives +Inf with status 0 (parameter-estimation.jl:61-64,134-136)."""
    from cude.engine import Engine
    arch, N = (2, 4, 2), 70
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    beta = c["beta"].copy()
    beta[5] = np.nan
    eng.set_params(c["nn"], beta)
    out = eng.forward()
    assert out["loss"] == np.inf
    assert eng.n_failed() == 1
    loss, _, _ = eng.loss_grad()
    assert loss == np.inf
    eng.close()


@pytest.mark.parametrize("model", ["cpep", "supp"])
def test_multistart_forward_matches_individual_evaluations(model):
    """cude_multistart_forward (screening loop of train / fit_suppression_model) == K separate forward calls
    == oracle, including a candidate that fails (+Inf) without disturbing the others."""
    import c_oracle as co
    from cude.engine import Engine
    import cude_oracle as o
    rng = np.random.default_rng(11)
    K = 37
    if model == "cpep":
        arch, N = (2, 4, 2), 57
        c = make_cpep_case(N, arch)
        eng = Engine("cpep", arch)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        ref = lambda nn, cd: co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, nn, cd, 30, 2,
                                     want_grad=False)["loss"]
    else:
        arch, N = (4, 3, 5), 70
        c = make_supp_case(N, arch)
        eng = Engine("supp", arch, lam=0.01)
        eng.set_population_supp(c["tp"], c["data"])
        ref = lambda nn, cd: co.supp(c["tp"], c["data"], arch, nn, cd, 0.01, 30, want_grad=False)["loss"]
    nn_sets = np.stack([o.glorot_params(arch, 100 + k) for k in range(K)])
    cond_sets = rng.uniform(-2, 0, (K, N))
    cond_sets[5, 3] = np.nan
    losses = eng.multistart_forward(nn_sets, cond_sets)
    assert losses[5] == np.inf
    for k in (0, 1, 17, 36):
        r = ref(nn_sets[k], cond_sets[k])
        assert abs(losses[k] - r) <= 1e-10 * abs(r)
        eng.set_params(nn_sets[k], cond_sets[k])
        assert abs(eng.forward()["loss"] - losses[k]) <= 1e-13 * abs(r)
    eng.close()


def test_full_size_properties_1e5_subjects():
    """BASELINE configs[2] size (1e5 subjects): properties that need no oracle run at that size --
    (i) shard additivity: loss/gradient of the whole population = N-weighted combination of two shards,
    (ii) bitwise determinism of repeated evaluations, (iii) directional derivative vs central differences,
    (iv) an oracle spot check on a random subset of subjects (per-subject SSE and dL/dbeta)."""
    import c_oracle as co
    from cude.engine import Engine
    arch, N = (2, 6, 2), 100000
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch, n_steps=30, n_state=3)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    loss, g_nn, g_cond = eng.loss_grad()
    loss2, g_nn2, g_cond2 = eng.loss_grad()
    assert loss == loss2 and np.array_equal(g_nn, g_nn2) and np.array_equal(g_cond, g_cond2)      # (ii)
    sse = eng.forward(want_sse=True)["sse"]
    assert abs(sse.sum() / N - loss) < 1e-12 * loss
    # (i) two shards of unequal size
    cut = 37123
    parts = []
    for lo, hi in ((0, cut), (cut, N)):
        e2 = Engine("cpep", arch, n_steps=30, n_state=3)
        e2.set_population_cpep(c["tp"], c["G"][lo:hi], c["obs"][lo:hi], c["age"][lo:hi], c["t2dm"][lo:hi])
        e2.set_params(c["nn"], c["beta"][lo:hi])
        l, gn, gc = e2.loss_grad()
        parts.append((hi - lo, l, gn, gc))
        e2.close()
    w = np.array([p[0] for p in parts]) / N
    assert abs(w[0] * parts[0][1] + w[1] * parts[1][1] - loss) < 1e-12 * loss
    assert np.max(np.abs(w[0] * parts[0][2] + w[1] * parts[1][2] - g_nn)) < 1e-11 * np.max(np.abs(g_nn))
    assert np.max(np.abs(np.concatenate([w[0] * parts[0][3], w[1] * parts[1][3]]) - g_cond)) < 1e-15
    # (iii) directional derivative
    rng = np.random.default_rng(0)
    d_nn, d_b = rng.standard_normal(g_nn.size), rng.standard_normal(N)
    eps = 1e-6
    eng.set_params(c["nn"] + eps * d_nn, c["beta"] + eps * d_b); lp = eng.forward()["loss"]
    eng.set_params(c["nn"] - eps * d_nn, c["beta"] - eps * d_b); lm = eng.forward()["loss"]
    assert abs((lp - lm) / (2 * eps) - (g_nn @ d_nn + g_cond @ d_b)) < 1e-7 * abs(g_nn @ d_nn + g_cond @ d_b)
    # (iv) oracle on 400 random subjects: per-subject SSE and dL/dbeta (which are local to a subject)
    idx = np.sort(rng.choice(N, 400, replace=False))
    ref = co.cpep(c["tp"], c["G"][idx], c["obs"][idx], c["age"][idx], c["t2dm"][idx], arch, c["nn"], c["beta"][idx],
                  30, 3)
    assert np.max(np.abs(sse[idx] - ref["sse"])) < 1e-11
    assert np.max(np.abs(g_cond[idx] * N - ref["g_beta"] * 400)) < 1e-9 * np.max(np.abs(ref["g_beta"] * 400))
    eng.close()


@pytest.mark.parametrize("n_steps,tp", [(30, [0.0, 30.0, 60.0, 90.0, 120.0]), (17, [0.0, 10.0, 45.0, 50.0, 120.0]),
                                         (60, [0.0, 15.0, 30.0, 45.0, 60.0, 75.0, 90.0, 120.0]), (5, [0.0, 120.0]),
                                         (1, [0.0, 60.0, 120.0]), (1200, [0.0, 60.0, 120.0])])
@pytest.mark.parametrize("arch", [(2, 4, 2), (2, 6, 2)])
def test_cpep_general_time_grids(n_steps, tp, arch):
    """Irregular observation times / step counts: observations inside, at the end of, and sharing a step; T=2;
    a single step holding every observation.  Width 6 runs the layer-1 exponent table: steps straddling glucose
    knots, runs of one step, pieces shorter than a step."""
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    N = 129
    rng = np.random.default_rng(4)
    tp = np.array(tp)
    age, t2 = rng.uniform(20, 79, N), rng.random(N) < 0.4
    G = 5.0 + np.abs(rng.standard_normal((N, tp.size))).cumsum(1)
    obs = 0.5 + rng.random((N, tp.size))
    nn, beta = o.glorot_params(arch, 9), rng.normal(-0.6, 0.5, N)
    ref = co.cpep(tp, G, obs, age, t2, arch, nn, beta, n_steps, 2, want_traj=True)
    eng = Engine("cpep", arch, n_steps=n_steps, n_state=2)
    eng.set_population_cpep(tp, G, obs, age, t2)
    eng.set_params(nn, beta)
    f = eng.forward(want_traj=True)
    assert np.max(np.abs(f["traj"] - ref["traj"].transpose(2, 1, 0))) < 1e-11 * np.max(np.abs(ref["traj"]))
    loss, g_nn, g_cond = eng.loss_grad()
    assert abs(loss - ref["loss"]) < 1e-10 * ref["loss"]
    assert np.max(np.abs(g_nn - ref["g_nn"])) < 1e-9 * np.max(np.abs(ref["g_nn"]))
    assert np.max(np.abs(g_cond - ref["g_beta"])) < 1e-9 * np.max(np.abs(ref["g_beta"]))
    eng.close()


@pytest.mark.parametrize("scale", [1.0, 6.0, 40.0])
def test_exponent_table_matches_direct_exponentials(scale, monkeypatch):
    """The layer-1 exponent recurrence (anchor x tabulated factor) against the same kernel with the table switched
    off, including saturated first-layer units (weights x40: |z| far beyond the tanh clamp) and n_state 2 / 3."""
    from cude.engine import Engine
    c = make_cpep_case(700, (2, 6, 2), nn_scale=1.0)
    nn = c["nn"].copy()
    nn[:18] *= scale                                   # first layer: W1 (12) and b1 (6)
    out = {}
    for mode in ("table", "direct"):
        if mode == "direct":
            monkeypatch.setenv("CUDE_NO_EXPTAB", "1")
        monkeypatch.setenv("CUDE_CPEP_PATH", "1")      # the fused one-lane kernel at this small size
        eng = Engine("cpep", (2, 6, 2), n_steps=30, n_state=3)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(nn, c["beta"])
        f = eng.forward(want_sse=True, want_traj=True)
        out[mode] = (f["traj"], f["sse"]) + eng.loss_grad()
        eng.close()
    t, d = out["table"], out["direct"]
    assert np.max(np.abs(t[0] - d[0])) <= 1e-11 * np.max(np.abs(d[0]))
    assert np.allclose(t[1], d[1], rtol=1e-10, atol=1e-14)
    assert abs(t[2] - d[2]) <= 1e-11 * abs(d[2])
    assert np.max(np.abs(t[3] - d[3])) <= 1e-9 * np.max(np.abs(d[3]))
    assert np.max(np.abs(t[4] - d[4])) <= 1e-9 * np.max(np.abs(d[4]))


@pytest.mark.parametrize("arch,n_state", [((2, 6, 2), 3), ((2, 4, 2), 2)])
def test_dense_output_matches_the_oracle_interpolant(arch, n_state):
    """cude_simulate: the reference's `simulate(...; timepoints = t0:0.1:tend)` (src/saem.jl:31-53).  1201 output
    times on [0, 120] against the oracle's fixed-step solve with its Tsit5 interpolant at the same times; at the
    observation times the dense output equals cude_forward's trajectory."""
    import cude_oracle as o
    from cude.engine import Engine
    c = make_cpep_case(70, arch)
    eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eng.set_params(c["nn"], c["beta"])
    dense = np.round(np.arange(0.0, 120.0 + 1e-9, 0.1), 10)
    got = eng.simulate(dense)
    assert got.shape == (n_state, dense.size, 70)
    at_obs = eng.simulate(c["tp"])
    assert np.array_equal(at_obs, eng.forward(want_traj=True)["traj"])
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    eb = np.exp(c["beta"])
    u0 = [pop.c0, (pop.k2 / pop.k1) * pop.c0] + ([pop.c0 * 0.0] if n_state == 3 else [])
    rhs = lambda t, u: o.cpep_rhs(np, pop, c["nn"], eb, arch, t, u, n_state)
    ref = o.solve_fixed(rhs, u0, list(dense), 30)
    for k in range(dense.size):
        for s in range(n_state):
            assert np.allclose(got[s, k], ref[k][s], rtol=1e-10, atol=1e-12), (k, s)
    with pytest.raises(Exception):
        eng.simulate([0.0, 130.0])                      # outside the span
    with pytest.raises(Exception):
        eng.simulate([10.0, 5.0])                       # decreasing
    eng.close()


def test_size_limits_and_missing_values():
    """Edges of the domain: the maximum number of observation times (32) against the oracle, T = 33 and an empty
    population rejected with a status, a one-subject population, and missing values (NaN in a subject's glucose /
    c-peptide series -- the reference drops such subjects during data preparation, c-peptide/00-prepare-data.jl)
    failing only that subject (loss = Inf, the others' SSE finite and unchanged)."""
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    from cude._lib import CudeError
    rng = np.random.default_rng(12)
    arch, N, T = (2, 6, 2), 70, 32
    tp = np.cumsum(rng.uniform(1.0, 9.0, T))
    tp -= tp[0]
    age, t2 = rng.uniform(20, 79, N), rng.random(N) < 0.4
    G = 5.0 + np.abs(rng.standard_normal((N, T))).cumsum(1) * 0.3
    obs = 0.5 + rng.random((N, T))
    nn, beta = o.glorot_params(arch, 3), rng.normal(-0.6, 0.5, N)
    ref = co.cpep(tp, G, obs, age, t2, arch, nn, beta, 64, 2)
    eng = Engine("cpep", arch, n_steps=64, n_state=2)
    eng.set_population_cpep(tp, G, obs, age, t2)
    eng.set_params(nn, beta)
    loss, g_nn, g_cond = eng.loss_grad()
    assert abs(loss - ref["loss"]) < 1e-10 * ref["loss"]
    assert np.max(np.abs(g_nn - ref["g_nn"])) < 1e-9 * np.max(np.abs(ref["g_nn"]))
    assert np.max(np.abs(g_cond - ref["g_beta"])) < 1e-9 * np.max(np.abs(ref["g_beta"]))
    clean = eng.forward(want_sse=True)["sse"]
    # missing values
    G2, obs2 = G.copy(), obs.copy()
    G2[7, 5] = np.nan
    obs2[30, 31] = np.nan
    eng.set_population_cpep(tp, G2, obs2, age, t2)
    eng.set_params(nn, beta)
    out = eng.forward(want_sse=True)
    bad = np.zeros(N, bool)
    bad[[7, 30]] = True
    assert out["loss"] == np.inf and eng.n_failed() == 2
    assert not np.any(np.isfinite(out["sse"][bad])) and np.array_equal(out["sse"][~bad], clean[~bad])
    # one subject
    eng.set_population_cpep(tp, G[:1], obs[:1], age[:1], t2[:1])
    eng.set_params(nn, beta[:1])
    one = eng.forward(want_sse=True)
    assert abs(one["sse"][0] - clean[0]) <= 1e-12 * clean[0] and abs(one["loss"] - clean[0]) <= 1e-12 * clean[0]
    # limits
    with pytest.raises(CudeError):
        eng.set_population_cpep(np.arange(33.0), np.ones((2, 33)), np.ones((2, 33)), np.ones(2) * 40, np.zeros(2))
    with pytest.raises((CudeError, ValueError)):
        eng.set_population_cpep(tp, G[:0], obs[:0], age[:0], t2[:0])
    eng.close()
    with pytest.raises(CudeError):
        Engine("cpep_sym", (1, 4, 2))                  # the symbolic model has no network
    with pytest.raises(CudeError):
        Engine("cpep", (2, 6, 2), n_steps=-1)
    with pytest.raises(CudeError):
        Engine("cpep", (2, 6, 2), n_steps=0, n_state=3)     # n_steps = 0 is the adaptive mode: 2-state model only


def test_argument_errors_are_statuses():
    from cude.engine import Engine
    from cude._lib import CudeError
    wide = Engine("cpep", (2, 9, 2))                   # no tuned kernel for width 9: the fallback kernel (round 5; an error before)
    assert wide.fallback_kernel
    wide.close()
    with pytest.raises(CudeError):
        Engine("cpep", (2, 200, 4))                    # ... which refuses what does not fit a workgroup's LDS -> CUDE_ERR_UNSUPPORTED
    with pytest.raises(CudeError):
        Engine("cpep", (4, 4, 2))                      # four network inputs: not a c-peptide network
    eng = Engine("cpep", (2, 4, 2))
    with pytest.raises(CudeError):
        eng.forward()                                  # population not set
    with pytest.raises(CudeError):
        eng.set_population_cpep([0.0, 30.0, 30.0], np.ones((3, 3)), np.ones((3, 3)), np.ones(3), np.zeros(3))
    eng.set_population_cpep([0.0, 30.0, 60.0], np.ones((3, 3)) * 5, np.ones((3, 3)), np.ones(3) * 40, np.zeros(3))
    with pytest.raises(CudeError):
        eng.forward()                                  # parameters not set
    with pytest.raises(CudeError):
        eng.adam_step()                                # adam_init missing
    eng.close()
