"""The CPU oracle checked against itself three ways (numpy values, torch reverse-mode autograd, C forward-mode
duals), against finite differences, and for the properties of the discretisation it restates."""
import math

import numpy as np
import pytest

from conftest import make_cpep_case, make_supp_case


def test_tsit5_tableau_order_conditions():
    import cude_oracle as o
    A = np.zeros((7, 7))
    for i in range(7):
        A[i, :len(o.A[i])] = o.A[i]
    c = np.array(o.C)
    b = A[6].copy()
    assert np.allclose(A.sum(1), c, atol=1e-15)
    Ac = A @ c
    conds = [(b.sum(), 1), (b @ c, 1 / 2), (b @ c**2, 1 / 3), (b @ Ac, 1 / 6), (b @ c**3, 1 / 4), (b @ (c * Ac), 1 / 8),
             (b @ (A @ c**2), 1 / 12), (b @ (A @ Ac), 1 / 24), (b @ c**4, 1 / 5), (20 * (b @ (A @ c**3)), 1),
             (60 * (b @ (A @ (A @ c**2))), 1), (120 * (b @ (A @ (A @ Ac))), 1)]
    for got, want in conds:
        assert abs(got - want) < 5e-15
    # dense output: b(1) = b, b(0) = 0, order-4 conditions at interior points
    assert np.allclose([sum(r) for r in o.R], b, atol=5e-15)
    for th in (0.3, 0.7):
        w = np.array([((o.R[i][3] * th + o.R[i][2]) * th + o.R[i][1]) * th * th + o.R[i][0] * th for i in range(7)])
        assert abs(w.sum() - th) < 1e-14 and abs(w @ c - th**2 / 2) < 1e-14 and abs(w @ c**2 - th**3 / 3) < 1e-14
    assert o.interp_weights(1.0)[:6] == o.A[6] and o.interp_weights(1.0)[6] == 0.0


def test_observation_location():
    import cude_oracle as o
    loc = o.locate_observations([0, 30, 60, 90, 120], 30)
    assert [n for n, _ in loc] == [0, 7, 14, 22, 29]
    assert loc[0][1] == 0.0 and abs(loc[1][1] - 0.5) < 1e-12 and abs(loc[2][1] - 1.0) < 1e-12


@pytest.mark.parametrize("arch,n_state", [((2, 6, 2), 3), ((2, 4, 2), 2), ((3, 4, 2), 2)])
def test_cpep_three_implementations_agree(arch, n_state):
    import cude_oracle as o
    import c_oracle as co
    c = make_cpep_case(12, arch)
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], covariate=(arch[0] == 3))
    L_np, sse_np = o.cpep_loss(np, c["nn"], c["beta"], pop, arch, 30, n_state)
    L_t, gn_t, gb_t, _ = o.cpep_loss_grad_torch(c["nn"], c["beta"], pop, arch, 30, n_state)
    r = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], 30, n_state,
                covariate=(arch[0] == 3))
    assert abs(L_np - r["loss"]) < 1e-13 * abs(L_np) and abs(L_t - r["loss"]) < 1e-13 * abs(L_np)
    assert np.max(np.abs(sse_np - r["sse"])) < 1e-13
    assert np.max(np.abs(gn_t - r["g_nn"])) < 1e-12 * np.max(np.abs(gn_t))
    assert np.max(np.abs(gb_t - r["g_beta"])) < 1e-12 * np.max(np.abs(gb_t))


def test_supp_three_implementations_agree():
    import cude_oracle as o
    import c_oracle as co
    c = make_supp_case(9)
    L_np, sse_np = o.supp_loss(np, c["nn"], c["theta"], c["data"], c["tp"], c["arch"], 30, 0.01)
    L_t, gn_t, gt_t, _ = o.supp_loss_grad_torch(c["nn"], c["theta"], c["data"], c["tp"], c["arch"], 30, 0.01)
    r = co.supp(c["tp"], c["data"], c["arch"], c["nn"], c["theta"], 0.01, 30)
    assert abs(L_np - r["loss"]) < 1e-12 * abs(L_np) and abs(L_t - r["loss"]) < 1e-12 * abs(L_np)
    assert np.max(np.abs(gn_t - r["g_nn"])) < 1e-11 * np.max(np.abs(gn_t))
    assert np.max(np.abs(gt_t - r["g_theta"])) < 1e-11 * np.max(np.abs(gt_t))


def test_gradient_matches_central_differences():
    import c_oracle as co
    arch = (2, 6, 2)
    c = make_cpep_case(20, arch)
    args = (c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch)
    r = co.cpep(*args, c["nn"], c["beta"], 30, 3)
    f = lambda nn, b: co.cpep(*args, nn, b, 30, 3, want_grad=False)["loss"]
    eps = 1e-6
    for q in (0, 7, 31, 60, 66):
        d = np.zeros_like(c["nn"]); d[q] = eps
        fd = (f(c["nn"] + d, c["beta"]) - f(c["nn"] - d, c["beta"])) / (2 * eps)
        assert abs(fd - r["g_nn"][q]) < 1e-7 * max(1.0, abs(fd))
    d = np.zeros(20); d[4] = eps
    fd = (f(c["nn"], c["beta"] + d) - f(c["nn"], c["beta"] - d)) / (2 * eps)
    assert abs(fd - r["g_beta"][4]) < 1e-8


def test_fixed_step_converges_and_brackets_adaptive():
    """S=30 is within ~1e-5 of the converged solution on the smooth synthetic forcing; the restated adaptive
    controller (reference tolerances) differs from it at the 1e-3 level, as DESIGN.md states."""
    import cude_oracle as o
    import c_oracle as co
    arch = (2, 4, 2)
    c = make_cpep_case(6, arch)
    args = (c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"])
    l30 = co.cpep(*args, 30, 2, want_grad=False, want_traj=True)
    l480 = co.cpep(*args, 480, 2, want_grad=False, want_traj=True)
    assert np.max(np.abs(l30["traj"] - l480["traj"])) < 2e-4 * np.max(np.abs(l480["traj"]))
    pop = o.CPepPopulation(c["tp"], c["G"][:1], c["obs"][:1], c["age"][:1], c["t2dm"][:1])
    eb = np.exp(c["beta"][:1])
    rhs = lambda t, u: [float(v[0]) if np.ndim(v) else float(v)
                        for v in o.cpep_rhs(np, pop, c["nn"], eb, arch, t, [np.array([u[0]]), np.array([u[1]])], 2)]
    sol = o.solve_adaptive(rhs, [pop.c0[0], pop.k2[0] / pop.k1[0] * pop.c0[0]], pop.timepoints)
    ad = np.array([s[0] for s in sol])
    assert np.max(np.abs(ad - l480["traj"][0, :, 0])) < 5e-3 * np.max(np.abs(ad))


def _mlp_matrix_form(x, p, arch):
    """Independent restatement of the SimpleChains network for this test only: weight matrices rebuilt with
    reshape(order="F") and applied with matmul (the oracle indexes the flat vector element by element)."""
    nin, w, d = arch
    off, h = 0, np.asarray(x, dtype=np.float64)
    for layer in range(d):
        fan = nin if layer == 0 else w
        W = p[off:off + w * fan].reshape((w, fan), order="F")
        b = p[off + w * fan:off + w * fan + w]
        off += w * fan + w
        h = np.tanh(W @ h + b)
    return float(np.log1p(np.exp(p[off:off + w] @ h + p[off + w])))


def test_against_scipy_integrators():
    """The oracle's own Tsit5 (tableau, FSAL, dense output) against an integrator it shares no code with: scipy's
    DOP853 at rtol 1e-12, restarted at the glucose knots, with the right-hand sides written independently here
    (matrix-form MLP, np.interp glucose).  The fixed-step oracle at S = 960 must agree to ~1e-9."""
    from scipy.integrate import solve_ivp
    import cude_oracle as o
    import c_oracle as co
    arch = (2, 6, 2)
    c = make_cpep_case(3, arch)
    fixed = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], 960, 2, want_grad=False,
                    want_traj=True)["traj"]
    k0, k1, k2 = o.van_cauter_parameters(c["age"], c["t2dm"])
    for i in range(3):
        c0, eb = c["obs"][i, 0], np.exp(c["beta"][i])
        base = _mlp_matrix_form([0.0, eb], c["nn"], arch)

        def rhs(t, u):
            dG = np.interp(t, c["tp"], c["G"][i]) - c["G"][i, 0]
            prod = _mlp_matrix_form([dG, eb], c["nn"], arch) - base
            return [-(k0[i] + k2[i]) * u[0] + k1[i] * u[1] + k0[i] * c0 + prod, -k1[i] * u[1] + k2[i] * u[0]]
        u = np.array([c0, k2[i] / k1[i] * c0])
        for j in range(1, len(c["tp"])):
            sol = solve_ivp(rhs, (c["tp"][j - 1], c["tp"][j]), u, method="DOP853", rtol=1e-12, atol=1e-14)
            u = sol.y[:, -1]
            assert abs(u[0] - fixed[i, j, 0]) < 2e-9 * abs(u[0]) and abs(u[1] - fixed[i, j, 1]) < 2e-9 * abs(u[1])
    # suppression model (state-dependent network input), smooth right-hand side: one integration over [0, 30]
    s = make_supp_case(2)
    fx = co.supp(s["tp"], s["data"], s["arch"], s["nn"], s["theta"], 0.0, 960, want_grad=False, want_traj=True)["traj"]
    for i in range(2):
        et = np.exp(s["theta"][i])

        def rhs(t, u):
            uh = _mlp_matrix_form([u[0], u[1], u[2], et], s["nn"], s["arch"])
            return [-0.4 * u[0], 0.4 * u[0] - uh, uh - 0.3 * u[2]]
        sol = solve_ivp(rhs, (s["tp"][0], s["tp"][-1]), s["data"][:, 0, i], method="DOP853", rtol=1e-12, atol=1e-14,
                        t_eval=s["tp"])
        assert np.max(np.abs(sol.y - fx[:, :, i])) < 2e-8 * np.max(np.abs(fx[:, :, i]))


def test_scalar_rhs_is_the_array_rhs():
    """cpep_rhs_scalar (plain floats, fed to solve_adaptive) against cpep_rhs, all three production terms."""
    import cude_oracle as o
    rng = np.random.default_rng(4)
    for arch, covariate in (((2, 4, 2), False), ((3, 4, 2), True), (o.SYMBOLIC, False)):
        c = make_cpep_case(6, arch if arch[1] else (2, 4, 2))
        pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], covariate=covariate)
        nn = o.glorot_params(arch, 3) if arch[1] else np.array([1.78])
        cond = np.exp(rng.normal(0.0, 1.0, pop.N))
        for t in (0.0, 7.3, 30.0, 61.9, 120.0):
            u = [rng.random(pop.N) + 0.5, rng.random(pop.N) + 0.5]
            ref = o.cpep_rhs(np, pop, nn, cond, arch, t, u, 2)
            for i in range(pop.N):
                got = o.cpep_rhs_scalar(pop, i, nn, cond[i], arch)(t, [float(u[0][i]), float(u[1][i])])
                assert abs(got[0] - ref[0][i]) < 1e-14 and abs(got[1] - ref[1][i]) < 1e-14


def test_c_adaptive_mode_is_the_python_adaptive_mode():
    """cude_oracle_cpep_adaptive (C, used to scan parameters finely) takes the same accepted steps as
    cude_oracle.solve_adaptive (the restatement the known answers pin): all three production terms, dense output."""
    import cude_oracle as o
    import c_oracle as co
    rng = np.random.default_rng(5)
    times = np.round(np.arange(0.0, 120.01, 2.5), 10)
    for arch, covariate in (((2, 4, 2), False), ((3, 4, 2), True), (o.SYMBOLIC, False)):
        c = make_cpep_case(5, arch if arch[1] else (2, 4, 2))
        pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], covariate=covariate)
        nn = o.glorot_params(arch, 3) if arch[1] else np.array([1.78])
        cond = np.exp(rng.normal(0.0, 1.0, pop.N)) * (30.0 if arch[1] == 0 else 1.0)
        got = co.cpep_adaptive(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, nn, cond, times,
                               covariate=covariate)
        for i in range(pop.N):
            c0 = float(pop.c0[i])
            ref = o.solve_adaptive(o.cpep_rhs_scalar(pop, i, nn, cond[i], arch), [c0, float(pop.k2[i] / pop.k1[i]) * c0],
                                   list(times))
            assert np.max(np.abs(got[i] - np.array([r[0] for r in ref]))) < 1e-11


def test_adaptive_gradient_checker_replay_and_complex_step():
    """The checker of the device's adaptive gradient (tests/test_gpu_adaptive_grad.py): (i) replaying the recorded
    accepted steps reproduces the adaptive solution exactly -- same arithmetic; (ii) its complex-step derivative equals
    torch's reverse-mode derivative of the same replay (a third algorithm) and central differences of it."""
    import torch
    import cude_oracle as o
    arch = (2, 4, 2)
    c = make_cpep_case(4, arch)
    pop = o.CPepPopulation(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    loss, g_nn, g_b, sse = o.cpep_adaptive_loss_grad(c["nn"], c["beta"], pop, arch)
    tl = torch.zeros((), dtype=torch.float64)
    nn_t = torch.tensor(c["nn"], requires_grad=True)
    b_t = torch.tensor(c["beta"], requires_grad=True)
    for i in range(pop.N):
        c0 = float(pop.c0[i])
        u0 = [c0, float(pop.k2[i] / pop.k1[i]) * c0]
        rec = []
        ref = o.solve_adaptive(o.cpep_rhs_scalar(pop, i, c["nn"], float(np.exp(c["beta"][i])), arch), u0, pop.timepoints,
                               record=rec)
        assert 8 <= len(rec) <= 60 and abs(rec[-1][0] + rec[-1][1] - pop.timepoints[-1]) < 1e-12
        again = o.replay_steps(o.cpep_rhs_scalar(pop, i, c["nn"], float(np.exp(c["beta"][i])), arch), u0, pop.timepoints, rec)
        assert np.max(np.abs(np.array(again) - np.array(ref))) == 0.0
        assert abs(sum((again[t][0] - pop.cpeptide[i, t]) ** 2 for t in range(pop.T)) - sse[i]) <= 1e-11 * sse[i]
        G = [float(v) for v in pop.glucose[i]]
        k0, k1, k2 = float(pop.k0[i]), float(pop.k1[i]), float(pop.k2[i])
        eb = torch.exp(b_t[i])

        def rhs(t, u, G=G, k0=k0, k1=k1, k2=k2, c0=c0, eb=eb):
            dG = o.linear_interp(pop.timepoints, G, t) - G[0]
            prod = o.mlp(torch, [dG + 0.0 * eb, eb], nn_t, arch) - o.mlp(torch, [0.0 * eb, eb], nn_t, arch)
            return [-(k0 + k2) * u[0] + k1 * u[1] + k0 * c0 + prod, -k1 * u[1] + k2 * u[0]]
        out = o.replay_steps(rhs, [torch.tensor(u0[0], dtype=torch.float64), torch.tensor(u0[1], dtype=torch.float64)],
                             pop.timepoints, rec)
        for t in range(1, pop.T):
            tl = tl + (out[t][0] - pop.cpeptide[i, t]) ** 2
    (tl / pop.N).backward()
    assert np.max(np.abs(nn_t.grad.numpy() - g_nn)) <= 1e-11 * np.max(np.abs(g_nn))
    assert np.max(np.abs(b_t.grad.numpy() - g_b)) <= 1e-11 * np.max(np.abs(g_b))
    # suppression model: complex step against central differences of the replayed loss in one direction
    s = make_supp_case(3)
    l0, gn, gt, _ = o.supp_adaptive_loss_grad(s["nn"], s["theta"], s["data"], s["tp"], s["arch"], 0.01)
    scale, tpl = o.supp_scale(s["data"]), [float(v) for v in s["tp"]]
    recs = []
    for i in range(3):
        rec = []
        o.solve_adaptive(lambda t, u: o.supp_rhs(math, [float(v) for v in s["nn"]], math.exp(float(s["theta"][i])), s["arch"], t, u),
                         [float(v) for v in s["data"][:, 0, i]], tpl, record=rec)
        recs.append(rec)

    def replayed(nn, theta):
        tot = 0.0
        for i in range(3):
            out = o.replay_steps(lambda t, u: o.supp_rhs(np, nn, float(np.exp(theta[i])), s["arch"], t, u),
                                 [float(v) for v in s["data"][:, 0, i]], tpl, recs[i])
            tot += sum(((out[t][k] - s["data"][k, t, i]) / scale[k]) ** 2 for t in range(1, len(tpl)) for k in range(3))
        return tot / 3 + 0.01 * float(nn @ nn)
    assert abs(replayed(s["nn"], s["theta"]) - l0) <= 1e-12 * l0
    rng = np.random.default_rng(0)
    dn, dth, h = rng.standard_normal(s["nn"].size), rng.standard_normal(3), 1e-6
    fd = (replayed(s["nn"] + h * dn, s["theta"] + h * dth) - replayed(s["nn"] - h * dn, s["theta"] - h * dth)) / (2 * h)
    assert abs(fd - (gn @ dn + gt @ dth)) <= 1e-7 * abs(fd)


def test_failure_convention_and_adam():
    import cude_oracle as o
    import c_oracle as co
    arch = (2, 4, 2)
    c = make_cpep_case(5, arch)
    beta = c["beta"].copy(); beta[2] = np.nan
    r = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], beta, 30, 2, want_grad=False)
    assert r["loss"] == np.inf and r["n_failed"] == 1
    x, m, v = o.adam_update(np.array([1.0]), np.array([0.5]), np.zeros(1), np.zeros(1), 1, 1e-2)
    assert abs(x[0] - (1.0 - 1e-2 * 0.5 / (0.5 + 1e-8))) < 1e-15


@pytest.mark.parametrize("arch,n_state", [((2, 6, 2), 3), ((2, 4, 2), 2), ((3, 4, 2), 2), ((2, 4, 3), 3), ((2, 8, 2), 2)])
def test_reverse_mode_oracle_agrees_with_forward_mode_cpep(arch, n_state):
    """oracle/cude_oracle_rev.c (per-subject discrete adjoint, the CPU baseline of bench.py) against the
    forward-mode duals of oracle/cude_oracle.c (the reference's own AD method): two algorithms, one answer."""
    import c_oracle as co
    from conftest import make_cpep_case
    c = make_cpep_case(37, arch)
    kw = dict(covariate=(arch[0] == 3))
    a = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], 30, n_state, **kw)
    for nthreads in (1, 3):
        b = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], 30, n_state,
                    method="reverse", nthreads=nthreads, **kw)
        assert abs(a["loss"] - b["loss"]) <= 1e-13 * abs(a["loss"])
        assert np.max(np.abs(a["sse"] - b["sse"])) <= 1e-13 * np.max(a["sse"])
        assert np.max(np.abs(a["g_nn"] - b["g_nn"])) <= 1e-12 * np.max(np.abs(a["g_nn"]))
        assert np.max(np.abs(a["g_beta"] - b["g_beta"])) <= 1e-12 * np.max(np.abs(a["g_beta"]))


@pytest.mark.parametrize("arch", [(4, 3, 5), (4, 3, 2), (4, 6, 2)])
def test_reverse_mode_oracle_agrees_with_forward_mode_supp(arch):
    import c_oracle as co
    from conftest import make_supp_case
    s = make_supp_case(29, arch)
    for lam in (0.0, 0.01):
        a = co.supp(s["tp"], s["data"], arch, s["nn"], s["theta"], lam, 30)
        b = co.supp(s["tp"], s["data"], arch, s["nn"], s["theta"], lam, 30, method="reverse")
        assert abs(a["loss"] - b["loss"]) <= 1e-13 * abs(a["loss"])
        assert np.max(np.abs(a["g_nn"] - b["g_nn"])) <= 1e-12 * np.max(np.abs(a["g_nn"]))
        assert np.max(np.abs(a["g_theta"] - b["g_theta"])) <= 1e-12 * np.max(np.abs(a["g_theta"]))
