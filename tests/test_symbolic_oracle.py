"""CPU checks of the oracle's restatement of the symbolic (analytic production) c-peptide model
(c-peptide/03-symreg.jl:37-40, src/c-peptide-models.jl:68-75,118-142, src/saem-symreg.jl:23-29)."""
import math
import os

import numpy as np

import cude_oracle as o

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pop(N=6, seed=4):
    tp, G, cp, age, t2, _, rng = o.synthetic_cpep_population(N, seed)
    k = np.exp(rng.normal(math.log(40.0), 0.5, N))
    pop0 = o.CPepPopulation(tp, G, cp, age, t2)
    traj = o.cpep_forward(np, np.array([1.78]), k, pop0, o.SYMBOLIC, 30, cond_space="raw")
    obs = np.stack([traj[t][0] for t in range(5)], axis=1) * (1 + 0.05 * rng.standard_normal((N, 5)))
    obs[:, 0] = cp[:, 0]
    return o.CPepPopulation(tp, G, obs, age, t2), k


def test_production_term_matches_the_reference_expression():
    dG = np.array([-1.0, 0.0, 0.5, 3.0, 12.0])
    got = o.symbolic_production(np, dG, 1.78, 21.8)
    want = [0.0, 0.0, 1.78 * 0.5 / (0.5 + 21.8), 1.78 * 3.0 / (3.0 + 21.8), 1.78 * 12.0 / (12.0 + 21.8)]
    assert np.array_equal(got, np.array(want))
    assert o.n_params(o.SYMBOLIC) == 1


def test_dose_response_table_is_close_to_the_symbolic_form():
    """Soft pin: the reference's own comparison (03-symreg.jl:56-61) draws 1.78 dG/(dG + k) with
    k = 167 beta^3 + 21.8 next to the network dose-response table it was regressed on."""
    t = np.load(os.path.join(GOLDEN, "ohashi_production.npz"))
    k = 167.0 * t["beta"] ** 3 + 21.8
    sym = o.symbolic_production(np, t["glucose"], 1.78, k)
    assert np.corrcoef(sym, t["production"])[0, 1] > 0.97
    assert np.sqrt(np.mean((sym - t["production"]) ** 2)) < 0.07      # table range 0 .. 0.74 nM/min


def test_gradient_against_central_differences():
    pop, k = _pop()
    p0 = np.array([1.7])
    for space, cond in (("raw", k * 1.2), ("log", np.log(k * 1.2))):
        loss, g_p, g_c, _ = o.cpep_loss_grad_torch(p0, cond, pop, o.SYMBOLIC, 30, 2, space)
        f = lambda p, c: float(o.cpep_loss(np, p, c, pop, o.SYMBOLIC, 30, 2, space)[0])
        e = 1e-6
        fd_p = (f(p0 + e, cond) - f(p0 - e, cond)) / (2 * e)
        assert abs(fd_p - g_p[0]) < 1e-6 * abs(g_p[0])
        for i in range(pop.N):
            d = np.zeros(pop.N)
            d[i] = e * max(1.0, abs(cond[i]))
            fd = (f(p0, cond + d) - f(p0, cond - d)) / (2 * d[i])
            assert abs(fd - g_c[i]) < 1e-5 * max(abs(g_c[i]), 1e-8)


def test_fixed_step_solution_is_close_to_the_adaptive_one():
    """The reference solves with the adaptive default (abstol 1e-6, reltol 1e-3).  The analytic production has a
    kink where glucose crosses its basal value (the dG >= 0 branch), so fixed steps converge more slowly than for
    the smooth network term: 4e-4 at 8 steps per observation interval, 7.5e-5 at 16 (the mirror's default for
    this model), 8e-6 at 32 -- all inside the reference solver's own tolerance."""
    pop, k = _pop(4)
    fixed = o.cpep_forward(np, np.array([1.78]), k, pop, o.SYMBOLIC, 64, cond_space="raw")
    for i in range(pop.N):
        one = o.CPepPopulation(pop.timepoints, pop.glucose[i:i + 1], pop.cpeptide[i:i + 1], pop.age[i:i + 1],
                               pop.t2dm[i:i + 1])
        rhs = lambda t, u: [float(v[0]) for v in o.cpep_rhs(np, one, np.array([1.78]), k[i:i + 1], o.SYMBOLIC, t,
                                                            [np.array([u[0]]), np.array([u[1]])], 2)]
        tight = o.solve_adaptive(rhs, [one.c0[0], one.k2[0] / one.k1[0] * one.c0[0]], one.timepoints,
                                 abstol=1e-10, reltol=1e-10)
        for t in range(5):
            assert abs(fixed[t][0][i] - tight[t][0]) < 2e-4 * abs(tight[t][0]) + 1e-6
