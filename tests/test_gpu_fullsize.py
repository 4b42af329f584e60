"""BASELINE configs[3] size on ONE GPU: 1e6 subjects (the 8-GPU job's global population) split into 8 shards of
125 000 evaluated one after the other must reproduce the single-population loss and gradient -- the arithmetic
identity behind the subject-sharded multi-GPU step (sum of the ranks' P+2 partial vectors, 1/N_global inside
every shard)."""
import numpy as np
import pytest
import torch  # noqa: F401

pytestmark = pytest.mark.gpu


def _population(n, seed):
    rng = np.random.default_rng(seed)
    tp = np.array([0.0, 30.0, 60.0, 90.0, 120.0])
    age = rng.uniform(20, 79, n)
    t2 = rng.random(n) < 0.44
    z = rng.standard_normal(n)
    G = np.maximum(3.2, np.array([5.22, 9.10, 10.44, 10.62, 10.35])[None, :]
                   + np.array([0.88, 1.99, 3.45, 4.58, 4.93])[None, :] * z[:, None])
    obs = np.maximum(0.2, rng.normal(0.62, 0.29, n))[:, None] * (1.0 + np.abs(rng.standard_normal((n, 5))).cumsum(1) * 0.4)
    beta = rng.normal(-0.63, 0.9, n)
    return tp, G, obs, age, t2, beta


def test_one_million_subjects_equals_sum_of_eight_shards():
    import cude_oracle as o
    from cude.engine import Engine
    arch, N, shards = (2, 6, 2), 1_000_000, 8
    tp, G, obs, age, t2, beta = _population(N, 99)
    nn = o.glorot_params(arch, 5)
    whole = Engine("cpep", arch, n_steps=30, n_state=3)
    whole.set_population_cpep(tp, G, obs, age, t2)
    whole.set_params(nn, beta)
    loss, g_nn, g_cond = whole.loss_grad()
    whole.close()
    assert np.isfinite(loss)
    n_loc = N // shards
    part_sum = np.zeros(g_nn.size + 2)
    g_parts = []
    for r in range(shards):
        s = slice(r * n_loc, (r + 1) * n_loc)
        eng = Engine("cpep", arch, n_steps=30, n_state=3)
        eng.set_population_cpep(tp, G[s], obs[s], age[s], t2[s])
        eng.set_global_subjects(N)
        eng.set_params(nn, beta[s])
        part, gc = eng.loss_grad_partial(want_cond_grad=True)
        eng.close()
        part_sum += part
        g_parts.append(gc)
    P = g_nn.size
    assert part_sum[P + 1] == 0
    assert abs(part_sum[P] / N - loss) < 1e-12 * loss
    assert np.max(np.abs(part_sum[:P] - g_nn)) < 1e-11 * np.max(np.abs(g_nn))
    # per-subject gradients are local.  The whole population runs as a mixed launch (7 machine-fills on the one-lane
    # kernel -- bitwise equal to the shards, which run on it entirely -- and the remainder time-split: rounding)
    gp = np.concatenate(g_parts)
    n_bulk = 7 * 2048 * 64
    assert np.array_equal(gp[:n_bulk], g_cond[:n_bulk]) or np.array_equal(gp, g_cond)
    assert np.max(np.abs(gp - g_cond)) <= 1e-12 * np.max(np.abs(g_cond))


def test_adaptive_gradient_at_1e5_subjects_properties():
    """BASELINE configs[2] size in the reference's own solver mode (adaptive Tsit5 + adjoint of the accepted steps,
    csrc/cude_adaptive.hip): size-independent properties -- repeatable bit for bit, per-subject results independent of
    how the population is sharded or ordered, loss / network gradient additive over shards, one failing subject fails
    the evaluation and nothing else."""
    import cude_oracle as o
    from cude.engine import CudeError, Engine
    arch, N = (2, 4, 2), 100_000
    tp, G, obs, age, t2, beta = _population(N, 7)
    nn = o.glorot_params(arch, 5)
    whole = Engine("cpep", arch, n_steps=0, n_state=2)
    whole.set_population_cpep(tp, G, obs, age, t2)
    whole.set_params(nn, beta)
    loss, g_nn, g_cond = whole.loss_grad()
    again = whole.loss_grad()
    assert np.isfinite(loss) and again[0] == loss and np.array_equal(again[1], g_nn) and np.array_equal(again[2], g_cond)
    n_steps = np.array([len(whole.adaptive_steps(i)[0]) for i in range(0, N, 997)])
    assert 5 <= n_steps.min() and n_steps.max() <= 64
    fwd = whole.forward(want_sse=True)
    assert abs(fwd["loss"] - loss) <= 1e-14 * loss
    with pytest.raises(CudeError):          # a forward launch overwrites the step counts the gradient's tape is read with
        whole.adaptive_steps(0)
    P = g_nn.size
    part_sum, g_parts = np.zeros(P + 2), []
    for s in (slice(0, 37_003), slice(37_003, N)):                # ragged shards: neither a multiple of the wave size
        eng = Engine("cpep", arch, n_steps=0, n_state=2)
        eng.set_population_cpep(tp, G[s], obs[s], age[s], t2[s])
        eng.set_global_subjects(N)
        eng.set_params(nn, beta[s])
        part, gc = eng.loss_grad_partial(want_cond_grad=True)
        eng.close()
        part_sum += part
        g_parts.append(gc)
    assert part_sum[P + 1] == 0 and abs(part_sum[P] / N - loss) < 1e-12 * loss
    assert np.max(np.abs(part_sum[:P] - g_nn)) < 1e-11 * np.max(np.abs(g_nn))
    assert np.array_equal(np.concatenate(g_parts), g_cond)
    rev = Engine("cpep", arch, n_steps=0, n_state=2)              # reversed subject order: lanes meet other neighbours
    rev.set_population_cpep(tp, G[::-1], obs[::-1], age[::-1], t2[::-1])
    rev.set_params(nn, beta[::-1])
    l2, gn2, gc2 = rev.loss_grad()
    rev.close()
    assert np.array_equal(gc2[::-1], g_cond) and abs(l2 - loss) < 1e-12 * loss
    assert np.max(np.abs(gn2 - g_nn)) < 1e-11 * np.max(np.abs(g_nn))
    bad = beta.copy()
    bad[54_321] = np.inf
    whole.set_params(nn, bad)
    l3, _, gc3 = whole.loss_grad()
    assert np.isinf(l3) and whole.n_failed() == 1
    ok = np.ones(N, bool)
    ok[54_321] = False
    assert np.array_equal(gc3[ok], g_cond[ok])
    whole.close()


def test_mixed_launch_above_one_machine_fill():
    """More than one machine-fill of subjects (> 131 072 on 256 CUs x 2 waves x 4 SIMDs): whole rounds run on the
    one-lane kernel, the remainder time-split (cude_api.hip: setup_chunks).  Same loss and gradient as the one-lane
    kernel alone to rounding (and NOT bit-identical: it is a different path), per-subject gradients included; against
    the CPU oracle (reverse-mode port) at the test-suite tolerances."""
    import os
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    arch, N = (2, 6, 2), 200_003
    tp, G, obs, age, t2, beta = _population(N, 11)
    nn = o.glorot_params(arch, 5)
    out = {}
    for tag, env in (("mixed", None), ("one", "1")):
        if env:
            os.environ["CUDE_CPEP_PATH"] = env                         # the one-lane kernel for every block
        try:
            eng = Engine("cpep", arch, n_steps=30, n_state=3)
            eng.set_population_cpep(tp, G, obs, age, t2)
        finally:
            os.environ.pop("CUDE_CPEP_PATH", None)
        eng.set_params(nn, beta)
        out[tag] = eng.loss_grad()
        fwd = eng.forward()["loss"]
        assert abs(fwd - out[tag][0]) <= 1e-13 * fwd
        eng.adam_init(1e-2)
        out[tag + "_trace"] = eng.adam_run(3)
        eng.close()
    (l1, g1, c1), (l0, g0, c0) = out["mixed"], out["one"]
    assert abs(l1 - l0) <= 1e-13 * l0 and np.max(np.abs(g1 - g0)) <= 1e-12 * np.max(np.abs(g0))
    assert np.max(np.abs(c1 - c0)) <= 1e-12 * np.max(np.abs(c0))
    assert not np.array_equal(c1, c0)                                  # the remainder really took the other path
    assert np.array_equal(c1[:131072], c0[:131072])                    # ... and the whole rounds the same one
    assert np.allclose(out["mixed_trace"], out["one_trace"], rtol=1e-12)
    ref = co.cpep(tp, G, obs, age, t2, arch, nn, beta, 30, 3, method="reverse")
    assert abs(l1 - ref["loss"]) <= 1e-10 * ref["loss"]
    assert np.max(np.abs(g1 - ref["g_nn"])) <= 1e-9 * np.max(np.abs(ref["g_nn"]))
    assert np.max(np.abs(c1 - ref["g_beta"])) <= 1e-9 * np.max(np.abs(ref["g_beta"]))


def test_mixed_launch_below_one_fill_failures_and_small_remainder():
    """66 000 subjects: the selector pairs 1024 workgroups on the one-lane kernel with a remainder of 8 (the last one
    ragged) time-split beside them.  Against the one-lane kernel alone; a failing subject in either part fails the
    evaluation and is counted, everybody else's gradient is untouched."""
    import os
    import cude_oracle as o
    from cude.engine import Engine
    arch, N = (2, 6, 2), 66_000
    tp, G, obs, age, t2, beta = _population(N, 3)
    nn = o.glorot_params(arch, 5)
    res = {}
    for tag in ("auto", "one"):
        if tag == "one":
            os.environ["CUDE_CPEP_PATH"] = "1"
        try:
            eng = Engine("cpep", arch, n_steps=30, n_state=3)
            eng.set_population_cpep(tp, G, obs, age, t2)
        finally:
            os.environ.pop("CUDE_CPEP_PATH", None)
        eng.set_params(nn, beta)
        res[tag] = eng.loss_grad()
        if tag == "auto":
            bad = beta.copy()
            bad[100] = np.nan                       # in the one-lane part
            bad[65_990] = np.inf                    # in the time-split remainder
            eng.set_params(nn, bad)
            lb, _, gcb = eng.loss_grad()
            assert np.isinf(lb) and eng.n_failed() == 2
            ok = np.ones(N, bool)
            ok[[100, 65_990]] = False
            # (the two waves that hold a failing lane leave the layer-1 exponent table -- a wave-uniform choice -- so
            # their other lanes agree to rounding; every other wave bit for bit)
            same = gcb == res["auto"][2]
            assert np.count_nonzero(~same[ok]) <= 2 * 63
            assert np.max(np.abs(gcb[ok] - res["auto"][2][ok])) <= 1e-12 * np.max(np.abs(res["auto"][2]))
        eng.close()
    (l1, g1, c1), (l0, g0, c0) = res["auto"], res["one"]
    assert abs(l1 - l0) <= 1e-13 * l0 and np.max(np.abs(g1 - g0)) <= 1e-12 * np.max(np.abs(g0))
    assert np.array_equal(c1[:65_536], c0[:65_536]) and not np.array_equal(c1[65_536:], c0[65_536:])
    assert np.max(np.abs(c1 - c0)) <= 1e-12 * np.max(np.abs(c0))


def test_baseline_config1_forward_only_at_its_own_size_and_launch_shape():
    """BASELINE configs[1] as it is quoted: 1e4 subjects, CPEP3 (2x6x6x1, 3 states, 30 steps), forward-only, through
    whatever path the library's selector takes at that size (time-split chunks: no CUDE_CPEP_PATH here).  Per-subject
    SSE against the C oracle on 400 random subjects (1e-10), the loss against their mean, the quadrature state, and
    bitwise repeatability of the whole call."""
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    arch, N = (2, 6, 2), 10_000
    tp, G, obs, age, t2, beta = _population(N, 2024)
    nn = o.glorot_params(arch, 5)
    eng = Engine("cpep", arch, n_steps=30, n_state=3)
    eng.set_population_cpep(tp, G, obs, age, t2)
    eng.set_params(nn, beta)
    f1 = eng.forward(want_sse=True)
    f2 = eng.forward(want_sse=True)
    assert np.isfinite(f1["loss"]) and eng.n_failed() == 0
    assert f1["loss"] == f2["loss"] and np.array_equal(f1["sse"], f2["sse"])          # bitwise repeatable
    assert abs(f1["loss"] - f1["sse"].mean()) <= 1e-13 * f1["loss"]
    idx = np.sort(np.random.default_rng(7).choice(N, 400, replace=False))
    ref = co.cpep(tp, G[idx], obs[idx], age[idx], t2[idx], arch, nn, beta[idx], 30, 3, want_grad=False)
    assert ref["n_failed"] == 0
    assert np.max(np.abs(f1["sse"][idx] - ref["sse"])) <= 1e-10 * max(1.0, float(np.max(ref["sse"])))
    # the same subjects on their own (another launch shape: 7 workgroups): the same per-subject numbers to rounding
    sub = Engine("cpep", arch, n_steps=30, n_state=3)
    sub.set_population_cpep(tp, G[idx], obs[idx], age[idx], t2[idx])
    sub.set_params(nn, beta[idx])
    fs = sub.forward(want_sse=True)
    assert np.max(np.abs(fs["sse"] - f1["sse"][idx])) <= 1e-12 * max(1.0, float(np.max(ref["sse"])))
    sub.close()
    eng.close()


def test_suppression_full_size_properties_1e5_subjects():
    """The reference's EnsembleThreads workload (suppression cUDE 4-3x5-1, T = 8, 30 steps) at 1e5 subjects -- the size the
    gradient kernel is measured at, where it runs two waves per SIMD (round 3) -- through properties that need no oracle
    run at that size: bitwise determinism, loss = mean of the per-subject SSEs, additivity over two unequal shards (the
    shards see the population's `scale` and 1/N through cude_set_global_subjects), the directional derivative against
    central differences, and an oracle spot check of per-subject SSE and dL/dtheta on 300 subjects."""
    import c_oracle as co
    from conftest import make_supp_case
    from cude.engine import Engine
    N, lam = 100_000, 0.01
    c = make_supp_case(N)
    arch = c["arch"]
    eng = Engine("supp", arch, n_steps=30, lam=lam)
    eng.set_population_supp(c["tp"], c["data"])
    eng.set_params(c["nn"], c["theta"])
    assert eng.grad_occupancy() >= 8                                  # two waves per SIMD granted by the runtime
    loss, g_nn, g_th = eng.loss_grad()
    loss2, g_nn2, g_th2 = eng.loss_grad()
    assert loss == loss2 and np.array_equal(g_nn, g_nn2) and np.array_equal(g_th, g_th2)
    sse = eng.forward(want_sse=True)["sse"]
    assert abs(sse.sum() / N + lam * float(c["nn"] @ c["nn"]) - loss) < 1e-12 * loss
    scale, n_glob = eng.get_scale()
    # two shards of unequal size with the whole population's scale and subject count
    cut, parts = 41_007, []
    for lo, hi in ((0, cut), (cut, N)):
        e2 = Engine("supp", arch, n_steps=30, lam=lam)
        e2.set_population_supp(c["tp"], c["data"][:, :, lo:hi])
        e2.set_global_subjects(N, scale)
        e2.set_params(c["nn"], c["theta"][lo:hi])
        part, gt = e2.loss_grad_partial(want_cond_grad=True)
        parts.append((part, gt))
        e2.close()
    P = g_nn.size
    tot = parts[0][0] + parts[1][0]
    assert tot[P + 1] == 0
    assert abs(tot[P] / N + lam * float(c["nn"] @ c["nn"]) - loss) < 1e-12 * loss
    assert np.max(np.abs(tot[:P] + 2 * lam * c["nn"] - g_nn)) < 1e-11 * np.max(np.abs(g_nn))
    assert np.max(np.abs(np.concatenate([parts[0][1], parts[1][1]]) - g_th)) <= 1e-15 * max(1.0, np.max(np.abs(g_th)) * N)
    # directional derivative
    rng = np.random.default_rng(3)
    d_nn, d_t = rng.standard_normal(P), rng.standard_normal(N)
    eps = 1e-6
    eng.set_params(c["nn"] + eps * d_nn, c["theta"] + eps * d_t); lp = eng.forward()["loss"]
    eng.set_params(c["nn"] - eps * d_nn, c["theta"] - eps * d_t); lm = eng.forward()["loss"]
    dd = g_nn @ d_nn + g_th @ d_t
    assert abs((lp - lm) / (2 * eps) - dd) < 2e-7 * abs(dd)
    # oracle on 300 random subjects: per-subject quantities are local once the population's scale is used.  The C oracle
    # takes its scale from the data it is given, so the subset is evaluated with the population's scale through the
    # product (a 300-subject engine with set_global_subjects) AND the oracle is held to that engine at its own size.
    idx = np.sort(rng.choice(N, 300, replace=False))
    sub = Engine("supp", arch, n_steps=30, lam=0.0)
    sub.set_population_supp(c["tp"], c["data"][:, :, idx])
    sub.set_params(c["nn"], c["theta"][idx])
    ls, gns, gts = sub.loss_grad()
    ref = co.supp(c["tp"], c["data"][:, :, idx], arch, c["nn"], c["theta"][idx], 0.0, 30)
    assert abs(ls - ref["loss"]) < 1e-10 * ref["loss"]
    assert np.max(np.abs(gns - ref["g_nn"])) < 1e-9 * np.max(np.abs(ref["g_nn"]))
    assert np.max(np.abs(gts - ref["g_theta"])) < 1e-9 * np.max(np.abs(ref["g_theta"]))
    # ... and the same subjects inside the big population give the same per-subject SSE once rescaled by scale^2
    s_sub, _ = sub.get_scale()
    sse_sub = sub.forward(want_sse=True)["sse"]
    # per-subject SSE is sum_s SSE_s / scale_s^2: not separable without the per-state parts, so compare through a third
    # engine that holds the subset with the BIG population's scale
    sub.set_global_subjects(300, scale)
    sse_big_scale = sub.forward(want_sse=True)["sse"]
    assert np.max(np.abs(sse_big_scale - sse[idx])) < 1e-11 * max(1.0, float(np.max(sse[idx])))
    assert not np.array_equal(sse_sub, sse_big_scale) or np.allclose(s_sub, scale)
    sub.close()
    eng.close()


def test_suppression_adaptive_gradient_at_1e5_subjects_properties():
    """The suppression cUDE in the reference's own solver mode (Tsit5 adaptive, suppression_model.jl:113,123) at 1e5
    subjects, through the kernel with unrolled stages that keeps the stage inputs on its tape
    (csrc/cude_adaptive_unrolled_supp.hip): bitwise repeatable, the forward half of the gradient launch is the forward
    launch, per-subject results independent of the launch order (cude_adaptive_regroup), additive over two ragged shards,
    the directional derivative against central differences of the adaptive objective, one failing subject fails the
    evaluation and nothing else, and the C oracle's adaptive solve on 200 subjects (solver-tolerance level: two correct
    controllers do not always accept the same steps)."""
    import c_oracle as co
    from conftest import make_supp_case
    from cude.engine import Engine
    N, lam = 100_000, 0.01
    c = make_supp_case(N)
    arch = c["arch"]
    eng = Engine("supp", arch, n_steps=0, lam=lam)
    eng.set_option("auto_regroup", 0)
    eng.set_population_supp(c["tp"], c["data"])
    eng.set_params(c["nn"], c["theta"])
    loss, g_nn, g_th = eng.loss_grad()
    loss2, g_nn2, g_th2 = eng.loss_grad()
    assert np.isfinite(loss) and loss == loss2 and np.array_equal(g_nn, g_nn2) and np.array_equal(g_th, g_th2)
    n_steps = np.array([len(eng.adaptive_steps(i)[0]) for i in range(0, N, 997)])     # (the gradient's: before any forward launch)
    assert 8 <= n_steps.min() and n_steps.max() <= 64
    fwd = eng.forward(want_sse=True)
    assert abs(fwd["loss"] - loss) <= 1e-14 * loss
    assert abs(fwd["sse"].sum() / N + lam * float(c["nn"] @ c["nn"]) - loss) < 1e-12 * loss
    scale, _ = eng.get_scale()
    # launch ordered by accepted-step count: per-subject results bit for bit, the shared gradient to rounding
    before, after = eng.adaptive_regroup()
    assert after <= before
    l3, gn3, gt3 = eng.loss_grad()
    assert np.array_equal(gt3, g_th) and abs(l3 - loss) < 1e-12 * loss
    assert np.max(np.abs(gn3 - g_nn)) < 1e-11 * np.max(np.abs(g_nn))
    # two ragged shards with the whole population's scale and subject count
    cut, parts = 41_007, []
    for lo, hi in ((0, cut), (cut, N)):
        e2 = Engine("supp", arch, n_steps=0, lam=lam)
        e2.set_population_supp(c["tp"], c["data"][:, :, lo:hi])
        e2.set_global_subjects(N, scale)
        e2.set_params(c["nn"], c["theta"][lo:hi])
        parts.append(e2.loss_grad_partial(want_cond_grad=True))
        e2.close()
    P = g_nn.size
    tot = parts[0][0] + parts[1][0]
    assert tot[P + 1] == 0
    assert abs(tot[P] / N + lam * float(c["nn"] @ c["nn"]) - loss) < 1e-12 * loss
    assert np.max(np.abs(tot[:P] + 2 * lam * c["nn"] - g_nn)) < 1e-11 * np.max(np.abs(g_nn))
    assert np.array_equal(np.concatenate([parts[0][1], parts[1][1]]), g_th)
    # directional derivative of the adaptive objective (accepted steps move with the parameters: solver-tolerance level)
    rng = np.random.default_rng(3)
    d_nn, d_t = rng.standard_normal(P), rng.standard_normal(N)
    eps = 1e-5
    eng.set_params(c["nn"] + eps * d_nn, c["theta"] + eps * d_t); lp = eng.forward()["loss"]
    eng.set_params(c["nn"] - eps * d_nn, c["theta"] - eps * d_t); lm = eng.forward()["loss"]
    dd = g_nn @ d_nn + g_th @ d_t
    assert abs((lp - lm) / (2 * eps) - dd) < 5e-3 * abs(dd)
    # one failing subject
    bad = c["theta"].copy()
    bad[54_321] = np.inf
    eng.set_params(c["nn"], bad)
    l4, _, gt4 = eng.loss_grad()
    assert np.isinf(l4) and eng.n_failed() == 1
    ok = np.ones(N, bool)
    ok[54_321] = False
    assert np.array_equal(gt4[ok], g_th[ok])
    eng.close()
    # the C oracle's adaptive solve on 200 subjects (its own scale: the engine holds the same subset)
    idx = np.sort(rng.choice(N, 200, replace=False))
    sub = Engine("supp", arch, n_steps=0, lam=0.0)
    sub.set_population_supp(c["tp"], c["data"][:, :, idx])
    sub.set_params(c["nn"], c["theta"][idx])
    traj = sub.forward(want_traj=True)["traj"]
    ref = co.supp_adaptive(c["tp"], c["data"][:, :, idx], arch, c["nn"], c["theta"][idx])      # 3 x T x N, as simul
    assert traj.shape == ref.shape and np.max(np.abs(traj - ref)) < 5e-7 * np.max(np.abs(ref))
    sub.close()
