"""Round-5 changes that must not change a bit: the tail of a time-split gradient evaluation in one launch (option
"fused_tail": network-gradient columns, loss / failure columns with the optimiser's state advance, and the chunks' shares of
the conditional gradient; three launches before), on the single-set path, under captured graphs and on the side-by-side
restarts of the reference's `train` (src/parameter-estimation.jl:340-386)."""
import numpy as np
import pytest

from conftest import make_cpep_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,arch,n_state", [(57, (2, 4, 2), 2), (1000, (2, 6, 2), 3), (117, (3, 4, 2), 2)])
def test_one_launch_tail_of_the_time_split_gradient_is_bit_identical(N, arch, n_state):
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    rng = np.random.default_rng(2)
    nn_sets = c["nn"][None, :] * (1.0 + 0.05 * rng.standard_normal((5, c["nn"].size)))
    b_sets = c["beta"][None, :] + 0.1 * rng.standard_normal((5, N))
    mask = np.ones(c["nn"].size)
    mask[3] = 0.0
    out = []
    for fused in (0, 1):
        eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
        eng.set_option("fused_tail", fused)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        eng.set_param_mask(mask)
        r = list(eng.loss_grad())
        eng.adam_init(1e-2)
        r.append(np.array([eng.adam_step() for _ in range(3)] + list(eng.adam_run(11))))
        r += list(eng.get_params())
        r += list(eng.multistart_loss_grad(nn_sets, b_sets))
        r += list(eng.train_restarts(nn_sets, b_sets, 20, 1e-2, 5))
        out.append(r)
        eng.close()
    assert out[0][1][3] == 0.0 and np.all(np.isfinite(out[0][3]))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("N,arch,n_state,tp", [(57, (2, 4, 2), 2, None), (700, (2, 6, 2), 3, None),
                                               (90, (2, 4, 2), 2, [0.0, 0.0001, 30.0, 31.0, 31.5, 120.0])],
                         ids=["57", "700-three-states", "several-observations-in-one-step"])
def test_scan_adjoint_recursion_as_a_linear_map(N, arch, n_state, tp):
    """The time-split path's per-subject scan runs the stage-adjoint recursion (no network in it: J_f = A) as the subject's
    linear map, precomputed from the same algebra on unit inputs (Cpep2Args::adj_map; option "scan_map" = 0: the
    stage-by-stage form).  Same numbers to rounding -- also when several observations fall into one step and when one sits
    a hair behind the initial time --, and the same as the oracle's reverse mode."""
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import Engine
    c = make_cpep_case(N, arch)
    if tp is not None:                         # a grid whose observation times crowd into two of the 30 steps
        t_old = c["tp"]
        G = np.stack([np.interp(tp, t_old, c["G"][i]) for i in range(N)])
        obs = np.stack([np.interp(tp, t_old, c["obs"][i]) for i in range(N)])
        c = dict(c, tp=np.array(tp), G=G, obs=obs)
    out = []
    for use_map in (1, 0):
        eng = Engine("cpep", arch, n_steps=30, n_state=n_state)
        eng.set_option("cpep_path", "2:6")
        eng.set_option("scan_map", use_map)
        eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
        eng.set_params(c["nn"], c["beta"])
        out.append(eng.loss_grad())
        eng.close()
    (l1, g1, b1), (l0, g0, b0) = out
    assert l1 == l0                                                     # (the forward half is untouched)
    assert np.max(np.abs(g1 - g0)) <= 1e-13 * np.max(np.abs(g0)) and np.max(np.abs(b1 - b0)) <= 1e-13 * np.max(np.abs(b0))
    ref = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], arch, c["nn"], c["beta"], 30, n_state)
    assert abs(l1 - ref["loss"]) <= 1e-10 * ref["loss"]
    assert np.max(np.abs(g1 - ref["g_nn"])) <= 1e-9 * np.max(np.abs(ref["g_nn"]))
    assert np.max(np.abs(b1 - ref["g_beta"])) <= 1e-9 * np.max(np.abs(ref["g_beta"]))
