"""cude_train_restarts with the optimiser state resident on the device (csrc/cude_train.hip) -- the reference's
`for p in initials[selected]: _optimize(...)` (src/parameter-estimation.jl:372-383, `_optimize` :170-183;
suppression_model.jl:140-170) for all restarts at once -- against the same two stages stated on the host
(option "train_host": cude::adam_update / cude::Lbfgs of csrc/cude_optim.h over cude_multistart_loss_grad, the path of
rounds 2-4).  Adam: the device kernels use the host statement's arithmetic without contraction, so the iterates are the
same BITS.  L-BFGS: element-wise the same arithmetic, inner products summed by a tree instead of left to right, so the
iterates agree to rounding over the first iterations (tests/test_gpu_lbfgs_oracle.py holds the device stage to the
independent oracle optimiser)."""
import numpy as np
import pytest
import torch  # noqa: F401

from conftest import make_cpep_case, make_supp_case

pytestmark = pytest.mark.gpu


def _engine(model, N, n_steps=30, lam=0.0):
    from cude.engine import Engine
    if model == "supp":
        c = make_supp_case(N)
        eng = Engine("supp", c["arch"], n_steps=n_steps, lam=lam)
        eng.set_population_supp(c["tp"], c["data"])
        return eng, c["nn"], c["theta"]
    arch, ns = ((2, 6, 2), 3) if model == "cpep" else ((2, 4, 2), 2)
    c = make_cpep_case(N, arch)
    eng = Engine("cpep", arch, n_steps=n_steps, n_state=ns if n_steps else 2)
    eng.set_population_cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"])
    return eng, c["nn"], c["beta"]


def _starts(nn0, cond0, K, seed):
    rng = np.random.default_rng(seed)
    return (nn0[None, :] * (1.0 + 0.2 * rng.standard_normal((K, nn0.size))),
            cond0[None, :] + 0.3 * rng.standard_normal((K, cond0.size)))


@pytest.mark.parametrize("model,N,lam", [("cpep4", 57, 0.0), ("cpep", 3000, 0.0), ("supp", 37, 0.01), ("supp", 2500, 0.0)])
def test_adam_stage_on_the_device_gives_the_host_statements_bits(model, N, lam):
    eng, nn0, cond0 = _engine(model, N, lam=lam)
    K = 5
    nn_s, cond_s = _starts(nn0, cond0, K, 3)
    cond_s[2, N // 2] = np.nan                                  # restart 2 fails from the start: dropped, the others go on
    dev = eng.train_restarts(nn_s, cond_s, 12, 1e-2, 0, want_trace=True)
    eng.set_option("train_host", 2)
    host = eng.train_restarts(nn_s, cond_s, 12, 1e-2, 0, want_trace=True)
    eng.close()
    for d, h in zip(dev, host):
        assert np.array_equal(d, h, equal_nan=True)
    nn_d, cond_d, obj_d, tr_d = dev
    assert np.isinf(obj_d[2]) and np.all(np.isnan(tr_d[2])) and np.all(np.isfinite(np.delete(obj_d, 2)))
    assert np.array_equal(nn_d[2], nn_s[2]) and np.array_equal(cond_d[2], cond_s[2], equal_nan=True)   # never updated
    ok = np.arange(K) != 2
    assert np.all(np.isfinite(tr_d[ok])) and np.all(tr_d[ok, -1] < tr_d[ok, 0])


@pytest.mark.parametrize("model,N", [("cpep4", 57), ("cpep4", 20000), ("supp", 37)])
def test_lbfgs_stage_on_the_device_follows_the_host_machines(model, N):
    """Same iterates to rounding while rounding has not been amplified yet (the first iterations), the same loss-trace
    layout, and an end point of the same quality."""
    eng, nn0, cond0 = _engine(model, N)
    K = 4
    nn_s, cond_s = _starts(nn0, cond0, K, 4)
    few = 4
    dev = eng.train_restarts(nn_s, cond_s, 3, 1e-2, few, want_trace=True)
    dev_long = eng.train_restarts(nn_s, cond_s, 3, 1e-2, 40, want_trace=True)
    eng.set_option("train_host", 1)                             # Adam on the device, L-BFGS vectors on the host
    host = eng.train_restarts(nn_s, cond_s, 3, 1e-2, few, want_trace=True)
    host_long = eng.train_restarts(nn_s, cond_s, 3, 1e-2, 40, want_trace=True)
    eng.close()
    scale = max(1.0, float(np.max(np.abs(host[0]))), float(np.max(np.abs(host[1]))))
    assert np.max(np.abs(dev[0] - host[0])) <= 1e-9 * scale and np.max(np.abs(dev[1] - host[1])) <= 1e-9 * scale
    assert np.max(np.abs(dev[2] / host[2] - 1.0)) <= 1e-10
    assert np.array_equal(np.isnan(dev[3]), np.isnan(host[3]))
    both = ~np.isnan(host[3])
    assert np.max(np.abs(dev[3][both] / host[3][both] - 1.0)) <= 1e-10
    assert np.array_equal(dev[3][:, :3], host[3][:, :3])         # the Adam part of the trace: the same bits
    # 40 iterations: a descent on every restart, and optima of the same quality (on the suppression objective the paths of
    # any two correct statements part after ~10 iterations, DESIGN.md 3: there the bar is a band, not a distance)
    band = 0.25 if model == "supp" else 0.05
    assert np.all(dev_long[2] <= dev[2]) and np.all(np.abs(dev_long[2] - host_long[2]) <= band * host_long[2] + 1e-6)
    # objective = loss at the returned point
    eng2, _, _ = _engine(model, N)
    for k in range(K):
        eng2.set_params(dev_long[0][k], dev_long[1][k])
        assert abs(eng2.forward()["loss"] - dev_long[2][k]) <= 1e-12 * dev_long[2][k]
    eng2.close()


def test_adaptive_restarts_on_the_device():
    """The reference's own solver mode (n_steps = 0) at a population large enough to be re-ordered by accepted-step count."""
    eng, nn0, cond0 = _engine("cpep4", 9000, n_steps=0)
    nn_s, cond_s = _starts(nn0, cond0, 3, 5)
    dev = eng.train_restarts(nn_s, cond_s, 6, 1e-2, 3, want_trace=True)
    eng.close()
    eng, _, _ = _engine("cpep4", 9000, n_steps=0)
    eng.set_option("train_host", 2)
    host = eng.train_restarts(nn_s, cond_s, 6, 1e-2, 3, want_trace=True)
    eng.close()
    assert np.all(np.isfinite(dev[2])) and np.all(dev[3][:, 5] < dev[3][:, 0])
    # Adam: the same bits (the shared gradient's summation order follows the launch order, which both runs change at the
    # same evaluations); L-BFGS: the adaptive objective turns last-place differences into accept / reject decisions of the
    # step controller, so rounding in the inner products shows at 1e-7 after three iterations
    assert np.array_equal(dev[3][:, :6], host[3][:, :6])
    assert np.max(np.abs(dev[2] / host[2] - 1.0)) <= 1e-5
    assert np.max(np.abs(dev[0] - host[0])) <= 1e-4 * max(1.0, float(np.max(np.abs(host[0]))))
