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
