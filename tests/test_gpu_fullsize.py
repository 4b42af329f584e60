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
    assert np.array_equal(np.concatenate(g_parts), g_cond)      # per-subject gradients are local: bitwise equal
