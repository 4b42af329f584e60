"""Host-side logic that needs no GPU: API mirror argument handling, Latin hypercube initials, subject sharding,
L-BFGS + backtracking, and the data-parallel training step over a 2-rank gloo group (driven with an
oracle-backed stand-in engine: the product engine itself needs a GPU and never falls back)."""
import os
import sys

import numpy as np
import pytest

from conftest import make_cpep_case, free_port


def test_chain_mirrors_reference_argument_errors():
    from cude import api
    assert api.chain(4, 2, "tanh").n_params == 37
    assert api.chain(6, 2, np.tanh).arch == (2, 6, 2)
    assert api.chain([3, 3, 3, 3, 3], "tanh", input_dims=4).n_params == 67
    assert api.neural_network_model(5, 3, input_dims=4).n_params == 67
    with pytest.raises(ValueError):
        api.chain([], "tanh")
    with pytest.raises(ValueError):
        api.chain([4, 4], ["tanh"])
    assert api.chain([4, 5], "tanh").arch == (2, 5, 2)                 # unequal widths: zero-padded + masked (class Chain)
    mixed = api.chain([4, 5], ["tanh", "relu"])                        # (round 5) per-layer functions: the general form,
    assert mixed.general and mixed.arch == (2, (4, 5), ("tanh", "relu"), "softplus")      # SimpleChains' own layout, no padding
    assert mixed.n_params == 2 * 4 + 4 + 4 * 5 + 5 + 5 + 1 and mixed.mask is None
    relu = api.chain(4, 2, "relu", output_activation="identity")       # (round 4) relu / sigmoid, softplus / identity
    assert (relu.activation, relu.output_activation) == ("relu", "identity") and relu.key != api.chain(4, 2, "tanh").key
    assert api.chain([4, 4], ["sigmoid", "sigmoid"]).activation == "sigmoid"
    with pytest.raises(NotImplementedError):
        api.chain(4, 2, "gelu")
    assert api.chain([4, 5], "relu").general                           # unequal widths with another function: general form too
    with pytest.raises(NotImplementedError):
        api.chain([4, 5], "tanh", output_dims=2)                       # one network output, as every model of the reference
    net = api.chain(4, 2, "tanh")
    with pytest.raises(ValueError):
        api.CPeptideConditionalUDEModel([1, 2, 3], [0, 1, 2], 40, net, [1, 2], False)
    m = api.CPeptideCUDEModel([5.0, 6.0, 7.0], [0.0, 30.0, 60.0], 40, net, [1.0, 2.0, 1.5], True)
    assert m.t2dm and m.age == 40.0


def test_init_params_and_latin_hypercube():
    from cude import api
    rng = np.random.default_rng(3)
    net = api.chain(6, 2, "tanh")
    p = api.init_params(net, rng)
    # Glorot normal over each layer's whole [W b] matrix: biases are random too, sigma = sqrt(2 / (out + in + 1))
    assert p.size == 67 and np.all(p != 0)
    big = api.init_params(api.chain(64, 3, "tanh"), rng)              # 2->64, 64->64 x2, 64->1
    hidden = big[192:192 + 64 * 65]
    assert abs(hidden.std() / np.sqrt(2.0 / 129) - 1) < 0.05 and abs(big[:192].std() / np.sqrt(2.0 / 67) - 1) < 0.15
    lhs = api.initial_parameters(7, -2.0, 0.0, 50, rng)
    assert lhs.shape == (7, 50) and lhs.min() >= -2.0 and lhs.max() <= 0.0
    for row in lhs:       # exactly one sample per stratum
        assert sorted(np.floor((row + 2.0) / 2.0 * 50).astype(int)) == list(range(50))


def test_find_confidence_intervals_semantics():
    """src/likelihood-profiles.jl:34-59: thresholds per target, outermost points, infinite ends, fallback target."""
    from cude import api
    values = np.linspace(-2.0, 2.0, 401)
    nll = 10.0 * values ** 2                                            # minimum 0 at 0
    lo, hi = api.find_confidence_intervals(nll, 0.0, values, target="cantelli95")
    assert abs(hi - np.sqrt(0.716)) < 0.011 and hi <= np.sqrt(0.716) and abs(lo + hi) < 1e-12
    lo90, hi90 = api.find_confidence_intervals(nll, 0.0, values, target="cantelli90")
    lo_r, hi_r = api.find_confidence_intervals(nll, 0.0, values, target="raue95")
    assert hi_r < hi90 < hi and abs(hi_r - np.sqrt(0.3841458820694124)) < 0.011
    assert api.find_confidence_intervals(nll, 0.0, values, target="nonsense") == (lo_r, hi_r)
    flat_right = np.where(values > 0, 0.0, nll)                         # unidentifiable upwards
    assert api.find_confidence_intervals(flat_right, 0.0, values) == (lo, np.inf)
    assert api.find_confidence_intervals(np.zeros(401), 0.0, values) == (-np.inf, np.inf)
    with pytest.raises(ValueError):
        api.find_confidence_intervals(nll + 100.0, 0.0, values)


def test_shard_bounds_cover_and_balance():
    from cude.parallel import shard_bounds
    for n, w in [(10, 3), (1000000, 8), (5, 8), (64, 2)]:
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_lbfgs_and_backtracking():
    from cude.lbfgs import lbfgs

    def rosen(x):
        f = 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
        g = np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2)])
        return f, g
    r = lbfgs(rosen, np.array([-1.2, 1.0]), maxiters=200)
    assert r["f"] < 1e-12 and np.allclose(r["x"], [1.0, 1.0], atol=1e-5)
    A = np.diag(np.arange(1.0, 21.0))
    r = lbfgs(lambda x: (0.5 * x @ A @ x, A @ x), np.ones(20), maxiters=100)
    assert r["f"] < 1e-14

    def with_inf(x):                       # solver-failure region: +Inf must be backed away from
        if x[0] > 2.0:
            return np.inf, np.zeros(1)
        return (x[0] - 1.5) ** 2, np.array([2 * (x[0] - 1.5)])
    r = lbfgs(with_inf, np.array([-30.0]), maxiters=50)
    assert abs(r["x"][0] - 1.5) < 1e-6


def test_batched_lbfgs_reproduces_serial_runs():
    """lbfgs_batched drives K copies of the same generator in lock step: every problem must see exactly the values
    it would see alone -- including problems that finish early or hit the Inf region -- whatever the others do."""
    from cude.lbfgs import lbfgs, lbfgs_batched
    rng = np.random.default_rng(0)
    scales = [1.0, 3.0, 0.2, 10.0, 1.0]
    starts = np.array([[-1.2, 1.0], [2.0, -1.0], [0.3, 0.3], [-0.5, 2.5], [1.0, 1.0]])     # the last one: converged at x0

    def rosen(x, a):
        if x[0] > 4.0:                                  # a solver-failure region
            return np.inf, np.zeros(2)
        f = a * (100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2)
        g = a * np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2)])
        return f, g
    n_batches = [0]

    def fg_batch(X):
        n_batches[0] += 1
        vals = [rosen(X[k], scales[k]) for k in range(len(scales))]
        return np.array([v[0] for v in vals]), np.stack([v[1] for v in vals])
    batched = lbfgs_batched(fg_batch, starts, maxiters=60)
    serial = [lbfgs(lambda x, a=a: rosen(x, a), starts[k], maxiters=60) for k, a in enumerate(scales)]
    for b, s in zip(batched, serial):
        assert np.array_equal(b["x"], s["x"]) and b["f"] == s["f"] and b["iterations"] == s["iterations"]
        assert b["f_calls"] == s["f_calls"] and b["converged"] == s["converged"]
    assert n_batches[0] == max(s["f_calls"] for s in serial)          # lock step: as many rounds as the longest run
    assert batched[-1]["iterations"] == 0 and batched[0]["f"] < 1e-10


def test_native_lbfgs_follows_the_python_statement():
    """cude_lbfgs_minimize (csrc/cude_optim.h: the same algorithm as a resumable C++ state machine, the engine of
    cude_train_restarts) against cude.lbfgs.lbfgs on host objectives.  The two differ only in the summation order
    of their dot products, so iterates agree closely and the counters exactly on these problems."""
    from cude.engine import lbfgs_minimize
    from cude.lbfgs import lbfgs

    def rosen(x):
        f = 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
        g = np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2)])
        return f, g
    a, b = lbfgs_minimize(rosen, [-1.2, 1.0], 200), lbfgs(rosen, np.array([-1.2, 1.0]), maxiters=200)
    assert a["f"] < 1e-12 and np.allclose(a["x"], [1.0, 1.0], atol=1e-5) and a["converged"]
    assert a["iterations"] == b["iterations"] and a["f_calls"] == b["f_calls"]
    assert np.allclose(a["x"], b["x"], rtol=0, atol=1e-9)
    A = np.diag(np.arange(1.0, 31.0))                     # more than m = 10 curvature pairs: the ring buffer wraps
    quad = lambda x: (0.5 * x @ A @ x, A @ x)
    a, b = lbfgs_minimize(quad, np.ones(30), 100), lbfgs(quad, np.ones(30), maxiters=100)
    assert a["f"] < 1e-14 and a["iterations"] == b["iterations"] and a["f_calls"] == b["f_calls"]

    def with_inf(x):                                      # solver-failure region: +Inf must be backed away from
        if x[0] > 2.0:
            return np.inf, np.zeros(1)
        return (x[0] - 1.5) ** 2, np.array([2 * (x[0] - 1.5)])
    a, b = lbfgs_minimize(with_inf, [-30.0], 50), lbfgs(with_inf, np.array([-30.0]), maxiters=50)
    assert abs(a["x"][0] - 1.5) < 1e-6 and a["iterations"] == b["iterations"] and a["f_calls"] == b["f_calls"]
    a = lbfgs_minimize(rosen, [1.0, 1.0], 50)             # already converged at x0
    assert a["iterations"] == 0 and a["f_calls"] == 1 and a["converged"]
    a, b = lbfgs_minimize(rosen, [-1.2, 1.0], 7), lbfgs(rosen, np.array([-1.2, 1.0]), maxiters=7)
    assert a["iterations"] == 7 and not a["converged"] and np.allclose(a["x"], b["x"], atol=1e-12)
    assert abs(a["f"] - b["f"]) < 1e-12


# ------------------------------------------------------------------ 2-rank gloo data-parallel step
class OracleEngine:
    """Test stand-in with the Engine interface used by ShardedTrainer (numerics from the CPU oracle)."""

    def __init__(self, case, lo, hi):
        import c_oracle as co
        self.co, self.c, self.lo, self.hi = co, case, lo, hi
        self.N, self.P = hi - lo, case["nn"].size
        self.nn, self.cond = case["nn"].copy(), case["beta"][lo:hi].copy()
        self.n_global = float(self.N)

    def set_global_subjects(self, n, scale=None):
        self.n_global = float(n)

    def loss_grad_partial(self, want_cond_grad=False):
        c, s = self.c, slice(self.lo, self.hi)
        r = self.co.cpep(c["tp"], c["G"][s], c["obs"][s], c["age"][s], c["t2dm"][s], c["arch"], self.nn, self.cond,
                         30, 3)
        self.g_cond = r["g_beta"] * self.N / self.n_global
        part = np.concatenate([r["g_nn"] * self.N / self.n_global, [r["sse"].sum(), float(r["n_failed"])]])
        return part, (self.g_cond if want_cond_grad else None)

    def adam_init(self, lr):
        self.lr, self.t = lr, 0
        self.m_n = np.zeros(self.P); self.v_n = np.zeros(self.P)
        self.m_c = np.zeros(self.N); self.v_c = np.zeros(self.N)

    def adam_apply(self, reduced):
        import cude_oracle as o
        self.t += 1
        self.nn, self.m_n, self.v_n = o.adam_update(self.nn, reduced[:self.P], self.m_n, self.v_n, self.t, self.lr)
        self.cond, self.m_c, self.v_c = o.adam_update(self.cond, self.g_cond, self.m_c, self.v_c, self.t, self.lr)
        return reduced[self.P] / self.n_global

    def get_params(self):
        return self.nn.copy(), self.cond.copy()

    # -- the rest of the Engine interface used by cude.parallel.saem_loop
    def set_params(self, nn=None, cond=None):
        if nn is not None:
            self.nn = np.array(nn, dtype=np.float64)
        if cond is not None:
            self.cond = np.array(cond, dtype=np.float64)

    def _solve(self, grad):
        c, s = self.c, slice(self.lo, self.hi)
        return self.co.cpep(c["tp"], c["G"][s], c["obs"][s], c["age"][s], c["t2dm"][s], c["arch"], self.nn,
                            self.cond, 30, 2, want_grad=grad)

    def forward(self, want_sse=False, want_traj=False):
        r = self._solve(False)
        return {"loss": r["loss"], "sse": r["sse"]}

    def loss_grad(self, want_cond_grad=True):
        r = self._solve(True)
        return r["loss"], r["g_nn"], r["g_beta"]

    def mh_estep(self, normals, uniforms, sigma, prior_mean, prior_sd, proposal_std, temperature=1.0, gamma=1.0):
        import cude_oracle as o
        c, s = self.c, slice(self.lo, self.hi)
        pop = o.CPepPopulation(c["tp"], c["G"][s], c["obs"][s], c["age"][s], c["t2dm"][s])
        self.cond, acc = o.mh_chain(self.nn, self.cond, pop, c["arch"], 30, sigma, prior_mean, prior_sd, proposal_std,
                                    temperature, gamma, normals, uniforms)
        return acc


def _rank_main(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "conditional-ude_amd"), os.path.join(here, "..", "oracle"), here):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from cude.parallel import ShardedTrainer, TorchCollective, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = make_cpep_case(n_total, (2, 6, 2))
    lo, hi = shard_bounds(n_total, world, rank)
    tr = ShardedTrainer(OracleEngine(case, lo, hi), TorchCollective(dist), transport="host")
    tr.sync_population_statistics()
    tr.adam_init(1e-2)
    losses = [tr.adam_step() for _ in range(4)]
    cond = tr.gather_conditional(n_total)
    nn, _ = tr.engine.get_params()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=losses, cond=cond, nn=nn)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_two_rank_gloo_training_matches_single_process(tmp_path, world):
    """world = 3 shards the 23 subjects unevenly (8 / 8 / 7); world = 8 is the node the headline is quoted on (3 / 3 / 3
    / 3 / 3 / 3 / 3 / 2 subjects)."""
    import torch.multiprocessing as mp
    import c_oracle as co
    import cude_oracle as o
    n_total = 23
    port = free_port()
    mp.spawn(_rank_main, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / f"rank{world - 1}.npz")
    assert np.array_equal(r0["losses"], r1["losses"]) and np.array_equal(r0["nn"], r1["nn"])
    # single-process reference: the same 4 Adam steps on the whole population
    c = make_cpep_case(n_total, (2, 6, 2))
    nn, beta = c["nn"].copy(), c["beta"].copy()
    m_n, v_n, m_b, v_b = np.zeros_like(nn), np.zeros_like(nn), np.zeros_like(beta), np.zeros_like(beta)
    ref_losses = []
    for t in range(1, 5):
        r = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], c["arch"], nn, beta, 30, 3)
        ref_losses.append(r["loss"])
        nn, m_n, v_n = o.adam_update(nn, r["g_nn"], m_n, v_n, t, 1e-2)
        beta, m_b, v_b = o.adam_update(beta, r["g_beta"], m_b, v_b, t, 1e-2)
    assert np.allclose(r0["losses"], ref_losses, rtol=1e-12)
    assert np.allclose(r0["nn"], nn, rtol=0, atol=1e-12)
    assert np.allclose(r0["cond"], beta, rtol=0, atol=1e-12)


# ------------------------------------------------------------------ sharded L-BFGS (second stage of _optimize)
def _lbfgs_rank_main(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "conditional-ude_amd"), os.path.join(here, "..", "oracle"), here):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from cude.parallel import ShardedTrainer, TorchCollective, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = make_cpep_case(n_total, (2, 4, 2))
    lo, hi = shard_bounds(n_total, world, rank)
    tr = ShardedTrainer(OracleEngine(case, lo, hi), TorchCollective(dist), transport="host")
    tr.sync_population_statistics()
    tr.adam_init(1e-2)
    for _ in range(2):
        tr.adam_step()
    res = tr.lbfgs(6)
    cond = tr.gather_conditional(n_total)
    nn, _ = tr.engine.get_params()
    np.savez(os.path.join(out_dir, f"lbfgs{rank}.npz"), f=res["f"], it=res["iterations"], calls=res["f_calls"],
             cond=cond, nn=nn)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_lbfgs_follows_the_single_process_iterates(tmp_path, world):
    """Adam x2 then L-BFGS x6 (`_optimize`, src/parameter-estimation.jl:170-183) with the subjects sharded over
    2 / 3 gloo ranks: every inner product of the L-BFGS recursion takes its conditional part through the collective
    (cude_lbfgs_minimize_sharded).  Must reproduce the run of one process holding all subjects -- same iteration and
    evaluation counts, iterates equal up to the summation order of the inner products."""
    import torch.multiprocessing as mp
    import c_oracle as co
    import cude_oracle as o
    from cude.engine import lbfgs_minimize
    n_total = 17
    port = free_port()
    mp.spawn(_lbfgs_rank_main, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "lbfgs0.npz"), np.load(tmp_path / f"lbfgs{world - 1}.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k                  # every rank walked the same path
    c = make_cpep_case(n_total, (2, 4, 2))
    nn, beta = c["nn"].copy(), c["beta"].copy()
    P = nn.size
    m_n, v_n, m_b, v_b = np.zeros_like(nn), np.zeros_like(nn), np.zeros_like(beta), np.zeros_like(beta)
    for t in range(1, 3):
        r = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], c["arch"], nn, beta, 30, 3)
        nn, m_n, v_n = o.adam_update(nn, r["g_nn"], m_n, v_n, t, 1e-2)
        beta, m_b, v_b = o.adam_update(beta, r["g_beta"], m_b, v_b, t, 1e-2)

    def fg(x):
        r = co.cpep(c["tp"], c["G"], c["obs"], c["age"], c["t2dm"], c["arch"], x[:P], x[P:], 30, 3)
        return r["loss"], np.concatenate([r["g_nn"], r["g_beta"]])
    one = lbfgs_minimize(fg, np.concatenate([nn, beta]), 6)
    assert int(r0["it"]) == one["iterations"] == 6 and int(r0["calls"]) == one["f_calls"]
    assert abs(float(r0["f"]) - one["f"]) <= 1e-10 * abs(one["f"])
    assert np.allclose(r0["nn"], one["x"][:P], rtol=0, atol=1e-9)
    assert np.allclose(r0["cond"], one["x"][P:], rtol=0, atol=1e-9)


# ------------------------------------------------------------------ 2-rank gloo SAEM (BASELINE configs[4] sharding)
_SAEM_KW = dict(sigma=0.4, prior_eta=-0.6, prior_omega=0.8, iterations=3, n_burnin_iterations=1, n_mcmc_steps=2,
                initial_mcmc_steps=3, proposal_std=0.3, m_step_iters=2)


def _saem_draws(n_total, lo, hi):
    def draws(it, steps):
        rng = np.random.default_rng(1000 + it)             # the same global stream on every rank, sliced per shard
        return rng.standard_normal((steps, n_total))[:, lo:hi], rng.random((steps, n_total))[:, lo:hi]
    return draws


def _saem_rank_main(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "conditional-ude_amd"), os.path.join(here, "..", "oracle"), here):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from cude.parallel import TorchCollective, saem_loop, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = make_cpep_case(n_total, (2, 4, 2))
    lo, hi = shard_bounds(n_total, world, rank)
    coll = TorchCollective(dist)
    res = saem_loop(OracleEngine(case, lo, hi), len(case["tp"]), case["nn"], collective=coll,
                    draws=_saem_draws(n_total, lo, hi), **_SAEM_KW)
    p_all = np.zeros(n_total)
    p_all[lo:hi] = res.p_individuals
    np.savez(os.path.join(out_dir, f"saem{rank}.npz"), nn=res.p_neural, p=coll.allreduce_sum(p_all), omega=res.Omega,
             sigma=res.sigma, eta=res.eta, nll=res.total_nll_values, acc=res.acceptance_rates)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_two_rank_gloo_saem_matches_single_process(tmp_path, world):
    """BASELINE configs[4] is SAEM on 8 GPUs: world = 8 shards the 11 subjects 2 / 2 / 2 / 1 / 1 / 1 / 1 / 1."""
    import torch.multiprocessing as mp
    from cude.parallel import saem_loop
    n_total = 11
    port = free_port()
    mp.spawn(_saem_rank_main, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "saem0.npz"), np.load(tmp_path / f"saem{world - 1}.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k                  # replicas stay identical
    case = make_cpep_case(n_total, (2, 4, 2))
    one = saem_loop(OracleEngine(case, 0, n_total), len(case["tp"]), case["nn"], collective=None,
                    draws=_saem_draws(n_total, 0, n_total), **_SAEM_KW)
    assert np.array_equal(r0["acc"], one.acceptance_rates)       # identical accept / reject decisions
    assert np.allclose(r0["p"], one.p_individuals, rtol=0, atol=1e-12)
    assert np.allclose(r0["nn"], one.p_neural, rtol=0, atol=1e-10)
    assert np.allclose(r0["nll"], one.total_nll_values, rtol=1e-11)
    assert abs(r0["sigma"] - one.sigma) < 1e-11 and abs(r0["omega"] - one.Omega) < 1e-12
    assert abs(r0["eta"] - one.eta) < 1e-13


def test_population_cache_is_keyed_on_content_and_bounded(monkeypatch):
    """cude.api builds one device population per CONTENT (a fresh `[model]` list per call used to miss the
    cache and leak an engine per call; different observations with an equal prefix used to hit a stale one)."""
    from cude import api

    class FakeEngine:
        live = 0

        def __init__(self, *a, **k):
            FakeEngine.live += 1

        def set_population_cpep(self, *a):
            pass

        def close(self):
            FakeEngine.live -= 1
    monkeypatch.setattr(api, "Engine", FakeEngine)
    api.clear_cache()
    tp = np.array([0.0, 30.0, 60.0, 90.0, 120.0])
    net = api.chain(4, 2)
    mk = lambda s: api.CPeptideConditionalUDEModel(5 + s + np.arange(5.0), tp, 40 + s, net, 1 + 0.1 * np.arange(5.0), False)
    models = [mk(s) for s in range(12)]
    data = np.stack([m.cpeptide for m in models])
    a = api._population([models[0]], tp, data[:1])
    assert api._population([models[0]], tp, data[:1]) is a and FakeEngine.live == 1      # fresh list, same content
    assert api._population([mk(0)], tp, data[:1]) is a                                   # equal model, new object
    d2 = data.copy()
    d2[7, 4] += 1e-3                                                                     # differs beyond any prefix
    b, c = api._population(models, tp, data), api._population(models, tp, d2)
    assert b is not c and api._population(models, tp, d2.copy()) is c
    with pytest.raises(ValueError):
        api._population(models, tp + 1.0, data)                                          # another time grid
    for s in range(12):                                                                  # LRU: evicted engines are closed
        api._population([models[s]], tp, data[s:s + 1])
    assert len(api._CACHE) == api._CACHE_MAX and FakeEngine.live == api._CACHE_MAX
    api.clear_cache()
    assert FakeEngine.live == 0


def test_philox_restatement_hits_the_published_known_answers():
    """Random123's known-answer vectors for philox4x32-10 (kat_vectors): the Python restatement in conftest.py, which
    tests/test_gpu_saem.py uses to predict the device-side draws bit for bit."""
    from conftest import philox4x32_10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_unequal_widths_are_the_zero_padded_equal_width_network():
    """`chain([w1, w2, ...], tanh)` (src/neural-network.jl:42-58): api.pad_network embeds it in the equal-width network
    the kernels are compiled for; outputs equal those of a general-width MLP written out here, for several shapes."""
    import cude_oracle as o
    from cude import api
    rng = np.random.default_rng(5)
    for nin, widths in ((2, [6, 3]), (2, [3, 6]), (3, [4, 2, 4]), (4, [5, 3]), (2, [7])):
        n = sum(w * f + w for w, f in zip(widths + [1], [nin] + widths))
        p = rng.standard_normal(n)
        net, P = api.pad_network(widths, p, input_dims=nin)
        assert net.arch == (nin, max(widths), len(widths)) and P.size == net.n_params
        assert np.array_equal(api.unpad_network(widths, P, input_dims=nin), p)
        assert np.count_nonzero(P) == n                                     # everything else is exactly zero
        x, h, at, fan = rng.standard_normal((nin, 9)), None, 0, nin
        h = x
        for k, w in enumerate(widths + [1]):
            Wm, b = p[at:at + w * fan].reshape(fan, w).T, p[at + w * fan:at + w * fan + w]
            at, fan = at + w * fan + w, w
            z = Wm @ h + b[:, None]
            h = np.tanh(z) if k < len(widths) else np.log1p(np.exp(z))
        assert np.max(np.abs(h[0] - o.mlp(np, x, P, net.arch))) < 1e-14
    with pytest.raises(ValueError):
        api.pad_network([6, 3], np.zeros(5))
    # chain(widths) itself: carried as the padded network + a mask of its live parameters
    net = api.chain([6, 3], "tanh")
    assert net.arch == (2, 6, 2) and net.widths == [6, 3] and net.n_params == 67 and int(net.mask.sum()) == 43
    p0 = api.init_params(net, np.random.default_rng(1))
    assert p0.size == 67 and np.array_equal(p0 != 0.0, net.mask == 1.0)
    assert api.chain([4, 4], "tanh").mask is None and api.chain(4, 2, "tanh").widths is None
    relu = api.chain([6, 3], "relu")            # (round 5) another function on unequal widths: the general form, UNPADDED layout
    assert relu.general and relu.mask is None and relu.n_params == 43 and api.init_params(relu, np.random.default_rng(1)).size == 43


def test_script_utilities_stratified_split_and_argmedian():
    """src/utils.jl:15-31,43-45 as the reference's scripts use them (02-conditional.jl:19,463)."""
    from cude import api
    types = np.array(["NGT"] * 37 + ["IGT"] * 25 + ["T2DM"] * 20)
    np.random.default_rng(0).shuffle(types)
    train, test = api.stratified_split(np.random.default_rng(1), types, 0.7)
    assert np.all(np.diff(train) > 0) and np.array_equal(np.sort(np.concatenate([train, test])), np.arange(82))
    for ty, n in (("NGT", 26), ("IGT", 18), ("T2DM", 14)):         # round(0.7 * count): 25.9 -> 26, 17.5 -> 18 (half to even), 14
        assert np.sum(types[train] == ty) == n
    again, _ = api.stratified_split(np.random.default_rng(1), types, 0.7)
    assert np.array_equal(again, train)
    assert api.argmedian([3.0, 9.0, 1.0, 4.0, 7.0]) == 3 and api.argmedian([2.0, 8.0]) == 0     # even length: first of the two nearest
    # map_objective = -(ll + log prior), src/saem.jl:68-72
    v = api.map_objective(0.3, 1.7, 5, 0.6, 0.9, prior_individual=-0.2)
    ll = -2.5 * np.log(0.36) - 1.7 / 0.72
    lp = -0.5 * (0.5 / 0.9) ** 2 - np.log(0.9) - 0.5 * np.log(2 * np.pi)
    assert abs(v + ll + lp) < 1e-14


def test_generate_data_mirrors_the_reference_generator():
    """generate_data / get_group_parameters / lsup! (suppression/src/suppression_model.jl:16-20,33-63; suppression.jl:27-36):
    noise-free output solves the ground-truth model (scipy DOP853 at 1e-12), the noise enters as the reference writes it,
    and the oracle's independent restatement gives the same noise-free trajectories for the same parameters."""
    from scipy.integrate import solve_ivp
    from cude import api
    tp = np.linspace(0.0, 30.0, 8)
    means, sizes = [0.5, 2.5, 5.0, 7.5, 10.0, 12.5], [3, 2, 2, 1, 2, 3]
    data, gt = api.generate_data(means, sizes, tp, rng=np.random.default_rng(4))
    assert data.shape == (3, 8, 13) and gt.shape == (13,) and np.all(data[:, 0] == np.array([10.0, 0.0, 0.0])[:, None])
    # the same draws again: group parameters are the first thing drawn in every group
    rng = np.random.default_rng(4)
    col = 0
    for mean, size in zip(means, sizes):
        p = api.get_group_parameters(mean, size, rng=rng)
        assert np.all(p >= 0.05) and np.array_equal(p[3], gt[col:col + size])
        for j in range(size):
            rng.standard_normal((3, 8)); rng.standard_normal((3, 8))          # the two noise fields of this subject
            sol = solve_ivp(lambda t, u: api.lsup(u, p[:, j]), (0.0, 30.0), [10.0, 0.0, 0.0], method="DOP853", t_eval=tp,
                            rtol=1e-12, atol=1e-14).y
            assert np.max(np.abs(sol - data[:, :, col])) < 1e-7
            col += 1
    noisy, gt2 = api.generate_data(means, sizes, tp, noise_multiplicative=0.1, rng=np.random.default_rng(4))
    assert np.array_equal(gt2, gt) and np.all(noisy >= 0.0)
    ratio = noisy[:, 1:] / data[:, 1:] - 1.0                                   # = 0.1 * randn wherever nothing was clamped
    assert 0.07 < np.std(ratio) < 0.13 and abs(np.mean(ratio)) < 0.03


# ------------------------------------------------------------------ the exchange's attach protocol (host logic, no GPU)
class _MailboxEngine:
    """Stand-in for the three exchange entry points: kind k of this rank's mailbox memory 'works' unless listed in `bad`
    (export fails: allocation / IPC export) or in `bad_attach` (a peer cannot use it: only attach finds out)."""

    def __init__(self, rank, bad=(), bad_attach=()):
        self.rank, self.bad, self.bad_attach = rank, set(bad), set(bad_attach)
        self.level, self.kind, self.attached, self.log = 0, None, False, []      # level: csrc/cude_comm.hip xchg_next_kind

    def xchg_export(self, n_ranks, rank):
        from cude._lib import CudeError
        k = self.level
        while k in self.bad:
            k += 1
        if k > 2:
            self.level = 3
            raise CudeError(-5, "exchange mailbox: every kind of device memory has been tried on this context")
        self.kind = k
        self.log.append(("export", self.kind))
        return bytes([0xC0 + self.kind]) * 128

    def xchg_attach(self, handles, timeout_s):
        from cude._lib import CudeError
        assert len(handles) > 1 and all(len(h) == 128 for h in handles)
        if any(h[0] == 0 for h in handles) or self.kind in self.bad_attach:
            self.level, self.kind = self.level + 1, None          # (the library releases a failed attach and moves one level on)
            raise CudeError(-1, "hipIpcOpenMemHandle / self-test failed")
        self.attached = True

    def xchg_detach(self):
        if self.kind is not None:
            self.level += 1
        self.kind, self.attached = None, False
        self.log.append(("detach",))


def _attach_rank_main(rank, world, port, scenario, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "conditional-ude_amd"))
    import torch.distributed as dist
    from cude.parallel import TorchCollective, attach_exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bad, bad_attach = scenario[rank]
    eng = _MailboxEngine(rank, bad, bad_attach)
    ok, why = attach_exchange(eng, TorchCollective(dist), 1.0)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ok=ok, kind=-1 if eng.kind is None else eng.kind,
             attached=eng.attached, exports=sum(1 for e in eng.log if e[0] == "export"), why=str(why))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario,ok,kinds,exports", [
    ({0: ((), ()), 1: ((), ()), 2: ((), ())}, True, (0, 0, 0), (1, 1, 1)),              # first kind works everywhere
    ({0: ((), ()), 1: ((), (0,)), 2: ((), ())}, True, (1, 1, 1), (2, 2, 2)),            # one rank cannot USE kind 0: all move on
    ({0: ((0,), ()), 1: ((), (0, 1)), 2: ((), ())}, True, (2, 2, 2), None),             # mixed: export falls through, attach fails twice
    ({0: ((), ()), 1: ((0, 1, 2), ()), 2: ((), ())}, False, (-1, -1, -1), None),        # a rank with no kind left: released everywhere
    ({0: ((), (0, 1, 2)), 1: ((), ()), 2: ((), ())}, False, (-1, -1, -1), (3, 3, 3)),   # every kind fails at attach on one rank
], ids=["first", "second", "mixed", "export-exhausted", "attach-exhausted"])
def test_exchange_attach_protocol_agrees_over_the_ranks(tmp_path, scenario, ok, kinds, exports):
    """cude/parallel.py attach_exchange (the loop bench.py and CUDEHip.jl attach_exchange! run; include/cude.h "cude_xchg_*"):
    whatever fails on whichever rank, every rank makes the same sequence of collective calls, and they end either ALL
    attached on their next common attempt or ALL released."""
    import torch.multiprocessing as mp
    world, port = 3, free_port()
    mp.spawn(_attach_rank_main, args=(world, port, scenario, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    assert all(bool(x["ok"]) == ok and bool(x["attached"]) == ok for x in r)
    assert tuple(int(x["kind"]) for x in r) == kinds
    if exports is not None:
        assert tuple(int(x["exports"]) for x in r) == exports
    if not ok:
        assert all(str(x["why"]) not in ("", "None") for x in r)
