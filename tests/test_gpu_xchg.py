"""The peer-write exchange (cude_xchg_*, conditional-ude_amd/csrc/cude_xchg.h + cude_comm.hip) on the GPU.

The reference has no distributed path (its only parallelism over trajectories is EnsembleThreads,
suppression/src/suppression_model.jl:113,123); sharding subjects over GPUs needs ONE sum of P+2 doubles per optimiser
step, and the exchange forms it inside the reduction kernels.  What a one-GPU box can show:
  * one rank: the path with the exchange (captured graphs included) is bit-identical to the plain one;
  * two (and three) PROCESSES on the one GPU, mailboxes shared through HIP IPC: every rank ends with bit-identical losses and
    network parameters, equal to a single engine on the whole population to rounding -- single steps, captured runs,
    the time-split and mixed launch paths, an L2 term, the suppression model, the L-BFGS stage, screening;
  * two contexts of ONE process (threads): the same, through plain addresses instead of IPC handles;
  * a peer that never shows up: the bounded wait ends, the call returns CUDE_ERR_COMM, nothing hangs.
The protocol's orderings for 3 and 8 ranks run on host threads in tests/test_xchg_protocol.py."""
import os
import threading

import numpy as np
import pytest

from conftest import make_cpep_case, make_supp_case, free_port

pytestmark = pytest.mark.gpu


def _engine(model, arch, c, lo, hi, lam=0.0, n_state=3, path=None):
    from cude.engine import Engine
    eng = Engine(model, arch, n_steps=30, n_state=n_state, lam=lam) if model == "cpep" else Engine("supp", arch, n_steps=30, lam=lam)
    if path:
        eng.set_option("cpep_path", path)
    return eng


def _upload(eng, model, c, lo, hi):
    if model == "cpep":
        eng.set_population_cpep(c["tp"], c["G"][lo:hi], c["obs"][lo:hi], c["age"][lo:hi], c["t2dm"][lo:hi])
        eng.set_params(c["nn"], c["beta"][lo:hi])
    else:
        eng.set_population_supp(c["tp"], c["data"][:, :, lo:hi])
        eng.set_params(c["nn"], c["theta"][lo:hi])


def _train(eng):
    """forward, gradient, single steps, a captured run (8 + 4 + 1 iterations), single steps again"""
    fwd = eng.forward()["loss"]
    loss, g_nn, _ = eng.loss_grad()
    eng.adam_init(1e-2)
    losses = [eng.adam_step() for _ in range(3)]
    losses += list(eng.adam_run(13))
    losses += [eng.adam_step() for _ in range(2)]
    nn, cond = eng.get_params()
    return dict(fwd=fwd, loss=loss, g_nn=g_nn, losses=np.array(losses), nn=nn, cond=cond)


@pytest.mark.parametrize("model,arch,lam", [("cpep", (2, 6, 2), 0.0), ("supp", (4, 3, 5), 0.01)])
def test_one_rank_exchange_is_bit_identical_to_the_plain_path(model, arch, lam):
    c = make_cpep_case(300, arch) if model == "cpep" else make_supp_case(300, arch)

    def run(with_xchg):
        eng = _engine(model, arch, c, 0, 300, lam)
        if with_xchg:
            eng.xchg_attach([eng.xchg_export(1, 0)], 5.0)
            assert eng.xchg_info()[:2] == (1, 0) and eng.xchg_info()[3] == 0
        _upload(eng, model, c, 0, 300)
        out = _train(eng)
        red = eng.allreduce_host([1.5, -2.5, 7.0])
        eng.close()
        return out, red
    (a, _), (b, red) = run(False), run(True)
    assert np.array_equal(red, [1.5, -2.5, 7.0])
    for k in a:
        assert np.array_equal(a[k], b[k]), k


# ------------------------------------------------------------------------------------------ two processes, one GPU
def _rank(rank, world, port, cfg, out_dir):
    import torch  # noqa: F401  (first: one shared HIP runtime)
    import torch.distributed as dist
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.join(os.path.dirname(here), "conditional-ude_amd"), os.path.join(os.path.dirname(here), "oracle")]
    from cude.parallel import ShardedTrainer, TorchCollective, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, arch, lam, n_total, path = cfg
    c = make_cpep_case(n_total, arch) if model == "cpep" else make_supp_case(n_total, arch)
    lo, hi = shard_bounds(n_total, world, rank)
    eng = _engine(model, arch, c, lo, hi, lam, path=path)
    ShardedTrainer.attach_xchg(eng, TorchCollective(dist), timeout_s=30.0)
    assert eng.xchg_info()[:2] == (world, rank)
    _upload(eng, model, c, lo, hi)                      # (the global subject count / scale go through the exchange here)
    out = _train(eng)
    tr = ShardedTrainer(eng, TorchCollective(dist), transport="xchg")
    out["cond_all"] = tr.gather_conditional(n_total)
    # the L-BFGS stage on the sharded population: every inner product's local part is summed through the exchange
    res = tr.lbfgs(6)
    out["lbfgs_f"] = res["f"]
    out["nn_lbfgs"], _ = eng.get_params()
    # screening: candidate losses are summed over the ranks before the top-k (cude_multistart_forward)
    rng = np.random.default_rng(5)
    nn_sets = c["nn"][None, :] * (1.0 + 0.1 * rng.standard_normal((4, c["nn"].size)))
    key = "beta" if model == "cpep" else "theta"
    cond_sets = np.repeat(c[key][None, lo:hi], 4, axis=0)
    out["ms"] = eng.multistart_forward(nn_sets, cond_sets)
    out["timeouts"] = eng.xchg_info()[3]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


CASES = [("cpep", (2, 6, 2), 0.0, 333, None),           # small population: the time-split path, two reduction launches
         ("cpep", (2, 6, 2), 0.0, 333, "1"),            # one lane per subject: one reduction launch
         ("cpep", (2, 4, 2), 0.0, 700, "3:3:5"),        # mixed launch: three reduction launches, the middle one accumulates
         ("supp", (4, 3, 5), 0.01, 200, None)]          # L2 term: the state advance rides in the L2 kernel


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("cfg", CASES, ids=["split", "one-lane", "mixed", "supp-l2"])
def test_processes_sharing_one_gpu_train_as_one_engine(cfg, world, tmp_path):
    import torch.multiprocessing as mp
    from cude.engine import Engine  # noqa: F401
    model, arch, lam, n_total, path = cfg
    if world == 3 and cfg is not CASES[0] and cfg is not CASES[3]:
        pytest.skip("three ranks: the time-split and the suppression case")
    port = free_port()
    mp.spawn(_rank, args=(world, port, cfg, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    # every rank formed the same sums in the same order: identical bits
    for k in ("fwd", "loss", "g_nn", "losses", "nn", "lbfgs_f", "nn_lbfgs", "ms"):
        assert all(np.array_equal(r[0][k], r[j][k]) for j in range(1, world)), k
    assert all(r[j]["timeouts"] == 0 for j in range(world))
    # ... and the same numbers as one engine on the whole population, to the rounding of a different summation order
    c = make_cpep_case(n_total, arch) if model == "cpep" else make_supp_case(n_total, arch)
    eng = _engine(model, arch, c, 0, n_total, lam, path=path if path != "3:3:5" else None)
    _upload(eng, model, c, 0, n_total)
    ref = _train(eng)
    nn_o, cond_o, obj = eng.train_restarts(ref["nn"][None, :], ref["cond"][None, :], 0, 1e-3, 6)
    rng = np.random.default_rng(5)
    nn_sets = c["nn"][None, :] * (1.0 + 0.1 * rng.standard_normal((4, c["nn"].size)))
    ms = eng.multistart_forward(nn_sets, np.repeat(c["beta" if model == "cpep" else "theta"][None, :], 4, axis=0))
    eng.close()
    assert abs(r[0]["fwd"] - ref["fwd"]) <= 1e-12 * abs(ref["fwd"])
    assert np.allclose(r[0]["g_nn"], ref["g_nn"], rtol=0, atol=1e-12 * np.max(np.abs(ref["g_nn"])))
    assert np.allclose(r[0]["losses"], ref["losses"], rtol=1e-10)
    assert np.allclose(r[0]["nn"], ref["nn"], rtol=0, atol=1e-10)
    assert np.allclose(r[0]["cond_all"], ref["cond"], rtol=0, atol=1e-10)
    assert np.allclose(r[0]["ms"], ms, rtol=1e-12)
    # L-BFGS amplifies rounding: on the c-peptide objective six iterations stay together to 1e-6; on the
    # suppression objective the paths of any two correct statements part within ten iterations (0.1 % here)
    assert abs(r[0]["lbfgs_f"] - obj[0]) <= (1e-6 if model == "cpep" else 1e-2) * abs(obj[0])
    assert r[0]["lbfgs_f"] < r[0]["losses"][-1]                      # (both include the L2 term)


# ------------------------------------------------------------------------------------------ two contexts, one process
def _two_contexts(_rank_index, out_dir):
    """(a process of its own: which hardware queue a stream lands on depends on every stream the process has made before,
    and two contexts whose streams share one queue cannot wait for each other -- in a long test session that happened
    about one run in four; ranks of a real job are processes, and a fresh one is what they look like)"""
    import torch  # noqa: F401  (first: one shared HIP runtime)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.join(os.path.dirname(here), "conditional-ude_amd"), os.path.join(os.path.dirname(here), "oracle")]
    from cude.engine import Engine
    arch, n_total = (2, 6, 2), 400
    c = make_cpep_case(n_total, arch)
    engs = [Engine("cpep", arch, n_steps=30, n_state=3) for _ in range(2)]
    handles = [e.xchg_export(2, k) for k, e in enumerate(engs)]
    out, err = [None, None], []

    def work(k):
        try:
            e = engs[k]
            e.xchg_attach(handles, 30.0)
            lo, hi = (0, 230) if k == 0 else (230, n_total)
            _upload(e, "cpep", c, lo, hi)
            e.adam_init(1e-2)
            losses = [e.adam_step() for _ in range(2)] + list(e.adam_run(9))
            out[k] = (np.array(losses), e.get_params()[0])
        except Exception as exc:  # noqa: BLE001
            err.append(exc)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    for e in engs:
        e.close()
    assert not err, err
    np.savez(os.path.join(out_dir, "two_contexts.npz"), l0=out[0][0], l1=out[1][0], nn0=out[0][1], nn1=out[1][1])


def test_two_contexts_of_one_process_exchange_through_plain_addresses(tmp_path):
    """Ranks of one process cannot open their own IPC handles: the handle carries the address.  Two threads drive two
    contexts (ctypes releases the interpreter lock inside the calls, and each call that waits for a peer needs the
    other thread to be inside its own)."""
    import torch.multiprocessing as mp
    from cude.engine import Engine
    arch, n_total = (2, 6, 2), 400
    c = make_cpep_case(n_total, arch)
    mp.spawn(_two_contexts, args=(str(tmp_path),), nprocs=1, join=True)
    r = np.load(tmp_path / "two_contexts.npz")
    out = [(r["l0"], r["nn0"]), (r["l1"], r["nn1"])]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    ref = Engine("cpep", arch, n_steps=30, n_state=3)
    _upload(ref, "cpep", c, 0, n_total)
    ref.adam_init(1e-2)
    want = [ref.adam_step() for _ in range(2)] + list(ref.adam_run(9))
    ref.close()
    assert np.allclose(out[0][0], want, rtol=1e-10)


def test_a_missing_peer_ends_in_an_error_not_in_a_hang():
    """Rank 0 of two attaches alone: the self-test's wait gives up after the time limit, cude_xchg_attach returns
    CUDE_ERR_COMM and the context works on as a single rank."""
    import time
    from cude.engine import Engine, CudeError
    arch = (2, 4, 2)
    c = make_cpep_case(100, arch)
    a, b = Engine("cpep", arch, n_steps=30, n_state=2), Engine("cpep", arch, n_steps=30, n_state=2)
    handles = [a.xchg_export(2, 0), b.xchg_export(2, 1)]
    t0 = time.perf_counter()
    with pytest.raises(CudeError) as ei:
        a.xchg_attach(handles, 0.3)
    dt = time.perf_counter() - t0
    assert ei.value.status == -5 and "exchange" in str(ei.value) and 0.25 < dt < 10.0
    assert a.xchg_info()[0] == 1
    _upload(a, "cpep", c, 0, 100)
    assert np.isfinite(a.loss_grad()[0])
    a.close()
    b.close()


# ------------------------------------------------------------------------------------------ the attach protocol's retry
def _rank_retry(rank, world, port, masks, out_dir):
    import torch  # noqa: F401
    import torch.distributed as dist
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.join(os.path.dirname(here), "conditional-ude_amd"), os.path.join(os.path.dirname(here), "oracle")]
    from cude.parallel import TorchCollective, attach_exchange, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arch, n_total = (2, 6, 2), 333
    c = make_cpep_case(n_total, arch)
    lo, hi = shard_bounds(n_total, world, rank)
    eng = _engine("cpep", arch, c, lo, hi)
    eng.set_option("xchg_fail_kinds", masks[rank])        # (test hook: this rank reports the listed kinds as failed)
    log = []
    ok, why = attach_exchange(eng, TorchCollective(dist), timeout_s=30.0, log=log.append)
    out = dict(ok=ok, why=str(why), attempts=len(log) + (1 if ok else 0), info=np.array(eng.xchg_info()))
    if not ok:                                            # released everywhere: the context works on as a single rank
        lo, hi = 0, n_total
    _upload(eng, "cpep", c, lo, hi)
    out.update(_train(eng))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("masks,kind,attempts", [((0, 1), 1, 2), ((3, 1), 0, 3), ((0, 0), 3, 1)],
                         ids=["second-kind", "third-kind", "first-kind"])
def test_a_failing_memory_kind_moves_every_rank_to_the_next(masks, kind, attempts, tmp_path):
    """One rank cannot use a kind of mailbox memory (forced through option xchg_fail_kinds; on a real node: a peer on
    another device that cannot open or write it): the ranks agree, ALL detach -- the one whose attach succeeded too -- and
    attach again with the next kind.  Afterwards they train as in test_processes_sharing_one_gpu_train_as_one_engine."""
    import torch.multiprocessing as mp
    from cude.engine import Engine  # noqa: F401
    port = free_port()
    mp.spawn(_rank_retry, args=(2, port, masks, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    for k in range(2):
        assert r[k]["ok"] and r[k]["attempts"] == attempts
        assert tuple(r[k]["info"][:2]) == (2, k) and r[k]["info"][2] == kind and r[k]["info"][3] == 0
    for k in ("fwd", "loss", "g_nn", "losses", "nn"):
        assert np.array_equal(r[0][k], r[1][k]), k
    arch, n_total = (2, 6, 2), 333
    c = make_cpep_case(n_total, arch)
    eng = _engine("cpep", arch, c, 0, n_total)
    _upload(eng, "cpep", c, 0, n_total)
    ref = _train(eng)
    eng.close()
    assert np.allclose(r[0]["losses"], ref["losses"], rtol=1e-10)


def test_no_usable_memory_kind_releases_the_exchange_on_every_rank(tmp_path):
    """All three kinds fail on one rank: attach_exchange returns False on BOTH ranks with nothing attached (the caller's
    cue to fall back to RCCL), and the contexts work on as single ranks."""
    import torch.multiprocessing as mp
    from cude.engine import Engine  # noqa: F401
    port = free_port()
    mp.spawn(_rank_retry, args=(2, port, (7, 0), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    for k in range(2):
        assert not r[k]["ok"] and r[k]["info"][0] == 1 and r[k]["attempts"] == 3
    assert "xchg_fail_kinds" in str(r[0]["why"])
    assert np.array_equal(r[0]["losses"], r[1]["losses"]) and np.all(np.isfinite(r[0]["losses"]))


def test_a_wait_that_gives_up_fails_the_call_that_synchronises_behind_it():
    """A peer stops calling (here: rank 1 simply does not make the call): rank 0's screening call -- whose result carries
    no NaN-able loss of its own but per-set sums -- returns CUDE_ERR_COMM instead of scoring the sets +Inf."""
    from cude.engine import Engine, CudeError
    arch, n_total = (2, 4, 2), 200
    c = make_cpep_case(n_total, arch)
    engs = [Engine("cpep", arch, n_steps=30, n_state=2) for _ in range(2)]
    handles = [e.xchg_export(2, k) for k, e in enumerate(engs)]
    err = []

    def attach(k):
        try:
            engs[k].xchg_attach(handles, 0.5)
            lo, hi = (0, 100) if k == 0 else (100, n_total)
            _upload(engs[k], "cpep", c, lo, hi)
        except Exception as exc:  # noqa: BLE001
            err.append(exc)
    th = [threading.Thread(target=attach, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not err, err
    nn_sets = np.repeat(c["nn"][None, :], 3, axis=0)
    with pytest.raises(CudeError) as ei:
        engs[0].multistart_forward(nn_sets, np.repeat(c["beta"][None, :100], 3, axis=0))
    assert ei.value.status == -5 and engs[0].xchg_info()[3] == 1
    with pytest.raises(CudeError):                      # ... and a gradient evaluation alike
        engs[0].loss_grad()
    for e in engs:
        e.close()
