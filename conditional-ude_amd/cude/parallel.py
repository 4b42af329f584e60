"""Subject sharding across GPUs (one process per GPU) for the population training step.

The reference has no distributed path; its only parallelism over trajectories is EnsembleThreads
(suppression/src/suppression_model.jl:113,123).  Subjects are independent, so they shard trivially; the
single coupled quantity is the shared network, whose gradient (with the loss sum and the failure count:
P+2 doubles) is summed across ranks once per optimiser step.  Two transports:

  * "rccl": the sum happens inside libcude_hip.so on the context's stream (cude_comm_init +
    cude_adam_step); the host only ships the 128-byte unique id once.
  * "xchg": the peer-write exchange (cude_xchg_*): the reduction kernels of every rank write their P+2 doubles into
    every peer's mailbox over xGMI and add what arrives in rank order -- no collective call at all, captured graphs
    keep working, sums are bitwise reproducible; the host only ships 128-byte handles once.
  * "host": the host sums the P+2 doubles with ANY collective it likes (torch.distributed gloo/nccl,
    MPI) between cude_loss_grad_partial and cude_adam_apply.

`engine` is anything with the Engine interface (tests drive this logic on CPU with a stand-in engine
over the gloo backend; production uses cude.engine.Engine).
"""
import numpy as np

from ._lib import XCHG_HANDLE_BYTES, CudeError


def shard_bounds(n_subjects, world_size, rank):
    """Contiguous block partition: the first (n mod world) ranks get one extra subject."""
    base, extra = divmod(int(n_subjects), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class TorchCollective:
    """Sum all-reduce of small host vectors through torch.distributed (any backend)."""

    def __init__(self, dist, device=None):
        self.dist, self.device = dist, device

    @property
    def world_size(self):
        return self.dist.get_world_size()

    @property
    def rank(self):
        return self.dist.get_rank()

    def allreduce_sum(self, vec):
        import torch
        t = torch.as_tensor(np.asarray(vec, dtype=np.float64).copy())
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allreduce(self, vec, op=0):
        """op 0 = sum, 1 = max (the cude_reduce_fn convention)."""
        import torch
        t = torch.as_tensor(np.asarray(vec, dtype=np.float64).copy())
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_bytes(self, payload):
        """Every rank's `payload` (equal lengths) in rank order."""
        import torch
        mine = torch.tensor(list(payload), dtype=torch.uint8)
        if self.device is not None:
            mine = mine.to(self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world_size)]
        self.dist.all_gather(out, mine)
        return [bytes(t.cpu().tolist()) for t in out]

    def broadcast_bytes(self, payload, src=0):
        import torch
        n = len(payload) if payload is not None else 0
        nt = torch.tensor([n], dtype=torch.int64)
        if self.device is not None:
            nt = nt.to(self.device)
        self.dist.broadcast(nt, src)
        buf = torch.zeros(int(nt.item()), dtype=torch.uint8)
        if self.rank == src:
            buf = torch.tensor(list(payload), dtype=torch.uint8)
        if self.device is not None:
            buf = buf.to(self.device)
        self.dist.broadcast(buf, src)
        return bytes(buf.cpu().tolist())


def attach_exchange(engine, collective, timeout_s=20.0, log=None):
    """The attach protocol of include/cude.h "cude_xchg_*": every rank exports its mailbox, the handles are all-gathered
    through the host collective, every rank maps its peers (collective: ends with a self-test), and the ranks AGREE on
    the outcome -- if any rank failed, all detach and go again with the next kind of mailbox memory (uncached ->
    fine-grained -> ordinary device memory).  Whether a peer on another device can open and write a given kind is only
    known at attach, and only to the rank that tried.  Returns (True, None) with the exchange attached on every rank, or
    (False, reason) with it released on every rank (the caller falls back to RCCL).  Every rank makes the same sequence
    of collective calls whatever fails locally."""
    why = None
    for _ in range(3):
        mine, err = bytes(XCHG_HANDLE_BYTES), None
        try:
            mine = engine.xchg_export(collective.world_size, collective.rank)
        except CudeError as exc:       # (no kind left, or none could be allocated / exported: peers get zeros)
            err = f"cude_xchg_export: {exc}"
        handles = collective.allgather_bytes(mine)
        if err is None:
            try:
                engine.xchg_attach(handles, timeout_s)
            except CudeError as exc:
                err = f"cude_xchg_attach: {exc}"
        exhausted = err is not None and err.startswith("cude_xchg_export")
        flags = collective.allreduce([0.0 if err is None else 1.0, 1.0 if exhausted else 0.0], op=1)
        if flags[0] == 0.0:
            return True, None
        why = err or why or "another rank could not attach"
        if log is not None:
            log(f"rank {collective.rank}: exchange attempt failed ({err or 'on another rank'})")
        engine.xchg_detach()
        if flags[1] != 0.0:
            break
    return False, why


class ShardedTrainer:
    """Data-parallel Adam training of one cUDE over subject shards.

    Each rank passes ITS shard's engine (population already uploaded) and the collective."""

    def __init__(self, engine, collective, transport="host"):
        self.engine, self.coll, self.transport = engine, collective, transport
        if transport not in ("host", "rccl", "xchg"):
            raise ValueError("transport must be 'host', 'rccl' or 'xchg'")

    @staticmethod
    def attach_rccl(engine_cls, engine, collective):
        """Bootstrap the built-in communicator: rank 0 creates the id, everyone joins.  Must run before the
        population is uploaded (the global subject count is all-reduced there)."""
        uid = engine_cls.comm_unique_id() if collective.rank == 0 else None
        uid = collective.broadcast_bytes(uid, 0)
        engine.comm_init(collective.world_size, collective.rank, uid)

    @staticmethod
    def attach_xchg(engine, collective, timeout_s=20.0):
        """Bootstrap the peer-write exchange (attach_exchange below); raises if no kind of mailbox memory works on every
        rank.  Before the population is uploaded, as attach_rccl."""
        ok, why = attach_exchange(engine, collective, timeout_s)
        if not ok:
            raise CudeError(f"peer-write exchange unavailable: {why}")

    def sync_population_statistics(self, scale_sums=None):
        """host transport: establish the global subject count (and SUPP's scale = mean_i max_t data)."""
        if self.transport != "host":
            return
        if scale_sums is None:
            tot = self.coll.allreduce_sum([float(self.engine.N)])
            self.engine.set_global_subjects(tot[0])
            self.n_global = float(tot[0])
        else:
            tot = self.coll.allreduce_sum(list(scale_sums) + [float(self.engine.N)])
            self.engine.set_global_subjects(tot[3], tot[:3] / tot[3])
            self.n_global = float(tot[3])

    def lbfgs(self, maxiters, lam=None):
        """Second stage of `_optimize` (src/parameter-estimation.jl:179-180: L-BFGS + BackTracking after Adam) on the
        sharded population, starting from the engine's current parameters; leaves the result in the engine and
        returns dict(f, iterations, f_calls, converged) -- identical on every rank.

        rccl: cude_train_restarts (one restart) reduces losses, network gradients and the conditional part of every
        inner product over its communicator.  host: the library's L-BFGS state machine
        (cude_lbfgs_minimize_sharded) with this trainer's collective as the reducer; every evaluation is
        cude_loss_grad_partial + one sum of P+2 doubles.  `lam` defaults to the engine's configured L2 weight (the host
        transport adds the term here, as cude_adam_apply does on the device), so both transports optimise the same
        objective.  On the rccl transport the library does not hand out its counters: iterations, f_calls and converged
        are None there."""
        if lam is None:
            lam = getattr(self.engine, "lam", 0.0)
        nn, cond = self.engine.get_params()
        P = nn.size
        if self.transport in ("rccl", "xchg"):
            nn_o, cond_o, obj = self.engine.train_restarts(nn[None, :], cond[None, :], 0, 1e-3, int(maxiters))
            self.engine.set_params(nn_o[0], cond_o[0])
            return dict(f=float(obj[0]), iterations=None, f_calls=None, converged=None)
        from .engine import lbfgs_minimize_sharded
        n_glob = self.n_global

        def fg(x):
            self.engine.set_params(x[:P], x[P:])
            part, g_cond = self.engine.loss_grad_partial(want_cond_grad=True)
            red = self.coll.allreduce_sum(part)
            f = red[P] / n_glob + lam * float(x[:P] @ x[:P]) if red[P + 1] == 0 and np.isfinite(red[P]) else np.inf
            return f, np.concatenate([red[:P] + 2.0 * lam * x[:P], g_cond])
        r = lbfgs_minimize_sharded(fg, np.concatenate([nn, cond]), P, self.coll.allreduce, maxiters)
        self.engine.set_params(r["x"][:P], r["x"][P:])
        return {k: r[k] for k in ("f", "iterations", "f_calls", "converged")}

    def adam_init(self, lr, **kw):
        self.engine.adam_init(lr, **kw)

    def adam_step(self):
        """One optimiser iteration; returns the GLOBAL loss (identical on every rank)."""
        if self.transport in ("rccl", "xchg"):
            return self.engine.adam_step()
        part, _ = self.engine.loss_grad_partial()
        return self.engine.adam_apply(self.coll.allreduce_sum(part))

    def gather_conditional(self, n_total):
        """All ranks' conditional parameters in global subject order (host side, for checkpoints)."""
        _, cond = self.engine.get_params()
        out = np.zeros(n_total)
        lo, hi = shard_bounds(n_total, self.coll.world_size, self.coll.rank)
        out[lo:hi] = cond
        return self.coll.allreduce_sum(out)


def saem_loop(engine, n_obs, initial_neural_params, *, collective=None, sigma=1.0, prior_eta=0.0, prior_omega=1.0,
              iterations=500, n_burnin_iterations=100, proposal_std=0.1, proposal_std_bounds=(1e-3, 1.0), alpha=0.7,
              n_mcmc_steps=1, initial_mcmc_steps=None, target_acceptance_rate=0.25, initial_temperature=10.0,
              temperature_decay=0.05, omega_learning_rate=0.04, rng=None, draws=None, m_step_iters=5, m_step_lr=1e-2,
              device_seed=None, subject_offset=0):
    """`SAEM` of src/saem.jl:134-237 over the subjects resident in `engine` -- the whole population
    (collective=None) or this rank's shard of it (BASELINE configs[4]: the E-step needs no communication).

    Per iteration the ranks exchange, as sums of small host vectors: [accepted, log-likelihood, sum p, sum p^2]
    after the E-step (acceptance rate, total NLL, Omega <- var(p), eta <- mean(p), :204-205) and the P+2 doubles
    of the network gradient in each of the `m_step_iters` Adam iterations of the M-step (:118-131); every rank
    then applies identical updates to its replica of (network, sigma, Omega, eta, proposal_std).
    `draws(iteration, steps) -> (normals, uniforms)` of shape (steps, N_local) overrides the rank-local rng;
    `device_seed` (an int, the same on every rank) makes the Metropolis draws come from the library's counter-based
    generator instead (cude_set_rng: nothing crosses PCIe; `subject_offset` = global index of this shard's first
    subject, so the chains do not depend on the sharding).
    Quirks of the reference are preserved: the stochastic-approximation update is applied inside the chain (:185),
    Omega is updated as a variance but used as a standard deviation (:91,:204); the 'current' likelihood it
    re-evaluates each Metropolis step is re-evaluated here only when gamma < 1 (for gamma = 1 the value is carried,
    bit-identical)."""
    import math
    from types import SimpleNamespace
    rng = np.random.default_rng() if rng is None else rng
    initial_mcmc_steps = n_mcmc_steps if initial_mcmc_steps is None else initial_mcmc_steps
    reduce = (lambda v: np.asarray(v, dtype=np.float64)) if collective is None else collective.allreduce_sum
    eng, N, T = engine, engine.N, int(n_obs)
    n_glob = float(reduce([float(N)])[0])
    if collective is not None:
        eng.set_global_subjects(n_glob)
    p_ind = np.full(N, float(prior_eta))
    p_nn = np.array(initial_neural_params, dtype=np.float64)
    P = p_nn.size
    omega = float(prior_omega)
    nll_values, acc_rates = [], []
    if device_seed is not None:
        eng.set_rng(int(device_seed), int(subject_offset))
    for it in range(1, iterations + 1):
        gamma = 1.0 if it <= n_burnin_iterations else 1.0 / (it - n_burnin_iterations) ** alpha
        steps = initial_mcmc_steps if it <= n_burnin_iterations else n_mcmc_steps
        temperature = max(1.0, initial_temperature * math.exp(-temperature_decay * it))
        # E-step (:177-186) fused on the device: all Metropolis steps queued on the stream, one synchronisation
        eng.set_params(p_nn, p_ind)
        if device_seed is not None:
            n_acc = eng.mh_estep(None, None, sigma, prior_eta, omega, proposal_std, temperature, gamma, n_mc=steps)
        else:
            z, u = draws(it, steps) if draws is not None else (rng.standard_normal((steps, N)), rng.random((steps, N)))
            n_acc = eng.mh_estep(z, u, sigma, prior_eta, omega, proposal_std, temperature, gamma)
        _, p_ind = eng.get_params()
        sse = eng.forward(want_sse=True)["sse"]
        ll = np.where(np.isfinite(sse), -(T / 2) * math.log(sigma ** 2) - sse / (2 * sigma ** 2), -np.inf)
        accepted, loglik, s1, s2 = reduce([float(np.sum(n_acc)), float(ll.sum()), float(p_ind.sum()),
                                           float((p_ind * p_ind).sum())])
        # M-step (:118-131): Adam iterations on (neural, sigma), random effects fixed
        x = np.concatenate([p_nn, [sigma]])
        m, v = np.zeros_like(x), np.zeros_like(x)
        for t in range(1, m_step_iters + 1):
            eng.set_params(x[:-1], p_ind)
            if collective is None:
                mean_sse, g_nn, _ = eng.loss_grad(want_cond_grad=False)      # mean SSE and its network gradient
            else:
                part = reduce(eng.loss_grad_partial()[0])
                mean_sse, g_nn = (part[P] / n_glob if part[P + 1] == 0 else np.inf), part[:P]
            s = x[-1]
            g = np.concatenate([g_nn * n_glob / (2 * s * s), [n_glob * T / s - mean_sse * n_glob / s ** 3]])
            m = 0.9 * m + 0.1 * g
            v = 0.999 * v + 0.001 * g * g
            x = x - m_step_lr * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
        sigma = float(x[-1])
        p_nn = (1 - gamma) * p_nn + gamma * x[:-1]
        mean = s1 / n_glob
        var = max(s2 - n_glob * mean * mean, 0.0) / (n_glob - 1) if n_glob > 1 else 0.0
        omega = (1 - omega_learning_rate) * omega + omega_learning_rate * var
        prior_eta = (1 - omega_learning_rate) * prior_eta + omega_learning_rate * mean
        rate = accepted / (n_glob * steps)
        nll_values.append(-loglik)
        acc_rates.append(rate)
        if it > n_burnin_iterations:
            proposal_std = float(np.clip(math.exp(math.log(proposal_std) + gamma * (rate - target_acceptance_rate)),
                                         proposal_std_bounds[0], proposal_std_bounds[1]))
    return SimpleNamespace(p_neural=p_nn, p_individuals=p_ind, Omega=omega, sigma=sigma, eta=prior_eta,
                           total_nll_values=nll_values, acceptance_rates=acc_rates)
