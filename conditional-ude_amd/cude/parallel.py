"""Subject sharding across GPUs (one process per GPU) for the population training step.

The reference has no distributed path; its only parallelism over trajectories is EnsembleThreads
(suppression/src/suppression_model.jl:113,123).  Subjects are independent, so they shard trivially; the
single coupled quantity is the shared network, whose gradient (with the loss sum and the failure count:
P+2 doubles) is summed across ranks once per optimiser step.  Two transports:

  * "rccl": the sum happens inside libcude_hip.so on the context's stream (cude_comm_init +
    cude_adam_step); the host only ships the 128-byte unique id once.
  * "host": the host sums the P+2 doubles with ANY collective it likes (torch.distributed gloo/nccl,
    MPI) between cude_loss_grad_partial and cude_adam_apply.

`engine` is anything with the Engine interface (tests drive this logic on CPU with a stand-in engine
over the gloo backend; production uses cude.engine.Engine).
"""
import numpy as np


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


class ShardedTrainer:
    """Data-parallel Adam training of one cUDE over subject shards.

    Each rank passes ITS shard's engine (population already uploaded) and the collective."""

    def __init__(self, engine, collective, transport="host"):
        self.engine, self.coll, self.transport = engine, collective, transport
        if transport not in ("host", "rccl"):
            raise ValueError("transport must be 'host' or 'rccl'")

    @staticmethod
    def attach_rccl(engine_cls, engine, collective):
        """Bootstrap the built-in communicator: rank 0 creates the id, everyone joins.  Must run before the
        population is uploaded (the global subject count is all-reduced there)."""
        uid = engine_cls.comm_unique_id() if collective.rank == 0 else None
        uid = collective.broadcast_bytes(uid, 0)
        engine.comm_init(collective.world_size, collective.rank, uid)

    def sync_population_statistics(self, scale_sums=None):
        """host transport: establish the global subject count (and SUPP's scale = mean_i max_t data)."""
        if self.transport != "host":
            return
        if scale_sums is None:
            tot = self.coll.allreduce_sum([float(self.engine.N)])
            self.engine.set_global_subjects(tot[0])
        else:
            tot = self.coll.allreduce_sum(list(scale_sums) + [float(self.engine.N)])
            self.engine.set_global_subjects(tot[3], tot[:3] / tot[3])

    def adam_init(self, lr, **kw):
        self.engine.adam_init(lr, **kw)

    def adam_step(self):
        """One optimiser iteration; returns the GLOBAL loss (identical on every rank)."""
        if self.transport == "rccl":
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
