#!/usr/bin/env python3
"""Headline benchmark: subject-trajectories/sec of the fused cUDE training step on MI355X.

One "step" = one pass of the hot path over the resident synthetic population:
forward fixed-step Tsit5 (30 steps) + discrete adjoint + reduction (+ RCCL all-reduce of the
P+2 doubles when N>1) + Adam update of the network (replicated) and of the per-subject
conditional parameters (sharded), with the loss returned to the host every step exactly as
Optimization.jl's loop needs it.  Workload: CPEP3 = reference c-peptide cUDE
(src/c-peptide-models.jl:7-14,86-94) + cumulative-secretion quadrature state, 2->6->6->1 MLP,
T=5 observations on [0,120] min, seeded synthetic population.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))

ARCH = (2, 6, 2)
N_STATE = 3
N_STEPS = 30
T_OBS = 5
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # 256 CU x 4 SIMD x 16 fp64 FMA lanes x 2 flop x 2.4 GHz
# algorithmic bytes per subject-trajectory of the dominant (forward+adjoint) kernel, SURVEY.md 8(d):
# reads k0,k1,k2,c0 (32) + beta (8) + glucose increments (40) + observations (40); writes dL/dbeta (8) + sse (8)
# + cumulative secretion (8, the CPEP3 quadrature output)
ALGO_BYTES_PER_SUBJECT = 32 + 8 + 8 * T_OBS + 8 * T_OBS + 8 + 8 + 8


def executed_flops_per_trajectory(arch=ARCH, n_steps=N_STEPS, n_obs=T_OBS):
    """fp64 flops the forward+adjoint kernel actually executes per subject (FMA = 2), counted from
    the kernel's structure (DESIGN.md "flop accounting"): evaluations x per-evaluation counts."""
    nin, w, d = arch
    # m_tanh_vec per neuron: min, mul, rndne, cvt, ldexp, add, bfi (7 x 1 flop) + 13 FMA (2 reduction, 10 Horner,
    # 1 final) ; per layer: one reciprocal (rcp + 3 FMA) + 3(W-1) multiplies
    tanh_layer_fl = w * (7 + 13 * 2) + (1 + 3 * 2) + 3 * (w - 1)
    softplus_fl = 26 * 2 + 22      # exp (12 FMA) + shared reciprocal (3 FMA) + atanh series (11 FMA) + 22 other ops
    fwd_eval = 2 * (w * 1 + (d - 1) * w * w + w) + d * tanh_layer_fl + softplus_fl
    bwd_eval = (2 * w + w + 2) + (d - 1) * w * (4 + 4 * w) + w * 6
    n_eval = 5 * n_steps + 1
    stage_fwd = 2 * (2 * 21 + 2 * 6 + 7 * 4) + 2 * 6          # stage sums, Y, A*Y+g, quadrature
    stage_rev = 2 * (2 * 21 + 6 * 4 + 12) + 12
    obs_fl = n_obs * (2 * 3 * 7 + 8) * 2
    total = n_eval * fwd_eval + n_eval * (fwd_eval + bwd_eval) + n_steps * (stage_fwd + stage_rev) + obs_fl
    if 6 <= w <= 7:
        # layer-1 exponent table (cude_device.h Mlp::HAS_TAB): inside a run of steps within one glucose piece the W
        # layer-1 exponentials (7 + 13 FMA each) and the W first-layer FMAs are replaced by one multiply + min + add
        # per neuron; per run and sweep: 6 W exponentials (table + anchor) and the range check; per step W multiplies
        n_tab_steps, n_runs = table_steps(n_steps, n_obs)
        saved_per_eval = w * (7 + 13 * 2) + 2 * w - 3 * w
        exp_fl = 6 + 12 * 2
        per_run = 6 * w * exp_fl + 10 * w
        total += 2 * (-5 * n_tab_steps * saved_per_eval + n_runs * per_run + n_tab_steps * w)
    return total


def table_steps(n_steps, n_obs):
    """(number of steps that lie inside one glucose piece, number of runs of such steps) for equidistant
    observation times -- the classification of step_tables() in csrc/cude_api.hip."""
    piece = []
    for n in range(n_steps):
        a, b = n * (n_obs - 1) / n_steps, (n + 1) * (n_obs - 1) / n_steps      # in units of one piece
        j = min(int(a + 1e-9), n_obs - 2)
        piece.append(j if b <= j + 1 + 1e-9 else -1)
    runs = sum(1 for n in range(n_steps) if piece[n] >= 0 and (n == 0 or piece[n - 1] != piece[n]))
    return sum(1 for p in piece if p >= 0), runs


def synthetic_population(n, seed):
    """Seeded synthetic population of the c-peptide shape; numpy only."""
    rng = np.random.default_rng(seed)
    age = rng.uniform(20, 79, n)
    t2dm = rng.random(n) < 0.44
    tp = np.array([0.0, 30.0, 60.0, 90.0, 120.0])
    mu = np.array([5.22, 9.10, 10.44, 10.62, 10.35])
    sd = np.array([0.88, 1.99, 3.45, 4.58, 4.93])
    z = rng.standard_normal(n)
    G = np.maximum(3.2, mu[None, :] + sd[None, :] * z[:, None])
    beta = rng.normal(-0.63, 0.9, n)
    c0 = np.maximum(0.2, rng.normal(0.62, 0.29, n))
    return tp, G, c0, age, t2dm, beta, rng


def glorot(arch, seed):
    rng = np.random.default_rng(seed)
    nin, w, d = arch
    parts, fan = [], nin
    for _ in range(d):
        parts += [rng.standard_normal(w * fan) * np.sqrt(2.0 / (fan + w)), np.zeros(w)]
        fan = w
    parts += [rng.standard_normal(fan) * np.sqrt(2.0 / (fan + 1)), np.zeros(1)]
    return np.concatenate(parts)


def cpu_baseline(tp, G, obs, age, t2dm, nn, beta, sample):
    """Times the CPU oracle ("port": forward-mode duals + OpenMP, oracle/cude_oracle.c) on a bounded
    sample of the same population.  Baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as co
    n = min(sample, G.shape[0])
    threads = co.num_threads()
    co.cpep(tp, G[:64], obs[:64], age[:64], t2dm[:64], ARCH, nn, beta[:64], N_STEPS, N_STATE)   # warm-up
    t0 = time.perf_counter()
    r = co.cpep(tp, G[:n], obs[:n], age[:n], t2dm[:n], ARCH, nn, beta[:n], N_STEPS, N_STATE, want_grad=True)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "subject-trajectories/s", "cores": threads, "kind": "port",
            "sample": f"{n} subjects of the same population, 1 loss+gradient evaluation "
                      f"(forward-mode duals, P+1 partials per subject, OpenMP static over subjects), {dt:.2f} s"}, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--subjects-per-gpu", type=int, default=125000,
                    help="125000/GPU = 1e6 subjects on 8 GPUs (BASELINE configs[3]); at N=1 this is the "
                         "configs[2] training step at that shard size")
    ap.add_argument("--cpu-sample", type=int, default=125000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # CUDE_BENCH_REHEARSAL=1 (development only): several ranks share the GPUs that exist and torch.distributed uses
    # gloo, so that the multi-rank control flow of this script can be exercised on a 1-GPU box
    rehearsal = os.environ.get("CUDE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    ctl = "cpu" if rehearsal else "cuda"          # device of the small control tensors handed to torch.distributed
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cude.engine import Engine   # after torch: one shared HIP runtime

    n_local = args.subjects_per_gpu
    tp, G, c0, age, t2dm, beta_true, rng = synthetic_population(n_local, 20250905 + rank)
    nn = glorot(ARCH, 1234)
    eng = Engine("cpep", ARCH, n_steps=N_STEPS, n_state=N_STATE, device=local_rank)
    transport = "rccl"          # all-reduce of the P+2 doubles inside libcude_hip.so (RCCL on the context's stream)
    if world > 1:
        # Every rank issues the same sequence of collectives whatever fails locally: first agree that librccl
        # is loadable everywhere (each rank draws an id; only rank 0's is used), then build the communicator.
        ok = torch.ones(1, device=ctl)
        my_id = bytes(128)
        if os.environ.get("CUDE_BENCH_TRANSPORT", "rccl") != "rccl":
            ok.zero_()
        else:
            try:
                my_id = Engine.comm_unique_id()
            except Exception as exc:  # e.g. librccl not loadable: fall back to the host-collective transport
                print(f"[rank {rank}] built-in RCCL communicator unavailable ({exc}); using torch.distributed",
                      file=sys.stderr)
                ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() != 0:
            uid = torch.tensor(list(my_id), dtype=torch.uint8, device=ctl)
            dist.broadcast(uid, 0)
            try:
                eng.comm_init(world, rank, bytes(uid.cpu().tolist()))
            except Exception as exc:
                print(f"[rank {rank}] cude_comm_init failed ({exc}); using torch.distributed", file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            transport = "host"  # cude_loss_grad_partial -> dist.all_reduce (RCCL via PyTorch) -> cude_adam_apply
            eng.close()
            eng = Engine("cpep", ARCH, n_steps=N_STEPS, n_state=N_STATE, device=local_rank)
    # observations = the device's own forward solve at the true betas + 5 % multiplicative noise
    cp0 = np.repeat(c0[:, None], T_OBS, axis=1)
    eng.set_population_cpep(tp, G, cp0, age, t2dm)
    eng.set_params(nn, beta_true)
    traj = eng.forward(want_traj=True)["traj"]
    obs = traj[0].T * (1.0 + 0.05 * rng.standard_normal((n_local, T_OBS)))
    obs[:, 0] = c0
    eng.set_population_cpep(tp, G, obs, age, t2dm)
    if transport == "host":
        eng.set_global_subjects(n_local * world)
    beta0 = beta_true + 0.3 * rng.standard_normal(n_local)
    eng.set_params(nn, beta0)
    eng.adam_init(1e-2)

    def train_step(want_loss=True):
        """One optimiser iteration over ALL ranks' subjects; returns the global loss."""
        if transport == "rccl":
            return eng.adam_step(want_loss=want_loss)
        part, _ = eng.loss_grad_partial()
        t = torch.from_numpy(part).to(ctl)
        dist.all_reduce(t)
        return eng.adam_apply(t.cpu().numpy())

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        train_step()
    eng.set_kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = train_step()              # returns the loss to the host every step
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, n_launch = eng.kernel_time_ms()
    eng.set_kernel_timing(False)

    # async variant (loss not read back per step), reported as extra information
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        train_step(want_loss=False)
    barrier()
    dt_async = time.perf_counter() - t1

    if world > 1:
        t = torch.tensor([dt, dt_async, kern_ms], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_async, kern_ms = (float(v) for v in t.cpu())

    if rank == 0:
        n_total = n_local * world
        value = n_total * args.steps / dt
        flops = executed_flops_per_trajectory()
        algo_bytes = ALGO_BYTES_PER_SUBJECT * n_local
        achieved_gbs = algo_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("subjects_per_gpu") == n_local:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "subject-trajectories/sec (fwd+adjoint, 3-state ODE, 30 steps)",
            "value": value, "unit": "subject-trajectories/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"CPEP3 cUDE training step (fwd Tsit5 x{N_STEPS} + discrete adjoint + Adam), "
                                   f"2x6x6x1 MLP, T={T_OBS}, {n_local} subjects/GPU "
                                   f"({n_total} total; BASELINE configs[2]/[3] shape)",
                       "subjects_per_gpu": n_local, "parallelism": f"subject-shard x{world}", "allreduce": transport if world > 1 else None},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "cpep_kernel<2,6,2,3,grad>", "kernel_ms": kern_ms, "launches": n_launch,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "path is fp64-VALU bound (SURVEY.md 8d), see roofline_valu"},
            "roofline_valu": {"bound": "valu_fp64", "achieved": flops * n_local / (kern_ms * 1e-3) / 1e12,
                              "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                              "frac": flops * n_local / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                              "flops_per_trajectory": flops},
            "async_value": n_total * args.steps / dt_async,
            "final_loss": loss,
        }
        if not args.no_cpu_baseline and world == 1:
            cb, _ = cpu_baseline(tp, G, obs, age, t2dm, nn, beta0, args.cpu_sample)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
