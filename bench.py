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

Prints ONE JSON line on rank 0.  Besides the contract fields: `roofline` (HBM, the contract's object),
`roofline_valu` (the binding one), `cpu_baseline` (reverse-mode CPU port; `cpu_baseline_forward_mode` = the
reference's own AD method), `extra` (the other BASELINE configs a single GPU can run: forward-only 1e4, the
reference-faithful 2->4->4->1 and suppression instances at 1e5, the SAEM E-step 1e4 x 100), and for N>1 the proof of
the transport: `config.allreduce` ("xchg" = the library's peer-write exchange inside the reduction kernels, "rccl" =
ncclAllReduce on the context's stream, "torch" = torch.distributed on a tensor aliasing the context's buffer), `xchg` (ranks,
memory kind, timed-out waits), `rccl_ranks` / `rccl_version` as the communicator reports them, `allreduce_check` (one
optimiser step through the library's transport vs through torch.distributed on a twin context), `transports` (the same K
steps through every transport that came up), per-rank kernel times; `cpu_baseline` / `parity_vs_cpu_baseline` are
rank 0's (its shard against the CPU port) while the other ranks wait at a barrier.
`launch_mode` says how the timed steps were queued: "plain+events/4" = ordinary launches with HIP events around every
4th gradient launch (the kernel time of the roofline objects is measured IN the timed region, which keeps the captured
graphs out of it); `ms_per_step_graph_replay` is the same K steps replayed from captured graphs right afterwards.
"""
import os

# dmabuf IPC for multi-process RCCL on this pool: must be in the environment before the first HIP call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse  # noqa: E402
import glob  # noqa: E402
import hashlib  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))

ARCH = (2, 6, 2)
N_STATE = 3
N_STEPS = 30
T_OBS = 5
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # 256 CU x 4 SIMD x 16 fp64 FMA lanes x 2 flop x 2.4 GHz
TIMING_PERIOD = 4              # kernel time of the timed region: HIP events around every 4th gradient launch (median quoted)
PREWARM_STEPS = 64             # untimed launches before the W warm-up steps: the GPU needs ~30 launches (20 ms) to
                               # reach its steady clock after an idle period (profiles/r02/clock_ramp.txt)


# ------------------------------------------------------------------------------------------ counted work
def cpep_algo_bytes(n_obs, n_state, grad):
    """Algorithmic bytes per subject-trajectory of the c-peptide kernels: reads k0,k1,k2,c0 (32) +
    beta (8) + glucose increments + observations (8 T each); writes sse (8) [+ dL/dbeta (8)] [+ cumulative secretion
    (8), the CPEP3 quadrature output]."""
    return 32 + 8 + 8 * n_obs + 8 * n_obs + 8 + (8 if grad else 0) + (8 if n_state == 3 else 0)


def supp_algo_bytes(n_obs, grad):
    """u0 is data[:, 0, :] (part of the observations): reads theta (8) + data (3 x 8 T); writes sse (8) [+ dL/dtheta]."""
    return 8 + 3 * 8 * n_obs + 8 + (8 if grad else 0)


TANH_TABLE_MAX_WIDTH_CPEP = 4   # csrc/cude_device.h CUDE_TANH_TAB_MAXW: fixed-step c-peptide kernels up to this width (and the
                                # fixed-step suppression kernel always) evaluate tanh by table + addition theorem


class Ops:
    """Executed vector instructions of a piece of a kernel, per lane, in three classes:
      fma    fused multiply-adds (2 floating-point operations, 1 issue slot each);
      fp1    other floating-point arithmetic: add / sub / mul and the reciprocal seed v_rcp_f64 (1 operation each; the
             reciprocal occupies four issue cycles, counted as ONE instruction here as the SQ counters count it);
      other  VALU instructions that do no floating-point arithmetic: min / max clamps, round-to-integer, int conversion,
             ldexp, sign transfer (v_bfi), compares and the two v_cndmask a double select takes, shifts.
    flops = 2 fma + fp1 is the numerator of roofline_valu; slots = fma + fp1 + other is what SQ_INSTS_VALU / SQ_WAVES
    counts per wave (register moves, accumulator seeds and address arithmetic excepted: the structural tally leaves
    those to the counter).  Counted from the kernels' source (cude_math.h, cude_device.h), one entry per operation."""

    def __init__(self, fma=0, fp1=0, other=0):
        self.fma, self.fp1, self.other = fma, fp1, other

    def __add__(self, o):
        return Ops(self.fma + o.fma, self.fp1 + o.fp1, self.other + o.other)

    def __sub__(self, o):
        return Ops(self.fma - o.fma, self.fp1 - o.fp1, self.other - o.other)

    def __mul__(self, k):
        return Ops(self.fma * k, self.fp1 * k, self.other * k)

    __rmul__ = __mul__

    @property
    def flops(self):
        return 2 * self.fma + self.fp1

    @property
    def slots(self):
        return self.fma + self.fp1 + self.other

    def __repr__(self):
        return f"Ops(fma={self.fma}, fp1={self.fp1}, other={self.other})"


RCP = Ops(fma=3, fp1=1)                 # m_rcp: v_rcp_f64 + one cubic Newton step
# m_exp2x_t: x * 2 log2 e (mul), rint, cvt, two reduction FMAs, ten Horner FMAs, ldexp
EXP2X = Ops(fma=12, fp1=1, other=3)
# per neuron, exponential form (m_tanh_den + m_tanh_vec): clamp (min), EXP2X, + 1, 1 - 2 inv (FMA), copysign (v_bfi)
TANH_EXP_NEURON = EXP2X + Ops(fma=1, fp1=1, other=2)
# per neuron, table form (m_tanh_cell + m_tanh_vec_tab): clamp (min); xa + magic, t - magic, xa - (.), b b, u + 10.5,
# b (.), u + 45, (.) 0.1 (8 fp1); fma(u, ., 105), numerator, denominator (3 FMA); index shift (other); the quotient's
# multiply (fp1); sign transfer (other)
TANH_TAB_NEURON = Ops(fma=3, fp1=9, other=3)
# m_softplus_t: -0.5 |x| (mul), clamp (max), EXP2X, 1 + e, compare, e - 1 / select (2 cndmask), e + 3, e + 2 / select,
# d den (mul), RCP, rp den, rp d, num (.), e inv_d / compare / select, s s, eight atanh FMAs, 2 s p (two muls), max(x, 0),
# + ln 2 / select, final sum
SOFTPLUS = EXP2X + RCP + Ops(fma=8, fp1=15, other=12)
# ... of which only the VALUE needs (the reverse sweeps re-evaluate the network for its derivative and never execute it):
# e - 1 / select, rp d, num (.), s s, the eight FMAs, the two muls of 2 s p, + ln 2 / select, max(x, 0), the final sum
SOFTPLUS_VALUE_ONLY = Ops(fma=8, fp1=8, other=5)


def tanh_layer(w, table_tanh):
    """one hidden layer's activations: W neurons, ONE shared reciprocal, 3 (W - 1) prefix / back-substitution multiplies"""
    return w * (TANH_TAB_NEURON if table_tanh else TANH_EXP_NEURON) + RCP + Ops(fp1=3 * (w - 1))


def mlp_ops(nv, w, d, n_dx=0, table_tanh=False):
    """(forward, backward) Ops of one network evaluation as the kernels execute it (Mlp::forward / backward,
    cude_device.h); n_dx = number of inputs whose derivative is formed."""
    fwd = Ops(fma=w * nv + (d - 1) * w * w + w) + d * tanh_layer(w, table_tanh) + SOFTPLUS
    out = Ops(fma=w, fp1=w + 2)                                           # dz = wgt sig; d/db; d/dw_out, dh = dz w_out
    hidden = (d - 1) * (w * Ops(fma=1, fp1=2) + w * w * Ops(fma=2) + Ops(fp1=w))   # d = dh (1 - h^2), d/db; d/dW, W^T d; s0 + s1
    first = w * (Ops(fma=1, fp1=2) + Ops(fma=nv))                         # d = dh (1 - h^2), d/dc; d/dW1 (varying inputs)
    dx = n_dx * Ops(fma=w, fp1=2)                                         # W1^T d for the wanted inputs: s0 + s1, dx +=
    return fwd, out + hidden + first + dx


def mlp_flops(nv, w, d, want_dx=False, table_tanh=False):
    """(forward, backward) floating-point operations of one network evaluation (FMA = 2)."""
    f, b = mlp_ops(nv, w, d, nv if want_dx else 0, table_tanh)
    return f.flops, b.flops


def table_steps(n_steps, n_obs):
    """(number of steps that lie inside one glucose piece, number of runs of such steps) for equidistant
    observation times -- the classification of step_tables() in csrc/cude_context.hip."""
    piece = []
    for n in range(n_steps):
        a, b = n * (n_obs - 1) / n_steps, (n + 1) * (n_obs - 1) / n_steps      # in units of one piece
        j = min(int(a + 1e-9), n_obs - 2)
        piece.append(j if b <= j + 1 + 1e-9 else -1)
    runs = sum(1 for n in range(n_steps) if piece[n] >= 0 and (n == 0 or piece[n - 1] != piece[n]))
    return sum(1 for p in piece if p >= 0), runs


def cpep_ops(arch=ARCH, n_steps=N_STEPS, n_obs=T_OBS, n_state=N_STATE, grad=True):
    """Ops the one-lane-per-subject c-peptide kernel executes per subject, counted from the kernel's structure
    (cude_cpep.hip): 5 S + 1 network evaluations per sweep, the Runge-Kutta algebra per step."""
    nin, w, d = arch
    table = w <= TANH_TABLE_MAX_WIDTH_CPEP and nin == 2
    fwd_eval, bwd_eval = mlp_ops(1, w, d, 0, table)
    n_eval = 5 * n_steps + 1
    stage_fwd = Ops(fma=2 * 21 + 2 * 6 + 7 * 4 + (6 if n_state == 3 else 0))   # stage sums, Y, A Y + g [, quadrature]
    stage_rev = Ops(fma=2 * 21 + 6 * 4 + 12, fp1=12)
    obs = n_obs * Ops(fma=(3 if n_state == 3 else 2) * 7, fp1=8)
    total = n_eval * fwd_eval + n_steps * stage_fwd + obs
    if grad:
        total = total + n_eval * (fwd_eval - SOFTPLUS_VALUE_ONLY + bwd_eval) + n_steps * stage_rev + obs
    if 6 <= w <= 7:
        # layer-1 exponent table (cude_device.h Mlp::HAS_TAB): inside a run of steps within one glucose piece the W
        # layer-1 exponentials (clamp + EXP2X + 1) and the W first-layer FMAs are replaced by one multiply + clamp + add
        # per neuron; per run and sweep: 6 W exponentials (table + anchor) and the range check (5 W compares / selects,
        # 5 W multiplies / adds); per step W multiplies
        n_tab_steps, n_runs = table_steps(n_steps, n_obs)
        saved_per_eval = w * (EXP2X + Ops(fp1=1, other=1) + Ops(fma=1)) - w * Ops(fp1=2, other=1)
        per_run = 6 * w * EXP2X + w * Ops(fp1=5, other=5)
        total = total + (2 if grad else 1) * (n_runs * per_run + n_tab_steps * Ops(fp1=w) - 5 * n_tab_steps * saved_per_eval)
    return total


def cpep_flops(arch=ARCH, n_steps=N_STEPS, n_obs=T_OBS, n_state=N_STATE, grad=True):
    return cpep_ops(arch, n_steps, n_obs, n_state, grad).flops


def supp_ops(arch, n_steps, n_obs, grad=True):
    """The same count for supp_kernel (cude_supp.hip): 6 S + 1 evaluations per sweep; the reverse sweep re-evaluates the
    network at the stored stage inputs (no kept activations at this size).  State 1 (du1 = -0.4 u1) is a table lookup:
    one multiply per evaluation, no Runge-Kutta sums, no adjoint -- the stage algebra is that of states 2 and 3, and the
    input-derivative of the first layer is formed for two of the three inputs."""
    nin, w, d = arch
    fwd_eval, bwd_eval = mlp_ops(3, w, d, 2, table_tanh=True)
    n_eval = 6 * n_steps + 1
    stage = Ops(fma=2 * 21 + 6 * 2)                   # per step: stage sums + Y per stage, two states
    rhs = Ops(fma=2, fp1=1)                           # per evaluation: u1 from the table; du2, du3 (one FMA each)
    total = n_eval * (fwd_eval + rhs) + n_steps * stage + n_obs * Ops(fma=2 * 7 + 2 + 3, fp1=2 * 2 + 2)
    if grad:
        vjp = Ops(fma=1, fp1=4)                       # u1 from the table; weight kb3 - kb2; ub2 += dx2; ub3 += -0.3 kb3 + dx3
        total = total + n_eval * (fwd_eval - SOFTPLUS_VALUE_ONLY + bwd_eval + vjp) + n_steps * stage + n_obs * Ops(fma=2 * 7, fp1=2 * 3)
    return total


def supp_flops(arch, n_steps, n_obs, grad=True):
    return supp_ops(arch, n_steps, n_obs, grad).flops


def kernel_source_sha():
    """Digest of the kernel sources: a PMC traffic record (profiles/pmc_traffic.json) is only quoted for the code it
    was measured on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "conditional-ude_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "conditional-ude_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_record(kernel_key, n_local):
    """The committed PMC record of a kernel (profiles/pmc_traffic.json), or (None, reason): quoted only for the code it
    was measured on and the population size it was measured at."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    rec = rec.get("kernels", {}).get(kernel_key) if "kernels" in rec else (rec if kernel_key == "headline" else None)
    if not rec:
        return None, f"no PMC record for {kernel_key}"
    if rec.get("source_sha") != kernel_source_sha():
        return None, f"stale PMC record (kernel sources changed since {rec.get('source_sha')})"
    if rec.get("subjects_per_gpu") != n_local:
        return None, f"PMC record is for {rec.get('subjects_per_gpu')} subjects"
    return rec, None


def pmc_traffic(kernel_key, n_local):
    """HBM bytes per launch from the committed PMC passes, or (None, reason)."""
    rec, why = pmc_record(kernel_key, n_local)
    return (rec.get("hbm_bytes_per_launch"), None) if rec else (None, why)


N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # the clock FP64_VALU_PEAK_TF is quoted at
VALU_CYCLES_PER_INSTRUCTION = 4     # a wave64 fp64 (or any full-rate VALU) instruction occupies its SIMD for four cycles


def rocprof_figures(rec):
    """The profiler's own durations of the kernel, from the committed `rocprofv3 --kernel-trace --stats` pass of the same
    command on the same sources (None without a current record): printed beside the HIP-event figure of this run."""
    if not rec or "rocprof_avg_ms" not in rec:
        return None
    return {"mean_ms": rec["rocprof_avg_ms"], "median_ms": rec.get("rocprof_median_ms"), "launches": rec.get("rocprof_calls"),
            "note": "every launch of the profiled process: the mean includes the clock-ramp launches at its start (the "
                    "first tens run 15-25 % long), the median does not; kernel_ms of this line is HIP events over the "
                    "timed region only"}


def rooflines(kernel, kern_ms, launches, n_subjects, bytes_per_subject, ops, traffic=None, note=None, sq=None,
              rocprof=None):
    """The contract's `roofline` object (HBM) and `roofline_valu` (the binding one).  `ops` = Ops per subject-trajectory:
    roofline_valu.frac is floating-point operations (FMA = 2, add / mul / rcp = 1) over the fp64 vector peak;
    clamps, conversions, selects and sign transfers occupy the same issue slots but are NOT floating-point work and
    are listed beside it (`other_valu_slots_per_trajectory`).  `valu_slot_utilisation` says how busy the vector issue
    slots were, whatever they issued: SQ_INSTS_VALU (committed PMC pass of these sources) x 4 cycles / (1024 SIMDs x
    kernel time x 2.4 GHz)."""
    gbs = bytes_per_subject * n_subjects / (kern_ms * 1e-3) / 1e9
    tfs = ops.flops * n_subjects / (kern_ms * 1e-3) / 1e12
    hbm = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "traffic": traffic, "kernel": kernel, "kernel_ms": kern_ms, "launches": launches,
           "algorithmic_bytes_per_launch": bytes_per_subject * n_subjects,
           "note": note or "path is fp64-VALU bound (SURVEY.md 8d), see roofline_valu"}
    if rocprof_figures(rocprof):
        hbm["kernel_ms_rocprof"] = rocprof_figures(rocprof)
    valu = {"bound": "valu_fp64", "achieved": tfs, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "frac": tfs / FP64_VALU_PEAK_TF, "flops_per_trajectory": ops.flops,
            "counted": {"fma": ops.fma, "other_fp_ops": ops.fp1, "other_valu_slots_per_trajectory": ops.other,
                        "valu_slots_per_trajectory": ops.slots,
                        "note": "flops = 2 fma + other_fp_ops (add, mul, rcp); other_valu_slots = clamps, rounding / "
                                "conversion, ldexp, selects, sign transfer: issue slots without floating-point work, "
                                "not in the numerator"},
            "valu_slot_utilisation": None}
    if sq and sq.get("SQ_INSTS_VALU") and sq.get("SQ_WAVES"):
        cycles = N_SIMD * kern_ms * 1e-3 * CLOCK_GHZ * 1e9
        valu["valu_slot_utilisation"] = sq["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INSTRUCTION / cycles
        valu["valu_instructions_per_wave"] = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]
        valu["valu_slot_note"] = ("SQ_INSTS_VALU of the committed PMC pass (profiles/pmc_traffic.json, same sources) x 4 "
                                  "cycles / (1024 SIMDs x this run's kernel time x 2.4 GHz)")
    return hbm, valu


# ------------------------------------------------------------------------------------------ synthetic data
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


def synthetic_suppression(n, seed):
    """Suppression-model data as the reference generates it (`generate_data`, suppression/src/suppression_model.jl:33-63,
    restated: SURVEY.md 8(d)): subjects cycle through the six suppression strengths of suppression.jl:27-28, parameters
    max(mu + sd * randn, 0.05) with mu = [0.4, 0.9, 0.3, mu_sup], sd = [0.1, 0.1, 0.1, mu_sup / 8], data = the ground-truth
    model lsup! (:16-20) from u0 = (10, 0, 0) at 8 times on [0, 30], 10 % multiplicative noise, clamped at 0.  (numpy's
    random stream and a fixed-step RK4 solve: inputs of the reference's distribution, not its numbers.)"""
    rng = np.random.default_rng(seed)
    T = 8
    tp = np.linspace(0.0, 30.0, T)
    mu_sup = np.array([0.5, 2.5, 5.0, 7.5, 10.0, 12.5])[np.arange(n) % 6]
    mu = np.stack([np.full(n, 0.4), np.full(n, 0.9), np.full(n, 0.3), mu_sup])
    sd = np.stack([np.full(n, 0.1), np.full(n, 0.1), np.full(n, 0.1), mu_sup / 8.0])
    p = np.maximum(mu + sd * rng.standard_normal((4, n)), 0.05)

    def f(u):
        a = p[1] * u[1] / (1.0 + p[3] * u[2])
        return np.stack([-p[0] * u[0], p[0] * u[0] - a, a - p[2] * u[2]])
    sub = 60                                             # RK4 sub-steps per observation interval (h = 0.071)
    h = (tp[1] - tp[0]) / sub
    u = np.stack([np.full(n, 10.0), np.zeros(n), np.zeros(n)])
    data = np.empty((3, T, n))
    data[:, 0] = u
    for k in range(1, T):
        for _ in range(sub):
            k1 = f(u)
            k2 = f(u + 0.5 * h * k1)
            k3 = f(u + 0.5 * h * k2)
            k4 = f(u + h * k3)
            u = u + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
        data[:, k] = u
    data = np.maximum(data * (1.0 + 0.1 * rng.standard_normal(data.shape)), 0.0)
    return tp, data, rng.standard_normal(n)              # theta ~ randn as in suppression.jl:38


def glorot(arch, seed):
    rng = np.random.default_rng(seed)
    nin, w, d = arch
    parts, fan = [], nin
    for _ in range(d):
        parts += [rng.standard_normal(w * fan) * np.sqrt(2.0 / (fan + w)), np.zeros(w)]
        fan = w
    parts += [rng.standard_normal(fan) * np.sqrt(2.0 / (fan + 1)), np.zeros(1)]
    return np.concatenate(parts)


def cpep_engine(Engine, arch, n_state, n, seed, device, nn):
    """Engine with a resident synthetic c-peptide population whose observations are the device's own forward solve at
    the true betas + 5 % multiplicative noise; parameters set to a perturbed start."""
    tp, G, c0, age, t2dm, beta_true, rng = synthetic_population(n, seed)
    eng = Engine("cpep", arch, n_steps=N_STEPS, n_state=n_state, device=device)
    T = len(tp)
    eng.set_population_cpep(tp, G, np.repeat(c0[:, None], T, axis=1), age, t2dm)
    eng.set_params(nn, beta_true)
    traj = eng.forward(want_traj=True)["traj"]
    obs = traj[0].T * (1.0 + 0.05 * rng.standard_normal((n, T)))
    obs[:, 0] = c0
    beta0 = beta_true + 0.3 * rng.standard_normal(n)
    return eng, dict(tp=tp, G=G, obs=obs, age=age, t2dm=t2dm, beta0=beta0, rng=rng)


# ------------------------------------------------------------------------------------------ CPU baseline
def usable_cores(n_logical):
    """Host cores this process may actually use: the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) when there
    is one, else the logical count.  The GPU box shows 256 logical CPUs to a 16-core share: 128 OpenMP threads there
    measured 1.5e5 traj/s against 2.3e5 with 16 (profiles/r03/cpu_baseline_scaling.txt)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, min(n_logical, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return max(1, min(n_logical, int(q / per + 0.5)))
    except (OSError, ValueError):
        pass
    try:
        return max(1, min(n_logical, len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        return n_logical


def cpu_baseline(pop, nn, sample, eng=None, world=1):
    """Times the CPU port on a bounded sample of the same population: the per-subject reverse-mode gradient with OpenMP
    static scheduling over subjects (oracle/cude_oracle_rev.c), and -- for the record -- the
    reference's own AD method, forward-mode duals with P+1 partials per subject (oracle/cude_oracle.c).  Baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as co
    n = min(sample, pop["G"].shape[0])
    threads = usable_cores(co.num_threads())
    if threads > 16:
        # no quota visible: a CPU share smaller than the logical count shows as anti-scaling -- probe, keep the fastest
        m = min(n, 4000)
        small = (pop["tp"], pop["G"][:m], pop["obs"][:m], pop["age"][:m], pop["t2dm"][:m], ARCH, nn, pop["beta0"][:m],
                 N_STEPS, N_STATE)
        best = None
        for cand in sorted({16, 32, 64, threads}):
            if cand > threads:
                continue
            co.cpep(*small, method="reverse", nthreads=cand)
            t0 = time.perf_counter()
            co.cpep(*small, method="reverse", nthreads=cand)
            dt_c = time.perf_counter() - t0
            if best is None or dt_c < best[0]:
                best = (dt_c, cand)
        threads = best[1]
    a = (pop["tp"], pop["G"][:n], pop["obs"][:n], pop["age"][:n], pop["t2dm"][:n], ARCH, nn, pop["beta0"][:n], N_STEPS,
         N_STATE)
    co.cpep(pop["tp"], pop["G"][:256], pop["obs"][:256], pop["age"][:256], pop["t2dm"][:256], ARCH, nn,
            pop["beta0"][:256], N_STEPS, N_STATE, method="reverse", nthreads=threads)  # warm-up (thread pool)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = co.cpep(*a, method="reverse", nthreads=threads)
    dt = (time.perf_counter() - t0) / reps
    rev = {"value": n / dt, "unit": "subject-trajectories/s", "cores": threads, "kind": "port",
           "sample": f"{n} subjects of the same population, {reps} loss+gradient evaluations (per-subject reverse-mode "
                     f"discrete adjoint, OpenMP static over subjects), {dt:.2f} s each"}
    parity = None
    if eng is not None and n == eng.N:
        # max relative error of loss and gradients against the CPU backend, same run, same inputs
        eng.set_params(nn, pop["beta0"])
        if world == 1:
            loss, g_nn, g_cond = eng.loss_grad()
        else:
            # this rank's shard only (no collective: the other ranks are waiting at a barrier): the un-reduced vector
            # carries 1/N_global, the CPU port on the shard 1/N_local
            part, g_cond = eng.loss_grad_partial(want_cond_grad=True)
            loss, g_nn, g_cond = part[-2] / n, part[:-2] * world, g_cond * world
        parity = {"subjects": n, "loss_rel": abs(loss - ref["loss"]) / abs(ref["loss"]),
                  "g_nn_rel_maxnorm": float(np.max(np.abs(g_nn - ref["g_nn"])) / np.max(np.abs(ref["g_nn"]))),
                  "g_cond_rel_maxnorm": float(np.max(np.abs(g_cond - ref["g_beta"])) / np.max(np.abs(ref["g_beta"]))),
                  "north_star_rtol": 1e-6}
    t0 = time.perf_counter()
    co.cpep(*a, nthreads=threads)
    dt = time.perf_counter() - t0
    fwd = {"value": n / dt, "unit": "subject-trajectories/s", "cores": threads, "kind": "port",
           "sample": f"{n} subjects, 1 loss+gradient evaluation by the reference's AD method (forward-mode duals, P+1 "
                     f"partials per subject; the reference itself carries N+P), {dt:.2f} s"}
    return rev, fwd, parity


# ------------------------------------------------------------------------------------------ extra configurations
def timed_adam(eng, n, steps, warm):
    """(seconds per queued step, kernel ms by HIP events, timed launches) -- the step time with kernel timing OFF (replayed
    graphs: the `ms_per_step_graph_replay` mode of the headline, NOT the mode of its timed region, which carries the
    sampled events and therefore plain launches), the kernel time in a separate pass of the same iterations with an event
    pair around every launch (plain launches: a pair costs ~4.5 us of stream time)."""
    for _ in range(warm):
        eng.adam_step(want_loss=False)
    eng.adam_run(steps)                              # captures the iteration graphs, sizes the loss trace
    eng.synchronize()
    t0 = time.perf_counter()
    losses = eng.adam_run(steps)                     # queued steps, device-side loss trace: as in the headline
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert np.all(np.isfinite(losses))
    eng.set_kernel_timing(True)
    eng.adam_run(steps)
    ms, launches = eng.kernel_time_ms()
    eng.set_kernel_timing(False)
    return dt, ms, launches


def extras(Engine, device, steps=20, warm=40):
    """The other BASELINE configs one GPU can run, each with its own kernel time, counted bytes / flops and both
    roofline fractions (SURVEY.md 8(d) "always also report")."""
    out = {}
    # ---- configs[1]: forward-only ensemble, 1e4 subjects, headline model
    n = 10000
    eng, pop = cpep_engine(Engine, ARCH, N_STATE, n, 777, device, glorot(ARCH, 1234))
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(glorot(ARCH, 1234), pop["beta0"])
    for _ in range(warm):
        eng.forward()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward()
    dt = (time.perf_counter() - t0) / steps        # (call time without the kernel-timing events; those in a second pass)
    eng.set_kernel_timing(True)
    for _ in range(steps):
        eng.forward()
    ms, launches = eng.kernel_time_ms()
    eng.set_kernel_timing(False)
    hbm, valu = rooflines("cpep2_fwd_kernel<2,6,2,3> + cpep2_scan_kernel (time-split forward)", ms, launches, n,
                          cpep_algo_bytes(T_OBS, N_STATE, False), cpep_ops(grad=False))
    out["forward_only_1e4"] = {"config": "BASELINE configs[1]: CPEP3 2x6x6x1, 1e4 subjects, forward-only loss",
                               "value": n / dt, "unit": "subject-trajectories/s", "ms_per_call": dt * 1e3,
                               "roofline": hbm, "roofline_valu": valu}
    eng.close()
    # ---- configs[2] at its exact size: the headline model, 1e5 subjects, fwd + adjoint + Adam (time-split path, L chosen
    # by the library's launch-cost model)
    n = 100000
    nn6 = glorot(ARCH, 1234)
    eng, pop = cpep_engine(Engine, ARCH, N_STATE, n, 776, device, nn6)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn6, pop["beta0"])
    eng.adam_init(1e-2)
    dt, ms, launches = timed_adam(eng, n, steps, warm)
    hbm, valu = rooflines("mixed gradient launch: cpep_kernel<Mlp<2,6,2,1>,3,grad> on 1024 workgroups (one long wave per "
                          "SIMD) beside cpep2_fwd / scan / rev <2,6,2,3> on the other 539 (library's path selector)", ms,
                          launches, n,
                          cpep_algo_bytes(T_OBS, N_STATE, True), cpep_ops())
    out["train_step_1e5"] = {"config": "BASELINE configs[2]: CPEP3 2x6x6x1, exactly 1e5 subjects, fwd + adjoint + Adam",
                             "value": n / dt, "unit": "subject-trajectories/s", "ms_per_step": dt * 1e3,
                             "roofline": hbm, "roofline_valu": valu}
    # ---- the reference's own training entry point at that size: `train` / `_optimize` (src/parameter-estimation.jl:340-386,
    # :170-183) = 25 selected restarts, Adam then L-BFGS, here side by side with the optimiser state resident on the device
    # (cude_train_restarts); per-iteration time by difference of two calls (the sets' upload / download drops out)
    K, a_it, l_it = 25, 8, 4
    rng = np.random.default_rng(11)
    nn_sets = nn6[None, :] * (1.0 + 0.1 * rng.standard_normal((K, nn6.size)))
    cond_sets = pop["beta0"][None, :] + 0.1 * rng.standard_normal((K, n))
    eng.train_restarts(nn_sets, cond_sets, 2, 1e-3, 0)
    t0 = time.perf_counter()
    eng.train_restarts(nn_sets, cond_sets, a_it, 1e-3, 0)
    t1 = time.perf_counter()
    eng.train_restarts(nn_sets, cond_sets, 2 * a_it, 1e-3, 0)
    t2 = time.perf_counter()
    per_it = ((t2 - t1) - (t1 - t0)) / a_it
    _, _, obj, tr = eng.train_restarts(nn_sets, cond_sets, 0, 1e-3, l_it, want_trace=True)
    t3 = time.perf_counter()
    out["train_restarts_1e5x25"] = {
        "config": "`train` second phase (parameter-estimation.jl:372-383): 25 restarts x 1e5 subjects side by side, CPEP3 "
                  "2x6x6x1, Adam then L-BFGS + BackTracking, optimiser state resident on the device",
        "value": K * n / per_it, "unit": "subject-trajectories/s", "ms_per_adam_iteration": per_it * 1e3,
        "adam_iteration_over_25_single_set_steps": per_it / (K * dt),
        "ms_per_lbfgs_iteration": (t3 - t2) / l_it * 1e3, "lbfgs_iterations_accepted": int(np.sum(np.isfinite(tr))),
        "host_device_bytes_per_adam_iteration": 0,
        "host_device_bytes_per_lbfgs_round": K * 184,
        "note": "one launch evaluates all 25 sets (39 075 waves: the chip stays filled, hence < 25 single-set steps); the "
                "L-BFGS vectors (iterate, gradient, direction, 10 (s, y) pairs per restart) stay on the device, one "
                "workgroup per restart runs Optim's iteration, the host reads the restarts' 184-byte states once per round"}
    eng.close()
    # ---- reference-faithful c-peptide instance: 2->4->4->1, 2 states, 1e5 subjects, fwd + adjoint + Adam
    n, arch = 100000, (2, 4, 2)
    nn4 = glorot(arch, 1234)
    eng, pop = cpep_engine(Engine, arch, 2, n, 778, device, nn4)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn4, pop["beta0"])
    eng.adam_init(1e-2)
    dt, ms, launches = timed_adam(eng, n, steps, warm)
    hbm, valu = rooflines("cpep gradient launch <2,4,2,2> (library's path selector)", ms, launches, n,
                          cpep_algo_bytes(T_OBS, 2, True), cpep_ops(arch, N_STEPS, T_OBS, 2, True))
    out["cpep2_4_1e5"] = {"config": "reference c-peptide cUDE (02-conditional.jl:22): 2x4x4x1, 2 states, 1e5 subjects, "
                                    "fwd + adjoint + Adam",
                          "value": n / dt, "unit": "subject-trajectories/s", "ms_per_step": dt * 1e3,
                          "roofline": hbm, "roofline_valu": valu}
    eng.close()
    # ---- the same instance as the reference itself solves and differentiates it: adaptive Tsit5 (abstol 1e-6, reltol
    # 1e-3) + the adjoint of the accepted step sequence (= AutoForwardDiff through solve) + Adam
    eng = Engine("cpep", arch, n_steps=0, n_state=2, device=device)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn4, pop["beta0"])
    eng.adam_init(1e-2)
    eng.set_option("auto_regroup", 0)                                       # (cude_adam_run would regroup by itself)
    dt_plain, ms_plain, _ = timed_adam(eng, n, steps, warm)                 # subjects in the caller's order
    eng.set_option("auto_regroup", 1)
    eng.loss_grad(want_cond_grad=False)
    spread = eng.adaptive_regroup()        # launch ordered by accepted-step count: a wave's lanes finish together
    dt, ms, launches = timed_adam(eng, n, steps, warm)
    acc_steps = np.array([len(eng.adaptive_steps(i)[0]) for i in range(0, n, n // 500)])
    out["cpep2_4_1e5_adaptive"] = {
        "config": "reference c-peptide cUDE, 2x4x4x1, 2 states, 1e5 subjects, ADAPTIVE Tsit5 forward (the reference's "
                  "solver settings) + adjoint of the accepted steps + Adam",
        "value": n / dt, "unit": "subject-trajectories/s", "ms_per_step": dt * 1e3,
        "kernel": "adaptive_unrolled_kernel<CpepAd<Mlp<2,4,2,1>>,grad>", "kernel_ms": ms, "launches": launches,
        "accepted_steps_per_subject": {"min": int(acc_steps.min()), "median": float(np.median(acc_steps)),
                                       "max": int(acc_steps.max())},
        "regrouped": {"note": "launch ordered by the accepted-step counts of the last evaluation (cude_adaptive_regroup; "
                              "cude_adam_run does it by itself for populations >= 8192; per-subject results unchanged bit "
                              "for bit)",
                      "mean_step_spread_within_a_wave": {"before": spread[0], "after": spread[1]},
                      "ms_per_step_in_the_callers_order": dt_plain * 1e3, "kernel_ms_in_the_callers_order": ms_plain},
        "note": "work per subject is data-dependent (a wave runs as long as its slowest lane): no fixed algorithmic "
                "flop count, hence no roofline fraction; 5 network evaluations per trial step forward, 5 VJPs per "
                "accepted step in reverse, 8 B/subject/step of tape (PMC: profiles/pmc_traffic.json)"}
    arec, _ = pmc_record("adaptive_grad", n)
    if arec and arec.get("sq", {}).get("SQ_INSTS_VALU"):
        a = out["cpep2_4_1e5_adaptive"]
        a["hbm_traffic_per_launch"] = arec.get("hbm_bytes_per_launch")
        a["valu_slot_utilisation"] = (arec["sq"]["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INSTRUCTION /
                                      (N_SIMD * ms * 1e-3 * CLOCK_GHZ * 1e9))
        a["valu_instructions_per_wave"] = arec["sq"]["SQ_INSTS_VALU"] / arec["sq"]["SQ_WAVES"]
        a["kernel_ms_rocprof"] = rocprof_figures(arec)
    eng.close()
    # ---- suppression instance: 4->3x5->1, 3 states, T = 8, 1e5 subjects, fwd + adjoint + Adam
    n, arch = 100000, (4, 3, 5)
    tp, data, theta = synthetic_suppression(n, 779)
    eng = Engine("supp", arch, n_steps=N_STEPS, lam=0.01, device=device)
    eng.set_population_supp(tp, data)
    eng.set_params(glorot(arch, 1234), theta)
    eng.adam_init(1e-3)
    dt, ms, launches = timed_adam(eng, n, steps, 10)
    scratch = 2 * (6 * N_STEPS + 1) * 2 * 8          # stage inputs (states 2, 3) written by the forward, read by the reverse sweep
    srec, _ = pmc_record("supp_stage_inputs", n)
    hbm, valu = rooflines("supp_kernel<3,5,grad>", ms, launches, n, supp_algo_bytes(8, True),
                          supp_ops(arch, N_STEPS, 8, True), traffic=srec.get("hbm_bytes_per_launch") if srec else None,
                          sq=srec.get("sq") if srec else None, rocprof=srec,
                          note=f"+ {scratch} B/subject of stage-input scratch (stored linearisation points instead of "
                               f"a recomputed forward sweep; profiles/r02/supp_scratch_tradeoff.txt)")
    out["supp_1e5"] = {"config": "suppression cUDE (suppression.jl:18): 4x3x3x3x3x3x1, 3 states, T=8, 1e5 subjects, "
                                 "fwd + adjoint + Adam, lambda=0.01",
                       "value": n / dt, "unit": "subject-trajectories/s", "ms_per_step": dt * 1e3,
                       "roofline": hbm, "roofline_valu": valu}
    eng.close()
    # the same kernel at a size that fills the chip exactly: 131 072 subjects = 2 048 waves on 2 048 wave slots (256 CUs x
    # 4 SIMDs x 2 resident waves), where 1e5 subjects are 1 563 waves = 76 % of one fill -- the fraction of the vector
    # peak at 1e5 mixes the kernel's own efficiency with that idle quarter
    n_fill = 131072
    tp_f, data_f, theta_f = synthetic_suppression(n_fill, 779)
    eng = Engine("supp", arch, n_steps=N_STEPS, lam=0.01, device=device)
    eng.set_population_supp(tp_f, data_f)
    eng.set_params(glorot(arch, 1234), theta_f)
    eng.adam_init(1e-3)
    dt_f, ms_f, launches_f = timed_adam(eng, n_fill, steps, 10)
    _, valu_f = rooflines("supp_kernel<3,5,grad>", ms_f, launches_f, n_fill, supp_algo_bytes(8, True),
                          supp_ops(arch, N_STEPS, 8, True))
    out["supp_1e5"]["at_one_full_fill"] = {
        "subjects": n_fill, "waves": n_fill // 64, "wave_slots": 2048, "fill_at_1e5": (n + 63) // 64 / 2048.0,
        "value": n_fill / dt_f, "ms_per_step": dt_f * 1e3, "kernel_ms": ms_f,
        "roofline_valu_frac": valu_f["frac"], "achieved_tflops": valu_f["achieved"]}
    eng.close()
    # ---- the same instance as the reference solves it (Tsit5 adaptive, EnsembleThreads: suppression_model.jl:113,123)
    eng = Engine("supp", arch, n_steps=0, lam=0.01, device=device)
    eng.set_population_supp(tp, data)
    eng.set_params(glorot(arch, 1234), theta)
    eng.adam_init(1e-3)
    eng.loss_grad(want_cond_grad=False)
    spread = eng.adaptive_regroup()
    dt, ms, launches = timed_adam(eng, n, steps, 10)
    acc_steps = np.array([len(eng.adaptive_steps(i)[0]) for i in range(0, n, n // 500)])
    asrec, _ = pmc_record("adaptive_supp_grad", n)
    out["supp_1e5_adaptive"] = {
        "config": "suppression cUDE, 4x3x3x3x3x3x1, 3 states, T=8, 1e5 subjects, ADAPTIVE Tsit5 forward (the reference's "
                  "solver) + adjoint of the accepted steps + Adam, lambda=0.01",
        "value": n / dt, "unit": "subject-trajectories/s", "ms_per_step": dt * 1e3,
        "kernel": "adaptive_unrolled_supp_kernel<SuppAd<3,5>,grad>", "kernel_ms": ms, "launches": launches,
        "accepted_steps_per_subject": {"min": int(acc_steps.min()), "median": float(np.median(acc_steps)),
                                       "max": int(acc_steps.max())},
        "mean_step_spread_within_a_wave": {"before": spread[0], "after": spread[1]},
        "hbm_traffic_per_launch": asrec.get("hbm_bytes_per_launch") if asrec else None,
        "kernel_ms_rocprof": rocprof_figures(asrec),
        "note": "6 network evaluations per trial step forward, 6 VJPs per accepted step in reverse at the stage inputs the "
                "forward sweep left on the tape (136 B per step and subject); data-dependent work: no roofline fraction"}
    eng.close()
    # ---- configs[4] on one GPU: SAEM E-step, 1e4 subjects x 100 Metropolis steps
    n, n_mc, arch = 10000, 100, (2, 4, 2)
    eng, pop = cpep_engine(Engine, arch, 2, n, 780, device, nn4)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn4, pop["beta0"])
    rng = np.random.default_rng(5)
    z, u = rng.standard_normal((n_mc, n)), rng.random((n_mc, n))
    eng.mh_estep(z[:10], u[:10], 0.5, -0.6, 0.8, 0.3)
    t0 = time.perf_counter()
    eng.mh_estep(z, u, 0.5, -0.6, 0.8, 0.3)          # host-supplied draws: 16 MB cross PCIe inside the call
    dt_host_draws = time.perf_counter() - t0
    def estep_best_of(e, reps=3):            # draws generated on the device (Philox4x32-10); the same chain every time
        best, acc_ = 1e9, None
        for _ in range(reps):
            e.set_params(nn4, pop["beta0"])
            e.set_rng(20250905)
            t0_ = time.perf_counter()
            acc_ = e.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
            best = min(best, time.perf_counter() - t0_)
        return best, acc_
    eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=10)
    dt, acc = estep_best_of(eng)              # the library's default: speculative steps where the size rule says so (option "mh_spec")
    _, state_spec = eng.get_params()
    # the same E-step with one Metropolis step per launch pair (mh_spec = 0): the launches the roofline figures below count
    eng.set_option("mh_spec", 0)
    eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=10)
    dt_plain, acc_plain = estep_best_of(eng)
    _, state_plain = eng.get_params()
    same_chain = bool(np.array_equal(acc, acc_plain) and np.array_equal(state_spec, state_plain))
    eng.set_kernel_timing(True)                                        # kernel time: a second E-step with the events on
    eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
    ms, launches = eng.kernel_time_ms()
    eng.set_kernel_timing(False)
    hbm, valu = rooflines("time-split forward launches <2,4,2,2> inside cude_mh_estep (mh_spec = 0)", ms, launches, n,
                          cpep_algo_bytes(T_OBS, 2, False), cpep_ops(arch, N_STEPS, T_OBS, 2, False))
    out["saem_estep_1e4x100"] = {"config": "BASELINE configs[4] on one GPU: SAEM E-step (saem.jl:86-108,177-186), 1e4 "
                                           "subjects x 100 Metropolis steps (gamma = 1: burn-in phase), 2x4x4x1, "
                                           "draws generated on the device (counter-based Philox4x32-10)",
                                 "ms_per_estep_with_host_supplied_draws": dt_host_draws * 1e3,
                                 "value": n * n_mc / dt, "unit": "Metropolis draws/s", "ms_per_estep": dt * 1e3,
                                 "timing": "best of 3 E-steps",
                                 "ms_per_estep_one_step_per_launch_pair": dt_plain * 1e3,
                                 "speculative_steps": "library default (option mh_spec = -1: depth by population size; "
                                                      "profiles/r05/estep_speculative.txt)",
                                 "same_chain_as_one_step_per_launch_pair": same_chain,
                                 "forward_solves_per_s": n * (n_mc + 1) / dt,
                                 "reference_equivalent_solves_per_s": 2 * n * n_mc / dt,
                                 "solves_note": "the reference solves the proposal AND the current state in every "
                                                "step (2 x 1e6 solves); with gamma = 1 the current state's SSE is "
                                                "carried over from the step that accepted it (identical bits), so "
                                                f"{n_mc + 1} ensemble launches do the work; HIP events time every 8th "
                                                f"Metropolis step ({launches} launches timed)",
                                 "acceptance_rate": float(acc.sum()) / (n * n_mc),
                                 "roofline": hbm, "roofline_valu": valu}
    eng.close()
    # one GPU's share of the same E-step on 8 GPUs (1 250 subjects): what the sharded configs[4] will run per rank
    n8 = n // 8
    eng, pop8 = cpep_engine(Engine, arch, 2, n8, 780, device, nn4)
    eng.set_population_cpep(pop8["tp"], pop8["G"], pop8["obs"], pop8["age"], pop8["t2dm"])
    pop = pop8
    eng.set_params(nn4, pop8["beta0"])
    eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=10)
    dt8, acc8 = estep_best_of(eng)
    eng.set_option("mh_spec", 0)
    eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=10)
    dt8_plain, acc8_plain = estep_best_of(eng)
    out["saem_estep_1e4x100"]["one_gpus_share_on_8_gpus"] = {
        "subjects": n8, "ms_per_estep": dt8 * 1e3, "ms_per_estep_one_step_per_launch_pair": dt8_plain * 1e3,
        "same_acceptance_counts": bool(np.array_equal(acc8, acc8_plain)),
        "note": "the E-step is a chain of dependent launches whose cost is one wave's latency: sharding alone shortens it "
                "only this much; speculative steps use the SIMDs a small shard leaves idle"}
    eng.close()
    # the same E-steps in the API mirrors' DEFAULT discretisation (the reference's adaptive solve, n_steps = 0): a Metropolis
    # step there is one wave's whole adaptive solve; the candidates of d steps run side by side as parameter sets of ONE
    # launch (option "mh_spec", depth by population size) -- the same chain
    ad = {}
    for n_ad, popd in ((n, None), (n8, pop8)):
        if popd is None:
            e0, popd = cpep_engine(Engine, arch, 2, n_ad, 780, device, nn4)
            e0.close()
        pop = popd
        res = {}
        for depth in (-1, 0):
            eng = Engine("cpep", arch, n_steps=0, n_state=2, device=device)
            eng.set_option("mh_spec", depth)
            eng.set_population_cpep(popd["tp"], popd["G"], popd["obs"], popd["age"], popd["t2dm"])
            eng.set_params(nn4, popd["beta0"])
            eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=10)
            res[depth] = estep_best_of(eng, reps=2)
            eng.close()
        ad[str(n_ad)] = {"ms_per_estep": res[-1][0] * 1e3, "ms_per_estep_one_step_per_launch": res[0][0] * 1e3,
                         "same_acceptance_counts": bool(np.array_equal(res[-1][1], res[0][1]))}
        # the stochastic-approximation phase (gamma < 1: the next state is a blend whose likelihood nothing carries): the
        # proposal and both possible next states in one launch, d steps per launch by speculation; fixed-step and adaptive
        for steps_g, tag in ((N_STEPS, "fixed"), (0, "adaptive")):
            eng = Engine("cpep", arch, n_steps=steps_g, n_state=2, device=device)
            eng.set_population_cpep(popd["tp"], popd["G"], popd["obs"], popd["age"], popd["t2dm"])
            best = 1e9
            for rep in range(3):
                eng.set_params(nn4, popd["beta0"])
                eng.set_rng(20250905)
                t0_ = time.perf_counter()
                eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, 1.0, 0.25, n_mc=n_mc)
                if rep:
                    best = min(best, time.perf_counter() - t0_)
            eng.close()
            ad[str(n_ad)][f"ms_per_estep_gamma_0.25_{tag}"] = best * 1e3
    out["saem_estep_1e4x100"]["adaptive_default_mode"] = dict(
        ad, note="n_steps = 0 (adaptive Tsit5, abstol 1e-6, reltol 1e-3): keys = subjects; speculative rounds put the "
                 "2^d - 1 candidate states of d steps into one adaptive launch + a resolver launch")
    return out


def saem_estep_sharded(Engine, device, world, rank, dist, torch, coll, agreed, n_total=10000, n_mc=100, reps=5):
    """BASELINE configs[4] on N GPUs: the SAEM E-step (src/saem.jl:86-108,177-186) of n_total subjects x n_mc Metropolis
    steps, subjects sharded over the ranks (the E-step itself needs no communication), followed by the one exchange a
    SAEM iteration makes: the sum over ranks of [accepted, sum p, sum p^2, n] (acceptance rate, eta <- mean(p), Omega <-
    var(p), saem.jl:196-205), timed inside the same region.  The sum goes through the library's own transport
    (cude_comm_allreduce_host: the peer-write exchange, else RCCL inside the library), torch.distributed only when
    neither comes up.  Returns this rank's view; the caller takes the max time."""
    from cude.parallel import ShardedTrainer, attach_exchange
    arch = (2, 4, 2)
    n = n_total // world + (1 if rank < n_total % world else 0)
    first = rank * (n_total // world) + min(rank, n_total % world)
    nn4 = glorot(arch, 4321)
    eng, pop = cpep_engine(Engine, arch, 2, n, 780 + rank, device, nn4)
    transport, why = None, None
    if os.environ.get("CUDE_BENCH_TRANSPORT", "xchg") == "xchg":
        ok, why = attach_exchange(eng, coll, 30.0)
        transport = "xchg" if ok else None
    if transport is None and os.environ.get("CUDE_BENCH_TRANSPORT", "xchg") in ("xchg", "rccl"):
        try:
            ShardedTrainer.attach_rccl(Engine, eng, coll)
            ok = True
        except Exception as exc:  # noqa: BLE001
            ok, why = False, f"{why}; cude_comm_init: {exc}"
        if agreed(ok):
            transport = "rccl"
        elif ok:                   # up here, not everywhere: start again without it
            eng.close()
            eng, pop = cpep_engine(Engine, arch, 2, n, 780 + rank, device, nn4)
    if transport is None:
        transport = "torch"
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn4, pop["beta0"])
    eng.set_rng(20250905, first)               # one global stream of draws: the chain does not depend on the sharding
    eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=10)
    ctl = torch.device("cuda", device)

    def total(v):
        if transport != "torch":
            return eng.allreduce_host(v)
        t = torch.tensor(v, dtype=torch.float64, device=ctl)
        dist.all_reduce(t)
        return t.cpu().numpy()
    total([0.0, 0.0, 0.0, 0.0])                # warm the transport
    torch.cuda.synchronize()
    dist.barrier()
    t_red = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        acc = eng.mh_estep(None, None, 0.5, -0.6, 0.8, 0.3, n_mc=n_mc)
        _, p = eng.get_params()
        t1 = time.perf_counter()
        tot = total([float(acc.sum()), float(p.sum()), float(p @ p), float(n)])
        t_red += time.perf_counter() - t1
    dt = (time.perf_counter() - t0) / reps
    lost = eng.xchg_info()[3] if transport == "xchg" else 0
    eng.close()
    return dict(dt=dt, dt_allreduce=t_red / reps, acceptance=float(tot[0]) / (tot[3] * n_mc), n_seen=float(tot[3]),
                mean_p=float(tot[1] / tot[3]), subjects_per_gpu=n, transport=transport, transport_note=why,
                timed_out_waits=lost)


# ------------------------------------------------------------------------------------------ main
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
    ap.add_argument("--no-extra", action="store_true", help="skip the other single-GPU configurations")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # CUDE_BENCH_FORCE_DIST=1 (development only): run the N > 1 code path with ONE rank -- process group, communicator
    # with its self-test, the cross-check of the two transports, the sharded SAEM E-step -- on a 1-GPU box
    dist_on = world > 1 or os.environ.get("CUDE_BENCH_FORCE_DIST") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # stdout carries ONE line, the JSON: libraries that print banners there (RCCL 2.26 announces its version, host name
    # and library path on stdout at communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        raise SystemExit("HSA_ENABLE_IPC_MODE_LEGACY must be 0 for multi-process RCCL on this pool")
    # CUDE_BENCH_REHEARSAL=1 (development only): several ranks share the GPUs that exist and torch.distributed uses
    # gloo, so that the multi-rank control flow of this script can be exercised on a 1-GPU box
    rehearsal = os.environ.get("CUDE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    ctl = "cpu" if rehearsal else "cuda"          # device of the small control tensors handed to torch.distributed
    torch.cuda.set_device(local_rank)
    rccl_log = None
    if dist_on:
        # RCCL's own account of a failure (WARN level: silent when all is well), per process, for `rccl_error` below
        rccl_log = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cude_rccl_{os.getpid()}.log")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cude.engine import Engine   # after torch: one shared HIP runtime

    n_local = args.subjects_per_gpu
    nn = glorot(ARCH, 1234)
    eng, pop = cpep_engine(Engine, ARCH, N_STATE, n_local, 20250905 + rank, local_rank, nn)
    # Transport of the one sum per step, in order of preference: "xchg" = the library's peer-write exchange (inside the
    # reduction kernels: no extra launch, captured graphs stay, sums in rank order), "rccl" = ncclAllReduce inside the
    # library on the context's stream, "torch" = torch.distributed on a tensor aliasing the context's P+2 doubles.
    # CUDE_BENCH_TRANSPORT=rccl|torch skips the ones in front.  Every rank issues the same sequence of collectives
    # whatever fails locally.
    want = os.environ.get("CUDE_BENCH_TRANSPORT", "xchg")
    transport = None
    rccl_info = None
    xchg_info = None
    rccl_error = None           # why a transport in front of the chosen one was not used (this rank's view), for the JSON line
    xchg_error = None

    def agreed(flag):
        t = torch.tensor([1.0 if flag else 0.0], device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.item() != 0

    coll = None
    xchg_attempts = []
    if dist_on:
        # ---- the exchange: export the mailbox, all-gather the 128-byte handles, map the peers (self-test inside), agree;
        # a memory kind that fails on any rank is given up by all (cude/parallel.py attach_exchange, include/cude.h)
        from cude.parallel import TorchCollective, attach_exchange
        coll = TorchCollective(dist, None if rehearsal else torch.device("cuda", local_rank))
        if agreed(want == "xchg"):
            ok, why = attach_exchange(eng, coll, 30.0, log=xchg_attempts.append)
            if ok:
                transport = "xchg"
                xchg_info = eng.xchg_info()
            else:
                xchg_error = why
                print(f"[rank {rank}] peer-write exchange unavailable ({why})", file=sys.stderr)
        else:
            xchg_error = "CUDE_BENCH_TRANSPORT skipped it"
        # ---- RCCL inside the library: the fallback, and (when the exchange is up) the second transport of `transports`
        my_id = bytes(128)
        ok = want in ("xchg", "rccl")
        if ok:
            try:
                my_id = Engine.comm_unique_id()
            except Exception as exc:  # e.g. librccl not loadable
                rccl_error = f"cude_comm_unique_id: {exc}"
                ok = False
        if agreed(ok):
            uid = torch.tensor(list(my_id), dtype=torch.uint8, device=ctl)
            dist.broadcast(uid, 0)
            try:
                eng.comm_init(world, rank, bytes(uid.cpu().tolist()))
                rccl_info = eng.comm_info()          # (ranks, rank, version) as RCCL itself reports them
                if rccl_info[0] != world or rccl_info[1] != rank:
                    raise RuntimeError(f"communicator reports {rccl_info}, expected ({world}, {rank})")
            except Exception as exc:
                rccl_error = f"cude_comm_init: {exc}"
                print(f"[rank {rank}] cude_comm_init failed ({exc})", file=sys.stderr)
                ok = False
            if agreed(ok):
                transport = transport or "rccl"
            else:
                rccl_info = None
                if rccl_error is None:
                    rccl_error = "another rank could not build the communicator"
        elif rccl_error is None:
            rccl_error = "CUDE_BENCH_TRANSPORT skipped it" if want == "torch" else "another rank could not load librccl"
        if rccl_error and rccl_log and os.path.exists(rccl_log):
            tail = open(rccl_log, errors="replace").read()[-1500:].strip()
            if tail:
                rccl_error += " | RCCL: " + tail
        if transport is None:
            # cude_loss_grad_partial_device -> dist.all_reduce on a tensor ALIASING the context's P+2 doubles (RCCL via
            # PyTorch, in place on the device) -> cude_adam_apply_device: no host copy in the step.  Under rehearsal
            # (gloo, CPU tensors) the vector goes through the host as before.
            transport = "torch" if not rehearsal else "host"
            eng.close()
            eng = Engine("cpep", ARCH, n_steps=N_STEPS, n_state=N_STATE, device=local_rank)
    # ---- N > 1: the exchange's first real steps, BEFORE anything is measured.  The self-test at attach moves a few hundred
    # doubles; a transport that passes it and then loses a word under load (a wait that runs out of time: CUDE_ERR_COMM)
    # would otherwise end this run without a JSON line.  Every rank makes the same calls whatever happens locally; if any
    # rank fails, all hand over to RCCL (or to torch.distributed on a fresh context) and the line says why.
    if dist_on and transport == "xchg":
        ok = True
        try:
            eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
            eng.set_params(nn, pop["beta0"])
            eng.adam_init(1e-2)
            for _ in range(4):
                eng.adam_step()
            eng.adam_run(16)                       # captured graphs
            eng.synchronize()
            ok = eng.xchg_info()[3] == 0
            if not ok:
                xchg_error = "a wait of the exchange ran out of time during its first steps"
        except Exception as exc:  # noqa: BLE001
            ok = False
            xchg_error = f"the exchange failed during its first steps: {exc}"
        if not agreed(ok):
            xchg_error = xchg_error or "the exchange failed during its first steps on another rank"
            print(f"[rank {rank}] {xchg_error}", file=sys.stderr)
            xchg_info = None
            if rccl_info is not None:
                try:
                    eng.xchg_enable(False)
                except Exception:  # noqa: BLE001  (already released on this rank)
                    pass
                transport = "rccl"
            else:
                transport = "torch" if not rehearsal else "host"
                eng.close()
                eng = Engine("cpep", ARCH, n_steps=N_STEPS, n_state=N_STATE, device=local_rank)
    lib_transport = transport in ("xchg", "rccl") or not dist_on     # the step is the library's own (cude_adam_step / _run)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])   # global count all-reduced here
    if not lib_transport:
        eng.set_global_subjects(n_local * world)
    eng.set_params(nn, pop["beta0"])
    eng.adam_init(1e-2)
    alias = {}                 # engine -> tensor aliasing its partial vector (device-resident torch transport)

    def host_step(e):
        if rehearsal:          # gloo on CPU tensors: through the host
            part, _ = e.loss_grad_partial()
            t = torch.from_numpy(part)
            dist.all_reduce(t)
            return e.adam_apply(t.numpy())
        if e not in alias:
            alias[e] = e.partial_tensor(torch, torch.device("cuda", local_rank))
        e.loss_grad_partial_device()          # (synchronises the context's stream)
        dist.all_reduce(alias[e])             # in place, on PyTorch's stream
        torch.cuda.current_stream().synchronize()
        return e.adam_apply_device()

    def train_step(want_loss=True):
        """One optimiser iteration over ALL ranks' subjects; returns the global loss."""
        if lib_transport:
            return eng.adam_step(want_loss=want_loss)
        return host_step(eng)

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()

    # ---- N > 1: the same optimiser step through both transports, from the same state (fresh Adam moments)
    allreduce_check = None
    if dist_on and lib_transport:
        twin = Engine("cpep", ARCH, n_steps=N_STEPS, n_state=N_STATE, device=local_rank)
        twin.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
        twin.set_global_subjects(n_local * world)
        twin.set_params(nn, pop["beta0"])
        twin.adam_init(1e-2)
        loss_a = eng.adam_step()
        loss_b = host_step(twin)
        (nn_a, cond_a), (nn_b, cond_b) = eng.get_params(), twin.get_params()
        twin.close()
        diff = max(abs(loss_a - loss_b) / abs(loss_b), float(np.max(np.abs(nn_a - nn_b)) / np.max(np.abs(nn_b))),
                   float(np.max(np.abs(cond_a - cond_b)) / np.max(np.abs(cond_b))))
        t = torch.tensor([diff], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_check = float(t.item())
        # (the same verdict on every rank.)  An exchange that passed its self-test but forms other sums than
        # torch.distributed does is not measured: the communicator takes over, and the line says why.
        if transport == "xchg" and not allreduce_check <= 1e-9 and rccl_info is not None:
            xchg_error = f"the exchange's step differs from torch.distributed's by {allreduce_check:.3g}: RCCL takes over"
            eng.xchg_enable(False)
            transport = "rccl"

    # ---- untimed: bring the GPU to its steady clock (same count on every rank: the collectives must pair up), then
    # the W warm-up steps of the contract
    for _ in range(PREWARM_STEPS):
        train_step(want_loss=False)
    # ---- N > 1 with both library transports up: the headline goes through the one that is faster HERE (a short untimed
    # trial of each, the slowest rank's time; the exchange has never run across devices before this node).  `transports`
    # below still reports both over the full K steps.
    transport_trial = None
    if dist_on and world > 1 and transport == "xchg" and rccl_info is not None and not rehearsal:
        def trial(k=16):
            eng.adam_run(k)
            barrier()
            tq = time.perf_counter()
            eng.adam_run(k)
            barrier()
            t = torch.tensor([(time.perf_counter() - tq) / k], dtype=torch.float64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        t_x = trial()
        eng.xchg_enable(False)
        t_r = trial()
        transport_trial = {"xchg_ms_per_step": t_x * 1e3, "rccl_ms_per_step": t_r * 1e3}
        if t_r < 0.98 * t_x:
            transport = "rccl"
            transport_trial["chosen"] = "rccl"
        else:
            eng.xchg_enable(True)
            transport_trial["chosen"] = "xchg"
    # HIP events around every TIMING_PERIOD-th gradient launch of the timed region (a pair costs ~4.5 us of stream time and
    # a gap on either side of the kernel: on every launch that is 2 % of the step being measured); the warm-up also
    # creates the events the timed steps will reuse
    eng.set_kernel_timing(TIMING_PERIOD)
    for _ in range(args.warmup):
        train_step()
    if lib_transport:
        eng.adam_run(max(args.steps, 1))  # untimed: sizes the event pool and the loss trace for K queued steps
    eng.kernel_time_ms()                  # drop the warm-up's timings
    # ---- timed region: EXACTLY K optimiser steps.  With the library's own transport (one GPU, or RCCL) they are
    # queued back to back by cude_adam_run: every step's loss is reduced on the device into a K-entry trace that is
    # handed over after the K-th step (what the reference's training callback does with the loss: push it to a
    # vector), so no step waits for the host.  With the torch.distributed transport the host is in every step.
    barrier()
    t0 = time.perf_counter()
    if lib_transport:
        losses = eng.adam_run(args.steps)
        loss = float(losses[-1])
        assert losses.size == args.steps and np.all(np.isfinite(losses))
    else:
        for _ in range(args.steps):
            loss = train_step()
    barrier()
    dt = time.perf_counter() - t0
    kern_mean, kern_ms, kern_min, n_launch = eng.kernel_time_stats()     # kern_ms = the MEDIAN of the sampled launches
    eng.set_kernel_timing(False)

    # the same K steps with the loss returned to the host after every step (reported as extra information)
    barrier()
    eng.set_kernel_timing(1)              # (this untimed pass carries events on EVERY launch: K samples beside the timed region's K/4)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        train_step()
    barrier()
    dt_sync = time.perf_counter() - t1
    sync_mean, sync_median, sync_min, sync_n = eng.kernel_time_stats()
    eng.set_kernel_timing(False)

    # the same K steps replayed from captured graphs (no kernel timing: cude_adam_run's production mode), and -- N > 1 --
    # through every other transport that came up, for the record (`transports`): all untimed by the contract's clock
    def timed_run(k):
        eng.adam_run(k)                                  # captures / warms
        barrier()
        tq = time.perf_counter()
        tr = eng.adam_run(k)
        barrier()
        assert np.all(np.isfinite(tr))
        return (time.perf_counter() - tq) / k

    dt_graph = None
    transports = {}
    if lib_transport:
        dt_graph = timed_run(args.steps)
        if dist_on:
            transports[transport] = {"ms_per_step": dt_graph * 1e3,
                                     "launch_mode": "graph replay" if transport == "xchg" else "plain"}
            if transport == "xchg" and rccl_info is not None:
                eng.xchg_enable(False)                   # RCCL on the context's stream; plain launches (not capturable)
                transports["rccl"] = {"ms_per_step": timed_run(args.steps) * 1e3, "launch_mode": "plain"}
                eng.xchg_enable(True)
            elif transport == "rccl" and transport_trial is not None:      # (the trial preferred RCCL: the exchange is still attached)
                eng.xchg_enable(True)
                transports["xchg"] = {"ms_per_step": timed_run(args.steps) * 1e3, "launch_mode": "graph replay"}
                eng.xchg_enable(False)
            if not rehearsal:
                tw = time.perf_counter()
                for _ in range(args.steps):
                    host_step(eng)
                barrier()
                transports["torch"] = {"ms_per_step": (time.perf_counter() - tw) / args.steps * 1e3,
                                       "launch_mode": "host in every step"}

    # rank 0's shard against the CPU port (the other ranks wait at the barrier behind it)
    cpu_rev = cpu_fwd = parity = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_rev, cpu_fwd, parity = cpu_baseline(pop, nn, args.cpu_sample, eng, world if dist_on else 1)
    if dist_on:
        eng.synchronize()
        dist.barrier()
        if xchg_info is not None:
            xchg_info = eng.xchg_info()                  # (timed-out waits over the whole run: must be 0)

    saem = None
    if dist_on and not rehearsal and not args.no_extra:
        eng.close()
        eng = None
        saem = saem_estep_sharded(Engine, local_rank, world, rank, dist, torch, coll, agreed)
        t = torch.tensor([saem["dt"], saem["dt_allreduce"]], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        saem["dt"], saem["dt_allreduce"] = float(t[0]), float(t[1])

    kern_min = kern_max = kern_ms
    if dist_on:
        vals = [dt, dt_sync, kern_ms, -kern_ms, dt_graph or 0.0] + [v["ms_per_step"] for v in transports.values()]
        t = torch.tensor(vals, dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_sync, kern_max, kern_min = float(t[0]), float(t[1]), float(t[2]), -float(t[3])
        dt_graph = float(t[4]) if dt_graph is not None else None
        for k, name in enumerate(transports):
            transports[name]["ms_per_step"] = float(t[5 + k])
        kern_ms = kern_max

    if rank == 0:
        n_total = n_local * world
        value = n_total * args.steps / dt
        hrec, traffic_note = pmc_record("headline", n_local)
        traffic = hrec.get("hbm_bytes_per_launch") if hrec else None
        hbm, valu = rooflines("cpep_kernel<Mlp<2,6,2,1>,3,grad>", kern_ms, n_launch, n_local,
                              cpep_algo_bytes(T_OBS, N_STATE, True), cpep_ops(), traffic, sq=hrec.get("sq") if hrec else None,
                              rocprof=hrec)
        if traffic is None:
            hbm["traffic_note"] = traffic_note
        out = {
            "metric": "subject-trajectories/sec (fwd+adjoint, 3-state ODE, 30 steps)",
            "value": value, "unit": "subject-trajectories/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"CPEP3 cUDE training step (fwd Tsit5 x{N_STEPS} + discrete adjoint + Adam), "
                                   f"2x6x6x1 MLP, T={T_OBS}, {n_local} subjects/GPU "
                                   f"({n_total} total; BASELINE configs[2]/[3] shape)",
                       "subjects_per_gpu": n_local, "parallelism": f"subject-shard x{world}",
                       "allreduce": transport if dist_on else None},
            "roofline": hbm, "roofline_valu": valu,
            "kernel_ms_samples": {"median": kern_ms, "mean": kern_mean, "min": kern_min, "launches": n_launch,
                                  "note": "HIP events around every TIMING_PERIOD-th gradient launch of the timed region; "
                                          "roofline.achieved uses the median (this rank's; N > 1: the slowest rank's)",
                                  "every_launch_of_the_read_back_pass": {"median": sync_median, "mean": sync_mean, "min": sync_min,
                                                                         "launches": sync_n}},
            "value_with_loss_read_back_every_step": n_total * args.steps / dt_sync,
            "final_loss": loss, "prewarm_steps": PREWARM_STEPS, "kernel_source_sha": kernel_source_sha(),
            "launch_mode": f"plain+events/{TIMING_PERIOD}" if lib_transport else "host in every step",
            "ms_per_step_graph_replay": dt_graph * 1e3 if dt_graph is not None else None,
        }
        if dist_on:
            out["rccl_ranks"] = rccl_info[0] if rccl_info else None
            out["rccl_version"] = rccl_info[2] if rccl_info else None
            out["allreduce_check"] = allreduce_check
            out["rccl_error"] = rccl_error
            out["xchg"] = ({"ranks": xchg_info[0], "memory_kind": {3: "uncached", 1: "fine-grained", 0: "device"}[xchg_info[2]],
                            "timed_out_waits": xchg_info[3], "attach_attempts": len(xchg_attempts) + 1,
                            "attempts_given_up": xchg_attempts} if xchg_info else None)
            out["xchg_error"] = xchg_error
            out["transport_trial"] = transport_trial
            out["transports"] = transports
            if saem is not None:
                out["saem_estep_1e4x100_sharded"] = {
                    "config": f"BASELINE configs[4]: SAEM E-step, 1e4 subjects x 100 Metropolis steps sharded over "
                              f"{world} GPUs ({saem['subjects_per_gpu']} subjects on rank 0), 2x4x4x1, device-side draws, "
                              f"+ the 4-double sum of a SAEM iteration through the library's transport "
                              f"(cude_comm_allreduce_host: {saem['transport']})",
                    "transport": saem["transport"], "transport_note": saem["transport_note"],
                    "timed_out_waits": saem["timed_out_waits"],
                    "value": 10000 * 100 / saem["dt"], "unit": "Metropolis draws/s", "ms_per_estep": saem["dt"] * 1e3,
                    "ms_allreduce_and_readback": saem["dt_allreduce"] * 1e3, "acceptance_rate": saem["acceptance"],
                    "subjects_seen": saem["n_seen"]}
            out["kernel_ms_per_rank"] = {"min": kern_min, "max": kern_max}
            out["hsa_ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
            if rehearsal:
                out["rehearsal"] = (f"{world} ranks time-share {torch.cuda.device_count()} GPU(s): a dry run of the "
                                    "N > 1 control flow, not a throughput measurement")
        if cpu_rev is not None:
            out["cpu_baseline"], out["cpu_baseline_forward_mode"] = cpu_rev, cpu_fwd
            if parity is not None:
                out["parity_vs_cpu_baseline"] = parity
        if not dist_on and not args.no_extra:
            eng.close()
            eng = None
            out["extra"] = extras(Engine, local_rank)
        sys.stdout.flush()
        with os.fdopen(json_fd, "w") as real_stdout:
            real_stdout.write(json.dumps(out) + "\n")
    if eng is not None:
        eng.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
