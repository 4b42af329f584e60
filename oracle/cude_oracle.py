"""CPU oracle for the conditional-UDE population ensemble solve + gradient.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this file; the product path
(conditional-ude_amd/) never does and fails loudly when its HIP library is absent.

PARITY STATUS.  The reference (Computational-Biology-TUe/conditional-ude) is pure Julia, ships
no tests, and Julia is not installed in the build image, so this restatement cannot be checked
against a RUN of the reference.  It is checked against the reference's STORED outputs:
  * suppression path -- PINNED by known answers: the 25 final objectives of the reference's
    lambda = 1 run (suppression/results/lambda=1.0.jld2) are functions of stored quantities only
    (collapsed networks, no dependence on the unsaved conditional parameters).  The adaptive
    mode below (OrdinaryDiffEq Tsit5 defaults restated: solve_adaptive) reproduces them to
    4e-10, the fixed-step mode to 1.26e-6 = the reference solver's own discretisation error
    (tests/test_known_answers.py).
    Likewise the 2 x 25 stored validation objectives of that run (7.6e-7 / 9.4e-7).
  * c-peptide path -- PINNED AT FIGURE RESOLUTION by the reference's own vector figures
    (tests/test_figure_pins.py, tests/golden/figure_traces.npz): 127 simulated c-peptide
    trajectories (stored networks of the conditional and the covariate model, and the symbolic
    model on the external data set; each a function of stored quantities and one scalar) are
    reproduced by the adaptive mode to 0.9e-4 ... 2.3e-4 nmol/L -- the 1/256 px quantisation of
    the figures, ~1e-4 of the plotted range -- and 234 fitted per-subject objectives to a median
    1.1e-4 ... 1.8e-4 of SSE (20 more, symbolic model on the external data, to ~1 %).  No finer c-peptide output of the reference
    exists (it stores no objective), so the fp64 tolerance 1e-6 is NOT pinned against the
    reference on this path; it is additionally soft-pinned by the stored training results of four
    runs (tests/test_soft_pins.py): conditional parameters recovered to ~1e-3, layout /
    input-order negative controls, stationarity.
Hard parity (rtol<=1e-6) is defined between this file, oracle/cude_oracle.c (forward-mode
duals, the reference's AD method) and the HIP kernels (discrete adjoint).

What is restated, with the reference lines followed (paths relative to /root/reference):
  softplus                      src/neural-network.jl:13-15  (naive log(1+exp(x)))
  MLP (SimpleChains TurboDense) src/neural-network.jl:42-58; params per layer
                                [vec_colmajor(W out x in); b], hidden tanh, output softplus
  van_cauter_parameters         src/c-peptide-models.jl:30-42
  c_peptide_kinetics!           src/c-peptide-models.jl:7-14
  conditional_production        src/c-peptide-models.jl:86-94 (NN([dG;e^b]) - NN([0;e^b]),
                                evaluated in every RHS call, as written)
  CPeptideConditionalUDEModel   src/c-peptide-models.jl:170-194 (u0, tspan, linear glucose)
  loss (single / population)    src/parameter-estimation.jl:56-68, 126-140
  loss_sigma                    src/parameter-estimation.jl:70-75
  ude_lsup!                     suppression/src/suppression_model.jl:88-95
  simul / suppression_loss      suppression/src/suppression_model.jl:107-130
  individual_log_likelihood     src/saem.jl:55-66
  mcmc_step                     src/saem.jl:86-108
The ODE solver is third-party (OrdinaryDiffEq Tsit5, not vendored): its published tableau
and dense-output polynomials (Tsitouras 2011) are restated here.  The
benchmark discretisation is FIXED-step Tsit5 (S uniform steps, observations from the free
4th-order interpolant); `solve_adaptive` restates the adaptive controller (abstol 1e-6,
reltol 1e-3) only to compare with the reference's stored results.

All functions take an array namespace `xp` (numpy or torch) so the same restatement gives
values (numpy) and reverse-mode gradients (torch float64 autograd).
"""
import math
import numpy as np

# ----------------------------------------------------------------------------- Tsit5
C = [0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0]
A = [
    [],
    [0.161],
    [-0.008480655492356989, 0.335480655492357],
    [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
    [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
    [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
     -0.028269050394068383],
    [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
     2.324710524099774],
]
BTILDE = [-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
          -0.1447110071732629, 0.5823571654525552, -0.45808210592918697,
          0.015151515151515152]
# dense output: b_i(th) = R[i][0] th + R[i][1] th^2 + R[i][2] th^3 + R[i][3] th^4
R = [
    [1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216],
    [0.0, 0.13169999999999998, -0.2234, 0.1017],
    [0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253],
    [0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902],
    [0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928],
    [0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661],
    [0.0, 1.5, -4.0, 2.5],
]


def interp_weights(theta):
    """b_i(theta), i=1..7.  theta == 1 returns the step weights (a7j, 0) exactly."""
    if abs(theta - 1.0) < 1e-12:
        return list(A[6]) + [0.0]
    return [((R[i][3] * theta + R[i][2]) * theta + R[i][1]) * theta * theta + R[i][0] * theta
            for i in range(7)]


def locate_observations(timepoints, n_steps):
    """For each observation time: (step index n, theta) with t_n < tau <= t_{n+1}
    (tau = t0 maps to step 0, theta 0).  t_n = t0 + n*h."""
    t0, t1 = float(timepoints[0]), float(timepoints[-1])
    h = (t1 - t0) / n_steps
    out = []
    for tau in timepoints:
        x = (float(tau) - t0) / h
        n = int(math.ceil(x - 1e-9)) - 1
        n = min(n_steps - 1, max(0, n))
        theta = (float(tau) - (t0 + n * h)) / h
        out.append((n, theta))
    return out


# ----------------------------------------------------------------------------- MLP
SYMBOLIC = (1, 0, 0)     # "architecture" of the analytic production model: one shared parameter p0


def general_arch(arch):
    """`chain(widths, activation_functions; input_dims, output_activation)` in its general form (src/neural-network.jl:
    42-58): arch = (nin, (w1, w2, ...), (act1, act2, ...), output_act) -- widths and activation names per hidden layer."""
    return len(arch) >= 2 and isinstance(arch[1], (list, tuple))


def n_params(arch):
    if general_arch(arch):
        p, fan = 0, arch[0]
        for w in list(arch[1]) + [1]:
            p += w * fan + w
            fan = w
        return p
    nin, width, depth = arch[:3]
    if width == 0:
        return 1
    p, fan_in = 0, nin
    for _ in range(depth):
        p += width * fan_in + width
        fan_in = width
    return p + fan_in + 1


def softplus(xp, x):
    return xp.log(1.0 + xp.exp(x))


def _relu(xp, z):
    if xp is np:
        return np.maximum(z, 0.0)
    if xp is math:
        return z if z > 0.0 else 0.0
    # torch: d relu / dz = 0 AT the kink, as ForwardDiff differentiates Julia's max(zero(x), x) (a tie returns the
    # constant); torch.clamp would pass the gradient through at z == 0, which a dead layer with zero biases hits exactly
    return z * (z > 0.0).to(z.dtype)


# `chain(widths, activations; output_activation)` (src/neural-network.jl:42-58) takes any activation functions; the
# reference's scripts pass tanh and softplus.  An `arch` tuple may name others behind its three sizes:
# (nin, width, depth, hidden, output) with hidden in HIDDEN_ACTS and output in OUTPUT_ACTS.
HIDDEN_ACTS = {"tanh": lambda xp, z: xp.tanh(z), "relu": _relu,
               "sigmoid": lambda xp, z: 1.0 / (1.0 + xp.exp(-z)), "identity": lambda xp, z: z, "softplus": softplus}
OUTPUT_ACTS = {"softplus": softplus, "identity": lambda xp, z: z}


def mlp(xp, inputs, p, arch):
    """inputs: list of nin arrays (N,) or scalars; p: (P,) parameter vector.
    Returns the scalar network output per subject (N,)."""
    if general_arch(arch):
        nin, widths = arch[0], list(arch[1])
        names = arch[2] if len(arch) > 2 else "tanh"
        acts = [HIDDEN_ACTS[names]] * len(widths) if isinstance(names, str) else [HIDDEN_ACTS[a] for a in names]
        out = HIDDEN_ACTS[arch[3] if len(arch) > 3 else "softplus"]
    else:
        nin, width, depth = arch[:3]
        widths = [width] * depth
        acts = [HIDDEN_ACTS[arch[3] if len(arch) > 3 else "tanh"]] * depth
        out = OUTPUT_ACTS[arch[4] if len(arch) > 4 else "softplus"]
    h = list(inputs)
    off, fan_in = 0, nin
    for width, act in zip(widths, acts):
        nxt = []
        for j in range(width):
            z = p[off + fan_in * width + j]                      # bias
            for i in range(fan_in):
                z = z + p[off + j + width * i] * h[i]            # W[j,i], column-major
            nxt.append(act(xp, z))
        off += fan_in * width + width
        h, fan_in = nxt, width
    z = p[off + fan_in]
    for i in range(fan_in):
        z = z + p[off + i] * h[i]
    return out(xp, z)


def symbolic_production(xp, dG, p0, k):
    """production(dG, k) = dG >= 0 ? 1.78dG/(dG + k) : 0.0  (c-peptide/03-symreg.jl:37-40, called through
    analytic_production src/c-peptide-models.jl:68-75; src/saem-symreg.jl:23-29 writes the same term with the
    indicator glucose(t) > glucose(0)).  p0 generalises the literal 1.78."""
    pos = dG >= 0
    den = xp.where(pos, dG + k, 1.0 + 0.0 * dG)     # the untaken branch must not poison autograd
    return xp.where(pos, (p0 * dG) / den, 0.0 * dG)


# ----------------------------------------------------------------------------- models
def van_cauter_parameters(age, t2dm):
    age = np.asarray(age, dtype=np.float64)
    t2dm = np.asarray(t2dm, dtype=bool)
    short = np.where(t2dm, 4.52, 4.95)
    frac = np.where(t2dm, 0.78, 0.76)
    long_ = 0.14 * age + 29.2
    k1 = frac * (math.log(2) / long_) + (1 - frac) * (math.log(2) / short)
    k0 = (math.log(2) / short) * (math.log(2) / long_) / k1
    k2 = (math.log(2) / short) + (math.log(2) / long_) - k0 - k1
    return k0, k1, k2


def linear_interp(knots_t, knots_u, t):
    """DataInterpolations.LinearInterpolation restated; knots_u: list of T arrays (N,)."""
    T = len(knots_t)
    j = int(np.searchsorted(np.asarray(knots_t), t, side="right")) - 1
    j = min(max(j, 0), T - 2)
    slope = (knots_u[j + 1] - knots_u[j]) / (knots_t[j + 1] - knots_t[j])
    return knots_u[j] + (t - knots_t[j]) * slope


class CPepPopulation:
    """SoA population of c-peptide cUDE subjects (numpy float64 arrays of length N)."""

    def __init__(self, timepoints, glucose, cpeptide, age, t2dm, covariate=False):
        self.timepoints = [float(t) for t in timepoints]
        self.glucose = np.ascontiguousarray(glucose, dtype=np.float64)    # N x T
        self.cpeptide = np.ascontiguousarray(cpeptide, dtype=np.float64)  # N x T
        self.age = np.asarray(age, dtype=np.float64)
        self.t2dm = np.asarray(t2dm, dtype=bool)
        self.k0, self.k1, self.k2 = van_cauter_parameters(self.age, self.t2dm)
        self.c0 = self.cpeptide[:, 0].copy()
        self.covariate = covariate
        self.N, self.T = self.glucose.shape


def _as(xp, a):
    if xp is np:
        return a
    return xp.as_tensor(a)


def cpep_rhs(xp, pop, nn, eb, arch, t, u, n_state):
    """combined! = kinetics + conditional production, as written (baseline term re-evaluated)."""
    G = [_as(xp, pop.glucose[:, j]) for j in range(pop.T)]
    dG = linear_interp(pop.timepoints, G, t) - linear_interp(pop.timepoints, G, pop.timepoints[0])
    if arch[1] == 0:                       # CPeptideODEModel: no baseline term (c-peptide-models.jl:68-75)
        prod = symbolic_production(xp, dG, nn[0], eb)
    elif arch[0] == 1:                     # CPeptideUDEModel: neural_network_production (c-peptide-models.jl:76-84)
        prod = mlp(xp, [dG], nn, arch) - mlp(xp, [dG * 0.0], nn, arch)
    elif pop.covariate:
        age = _as(xp, pop.age)
        prod = mlp(xp, [dG, eb, age], nn, arch) - mlp(xp, [dG * 0.0, eb, age], nn, arch)
    else:
        prod = mlp(xp, [dG, eb], nn, arch) - mlp(xp, [dG * 0.0, eb], nn, arch)
    k0, k1, k2, c0 = (_as(xp, pop.k0), _as(xp, pop.k1), _as(xp, pop.k2), _as(xp, pop.c0))
    du1 = -(k0 + k2) * u[0] + k1 * u[1] + k0 * c0 + prod
    du2 = -k1 * u[1] + k2 * u[0]
    if n_state == 3:                       # CPEP3: cumulative-secretion quadrature state
        return [du1, du2, prod]
    return [du1, du2]


def cpep_rhs_scalar(pop, i, nn, cond, arch):
    """cpep_rhs for subject i of `pop` in plain Python floats -- the same `mlp` / `linear_interp` code with `math` as
    the array module -- as a closure rhs(t, u) for solve_adaptive (an order of magnitude faster than 1-element
    arrays).  cond = exp(beta) resp. k.  tests/test_oracle.py checks it against cpep_rhs."""
    tpl = [float(v) for v in pop.timepoints]
    G = [float(v) for v in pop.glucose[i]]
    k0, k1, k2, c0, age = (float(pop.k0[i]), float(pop.k1[i]), float(pop.k2[i]), float(pop.c0[i]), float(pop.age[i]))
    p, cond = [float(v) for v in nn], float(cond)

    def rhs(t, u):
        dG = linear_interp(tpl, G, t) - G[0]
        if arch[1] == 0:
            prod = (p[0] * dG) / (dG + cond) if dG >= 0 else 0.0
        elif arch[0] == 1:
            prod = mlp(math, [dG], p, arch) - mlp(math, [0.0], p, arch)
        elif pop.covariate:
            prod = mlp(math, [dG, cond, age], p, arch) - mlp(math, [0.0, cond, age], p, arch)
        else:
            prod = mlp(math, [dG, cond], p, arch) - mlp(math, [0.0, cond], p, arch)
        return [-(k0 + k2) * u[0] + k1 * u[1] + k0 * c0 + prod, -k1 * u[1] + k2 * u[0]]
    return rhs


def supp_rhs(xp, nn, etheta, arch, t, u):
    uhat = mlp(xp, [u[0], u[1], u[2], etheta], nn, arch)
    return [-0.4 * u[0], 0.4 * u[0] - uhat, uhat - 0.3 * u[2]]


# ----------------------------------------------------------------------------- solver
def solve_fixed(rhs, u0, timepoints, n_steps):
    """Fixed-step Tsit5 from timepoints[0] to timepoints[-1]; returns the interpolated
    state at every observation time: list over T of list over states."""
    t0, t1 = float(timepoints[0]), float(timepoints[-1])
    h = (t1 - t0) / n_steps
    loc = locate_observations(timepoints, n_steps)
    ns = len(u0)
    y = list(u0)
    out = [None] * len(timepoints)
    k = [None] * 7
    k[0] = rhs(t0, y)
    for n in range(n_steps):
        tn = t0 + n * h
        for i in range(1, 7):
            Y = []
            for s in range(ns):
                acc = A[i][0] * k[0][s]
                for j in range(1, i):
                    acc = acc + A[i][j] * k[j][s]
                Y.append(y[s] + h * acc)
            if i < 6:
                k[i] = rhs(tn + C[i] * h, Y)
            else:
                ynew = Y
                k[6] = rhs(t0 + (n + 1) * h, ynew)
        for ti, (nn_, theta) in enumerate(loc):
            if nn_ == n:
                w = interp_weights(theta)
                ys = []
                for s in range(ns):
                    acc = w[0] * k[0][s]
                    for j in range(1, 7):
                        acc = acc + w[j] * k[j][s]
                    ys.append(y[s] + h * acc)
                out[ti] = ys
        y = ynew
        k[0] = k[6]
    return out


def solve_adaptive(rhs, u0, timepoints, abstol=1e-6, reltol=1e-3, max_steps=100000, record=None):
    """Adaptive Tsit5 with OrdinaryDiffEq-style PI control for ONE trajectory (numpy scalars/
    0-d arrays).  Used only for the soft pins against stored reference results.
    record: a list that receives (t_n, dt_n) of every ACCEPTED step (for replay_steps)."""
    t0, t1 = float(timepoints[0]), float(timepoints[-1])
    y = [float(v) for v in u0]
    ns = len(y)
    out = [None] * len(timepoints)
    out[0] = list(y)
    nxt = 1
    k = [None] * 7
    k[0] = rhs(t0, y)
    # initial step (Hairer's heuristic, as OrdinaryDiffEq)
    sk = [abstol + reltol * abs(v) for v in y]
    d0 = math.sqrt(sum((y[s] / sk[s]) ** 2 for s in range(ns)) / ns)
    d1 = math.sqrt(sum((k[0][s] / sk[s]) ** 2 for s in range(ns)) / ns)
    dt = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    y1 = [y[s] + dt * k[0][s] for s in range(ns)]
    f1 = rhs(t0 + dt, y1)
    d2 = math.sqrt(sum(((f1[s] - k[0][s]) / sk[s]) ** 2 for s in range(ns)) / ns) / dt
    dt1 = max(1e-6, dt * 1e-3) if max(d1, d2) <= 1e-15 else (0.01 / max(d1, d2)) ** (1 / 5)
    dt = min(100 * dt, dt1, t1 - t0)
    t, qold = t0, 1e-4
    beta1, beta2, gamma, qmin, qmax = 7 / 50, 2 / 25, 0.9, 0.2, 10.0
    for _ in range(max_steps):
        if t >= t1 - 1e-14 * max(1.0, abs(t1)):
            break
        dt = min(dt, t1 - t)
        for i in range(1, 7):
            Y = [y[s] + dt * sum(A[i][j] * k[j][s] for j in range(i)) for s in range(ns)]
            if i < 6:
                k[i] = rhs(t + C[i] * dt, Y)
            else:
                ynew = Y
                k[6] = rhs(t + dt, ynew)
        err = [dt * sum(BTILDE[j] * k[j][s] for j in range(7)) for s in range(ns)]
        est = math.sqrt(sum((err[s] / (abstol + reltol * max(abs(y[s]), abs(ynew[s])))) ** 2
                            for s in range(ns)) / ns)
        if not math.isfinite(est):
            return None
        if est <= 1.0:
            while nxt < len(timepoints) and timepoints[nxt] <= t + dt + 1e-12:
                theta = min(1.0, (timepoints[nxt] - t) / dt)
                w = interp_weights(theta)
                out[nxt] = [y[s] + dt * sum(w[j] * k[j][s] for j in range(7)) for s in range(ns)]
                nxt += 1
            q11 = est ** beta1 if est > 0 else 1e-12
            q = q11 / (qold ** beta2)
            q = max(1 / qmax, min(1 / qmin, q / gamma))
            if record is not None:
                record.append((t, dt))
            t, y, k[0] = t + dt, ynew, k[6]
            qold = max(est, 1e-4)
            dt = dt / q
        else:
            q11 = est ** beta1
            q = q11 / (qold ** beta2)
            q = max(1 / qmax, min(1 / qmin, q / gamma))
            dt = dt / min(1 / qmin, q11 / gamma)
    if nxt < len(timepoints):
        return None
    return out


def replay_steps(rhs, u0, timepoints, steps):
    """Tsit5 over a GIVEN sequence of accepted steps [(t_n, dt_n)] with solve_adaptive's `saveat` rule -- the map the
    reference's gradient differentiates: under ForwardDiff the solver's time variables stay Float64 (tspan and dt
    carry no partials; src/parameter-estimation.jl:59 passes only p = theta as duals), so the derivative of an adaptive
    solve is the derivative of this fixed sequence of arithmetic.  Generic in the number type (floats, complex
    perturbation batches, torch tensors): no comparison touches a state.  u0 / rhs values: per-state scalars or arrays."""
    y = list(u0)
    ns = len(y)
    out = [None] * len(timepoints)
    out[0] = list(y)
    nxt = 1
    k = [None] * 7
    k[0] = rhs(float(timepoints[0]), y)
    for (t, dt) in steps:
        for i in range(1, 7):
            Y = []
            for s in range(ns):
                acc = A[i][0] * k[0][s]
                for j in range(1, i):
                    acc = acc + A[i][j] * k[j][s]
                Y.append(y[s] + dt * acc)
            if i < 6:
                k[i] = rhs(t + C[i] * dt, Y)
            else:
                ynew = Y
                k[6] = rhs(t + dt, ynew)
        while nxt < len(timepoints) and timepoints[nxt] <= t + dt + 1e-12:
            w = interp_weights(min(1.0, (timepoints[nxt] - t) / dt))
            o = []
            for s in range(ns):
                acc = w[0] * k[0][s]
                for j in range(1, 7):
                    acc = acc + w[j] * k[j][s]
                o.append(y[s] + dt * acc)
            out[nxt] = o
            nxt += 1
        y, k[0] = ynew, k[6]
    if nxt < len(timepoints):
        return None
    return out


# ----------------------------------------------------------------------------- losses
def cpep_forward(xp, nn, beta, pop, arch, n_steps, n_state=2, cond_space="log"):
    """Returns (trajectory [T][state] of (N,) arrays).  cond_space = "raw" (symbolic model only): the
    conditional parameter is k itself (03-symreg.jl:99-106) instead of its logarithm."""
    nn = _as(xp, nn)
    beta = _as(xp, beta)
    eb = xp.exp(beta) if cond_space == "log" else beta
    c0, k1, k2 = _as(xp, pop.c0), _as(xp, pop.k1), _as(xp, pop.k2)
    u0 = [c0, (k2 / k1) * c0]
    if n_state == 3:
        u0.append(c0 * 0.0)
    rhs = lambda t, u: cpep_rhs(xp, pop, nn, eb, arch, t, u, n_state)
    return solve_fixed(rhs, u0, pop.timepoints, n_steps)


def cpep_loss(xp, nn, beta, pop, arch, n_steps, n_state=2, cond_space="log"):
    """Population loss (src/parameter-estimation.jl:126-140): mean over subjects of the SSE on
    state 1.  Returns (loss, per-subject SSE).  Non-finite -> +Inf as the reference."""
    traj = cpep_forward(xp, nn, beta, pop, arch, n_steps, n_state, cond_space)
    sse = 0.0
    for ti in range(pop.T):
        r = traj[ti][0] - _as(xp, pop.cpeptide[:, ti])
        sse = sse + r * r
    loss = sse.sum() / pop.N
    return loss, sse


def nll_sigma(sse, n_obs, sigma):
    """loss_sigma (src/parameter-estimation.jl:70-75)."""
    return (n_obs / 2) * np.log(sigma ** 2) + sse / (2 * sigma ** 2)


def individual_log_likelihood(sse, n_obs, sigma):
    """src/saem.jl:55-66."""
    return -(n_obs / 2) * np.log(sigma ** 2) - sse / (2 * sigma ** 2)


def supp_scale(data):
    """scale = mean(maximum(data, dims=2), dims=3)  (suppression_model.jl:126). data: 3 x T x N."""
    return data.max(axis=1).mean(axis=1)


def supp_forward(xp, nn, theta, data, timepoints, arch, n_steps):
    nn = _as(xp, nn)
    theta = _as(xp, theta)
    et = xp.exp(theta)
    u0 = [_as(xp, np.ascontiguousarray(data[s, 0, :])) for s in range(3)]
    rhs = lambda t, u: supp_rhs(xp, nn, et, arch, t, u)
    return solve_fixed(rhs, u0, [float(t) for t in timepoints], n_steps)


def supp_loss(xp, nn, theta, data, timepoints, arch, n_steps, lam):
    """suppression_loss (suppression_model.jl:117-130)."""
    traj = supp_forward(xp, nn, theta, data, timepoints, arch, n_steps)
    scale = supp_scale(data)
    N = data.shape[2]
    sse = 0.0
    for ti in range(len(timepoints)):
        for s in range(3):
            r = (traj[ti][s] - _as(xp, np.ascontiguousarray(data[s, ti, :]))) / scale[s]
            sse = sse + r * r
    nnv = _as(xp, nn)
    loss = sse.sum() / N + lam * (nnv * nnv).sum()
    return loss, sse


# ----------------------------------------------------------------------------- gradients
def cpep_loss_grad_torch(nn, beta, pop, arch, n_steps, n_state=2, cond_space="log"):
    """Reverse-mode (torch float64 autograd) gradient of the same discretisation."""
    import torch
    nn_t = torch.tensor(np.asarray(nn, dtype=np.float64), requires_grad=True)
    be_t = torch.tensor(np.asarray(beta, dtype=np.float64), requires_grad=True)
    loss, sse = cpep_loss(torch, nn_t, be_t, pop, arch, n_steps, n_state, cond_space)
    loss.backward()
    g_beta = be_t.grad.numpy().copy() if be_t.grad is not None else np.zeros(be_t.shape)   # (1-input network: no beta)
    return (float(loss.detach()), nn_t.grad.numpy().copy(), g_beta, sse.detach().numpy().copy())


def supp_loss_grad_torch(nn, theta, data, timepoints, arch, n_steps, lam):
    import torch
    nn_t = torch.tensor(np.asarray(nn, dtype=np.float64), requires_grad=True)
    th_t = torch.tensor(np.asarray(theta, dtype=np.float64), requires_grad=True)
    loss, sse = supp_loss(torch, nn_t, th_t, data, timepoints, arch, n_steps, lam)
    loss.backward()
    return (float(loss.detach()), nn_t.grad.numpy().copy(), th_t.grad.numpy().copy(),
            sse.detach().numpy().copy())


def _complex_step_batch(nn, cond_log, cond_space="log"):
    """Parameter batch for complex-step differentiation: column b < P perturbs nn[b], column P the conditional
    parameter, by i*1e-30 (the derivative is Im f / 1e-30 to machine precision: no subtraction)."""
    P = len(nn)
    p = np.asarray(nn, dtype=np.complex128)[:, None] + 1e-30j * np.eye(P, P + 1)
    c = np.full(P + 1, complex(cond_log))
    c[P] += 1e-30j
    return p, (np.exp(c) if cond_space == "log" else c)


def cpep_replay_loss_grad(nn, beta, pop, arch, steps, cond_space="log"):
    """Loss and gradient of the c-peptide population loss over GIVEN accepted-step sequences, steps[i] = [(t_n, dt_n)]
    (replay_steps), by the complex-step method over a batch of P + 1 perturbations.  Returns (loss, g_nn, g_beta, sse)."""
    nn = np.asarray(nn, dtype=np.float64)
    P, N = len(nn), pop.N
    tpl = [float(v) for v in pop.timepoints]
    g_nn, g_b, sse = np.zeros(P), np.zeros(N), np.zeros(N)
    for i in range(N):
        c0 = float(pop.c0[i])
        pb, cb = _complex_step_batch(nn, beta[i], cond_space)
        G = [float(v) for v in pop.glucose[i]]
        k0, k1, k2, age = float(pop.k0[i]), float(pop.k1[i]), float(pop.k2[i]), float(pop.age[i])

        def rhs(t, u):
            dG = linear_interp(tpl, G, t) - G[0]
            if arch[1] == 0:
                prod = (pb[0] * dG) / (dG + cb) if dG >= 0 else 0.0 * cb
            elif pop.covariate:
                prod = mlp(np, [dG, cb, age], pb, arch) - mlp(np, [0.0, cb, age], pb, arch)
            else:
                prod = mlp(np, [dG, cb], pb, arch) - mlp(np, [0.0, cb], pb, arch)
            return [-(k0 + k2) * u[0] + k1 * u[1] + k0 * c0 + prod, -k1 * u[1] + k2 * u[0]]
        out = replay_steps(rhs, [c0 + 0.0 * cb, (k2 / k1) * c0 + 0.0 * cb], tpl, steps[i])
        if out is None:
            return float("inf"), np.full(P, np.nan), np.full(N, np.nan), sse
        e = sum((out[ti][0] - pop.cpeptide[i, ti]) ** 2 for ti in range(1, pop.T)) + (c0 - pop.cpeptide[i, 0]) ** 2
        sse[i] = e[0].real
        g_nn += e.imag[:P] / 1e-30
        g_b[i] = e.imag[P] / 1e-30
    return sse.sum() / N, g_nn / N, g_b / N, sse


def cpep_adaptive_loss_grad(nn, beta, pop, arch, abstol=1e-6, reltol=1e-3, cond_space="log"):
    """Loss and gradient of the ADAPTIVE solve as the reference's AutoForwardDiff sees it
    (src/parameter-estimation.jl:56-68,126-140,165): per subject, the accepted steps of the plain adaptive solve are
    recorded and the fixed sequence is differentiated (cpep_replay_loss_grad).  Returns (loss, g_nn, g_beta, sse);
    loss = +Inf and NaN gradients when a solve fails.  Checker for the device's adjoint of the adaptive solve (a
    different algorithm: reverse-mode, hand-written)."""
    nn = np.asarray(nn, dtype=np.float64)
    steps = []
    for i in range(pop.N):
        cond = float(np.exp(beta[i])) if cond_space == "log" else float(beta[i])
        c0 = float(pop.c0[i])
        rec = []
        if solve_adaptive(cpep_rhs_scalar(pop, i, nn, cond, arch), [c0, float(pop.k2[i] / pop.k1[i]) * c0],
                          [float(v) for v in pop.timepoints], abstol, reltol, record=rec) is None:
            return float("inf"), np.full(len(nn), np.nan), np.full(pop.N, np.nan), np.zeros(pop.N)
        steps.append(rec)
    return cpep_replay_loss_grad(nn, beta, pop, arch, steps, cond_space)


def supp_replay_loss_grad(nn, theta, data, timepoints, arch, lam, steps):
    """The same for suppression_loss over given step sequences.  Returns (loss, g_nn, g_theta, sse)."""
    nn = np.asarray(nn, dtype=np.float64)
    P, N = len(nn), data.shape[2]
    tpl = [float(v) for v in timepoints]
    scale = supp_scale(data)
    g_nn, g_t, sse = np.zeros(P), np.zeros(N), np.zeros(N)
    for i in range(N):
        pb, cb = _complex_step_batch(nn, theta[i])
        out = replay_steps(lambda t, u: supp_rhs(np, pb, cb, arch, t, u), [float(data[s, 0, i]) + 0.0 * cb for s in range(3)],
                           tpl, steps[i])
        if out is None:
            return float("inf"), np.full(P, np.nan), np.full(N, np.nan), sse
        e = 0.0 * cb
        for ti in range(1, len(tpl)):
            for s in range(3):
                e = e + ((out[ti][s] - data[s, ti, i]) / scale[s]) ** 2
        sse[i] = e[0].real
        g_nn += e.imag[:P] / 1e-30
        g_t[i] = e.imag[P] / 1e-30
    return sse.sum() / N + lam * float(nn @ nn), g_nn / N + 2.0 * lam * nn, g_t / N, sse


def supp_adaptive_loss_grad(nn, theta, data, timepoints, arch, lam, abstol=1e-6, reltol=1e-3):
    """suppression_loss with the reference's AutoForwardDiff gradient (suppression/src/suppression_model.jl:117-130,
    :155): adaptive solve, accepted steps recorded, sequence differentiated.  Returns (loss, g_nn, g_theta, sse)."""
    nn = np.asarray(nn, dtype=np.float64)
    steps = []
    for i in range(data.shape[2]):
        et = math.exp(float(theta[i]))
        rec = []
        if solve_adaptive(lambda t, u: supp_rhs(math, [float(v) for v in nn], et, arch, t, u),
                          [float(data[s, 0, i]) for s in range(3)], [float(v) for v in timepoints], abstol, reltol,
                          record=rec) is None:
            return float("inf"), np.full(len(nn), np.nan), np.full(data.shape[2], np.nan), np.zeros(data.shape[2])
        steps.append(rec)
    return supp_replay_loss_grad(nn, theta, data, timepoints, arch, lam, steps)


# ----------------------------------------------------------------------------- Adam / MH
def adam_update(theta, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """Optimisers.jl Adam restated: returns (theta, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** t)
    vhat = v / (1 - b2 ** t)
    return theta - lr * mhat / (np.sqrt(vhat) + eps), m, v


def log_normal_pdf(x, mu, sd):
    return -0.5 * ((x - mu) / sd) ** 2 - np.log(sd) - 0.5 * math.log(2 * math.pi)


def mh_chain(nn, beta0, pop, arch, n_steps, sigma, prior_eta, omega, proposal_std,
             temperature, gamma, normals, uniforms, samples=None):
    """E-step of SAEM for every subject (src/saem.jl:86-108 and :177-186) with host-supplied
    draws: normals/uniforms are (n_mcmc, N).  Returns (beta, n_accepted per subject).
    The 'current' log-likelihood is recomputed every step as the reference does."""
    beta = np.array(beta0, dtype=np.float64)
    acc = np.zeros(pop.N, dtype=np.int64)
    T = pop.T
    for s in range(normals.shape[0]):
        prop = beta + normals[s] * proposal_std
        prior_ratio = log_normal_pdf(prop, prior_eta, omega) - log_normal_pdf(beta, prior_eta, omega)
        _, sse_new = cpep_loss(np, nn, prop, pop, arch, n_steps)
        _, sse_cur = cpep_loss(np, nn, beta, pop, arch, n_steps)
        ll_new = individual_log_likelihood(sse_new, T, sigma)
        ll_cur = individual_log_likelihood(sse_cur, T, sigma)
        ll_new = np.where(np.isfinite(ll_new), ll_new, -np.inf)
        ratio = ll_new / temperature - ll_cur / temperature
        accept = np.log(uniforms[s]) < (prior_ratio + ratio)
        acc += accept
        newb = np.where(accept, prop, beta)
        beta = (1 - gamma) * beta + gamma * newb
        if samples is not None:            # individual_samples of c-peptide/06-saem.jl:107-112
            samples.append(beta.copy())
    return beta, acc


# ----------------------------------------------------------------------------- synthetic data
def generate_suppression_data(group_means, group_sizes, timepoints, noise_multiplicative=0.1, seed=232705, n_steps=240):
    """`generate_data` of suppression/src/suppression_model.jl:33-63 restated (BASELINE configs[0] / SURVEY.md 8(d)):
    per group, parameters max(mu + sd * randn, 0.05) with mu = [0.4, 0.9, 0.3, mu_sup], sd = [0.1, 0.1, 0.1, mu_sup / 8]
    (get_group_parameters :33-37); every subject's data = solution of the ground-truth model lsup! (:16-20) from
    u0 = (10, 0, 0) at `timepoints`, times (1 + noise * randn), clamped at 0.  Returns (data 3 x T x N, the subjects'
    fourth parameter).  The random stream is numpy's, not StableRNG's, and the solve is fixed-step Tsit5 (vectorised over
    subjects) -- this generates inputs of the reference's distribution, it reproduces no stored number (the adaptive
    restatement does that: tests/test_known_answers.py)."""
    rng = np.random.default_rng(seed)
    tp = np.asarray(timepoints, dtype=np.float64)
    cols, sup = [], []
    for mu_sup, size in zip(group_means, group_sizes):
        mu = np.array([0.4, 0.9, 0.3, mu_sup])[:, None]
        sd = np.array([0.1, 0.1, 0.1, mu_sup / 8.0])[:, None]
        p = np.maximum(mu + sd * rng.standard_normal((4, size)), 0.05)

        def rhs(t, u, p=p):
            a = p[1] * u[1] / (1.0 + p[3] * u[2])
            return np.stack([-p[0] * u[0], p[0] * u[0] - a, a - p[2] * u[2]])
        u0 = np.stack([np.full(size, 10.0), np.zeros(size), np.zeros(size)])
        sol = np.asarray(solve_fixed(rhs, u0, tp, n_steps))                 # (T, 3, size) or (3, T, size)
        if sol.shape[0] != 3:
            sol = np.moveaxis(sol, 0, 1)
        sol = sol + noise_multiplicative * sol * rng.standard_normal(sol.shape)
        cols.append(np.maximum(sol, 0.0))
        sup.append(p[3])
    return np.concatenate(cols, axis=2), np.concatenate(sup)


def synthetic_cpep_population(N, seed=20250905):
    """Seeded synthetic population of the c-peptide shape; observations are
    filled with a smooth placeholder and should be replaced by a forward solve + noise."""
    rng = np.random.default_rng(seed)
    age = rng.uniform(20, 79, N)
    t2dm = rng.random(N) < 0.44
    tp = [0.0, 30.0, 60.0, 90.0, 120.0]
    muG = np.array([5.22, 9.10, 10.44, 10.62, 10.35])
    sdG = np.array([0.88, 1.99, 3.45, 4.58, 4.93])
    z = rng.standard_normal(N)
    G = np.maximum(3.2, muG[None, :] + sdG[None, :] * z[:, None])
    beta_true = rng.normal(-0.63, 0.9, N)
    c0 = np.maximum(0.2, rng.normal(0.62, 0.29, N))
    cpep = np.repeat(c0[:, None], 5, axis=1)
    return tp, G, cpep, age, t2dm, beta_true, rng


def glorot_params(arch, seed):
    rng = np.random.default_rng(seed)
    if general_arch(arch):
        parts, fan_in = [], arch[0]
        for width in list(arch[1]) + [1]:
            parts.append(rng.standard_normal(width * fan_in) * math.sqrt(2.0 / (fan_in + width)))
            parts.append(0.1 * rng.standard_normal(width))       # (non-zero biases: every entry of the gradient is exercised)
            fan_in = width
        return np.concatenate(parts)
    nin, width, depth = arch[:3]
    parts, fan_in = [], nin
    for _ in range(depth):
        parts.append(rng.standard_normal(width * fan_in) * math.sqrt(2.0 / (fan_in + width)))
        parts.append(np.zeros(width))
        fan_in = width
    parts.append(rng.standard_normal(fan_in) * math.sqrt(2.0 / (fan_in + 1)))
    parts.append(np.zeros(1))
    return np.concatenate(parts)
