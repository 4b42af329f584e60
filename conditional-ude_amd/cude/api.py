"""Host-side mirror of the reference's Julia API for the accelerated path (same names, argument
meaning and error behaviour), implemented on top of the C ABI (libcude_hip.so).  No numerics live
here except optimiser bookkeeping on small vectors; every loss, gradient and trajectory comes from
the HIP kernels, and a missing library / GPU raises CudeError (there is no CPU fallback).

Reference items mirrored (paths relative to the reference repo):
  chain / softplus                         src/neural-network.jl:13-15,42-58,85-87,105-107
  neural_network_model                     suppression/src/suppression_model.jl:78-85
  CPeptideConditionalUDEModel (+Covariate) src/c-peptide-models.jl:170-220, src/types.jl:16-19
  CPeptideUDEModel, its loss and train     src/c-peptide-models.jl:76-84,144-168; src/parameter-estimation.jl:56-68,144-159,205-247
  loss / loss_sigma (3 methods each)       src/parameter-estimation.jl:56-75,93-109,126-140
  initial_parameters                       src/parameter-estimation.jl:22-24,36-38
  train (population / fixed-NN), train_with_sigma, evaluate_model
                                           src/parameter-estimation.jl:272-307,340-433
  suppression_loss / simul / fit_suppression_model
                                           suppression/src/suppression_model.jl:107-177
  individual_log_likelihood / mcmc_step / total_nll / update_population_parameters / SAEM
                                           src/saem.jl:55-66,86-131,134-237
  likelihood_profile                       src/likelihood-profiles.jl:4-17
  CPeptideODEModel / production (symbolic) src/c-peptide-models.jl:68-75,118-142; c-peptide/03-symreg.jl:37-40
  train_symbolic (per-subject k, sigma)    c-peptide/03-symreg.jl:94-106
  SAEM_symbolic                            src/saem-symreg.jl:31-66,86-131,134-229
Discretisation: every function here solves, like the reference, with ADAPTIVE Tsit5 at OrdinaryDiffEq's default
tolerances unless told otherwise (`default_steps`; the gradient is that of the accepted step sequence taken as fixed
arithmetic, which is what ForwardDiff through `solve` yields up to the controller's weighting of the partials,
INTEGRATION.md).  The fixed-step mode -- a smooth loss with an exact discrete adjoint and the time-split kernels for small
populations, 1e-4 ... 1e-3 away from the reference's values -- is the explicit fast option: `n_steps=
fixed_steps(timepoints)` (c-peptide: 8 steps per observation interval; suppression: 30) per call, or
`set_default_steps("fixed")` for the module.  julia/CUDEHip.jl has the same default and the same functions.
Deliberate differences, documented in DESIGN.md: gradients are a discrete adjoint instead of ForwardDiff; the
per-subject 1-D fits use a bracketing search instead of
Fminbox(LBFGS).
"""
import hashlib
import math
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np

from .engine import Engine
from .lbfgs import lbfgs, lbfgs_batched

DEFAULT_STEPS = 30
ADAPTIVE = 0      # n_steps = ADAPTIVE: the reference's own adaptive Tsit5 (abstol 1e-6, reltol 1e-3)
_DEFAULT_MODE = [ADAPTIVE]      # what default_steps returns: ADAPTIVE, "fixed" or a step count (set_default_steps)


def fixed_steps(timepoints, per_interval=8):
    """The fixed-step grid of the fast mode (`n_steps=fixed_steps(timepoints)`): `per_interval` Tsit5 steps per
    observation interval when the observation times are equidistant, so that every step boundary coincides with a knot
    of the piecewise-linear glucose forcing (on the Ohashi data S=32 is ~8x more accurate than S=30, whose steps
    straddle the knots at 30 and 90 min: max rel. trajectory error 6e-6 vs 5e-5); otherwise DEFAULT_STEPS.
    julia/CUDEHip.jl `fixed_steps` is the same rule."""
    tp = np.asarray(timepoints, dtype=np.float64)
    d = np.diff(tp)
    if tp.size >= 2 and np.allclose(d, d[0], rtol=1e-12, atol=0.0):
        return per_interval * (tp.size - 1)
    return DEFAULT_STEPS


def default_steps(timepoints=None, per_interval=8):
    """Discretisation of every function of this module whose caller passes no `n_steps`: ADAPTIVE -- the reference's
    `solve(prob, Tsit5())` at OrdinaryDiffEq's default tolerances (src/parameter-estimation.jl:59,
    suppression_model.jl:113,123, saem.jl:52), the mode that reproduces the reference's stored numbers (2e-9 on its 75
    stored suppression objectives).  The fixed-step mode (a smooth loss, exact discrete adjoint, time-split kernels for
    small populations; 1e-4 ... 1e-3 away from the reference's values) is asked for explicitly: `n_steps=
    fixed_steps(timepoints)` per call or `set_default_steps("fixed")` for the module.  julia/CUDEHip.jl
    `default_steps` / `set_default_steps!` are the same functions (tests/test_julia_shim.py compares the mirrors)."""
    mode = _DEFAULT_MODE[0]
    if mode == "fixed":
        if timepoints is None:
            return DEFAULT_STEPS
        return fixed_steps(timepoints, per_interval)
    return int(mode)


def set_default_steps(mode):
    """mode: ADAPTIVE (0, the default), "fixed" (fixed_steps(timepoints) per population; 30 for the suppression model)
    or a positive step count.  Returns the previous mode."""
    if not (mode == "fixed" or (isinstance(mode, (int, np.integer)) and mode >= 0)):
        raise ValueError('set_default_steps: ADAPTIVE (0), "fixed" or a positive step count')
    prev = _DEFAULT_MODE[0]
    _DEFAULT_MODE[0] = mode if mode == "fixed" else int(mode)
    return prev


# ----------------------------------------------------------------------------- network
def softplus(x):
    return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0.0)


class Chain:
    """SimpleChain(static(input_dims), TurboDense{true}(tanh, width) x depth, TurboDense{true}(softplus, 1)).

    `widths` (unequal hidden widths, src/neural-network.jl:42-58): the network is carried as the equal-width network
    of width max(widths) whose extra units have zero weights, frozen by the library's parameter mask
    (cude_set_param_mask): parameter vectors have THAT layout (n_params entries, `mask` marks the live ones;
    pad_network / unpad_network convert to and from SimpleChains' own layout)."""

    def __init__(self, input_dims, width, depth, widths=None, activation="tanh", output_activation="softplus",
                 layer_activations=None):
        self.input_dims, self.width, self.depth = int(input_dims), int(width), int(depth)
        self.widths = None if widths is None else [int(w) for w in widths]
        # hidden tanh | relu | sigmoid, output softplus | identity: what the tuned kernels compile (the reference's
        # scripts build tanh / softplus networks only; src/neural-network.jl:42-58 accepts any function)
        self.activation, self.output_activation = activation, output_activation
        # the general form -- one activation function PER hidden layer, any of ACTIVATIONS, any widths: evaluated by the
        # library's fallback kernel (cude_set_network); parameter vectors then have SimpleChains' own layout, unpadded
        self.layer_activations = None if layer_activations is None else [str(a) for a in layer_activations]
        if self.layer_activations is not None and self.widths is None:
            self.widths = [self.width] * self.depth

    @property
    def general(self):
        return self.layer_activations is not None

    @property
    def mask(self):
        """1 for the live entries of the padded parameter vector, 0 for the padding; None for equal widths."""
        if self.widths is None or self.general:
            return None
        n = sum(w * f + w for w, f in zip(self.widths + [1], [self.input_dims] + self.widths))
        return (pad_network(self.widths, np.ones(n), input_dims=self.input_dims)[1] != 0.0).astype(np.float64)

    @property
    def arch(self):
        if self.general:        # (Engine: (nn_in, [widths], [activation per hidden layer], output activation))
            return (self.input_dims, tuple(self.widths), tuple(self.layer_activations), self.output_activation)
        return (self.input_dims, self.width, self.depth)

    @property
    def key(self):
        if self.general:
            return self.arch
        return self.arch + (tuple(self.widths) if self.widths else ()) + (self.activation, self.output_activation)

    def configure(self, engine):
        """tell a fresh engine the activation functions (before the population is uploaded)"""
        if self.general:        # (the engine was built from `arch`: cude_set_network has been called)
            return
        if self.activation != "tanh":
            engine.set_option("hidden_activation", self.activation)
        if self.output_activation != "softplus":
            engine.set_option("output_activation", self.output_activation)

    @property
    def n_params(self):
        if self.general:
            return _layer_slices(self.input_dims, self.widths)[1]
        p, fan = 0, self.input_dims
        for _ in range(self.depth):
            p += self.width * fan + self.width
            fan = self.width
        return p + fan + 1


_HIDDEN = ("tanh", "relu", "sigmoid")
_OUTPUT = ("softplus", "identity")
ACTIVATIONS = ("tanh", "relu", "sigmoid", "softplus", "identity")       # the fallback kernel: any of these in any layer


def chain(width, depth=None, activation="tanh", *, input_dims=2, output_dims=1, output_activation="softplus"):
    """chain(width, depth, act) / chain(widths::Vector, act) / chain(widths, acts::Vector) (src/neural-network.jl:42-58,
    85-87, 105-107).  What the kernels implement is accepted: ONE activation function for all hidden layers out of tanh,
    relu, sigmoid (functions are recognised by name: `sigmoid` also as `σ`, `sigmoid_fast`), one output unit with softplus
    or identity; unequal hidden widths with tanh / softplus only (zero-padded + masked, class Chain)."""
    if isinstance(width, (list, tuple, np.ndarray)):
        widths = list(width)
        if depth is not None and not isinstance(depth, (int, np.integer)):
            activation = depth
        if len(widths) == 0:
            raise ValueError("Input widths must be non-empty.")
        if output_dims != 1:
            raise NotImplementedError("the models of the reference use ONE network output (output_dims = 1)")
        per_layer = None
        if isinstance(activation, (list, tuple)):
            if len(activation) != len(widths):
                raise ValueError("The number of widths must match the number of activation functions.")
            per_layer = [_act_name(a) for a in activation]
            if len(set(per_layer)) == 1:
                activation, per_layer = per_layer[0], None
        out = _act_name(output_activation)
        act = None if per_layer is not None else _act_name(activation)
        # what only the fallback kernel evaluates: different functions per layer, softplus / identity hidden layers, other
        # output functions, unequal widths with anything but tanh / softplus
        if (per_layer is not None or act not in _HIDDEN or out not in _OUTPUT
                or (len(set(widths)) != 1 and (act != "tanh" or out != "softplus"))):
            names = per_layer if per_layer is not None else [act] * len(widths)
            bad = [a for a in names + [out] if a not in ACTIVATIONS]
            if bad:
                raise NotImplementedError(f"activation functions of the library: {ACTIVATIONS} (got {bad})")
            return Chain(input_dims, max(widths), len(widths), widths=widths, output_activation=out, layer_activations=names)
        if len(set(widths)) != 1:
            return Chain(input_dims, max(widths), len(widths), widths=widths)     # zero-padded + masked (class Chain)
        width, depth = widths[0], len(widths)
    act, out = _act_name(activation), _act_name(output_activation)
    if output_dims != 1:
        raise NotImplementedError("the models of the reference use ONE network output (output_dims = 1)")
    if act not in _HIDDEN or out not in _OUTPUT:
        if act not in ACTIVATIONS or out not in ACTIVATIONS:
            raise NotImplementedError(f"activation functions of the library: {ACTIVATIONS} (got {act!r}, {out!r})")
        return Chain(input_dims, width, depth, output_activation=out, layer_activations=[act] * int(depth))
    return Chain(input_dims, width, depth, activation=act, output_activation=out)


def _layer_slices(input_dims, widths):
    """(weight slice, bias slice, out, in) of every layer of a SimpleChains parameter vector, output layer last."""
    out, at, fan = [], 0, int(input_dims)
    for w in list(widths) + [1]:
        out.append((slice(at, at + w * fan), slice(at + w * fan, at + w * fan + w), w, fan))
        at += w * fan + w
        fan = w
    return out, at


def pad_network(widths, params, *, input_dims=2):
    """Unequal hidden widths, `chain([w1, w2, ...], tanh)` (src/neural-network.jl:42-58), on kernels that are compiled
    for equal widths: the network IS the equal-width network of width W = max(widths) whose extra units have zero
    weights and biases: nothing they compute reaches the output, and the gradient entries touching them vanish --
    exactly for the weights into a padded unit, at the rounding of the device's tanh(0) (~1e-16) for the weights
    leaving one.  Use it for loss / gradient / simulation calls and optimise in the UNPADDED parameters (gradient =
    unpad_network(gradient of the padded network): exact).  The library's own optimisers (train, cude_train_restarts)
    work on the padded vector, and Adam's scale invariance amplifies those ~1e-17 gradients until the padding comes
    alive: they then train the equal-width network, not the unequal-width one (tests/test_gpu_api.py).
    Returns (Chain(input_dims, W, depth), padded parameter vector)."""
    widths = [int(w) for w in widths]
    p = np.asarray(params, dtype=np.float64).reshape(-1)
    small, n_small = _layer_slices(input_dims, widths)
    if p.size != n_small:
        raise ValueError(f"expected {n_small} parameters for widths {widths}, got {p.size}")
    W = max(widths)
    big, n_big = _layer_slices(input_dims, [W] * len(widths))
    out = np.zeros(n_big)
    for (ws, bs, o, i), (wb, bb, ob, ib) in zip(small, big):
        M = np.zeros((ob, ib))
        M[:o, :i] = p[ws].reshape(i, o).T                    # SimpleChains stores W column-major: vec(W)[j + o*k] = W[j,k]
        out[wb] = M.T.reshape(-1)
        out[bb][:o] = p[bs]
    return Chain(input_dims, W, len(widths)), out


def unpad_network(widths, padded, *, input_dims=2):
    """Inverse of pad_network for a parameter (or gradient) vector of the padded network."""
    widths = [int(w) for w in widths]
    P = np.asarray(padded, dtype=np.float64).reshape(-1)
    small, n_small = _layer_slices(input_dims, widths)
    big, n_big = _layer_slices(input_dims, [max(widths)] * len(widths))
    if P.size != n_big:
        raise ValueError(f"expected {n_big} parameters of the padded network, got {P.size}")
    out = np.empty(n_small)
    for (ws, bs, o, i), (wb, bb, ob, ib) in zip(small, big):
        out[ws] = P[wb].reshape(ib, ob).T[:o, :i].T.reshape(-1)
        out[bs] = P[bb][:o]
    return out


def _act_name(a):
    name = a if isinstance(a, str) else getattr(a, "__name__", str(a))
    return {"σ": "sigmoid", "sigmoid_fast": "sigmoid", "tanh_fast": "tanh", "identity": "identity"}.get(name, name)


def neural_network_model(depth, width, *, input_dims=2):
    return Chain(input_dims, width, depth)


def init_params(net, rng=None):
    """SimpleChains.init_params restated.  A `TurboDense{true}` layer keeps weights and bias in ONE out x (in + 1)
    matrix [W b] and initialises all of it Glorot-normal, sigma = sqrt(2 / (out + in + 1)) -- the biases are random
    too.  (SimpleChains is third-party; this reading is the one under which the reference's suppression experiment
    is reproduced in distribution -- with zero biases 4 ... 9 of the 25 kept runs stall on the constant-production
    plateau above 0.7, which none of the reference's 125 stored runs does: profiles/r02/e2e_suppression.txt.)"""
    rng = np.random.default_rng() if rng is None else rng
    parts, fan = [], net.input_dims
    for out in (net.widths if getattr(net, "widths", None) else [net.width] * net.depth) + [1]:
        sigma = math.sqrt(2.0 / (out + fan + 1))
        parts += [rng.standard_normal(out * fan) * sigma, rng.standard_normal(out) * sigma]
        fan = out
    p = np.concatenate(parts)
    if getattr(net, "general", False):
        return p
    return pad_network(net.widths, p, input_dims=net.input_dims)[1] if getattr(net, "widths", None) else p


def ComponentArray(**kw):
    """ComponentArrays.ComponentArray stand-in: attribute access to named parts (neural, conditional, ...)."""
    return SimpleNamespace(**{k: (np.array(v, dtype=np.float64) if not np.isscalar(v) else float(v)) for k, v in kw.items()})


class OptimizationSolution(SimpleNamespace):
    """Fields used downstream of `train` in the reference: .u (ComponentArray) and .objective."""


# ----------------------------------------------------------------------------- c-peptide models
class CPeptideConditionalUDEModel:
    """CPeptideConditionalUDEModel(glucose, timepoints, age, chain, cpeptide, t2dm) -- holds one subject's data;
    the ODE problem itself lives on the GPU once the subject is part of a population."""

    def __init__(self, glucose_data, glucose_timepoints, age, network, cpeptide_data, t2dm, *, covariate=False):
        self.glucose = np.asarray(glucose_data, dtype=np.float64)
        self.timepoints = np.asarray(glucose_timepoints, dtype=np.float64)
        self.age = float(age)
        self.chain = network
        self.cpeptide = np.asarray(cpeptide_data, dtype=np.float64)
        self.t2dm = bool(t2dm)
        self.covariate = covariate
        if self.glucose.shape != self.timepoints.shape or self.cpeptide.shape != self.timepoints.shape:
            raise ValueError("glucose, cpeptide and timepoints must have the same length")
        if network.input_dims != (3 if covariate else 2):
            raise ValueError("network input_dims does not match the model (2: [dG, beta]; 3: [dG, beta, age])")
        self._key = _model_key(self, ("cude", network.key, covariate))


CPeptideCUDEModel = CPeptideConditionalUDEModel        # name used in the reference's docstrings / stale script


def embed_single_input(width, params):
    """Parameter (or mask) vector of a 1-input network -> that of its 2-input carrier: the first layer [vec(W1); b1]
    with W1 of shape W x 1 becomes W x 2 with a zero second column (column-major: W zeros behind the first W entries)."""
    q = np.asarray(params, dtype=np.float64).reshape(-1)
    return np.concatenate([q[:width], np.zeros(width), q[width:]])


def extract_single_input(width, carried):
    """Inverse of embed_single_input (parameters or gradients of the carrier -> those of the 1-input network)."""
    q = np.asarray(carried, dtype=np.float64).reshape(-1)
    return np.concatenate([q[:width], q[2 * width:]])


class CPeptideUDEModel:
    """CPeptideUDEModel(glucose, timepoints, age, chain(...; input_dims = 1), cpeptide, t2dm) -- the non-conditional
    UDE (src/c-peptide-models.jl:144-168): production = network([dG]) - network([0]) (neural_network_production,
    :76-84), parameters = the network's alone.

    On the device it IS the conditional model's kernel: the 1-input network is carried as the 2-input network whose
    first-layer weights of the second input are zero and frozen by the library's parameter mask, so that
    network([dG; e^beta]) == network([dG]) exactly (a product with an exact zero adds +0.0) and the conditional
    parameter -- kept at 0 -- has a vanishing gradient.  Parameter vectors handed in and out have the 1-input
    SimpleChains layout (embed_single_input / extract_single_input convert)."""

    def __init__(self, glucose_data, glucose_timepoints, age, network, cpeptide_data, t2dm):
        self.glucose = np.asarray(glucose_data, dtype=np.float64)
        self.timepoints = np.asarray(glucose_timepoints, dtype=np.float64)
        self.age = float(age)
        self.chain = network
        self.cpeptide = np.asarray(cpeptide_data, dtype=np.float64)
        self.t2dm = bool(t2dm)
        self.covariate = False
        if self.glucose.shape != self.timepoints.shape or self.cpeptide.shape != self.timepoints.shape:
            raise ValueError("glucose, cpeptide and timepoints must have the same length")
        if network.input_dims != 1:
            raise ValueError("network input_dims does not match the model (1: [dG])")
        W = network.width
        base = network.mask if network.mask is not None else np.ones(network.n_params)
        mask = embed_single_input(W, base)                     # (zero where the second input's weights sit)
        self._carrier = SimpleNamespace(arch=(2, W, network.depth), mask=mask, key=("ude",) + tuple(network.key),
                                        configure=network.configure)
        self._key = _model_key(self, ("ude", network.key))

    def embed(self, params):
        return embed_single_input(self.chain.width, params)

    def extract(self, carried):
        return extract_single_input(self.chain.width, carried)


def CPeptideConditionalCovariateUDEModel(glucose_data, glucose_timepoints, age, network, cpeptide_data, t2dm):
    return CPeptideConditionalUDEModel(glucose_data, glucose_timepoints, age, network, cpeptide_data, t2dm,
                                       covariate=True)


class MichaelisMentenProduction:
    """production(dG, k) = dG >= 0 ? 1.78dG/(dG + k[1]) : 0.0 -- the analytic production term found by symbolic
    regression (c-peptide/03-symreg.jl:37-40).  The reference passes an arbitrary Julia function to
    CPeptideODEModel; this is the one form compiled into the HIP kernel (vmax generalises the literal 1.78)."""

    def __init__(self, vmax=1.78):
        self.vmax = float(vmax)

    def __call__(self, dG, k):
        dG = np.asarray(dG, dtype=np.float64)
        k = np.asarray(k, dtype=np.float64)
        pos = dG >= 0
        return np.where(pos, self.vmax * dG / np.where(pos, dG + k, 1.0), 0.0)


production = MichaelisMentenProduction()


class CPeptideODEModel:
    """CPeptideODEModel(glucose, timepoints, age, production, cpeptide, t2dm) (src/c-peptide-models.jl:118-142):
    van Cauter kinetics + analytic production; the model parameter is the per-subject k."""

    def __init__(self, glucose_data, glucose_timepoints, age, production_function, cpeptide_data, t2dm):
        if not isinstance(production_function, MichaelisMentenProduction):
            raise NotImplementedError("only MichaelisMentenProduction (vmax*dG/(dG+k)) is compiled for the GPU")
        self.glucose = np.asarray(glucose_data, dtype=np.float64)
        self.timepoints = np.asarray(glucose_timepoints, dtype=np.float64)
        self.age = float(age)
        self.production = production_function
        self.cpeptide = np.asarray(cpeptide_data, dtype=np.float64)
        self.t2dm = bool(t2dm)
        if self.glucose.shape != self.timepoints.shape or self.cpeptide.shape != self.timepoints.shape:
            raise ValueError("glucose, cpeptide and timepoints must have the same length")
        self._key = _model_key(self, ("sym", production_function.vmax))


def _model_key(m, tag):
    """Content digest of one subject's model (what the device population is built from)."""
    h = hashlib.blake2b(repr(tag).encode(), digest_size=16)
    for a in (m.glucose, m.timepoints, m.cpeptide, [m.age, float(m.t2dm)]):
        h.update(np.ascontiguousarray(a, dtype=np.float64))
    return h.digest()


class _Pop:
    """Device-resident population built from a list of models (cached by content, see _population)."""

    def __init__(self, models, timepoints, cpeptide_data, n_steps, n_state, device, cond_space="log"):
        tp = np.asarray(timepoints, dtype=np.float64)
        for m in models:
            # the reference solves model.problem (glucose knots = the model's timepoints) and saves at `timepoints`;
            # the kernels take ONE grid for both
            if m.timepoints.shape != tp.shape or not np.array_equal(m.timepoints, tp):
                raise ValueError("timepoints must equal the timepoints the models were built with")
        if isinstance(models[0], CPeptideODEModel):
            self.engine = Engine("cpep_sym", n_steps=n_steps, n_state=n_state, device=device, cond_space=cond_space)
            self.shared = np.array([models[0].production.vmax])
        else:
            net = getattr(models[0], "_carrier", models[0].chain)      # (CPeptideUDEModel: its 2-input carrier)
            self.engine = Engine("cpep", net.arch, n_steps=n_steps, n_state=n_state, device=device)
            net.configure(self.engine)
            if net.mask is not None:
                self.engine.set_param_mask(net.mask)
        G = np.stack([m.glucose for m in models])
        cp = np.asarray(cpeptide_data, dtype=np.float64).reshape(len(models), -1)
        self.engine.set_population_cpep(tp, G, cp, [m.age for m in models], [m.t2dm for m in models])
        self.N, self.T = cp.shape


_CACHE = OrderedDict()        # content key -> population, least recently used first
_CACHE_MAX = 8                # engines kept alive; an evicted one is closed (stream + device buffers freed)
_DEVICE = 0


def set_device(device):
    global _DEVICE
    _DEVICE = int(device)


def clear_cache():
    for p in _CACHE.values():
        p.engine.close()
    _CACHE.clear()


def _cached(key, build):
    pop = _CACHE.get(key)
    if pop is None:
        pop = build()
        _CACHE[key] = pop
        while len(_CACHE) > _CACHE_MAX:
            _, old = _CACHE.popitem(last=False)
            old.engine.close()
    else:
        _CACHE.move_to_end(key)
    return pop


def _digest(*arrays):
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a, dtype=np.float64)
        h.update(str(a.shape).encode())
        h.update(a)
    return h.digest()


def _population(models, timepoints, cpeptide_data, n_steps=None, n_state=2, cond_space="log"):
    """The device population of (models, timepoints, data), built once per CONTENT: the key is a digest of every
    model's data (a fresh list of the same models -- `[model]`, `[model] * steps` -- is the same population), of the
    time grid and of all observations, so that calling `loss` in a loop neither rebuilds nor leaks engines, and
    changed data is never answered from a stale population."""
    if n_steps is None:       # (fixed mode: the analytic production has a kink at dG = 0: twice the steps of the smooth model)
        n_steps = default_steps(timepoints, 16 if isinstance(models[0], CPeptideODEModel) else 8)
    cp = np.asarray(cpeptide_data, dtype=np.float64)
    h = hashlib.blake2b(digest_size=16)
    for m in models:
        h.update(m._key)
    key = ("cpep", len(models), h.digest(), int(n_steps), n_state, cond_space, _digest(timepoints, cp))
    return _cached(key, lambda: _Pop(models, timepoints, cp, n_steps, n_state, _DEVICE, cond_space))


def _is_model(x):
    return isinstance(x, (CPeptideConditionalUDEModel, CPeptideODEModel, CPeptideUDEModel))


def _first(theta):
    """theta[1] of a Julia ComponentArray(ode=[k], sigma=s) / vector / scalar."""
    return float(np.ravel(getattr(theta, "ode", theta))[0])


def loss(theta, args, *, n_steps=None):
    """loss(theta, (models, timepoints, cpeptide_data))            population, mean SSE (:126-140)
       loss(theta, (model, timepoints, cpeptide_data))             single subject SSE (:56-68)
       loss(beta,  (model, timepoints, cpeptide_data, nn_params))  single subject, fixed network (:93-99)
    Returns +Inf when any trajectory is non-finite (the reference's solver-failure convention)."""
    if len(args) == 4:
        model, timepoints, data, nn = args
        pop = _population([model], timepoints, np.asarray(data)[None, :], n_steps)
        pop.engine.set_params(nn, np.atleast_1d(np.asarray(theta, dtype=np.float64))[:1])
        out = pop.engine.forward(want_sse=True)
        return out["sse"][0] if np.isfinite(out["loss"]) else np.inf
    models, timepoints, data = args
    if isinstance(models, CPeptideUDEModel):        # theta = the network's parameters (:56-68 with neural_network_production)
        pop = _population([models], timepoints, np.asarray(data)[None, :], n_steps)
        pop.engine.set_params(models.embed(theta), [0.0])
        return pop.engine.forward()["loss"]           # N = 1: mean SSE == SSE
    if isinstance(models, CPeptideODEModel):        # p = theta, production(dG, p) reads p[1] (03-symreg.jl:38)
        pop = _population([models], timepoints, np.asarray(data)[None, :], n_steps, cond_space="raw")
        pop.engine.set_params(pop.shared, [_first(theta)])
        return pop.engine.forward()["loss"]
    if _is_model(models):
        pop = _population([models], timepoints, np.asarray(data)[None, :], n_steps)
        pop.engine.set_params(theta.neural, np.atleast_1d(theta.conditional)[:1])
        return pop.engine.forward()["loss"]           # N = 1: mean SSE == SSE
    pop = _population(models, timepoints, data, n_steps)
    pop.engine.set_params(theta.neural, np.asarray(theta.conditional).reshape(-1)[:pop.N])
    return pop.engine.forward()["loss"]


def loss_sigma(theta, args, *, n_steps=None):
    """(n/2) log sigma^2 + SSE / (2 sigma^2)   (src/parameter-estimation.jl:70-75,101-109)."""
    n = len(args[1])
    if len(args) == 4:
        err = loss(theta.ode, args, n_steps=n_steps)
    else:
        err = loss(theta, args, n_steps=n_steps)
    return (n / 2) * math.log(theta.sigma ** 2) + err / (2 * theta.sigma ** 2)


def loss_and_gradient(theta, args, *, n_steps=None):
    """Replaces ForwardDiff.gradient(loss, theta) (AutoForwardDiff, :370) for the population loss (and, for a
    CPeptideUDEModel, :232: value and gradient with respect to the network's parameters)."""
    models, timepoints, data = args
    if isinstance(models, CPeptideUDEModel):
        pop = _population([models], timepoints, np.asarray(data)[None, :], n_steps)
        pop.engine.set_params(models.embed(theta), [0.0])
        val, g_nn, _ = pop.engine.loss_grad()
        return val, models.extract(g_nn)
    pop = _population(models, timepoints, data, n_steps)
    pop.engine.set_params(theta.neural, np.asarray(theta.conditional).reshape(-1)[:pop.N])
    val, g_nn, g_cond = pop.engine.loss_grad()
    return val, ComponentArray(neural=g_nn, conditional=g_cond.reshape(np.asarray(theta.conditional).shape))


def initial_parameters(*args, rng=None):
    """initial_parameters(chain, n_initials; rng) -> list of network parameter vectors
       initial_parameters(n_models, lhs_lb, lhs_ub, n_initials, rng) -> (n_models x n_initials) Latin hypercube."""
    if isinstance(args[0], Chain):
        net, n = args[0], args[1]
        return [init_params(net, rng) for _ in range(n)]
    n_models, lb, ub, n = args[:4]
    rng = args[4] if len(args) > 4 else (np.random.default_rng() if rng is None else rng)
    # QuasiMonteCarlo.LatinHypercubeSample: one stratified draw per interval and dimension, shuffled
    u = (rng.permuted(np.tile(np.arange(n), (n_models, 1)), axis=1) + rng.random((n_models, n))) / n
    return lb + (ub - lb) * u


class _LatinHypercube:
    """QuasiMonteCarlo.LatinHypercubeSample(n_models x n_initials) generated column block by column block: the
    stratum permutations (int32) are drawn once, the jitter per block, so the (n_models x n_initials) sample itself
    is never held (above 2^28 entries even the permutations are dropped for independent uniform draws)."""

    def __init__(self, n_models, n, rng):
        self.n_models, self.n, self.rng = n_models, n, rng
        self.perm = rng.permuted(np.tile(np.arange(n, dtype=np.int32), (n_models, 1)), axis=1) \
            if n_models * n <= 2 ** 28 else None

    def columns(self, first, count):
        """(n_models, count) block of the unit-cube sample."""
        u = self.rng.random((self.n_models, count))
        return u if self.perm is None else (self.perm[:, first:first + count] + u) / self.n


class _Columns:
    """ode_inits[:, k] of the selected candidates only (the full table no longer exists)."""

    def __init__(self, cols):
        self.cols = cols

    def __getitem__(self, key):
        return self.cols[key[1]]


# ----------------------------------------------------------------------------- training drivers
def _adam_then_lbfgs(eng, nn0, cond0, adam_iters, lbfgs_iters, lr, callback=None):
    eng.set_params(nn0, cond0)
    eng.adam_init(lr)
    if callback is None and adam_iters > 0:
        trace = eng.adam_run(adam_iters)          # all iterations queued as one replayed hipGraph
        if not np.all(np.isfinite(trace)):
            raise FloatingPointError("solver failure (non-finite loss) during Adam")
    else:
        for _ in range(adam_iters):
            last = eng.adam_step()
            if callback(None, last):
                break
            if not np.isfinite(last):
                raise FloatingPointError("solver failure (non-finite loss) during Adam")
    nn, cond = eng.get_params()
    P = nn.size

    def fg(x):
        eng.set_params(x[:P], x[P:])
        val, g_nn, g_cond = eng.loss_grad()
        return val, np.concatenate([g_nn, g_cond])

    res = lbfgs(fg, np.concatenate([nn, cond]), maxiters=lbfgs_iters, callback=callback)
    return res["x"][:P], res["x"][P:], res["f"]


def _batched_adam_then_lbfgs(eng, nn_inits, cond_inits, adam_iters, lbfgs_iters, lr, traces=None, native=True):
    """The K selected restarts trained SIDE BY SIDE instead of one after the other (the reference's loop,
    src/parameter-estimation.jl:372-383 / suppression_model.jl:140-170): every optimiser iteration evaluates the K
    current points with one cude_multistart_loss_grad launch.  Adam (Optimisers.jl update rule) is vectorised over
    the restarts; the K L-BFGS runs are the serial algorithm driven in lock step (cude.lbfgs.lbfgs_batched), so each
    restart follows the path it would follow alone.  A restart whose loss becomes non-finite during Adam is
    dropped, as the reference skips a failed optimisation.  Returns a list of (nn, cond, objective) or None."""
    if native:                                # the same two stages inside the library (cude_train_restarts)
        out = eng.train_restarts(nn_inits, cond_inits, adam_iters, lr, lbfgs_iters, want_trace=traces is not None)
        nn, cond, obj = out[:3]
        if traces is not None:
            for k in range(len(obj)):
                traces[k].extend(float(v) for v in out[3][k] if not np.isnan(v))
        return [(nn[k], cond[k], float(obj[k])) if np.isfinite(obj[k]) else None for k in range(len(obj))]
    X = np.concatenate([np.asarray(nn_inits, dtype=np.float64), np.asarray(cond_inits, dtype=np.float64)], axis=1)
    K, P = X.shape[0], eng.P
    alive = np.ones(K, bool)
    M, V = np.zeros_like(X), np.zeros_like(X)
    b1, b2, eps = 0.9, 0.999, 1e-8
    for t in range(1, adam_iters + 1):
        f, g_nn, g_cond = eng.multistart_loss_grad(X[:, :P], X[:, P:])
        alive &= np.isfinite(f)
        if traces is not None:
            for k in range(K):
                if alive[k]:
                    traces[k].append(float(f[k]))
        G = np.concatenate([g_nn, g_cond], axis=1)
        G[~alive] = 0.0
        M = b1 * M + (1 - b1) * G
        V = b2 * V + (1 - b2) * G * G
        step = lr * (M / (1 - b1 ** t)) / (np.sqrt(V / (1 - b2 ** t)) + eps)
        X[alive] -= step[alive]
    if not alive.any():
        return [None] * K
    idx = np.flatnonzero(alive)
    Xa = X[idx]

    def fg_batch(Xq):
        f, g_nn, g_cond = eng.multistart_loss_grad(Xq[:, :P], Xq[:, P:])
        return f, np.concatenate([g_nn, g_cond], axis=1)
    cbs = None if traces is None else [(lambda _x, l, k=k: traces[k].append(float(l)) and False) for k in idx]
    res = lbfgs_batched(fg_batch, Xa, maxiters=lbfgs_iters, callbacks=cbs)
    out = [None] * K
    for k, r in zip(idx, res):
        out[k] = (r["x"][:P], r["x"][P:], r["f"])
    return out


def train(models, timepoints, cpeptide_data, rng_or_nn, *, initial_guesses=None, selected_initials=None,
          lhs_lower_bound=-2.0, lhs_upper_bound=0.0, n_conditional_parameters=1, number_of_iterations_adam=1000,
          number_of_iterations_lbfgs=1000, learning_rate_adam=1e-2, initial_beta=-2.0, lbfgs_lower_bound=-4.0,
          lbfgs_upper_bound=1.0, lbfgs_iterations=1000, n_steps=None, side_by_side=True):
    """Three methods of the reference, selected by the types of the 1st and 4th argument as Julia's dispatch does:
    * a single CPeptideUDEModel + rng: the conventional UDE on one (mean) subject (:205-247): `initial_guesses`
      (default 10 000) initialisations screened by their loss, the best `selected_initials` (10) -> Adam -> L-BFGS.
    * rng (numpy Generator): population training, unknown network (:340-386): LHS + init screening of
      `initial_guesses` candidates (forward-only), best `selected_initials` -> Adam -> L-BFGS; the selected
      restarts are trained side by side (one launch per optimiser iteration for all of them) unless
      side_by_side=False.
    * array of network parameters: per-subject estimation of the conditional parameter with the network
      frozen (:272-288)."""
    if isinstance(models, CPeptideUDEModel):
        return _train_ude(models, timepoints, cpeptide_data, rng_or_nn,
                          initial_guesses=10_000 if initial_guesses is None else initial_guesses,
                          selected_initials=10 if selected_initials is None else selected_initials,
                          number_of_iterations_adam=number_of_iterations_adam,
                          number_of_iterations_lbfgs=number_of_iterations_lbfgs, learning_rate_adam=learning_rate_adam,
                          n_steps=n_steps, side_by_side=side_by_side)
    initial_guesses = 25_000 if initial_guesses is None else initial_guesses
    selected_initials = 25 if selected_initials is None else selected_initials
    if isinstance(rng_or_nn, np.random.Generator):
        rng = rng_or_nn
        pop = _population(models, timepoints, cpeptide_data, n_steps)
        eng, N = pop.engine, pop.N
        # screening (:351-372): candidates are generated chunk by chunk and never held all at once; the losses and the
        # selection `partialsortperm(losses_initial, 1:selected_initials)` stay on the device (cude_screen_candidates)
        strata = _LatinHypercube(N, initial_guesses, rng)
        net = models[0].chain

        def candidates(first, count):
            nn = np.stack([init_params(net, rng) for _ in range(count)])
            return nn, lhs_lower_bound + (lhs_upper_bound - lhs_lower_bound) * strata.columns(first, count).T
        order, _, nn_sel, cond_sel = eng.screen_candidates(initial_guesses, selected_initials, candidates)
        nn_inits = dict(zip(order, nn_sel))
        ode_inits = _Columns(dict(zip(order, cond_sel)))
        sols = []
        if side_by_side and len(order) > 1:
            fits = _batched_adam_then_lbfgs(eng, nn_sel, cond_sel,
                                            number_of_iterations_adam, number_of_iterations_lbfgs, learning_rate_adam)
            for fit in fits:
                if fit is None:
                    print("Optimization failed... Skipping")
                    continue
                nn, cond, obj = fit
                sols.append(OptimizationSolution(
                    u=ComponentArray(neural=nn, conditional=np.repeat(cond[:, None], n_conditional_parameters, 1)),
                    objective=obj))
            return sols
        for k in order:
            try:
                nn, cond, obj = _adam_then_lbfgs(eng, nn_inits[k], ode_inits[:, k], number_of_iterations_adam,
                                                 number_of_iterations_lbfgs, learning_rate_adam)
                sols.append(OptimizationSolution(
                    u=ComponentArray(neural=nn, conditional=np.repeat(cond[:, None], n_conditional_parameters, 1)),
                    objective=obj))
            except FloatingPointError:
                print("Optimization failed... Skipping")
        return sols
    nn = np.asarray(rng_or_nn, dtype=np.float64)
    beta, sse = estimate_conditional(models, timepoints, cpeptide_data, nn, initial_beta=initial_beta,
                                     lower=lbfgs_lower_bound, upper=lbfgs_upper_bound, n_steps=n_steps)
    return [OptimizationSolution(u=np.array([b]), objective=s) for b, s in zip(beta, sse)]


def _train_ude(model, timepoints, cpeptide_data, rng, *, initial_guesses, selected_initials, number_of_iterations_adam,
               number_of_iterations_lbfgs, learning_rate_adam, n_steps, side_by_side):
    """train(model::CPeptideUDEModel, timepoints, cpeptide_data, rng) (src/parameter-estimation.jl:205-247): the
    screening loop, `partialsortperm(losses_initial, 1:selected_initials)` and the two optimiser stages of
    `_optimize` (:144-159) run through the same device paths as the conditional model's (cude_screen_candidates,
    cude_train_restarts) on a population of one subject; solutions carry the 1-input parameter vector."""
    if not isinstance(rng, np.random.Generator):
        raise TypeError("train(model::CPeptideUDEModel, timepoints, cpeptide_data, rng): rng must be a numpy Generator")
    pop = _population([model], timepoints, np.asarray(cpeptide_data, dtype=np.float64)[None, :], n_steps)
    eng, net = pop.engine, model.chain

    def candidates(first, count):
        return np.stack([model.embed(init_params(net, rng)) for _ in range(count)]), np.zeros((count, 1))
    order, _, nn_sel, cond_sel = eng.screen_candidates(initial_guesses, selected_initials, candidates)
    sols = []
    if side_by_side and len(order) > 1:
        fits = _batched_adam_then_lbfgs(eng, nn_sel, cond_sel, number_of_iterations_adam, number_of_iterations_lbfgs,
                                        learning_rate_adam)
    else:
        fits = []
        for k in range(len(order)):
            try:
                fits.append(_adam_then_lbfgs(eng, nn_sel[k], cond_sel[k], number_of_iterations_adam,
                                             number_of_iterations_lbfgs, learning_rate_adam))
            except FloatingPointError:
                fits.append(None)
    for fit in fits:
        if fit is None:
            print("Optimization failed... Skipping")
            continue
        sols.append(OptimizationSolution(u=model.extract(fit[0]), objective=fit[2]))
    return sols


def estimate_conditional(models, timepoints, cpeptide_data, nn, *, initial_beta=-2.0, lower=-4.0, upper=1.0,
                         n_steps=None, n_grid=41, iters=48):
    """All N independent 1-D problems min_beta SSE_i(beta) at once (cude_fit_conditional: coarse scan + golden
    section with the search state on the device, one forward launch over the population per probe).  Box
    [lower, upper]; infinite bounds are replaced by initial_beta -/+ 6.  Returns (beta[N], SSE[N])."""
    pop = _population(models, timepoints, cpeptide_data, n_steps)
    eng = pop.engine
    lo = lower if np.isfinite(lower) else np.min(initial_beta) - 6.0
    hi = upper if np.isfinite(upper) else np.max(initial_beta) + 6.0
    eng.set_params(nn, None)
    x, _, sse = eng.fit_conditional(lo, hi, n_grid, iters)
    return x, sse


def train_with_sigma(models, timepoints, cpeptide_data, nn, *, initial_beta=-2.0, lbfgs_lower_bound=-4.0,
                     lbfgs_upper_bound=1.0, lbfgs_iterations=1000, n_steps=None):
    """Joint (beta, sigma) estimate per subject (:290-307).  For fixed beta the NLL is minimised by
    sigma^2 = SSE/n, and beta minimises SSE, so the 2-D problem separates."""
    beta, sse = estimate_conditional(models, timepoints, cpeptide_data, nn, initial_beta=initial_beta,
                                     lower=lbfgs_lower_bound, upper=lbfgs_upper_bound, n_steps=n_steps)
    n = len(timepoints)
    sigma = np.sqrt(np.maximum(sse, 1e-300) / n)
    obj = (n / 2) * np.log(sigma ** 2) + sse / (2 * sigma ** 2)
    return [OptimizationSolution(u=ComponentArray(ode=np.array([b]), sigma=s), objective=o)
            for b, s, o in zip(beta, sigma, obj)]


def evaluate_model(models, timepoints, cpeptide_data, neural_network_parameters, betas_train, *, n_steps=None):
    """Validation objectives of every candidate network (:406-433): (n_subjects x n_networks)."""
    cols = []
    for betas, p_nn in zip(betas_train, neural_network_parameters):
        try:
            sols = train(models, timepoints, cpeptide_data, np.asarray(p_nn), initial_beta=float(np.mean(betas)),
                         lbfgs_lower_bound=-np.inf, lbfgs_upper_bound=np.inf, n_steps=n_steps)
            cols.append([s.objective for s in sols])
        except Exception:
            cols.append([np.inf] * len(models))
    return np.array(cols).T


def likelihood_profile(beta, neural_network_parameters, model, timepoints, cpeptide_data, lower_bound, upper_bound,
                       sigma, *, steps=1000, n_steps=None):
    """src/likelihood-profiles.jl:4-17 -- the `steps` probes run as ONE population of copies of the subject."""
    copies = [model] * steps
    data = np.tile(np.asarray(cpeptide_data, dtype=np.float64), (steps, 1))
    pop = _population(copies, timepoints, data, n_steps)
    values = np.linspace(lower_bound, upper_bound, steps)
    pop.engine.set_params(neural_network_parameters, values)
    nll = pop.engine.forward(want_sse=True)["sse"] / (2 * sigma ** 2)
    nll_min = loss(beta, (model, timepoints, cpeptide_data, neural_network_parameters), n_steps=n_steps) / (2 * sigma ** 2)
    return nll, nll_min, values


_CI_THRESHOLDS = {"cantelli95": 7.16, "cantelli90": 5.24, "raue95": 3.841458820694124}   # last: quantile(Chisq(1), 0.95)


def find_confidence_intervals(loss_values, loss_minimum, parameter_values, *, target="cantelli95"):
    """src/likelihood-profiles.jl:34-59: the outermost profile points whose value is within the target's threshold of
    the minimum; an end of the interval that coincides with an end of the profiled range is reported as -/+Inf.
    An unknown target falls back to raue95, as the reference does."""
    loss_values = np.asarray(loss_values, dtype=np.float64)
    threshold = loss_minimum + _CI_THRESHOLDS.get(target, _CI_THRESHOLDS["raue95"])
    idx = np.flatnonzero(loss_values <= threshold)
    if idx.size == 0:
        raise ValueError("no profile point lies within the threshold of the minimum")     # Julia: minimum of empty
    lo = -np.inf if idx[0] == 0 else float(parameter_values[idx[0]])
    hi = np.inf if idx[-1] == len(loss_values) - 1 else float(parameter_values[idx[-1]])
    return lo, hi


def likelihood_profiles(betas, neural_network_parameters, models, timepoints, cpeptide_data, lower_bound, upper_bound,
                        sigma, *, steps=1000, n_steps=None):
    """The loop `[likelihood_profile(betas[i], nn, models[i], ...) for i in ...]` of c-peptide/02-conditional.jl:186-188
    for ALL models with one launch (cude_profile_conditional: the scan value is the grid's second dimension).
    Returns (nll (N, steps), nll_min (N,), values (steps,)) -- row i is likelihood_profile of subject i."""
    pop = _population(models, timepoints, cpeptide_data, n_steps)
    eng = pop.engine
    values = np.linspace(lower_bound, upper_bound, steps)
    eng.set_params(neural_network_parameters, np.asarray(betas, dtype=np.float64).reshape(-1))
    nll_min = eng.forward(want_sse=True)["sse"] / (2 * sigma ** 2)
    nll = eng.profile_conditional(values).T / (2 * sigma ** 2)
    return nll, nll_min, values


# ----------------------------------------------------------------------------- suppression model
class _SuppPop:
    def __init__(self, data, timepoints, net, lam, n_steps, device):
        self.engine = Engine("supp", net.arch, n_steps=n_steps, lam=lam, device=device)
        net.configure(self.engine)
        if net.mask is not None:
            self.engine.set_param_mask(net.mask)
        self.engine.set_population_supp(np.asarray(timepoints, dtype=np.float64), data)
        self.data = data


def _supp_population(prob, data, timepoints, lam, n_steps=None):
    n_steps = default_steps() if n_steps is None else n_steps       # (fixed mode: DEFAULT_STEPS = 30)
    data = np.asarray(data, dtype=np.float64)
    key = ("supp", prob.network.key, float(lam), int(n_steps), _digest(timepoints, data))
    return _cached(key, lambda: _SuppPop(data, timepoints, prob.network, lam, n_steps, _DEVICE))


def lsup(u, p):
    """lsup! (suppression_model.jl:16-20): the ground-truth model that generates the suppression data; u, p: arrays whose
    first axis is the state / parameter index."""
    a = p[1] * u[1] / (1.0 + p[3] * u[2])
    return np.stack([-p[0] * u[0], p[0] * u[0] - a, a - p[2] * u[2]])


def get_group_parameters(mu_sup, n_samples, *, rng):
    """get_group_parameters (:33-37): max(mu + std * randn(4, n), 0.05), mu = [0.4, 0.9, 0.3, mu_sup],
    std = [0.1, 0.1, 0.1, mu_sup / 8].  `rng`: numpy Generator (the reference's StableRNG stream is not reproducible
    here: same distribution, other numbers)."""
    mu = np.array([0.4, 0.9, 0.3, mu_sup])[:, None]
    sd = np.array([0.1, 0.1, 0.1, mu_sup / 8.0])[:, None]
    return np.maximum(mu + sd * rng.standard_normal((4, int(n_samples))), 0.05)


def generate_data(group_means, group_sizes, timepoints, *, noise_additive=0.0, noise_multiplicative=0.0, rng=None,
                  substeps=200):
    """generate_data (:39-63): per group the parameters of get_group_parameters, per subject the solution of lsup! from
    u0 = (10, 0, 0) at `timepoints` plus additive and multiplicative Gaussian noise, clamped at 0.  Returns
    (data 3 x T x N, ground-truth suppression parameters).  Data generation is not on the device path: a classical RK4
    solve with `substeps` steps per observation interval on the host (error ~1e-8 for the reference's grid), draws
    in the reference's order (group parameters, then per subject the additive and the multiplicative field)."""
    rng = np.random.default_rng(232705) if rng is None else rng
    tp = np.asarray(timepoints, dtype=np.float64)
    T, N = tp.size, int(np.sum(group_sizes))
    data, gt, col = np.zeros((3, T, N)), [], 0
    for mean, size in zip(group_means, group_sizes):
        p = get_group_parameters(mean, size, rng=rng)
        u = np.stack([np.full(size, 10.0), np.zeros(size), np.zeros(size)])
        sol = np.empty((3, T, size))
        sol[:, 0] = u
        for k in range(1, T):
            h = (tp[k] - tp[k - 1]) / substeps
            for _ in range(substeps):
                k1 = lsup(u, p)
                k2 = lsup(u + 0.5 * h * k1, p)
                k3 = lsup(u + 0.5 * h * k2, p)
                k4 = lsup(u + h * k3, p)
                u = u + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
            sol[:, k] = u
        for j in range(size):
            s = sol[:, :, j]
            s = s + noise_additive * rng.standard_normal(s.shape) + noise_multiplicative * s * rng.standard_normal(s.shape)
            data[:, :, col] = np.maximum(s, 0.0)
            col += 1
        gt.extend(p[3].tolist())
    return data, np.asarray(gt)


def SuppressionProblem(network):
    """Stand-in for `ODEProblem(ude_lsup!, [10,0,0], (0,30))` with the network closed over
    (suppression/suppression.jl:18-20): carries the network shape; u0 comes from the data (:119)."""
    return SimpleNamespace(network=network)


def suppression_loss(p, args, *, n_steps=None):
    """suppression_loss(p, (prob, individual_data, timepoints, lambda)) with p.theta[N], p.neural[P] (:117-130)."""
    prob, data, timepoints, lam = args
    pop = _supp_population(prob, data, timepoints, lam, n_steps)
    pop.engine.set_params(p.neural, p.theta)
    return pop.engine.forward()["loss"]


def suppression_loss_and_gradient(p, args, *, n_steps=None):
    prob, data, timepoints, lam = args
    pop = _supp_population(prob, data, timepoints, lam, n_steps)
    pop.engine.set_params(p.neural, p.theta)
    val, g_nn, g_th = pop.engine.loss_grad()
    return val, ComponentArray(theta=g_th, neural=g_nn)


def simul(p, prob, individual_data, timepoints, *, n_steps=None):
    """simul(p, prob, data, timepoints) -> 3 x T x N array (:107-115)."""
    pop = _supp_population(prob, individual_data, timepoints, 0.0, n_steps)
    pop.engine.set_params(p.neural, p.theta)
    return pop.engine.forward(want_traj=True)["traj"]


def fit_suppression_model(p_init, prob, data, timepoints, lam, *, select_best_n=1, adam_iters=2000, lbfgs_iters=2000,
                          n_steps=None, side_by_side=True):
    """fit_suppression_model (:132-177): screen all initials, keep the best n, Adam() [eta = 1e-3] then L-BFGS.
    The kept restarts are trained side by side (one launch per optimiser iteration for all of them) unless
    side_by_side=False."""
    pop = _supp_population(prob, data, timepoints, lam, n_steps)
    eng = pop.engine
    init_losses = eng.multistart_forward(np.stack([p.neural for p in p_init]), np.stack([p.theta for p in p_init]))
    best = np.argsort(init_losses, kind="stable")[:max(1, select_best_n)]
    sols, traces = [], []
    if side_by_side and len(best) > 1:
        traces = [[] for _ in best]
        fits = _batched_adam_then_lbfgs(eng, np.stack([p_init[k].neural for k in best]),
                                        np.stack([p_init[k].theta for k in best]), adam_iters, lbfgs_iters, 1e-3,
                                        traces=traces)
        for fit in fits:
            if fit is None:
                print("Optimization failed")
                continue
            nn, th, obj = fit
            sols.append(OptimizationSolution(u=ComponentArray(theta=th, neural=nn), objective=obj))
        return sols, traces
    for k in best:
        trace = []
        try:
            nn, th, obj = _adam_then_lbfgs(eng, p_init[k].neural, p_init[k].theta, adam_iters, lbfgs_iters, 1e-3,
                                           callback=lambda _x, l: trace.append(l) and False)
            sols.append(OptimizationSolution(u=ComponentArray(theta=th, neural=nn), objective=obj))
        except FloatingPointError:
            print("Optimization failed")
        traces.append(trace)
    return sols, traces


def validate_suppression_model(p_init, prob, data, timepoints, network_params, *, n_steps=None, lower=-8.0,
                               upper=5.0):
    """validate_suppression_model(p_init, prob, data, timepoints, network_params) (:179-222): conditional parameters
    of new subjects with the network frozen, returns (theta, objective).  With the network fixed the loss separates
    per subject, so instead of one L-BFGS run from the best of `p_init` all subjects are solved together by a
    bracketing search over [lower, upper] (one forward launch per probe) -- the global minimum per subject, hence an
    objective <= the reference's.  p_init is accepted for signature compatibility and only widens the bracket."""
    pop = _supp_population(prob, data, timepoints, 0.0, n_steps)
    eng, N = pop.engine, np.asarray(data).shape[2]
    if p_init is not None and len(p_init):
        lower = min(lower, float(np.min(p_init)))
        upper = max(upper, float(np.max(p_init)))
    eng.set_params(network_params, None)
    theta, _, best = eng.fit_conditional(lower, upper, 161, 48)
    return theta, float(best.sum() / N)


def validate_suppression_model_sigma(p_init, prob, data, timepoints, network_params, *, n_steps=None, lower=-8.0,
                                     upper=5.0, n_grid=261, iters=50):
    """validate_suppression_model_sigma(p_init, prob, data, timepoints, network_params) (:224-275, used by
    suppression/figures.jl:46): theta and one noise level per state for a test subject, minimising
    sum_s (n/2) log sigma_s^2 + SSE_s(theta) / (2 sigma_s^2) on the UNSCALED residuals with the network frozen.
    `data` is one subject (3 x T, as in the reference) or many (3 x T x N: all of them at once).  For given theta the
    optimal sigma_s^2 is SSE_s / n, so the problem is 1-D per subject in theta: a grid over [lower, upper] and
    golden-section refinements, all subjects in lock step, one forward launch (trajectories) per probe.  Returns
    (ComponentArray(ode=theta, sigma=(3,) or (N, 3)), objective) -- the global minimum per subject, hence <= what the
    reference's L-BFGS run from the best of `p_init` reaches; p_init only widens the bracket."""
    data = np.asarray(data, dtype=np.float64)
    single = data.ndim == 2
    if single:
        data = data[:, :, None]
    pop = _supp_population(prob, data, timepoints, 0.0, n_steps)
    eng, N, n = pop.engine, data.shape[2], data.shape[1]
    if p_init is not None and len(p_init):
        lower = min(lower, float(np.min(p_init)))
        upper = max(upper, float(np.max(p_init)))

    def nll(theta):
        eng.set_params(network_params, theta)
        traj = eng.forward(want_traj=True)["traj"]                       # (3, T, N)
        sse = np.sum((traj - data) ** 2, axis=1)                         # (3, N)
        with np.errstate(divide="ignore", invalid="ignore"):
            v = np.sum(0.5 * n * (np.log(sse / n) + 1.0), axis=0)
        return np.where(np.isfinite(v), v, np.inf), sse

    # sum_s log SSE_s(theta) has a narrow dip wherever one state is fitted well: a fine grid, then the three deepest
    # local minima of every subject are refined and the best kept
    grid = np.linspace(lower, upper, n_grid)
    vals = np.stack([nll(np.full(N, g))[0] for g in grid])              # (n_grid, N)
    pad = np.concatenate([np.full((1, N), np.inf), vals, np.full((1, N), np.inf)])
    is_min = (vals <= pad[:-2]) & (vals <= pad[2:])                      # the ends of the bracket count
    score = np.where(is_min & np.isfinite(vals), vals, np.inf)
    r = (math.sqrt(5.0) - 1.0) / 2.0
    theta, best = np.full(N, grid[0]), np.full(N, np.inf)
    for rank in range(3):
        k = np.argsort(score, axis=0, kind="stable")[rank]              # (N,) grid index of the rank-th deepest minimum
        k = np.where(np.isfinite(score[k, np.arange(N)]), k, np.argmin(vals, axis=0))
        a, b = grid[np.maximum(k - 1, 0)], grid[np.minimum(k + 1, n_grid - 1)]
        x1, x2 = b - r * (b - a), a + r * (b - a)
        f1, f2 = nll(x1)[0], nll(x2)[0]
        for _ in range(iters):
            left = f1 < f2
            b = np.where(left, x2, b)
            a = np.where(left, a, x1)
            probe = np.where(left, b - r * (b - a), a + r * (b - a))    # the surviving interior point is reused
            fp = nll(probe)[0]
            x1, f1, x2, f2 = (np.where(left, probe, x2), np.where(left, fp, f2), np.where(left, x1, probe),
                              np.where(left, f1, fp))
        cand, fc = np.where(f1 < f2, x1, x2), np.minimum(f1, f2)
        gk = vals[k, np.arange(N)]                                       # (a minimum AT an end of the bracket)
        cand, fc = np.where(gk < fc, grid[k], cand), np.minimum(gk, fc)
        theta, best = np.where(fc < best, cand, theta), np.minimum(fc, best)
    best, sse = nll(theta)
    sigma = np.sqrt(sse / n).T                                           # (N, 3)
    if single:
        return ComponentArray(ode=float(theta[0]), sigma=sigma[0]), float(best[0])
    return ComponentArray(ode=theta, sigma=sigma), best


# ----------------------------------------------------------------------------- utilities of the scripts (src/utils.jl)
def stratified_split(rng, types, f_train):
    """stratified_split(rng, types, f_train) (src/utils.jl:15-31): per type, round(f_train * count) subjects sampled
    without replacement for training; returns (sorted training indices, the complement), 0-based.  `rng` is a
    numpy Generator (the reference's StableRNG stream is not reproducible here: the split is of the same
    distribution, not the same subjects)."""
    types = np.asarray(types)
    train = []
    for ty in dict.fromkeys(types.tolist()):                              # unique(), first-occurrence order
        idx = np.flatnonzero(types == ty)
        n_train = int(round(f_train * idx.size))                          # round half to even, as Julia's round
        train.extend(rng.choice(idx, size=n_train, replace=False).tolist())
    train = np.sort(np.asarray(train, dtype=np.int64))
    return train, np.setdiff1d(np.arange(types.size), train)


def argmedian(x):
    """argmin(abs.(x .- median(x))) (src/utils.jl:43-45)."""
    x = np.asarray(x, dtype=np.float64)
    return int(np.argmin(np.abs(x - np.median(x))))


# ----------------------------------------------------------------------------- SAEM
def map_objective(p_individual, sse, n_obs, sigma, omega, *, prior_individual=0.0):
    """map_objective (src/saem.jl:68-72): -(log-likelihood + log N(p; prior_individual, omega)) given the subject's SSE."""
    return -(individual_log_likelihood(sse, n_obs, sigma) + _log_normal(p_individual, prior_individual, omega))


def compute_individual_maps(p_individuals, p_neural, models, timepoints, cpeptide_data, sigma, omega, *,
                            prior_individual=0.0, lower=-6.0, upper=4.0, n_steps=None):
    """compute_individual_maps (src/saem.jl:74-84): every subject's maximum-a-posteriori conditional parameter with
    the network frozen.  argmin_x -(ll + logprior) = argmin_x SSE(x) + (sigma / omega)^2 (x - prior)^2: one penalised
    per-subject search on the device for all subjects (cude_fit_conditional) instead of an L-BFGS run per subject;
    p_individuals (the reference's starting points) only widen the bracket."""
    models = [models] if _is_model(models) else list(models)
    data = np.asarray(cpeptide_data, dtype=np.float64)
    data = data[None, :] if data.ndim == 1 else data
    pop = _population(models, timepoints, data, n_steps)
    if p_individuals is not None and np.size(p_individuals):
        lower = min(lower, float(np.min(p_individuals)))
        upper = max(upper, float(np.max(p_individuals)))
    pop.engine.set_params(p_neural, None)
    x, _, _ = pop.engine.fit_conditional(lower, upper, 81, 48, penalty_weight=(sigma / omega) ** 2,
                                         penalty_center=prior_individual)
    return x


def individual_log_likelihood(sse, n_obs, sigma):
    """-(n/2) log sigma^2 - SSE/(2 sigma^2), -Inf on solver failure (src/saem.jl:55-66)."""
    ll = -(n_obs / 2) * math.log(sigma ** 2) - sse / (2 * sigma ** 2)
    return np.where(np.isfinite(sse), ll, -np.inf)


def _log_normal(x, mu, sd):
    return -0.5 * ((x - mu) / sd) ** 2 - math.log(sd) - 0.5 * math.log(2 * math.pi)


def mcmc_steps(sse_of, p_individual, n_obs, sigma, omega, proposal_std, prior_individual, temperature, gamma,
               normals, uniforms):
    """`mcmc_step` (src/saem.jl:86-108) applied `len(normals)` times to EVERY subject at once, with the
    stochastic-approximation update of the chain state (:185).  `sse_of(beta[N]) -> SSE[N]` is one forward launch
    over the population; normals / uniforms are (steps, N) draws (randn() / rand() of the reference).  As in the
    reference the current state's likelihood is re-evaluated every step.  Returns (p_individual, accepted[N])."""
    p = np.array(p_individual, dtype=np.float64)
    acc_count = np.zeros(p.size, dtype=np.int64)
    for z, u in zip(normals, uniforms):
        prop = p + z * proposal_std
        prior_ratio = _log_normal(prop, prior_individual, omega) - _log_normal(p, prior_individual, omega)
        ll_new = individual_log_likelihood(sse_of(prop), n_obs, sigma)
        ll_cur = individual_log_likelihood(sse_of(p), n_obs, sigma)
        acc = np.log(u) < prior_ratio + (ll_new / temperature - ll_cur / temperature)
        acc_count += acc
        p = (1 - gamma) * p + gamma * np.where(acc, prop, p)
    return p, acc_count


def SAEM(models, timepoints, cpeptide_data, initial_neural_params, *, sigma=1.0, prior_eta=0.0, prior_omega=1.0,
         iterations=500, n_burnin_iterations=100, proposal_std=0.1, proposal_std_bounds=(1e-3, 1.0), alpha=0.7,
         n_mcmc_steps=1, initial_mcmc_steps=None, target_acceptance_rate=0.25, initial_temperature=10.0,
         temperature_decay=0.05, omega_learning_rate=0.04, rng=None, n_steps=None, m_step_iters=5, m_step_lr=1e-2,
         collective=None, device_seed=None, subject_offset=0):
    """SAEM(individuals, initial_neural_params, network; ...) (src/saem.jl:134-237) with the population on the
    GPU: every Metropolis step evaluates all subjects in one forward launch; the M-step's 5 Adam iterations on
    (network, sigma) use the device gradient.  The loop itself is cude.parallel.saem_loop, shared with the
    subject-sharded multi-GPU form (`collective`: this process holds one shard of `models`).  device_seed: the
    Metropolis draws are generated on the device (counter-based, independent of the sharding given subject_offset)
    instead of by `rng` on the host."""
    from .parallel import saem_loop
    pop = _population(models, timepoints, cpeptide_data, n_steps)
    return saem_loop(pop.engine, pop.T, initial_neural_params, collective=collective, sigma=sigma,
                     prior_eta=prior_eta, prior_omega=prior_omega, iterations=iterations,
                     n_burnin_iterations=n_burnin_iterations, proposal_std=proposal_std,
                     proposal_std_bounds=proposal_std_bounds, alpha=alpha, n_mcmc_steps=n_mcmc_steps,
                     initial_mcmc_steps=initial_mcmc_steps, target_acceptance_rate=target_acceptance_rate,
                     initial_temperature=initial_temperature, temperature_decay=temperature_decay,
                     omega_learning_rate=omega_learning_rate, rng=rng, m_step_iters=m_step_iters, m_step_lr=m_step_lr,
                     device_seed=device_seed, subject_offset=subject_offset)


def individual_effects(models, timepoints, cpeptide_data, saem_result, *, n_samples=3000, proposal_std=0.3,
                       rng=None, n_steps=None, lower=-6.0, upper=4.0):
    """The per-individual loop that follows SAEM in c-peptide/06-saem.jl:97-135, for ALL individuals at once:
    posterior samples of the conditional parameter (n_samples Metropolis steps from the population mean, every
    state kept, :107-112), the MAP mode (minimiser of -(log-likelihood + log prior), :114-119), the MLE estimate
    (:121-126) and `mse = -2 individual_log_likelihood(mode; sigma = 1)` = the SSE at the mode (:129).  The two 1-D
    optimisations are bracketing searches over [lower, upper] with one forward launch per probe."""
    rng = np.random.default_rng() if rng is None else rng
    pop = _population(models, timepoints, cpeptide_data, n_steps)
    eng, N, T = pop.engine, pop.N, pop.T
    prior, omega, sigma = float(saem_result.eta), float(saem_result.Omega), float(saem_result.sigma)
    eng.set_params(saem_result.p_neural, np.full(N, prior))
    acc, samples = eng.mh_chain(rng.standard_normal((n_samples, N)), rng.random((n_samples, N)), sigma, prior, omega,
                                proposal_std)

    # -(log-likelihood + log prior) = [SSE + (sigma/Omega)^2 (b - prior)^2] / (2 sigma^2) + const
    modes, _, mse = eng.fit_conditional(lower, upper, 81, 48, (sigma / omega) ** 2, prior)
    mle, _, _ = eng.fit_conditional(lower, upper, 81, 48)
    return SimpleNamespace(samples=samples, modes=modes, mle=mle, mse=mse,
                           acceptance_rate=float(acc.sum()) / (n_samples * N))


# ----------------------------------------------------------------------------- symbolic (Michaelis-Menten) model
def train_symbolic(models, timepoints, cpeptide_data, *, lower=0.0, upper=1000.0, n_steps=None):
    """The per-subject loop of c-peptide/03-symreg.jl:94-106: minimise loss_sigma over (k, sigma) with
    0 <= k <= 1000 for every CPeptideODEModel.  For fixed k the optimum is sigma^2 = SSE/n, and k minimises the
    SSE, so all N problems are solved together by a bracketing search on log k (one forward launch per probe).
    Returns OptimizationSolution(u = ComponentArray(ode=[k], sigma), objective) per subject."""
    pop = _population(models, timepoints, cpeptide_data, n_steps, cond_space="log")
    eng, N = pop.engine, pop.N
    eng.set_params(pop.shared, np.zeros(N))
    lo = math.log(max(lower, 1e-6))
    hi = math.log(upper)

    logk, _, val = eng.fit_conditional(lo, hi, 61, 48)
    n = len(timepoints)
    sigma = np.sqrt(np.maximum(val, 1e-300) / n)
    obj = (n / 2) * np.log(sigma ** 2) + val / (2 * sigma ** 2)
    return [OptimizationSolution(u=ComponentArray(ode=np.array([k]), sigma=s), objective=o)
            for k, s, o in zip(np.exp(logk), sigma, obj)]


def SAEM_symbolic(models, timepoints, cpeptide_data, initial_population_parameter, *, sigma=1.0, prior_eta=0.0,
                  prior_omega=1.0, iterations=500, n_burnin_iterations=100, proposal_std=0.1,
                  proposal_std_bounds=(1e-3, 1.0), alpha=0.7, n_mcmc_steps=1, initial_mcmc_steps=None,
                  target_acceptance_rate=0.25, initial_temperature=10.0, temperature_decay=0.05,
                  omega_learning_rate=0.04, rng=None, n_steps=None, m_step_iters=5):
    """SAEM(individuals, initial_population_parameter; ...) of src/saem-symreg.jl:134-229: random effects eta_i
    with k_i = km_pop * exp(eta_i) (:57-59), Metropolis E-step (:86-108) on the device through the log-space
    conditional log k_i = log km_pop + eta_i (prior Normal(0, Omega) on eta), M-step = `m_step_iters` L-BFGS
    iterations on (km, sigma) of total_nll (:110-131) with the device gradient."""
    rng = np.random.default_rng() if rng is None else rng
    initial_mcmc_steps = n_mcmc_steps if initial_mcmc_steps is None else initial_mcmc_steps
    pop = _population(models, timepoints, cpeptide_data, n_steps, cond_space="log")
    eng, N, T = pop.engine, pop.N, pop.T
    eta = np.full(N, float(prior_eta))
    km = float(initial_population_parameter)
    omega = float(prior_omega)
    nll_values, acc_rates = [], []
    for it in range(1, iterations + 1):
        gamma = 1.0 if it <= n_burnin_iterations else 1.0 / (it - n_burnin_iterations) ** alpha
        steps = initial_mcmc_steps if it <= n_burnin_iterations else n_mcmc_steps
        temperature = max(1.0, initial_temperature * math.exp(-temperature_decay * it))
        # E-step: the chain state on the device is log k = log km + eta, so the prior is centred on log km
        eng.set_params(pop.shared, math.log(km) + eta)
        n_acc = eng.mh_estep(rng.standard_normal((steps, N)), rng.random((steps, N)), sigma, math.log(km), omega,
                             proposal_std, temperature, gamma)
        eta = eng.get_params()[1] - math.log(km)
        sse = eng.forward(want_sse=True)["sse"]
        loglik = float(individual_log_likelihood(sse, T, sigma).sum())

        def fg(x):                                   # total_nll(km, eta, individuals, sigma) and its gradient
            k_pop, s = x
            if k_pop <= 0 or s <= 0:
                return np.inf, np.zeros(2)
            eng.set_params(None, math.log(k_pop) + eta)
            mean_sse, _, g_log = eng.loss_grad()
            val = N * T / 2 * math.log(s * s) + N * mean_sse / (2 * s * s)
            return val, np.array([N * g_log.sum() / k_pop / (2 * s * s), N * T / s - N * mean_sse / s ** 3])
        res = lbfgs(fg, np.array([km, sigma]), maxiters=m_step_iters)
        km_new, sigma = float(res["x"][0]), float(res["x"][1])
        km = (1 - gamma) * km + gamma * km_new
        omega = (1 - omega_learning_rate) * omega + omega_learning_rate * float(np.var(eta, ddof=1))
        rate = int(n_acc.sum()) / (N * steps)
        nll_values.append(-loglik)
        acc_rates.append(rate)
        if it > n_burnin_iterations:
            proposal_std = float(np.clip(math.exp(math.log(proposal_std) + gamma * (rate - target_acceptance_rate)),
                                         proposal_std_bounds[0], proposal_std_bounds[1]))
    return SimpleNamespace(km_pop=km, eta=eta, Omega=omega, sigma=sigma, total_nll_values=nll_values,
                           acceptance_rates=acc_rates)


def simulate(p_neural, p_individuals, models, timepoints, cpeptide_data, *, out_timepoints=None, n_steps=None,
             save_idxs=1):
    """`simulate(p_neural, p_individual, individual, network; timepoints)` of src/saem.jl:31-53 (and
    src/saem-symreg.jl:31-53 for CPeptideODEModel lists, p_neural = None) for ALL individuals at once: plasma
    c-peptide (state `save_idxs`, 1-based as in Julia) at `out_timepoints` -- any non-decreasing times inside the
    span of `timepoints`, e.g. the dense grids t0:0.1:tend of the model-fit figures -- as an (N, n_times) array."""
    sym = isinstance(models[0], CPeptideODEModel)
    pop = _population(models, timepoints, cpeptide_data, n_steps, cond_space="raw" if sym else "log")
    if isinstance(models[0], CPeptideUDEModel):      # `solve(model.problem, p = neural_network_parameters)` per subject
        pop.engine.set_params(models[0].embed(p_neural), np.zeros(len(models)))   # (01-non-conditional.jl:64,74)
    else:
        pop.engine.set_params(pop.shared if sym else p_neural, np.asarray(p_individuals, dtype=np.float64).reshape(-1))
    times = np.asarray(timepoints if out_timepoints is None else out_timepoints, dtype=np.float64)
    return pop.engine.simulate(times)[save_idxs - 1].T


# ----------------------------------------------------------------------------- checkpoints / data files (JLD2)
def load_data(path):
    """`jldopen("data/ohashi.jld2") do file; file["train"], file["test"]; end` (c-peptide/02-conditional.jl:15-17):
    every top-level entry of a JLD2 file; NamedTuples become SimpleNamespaces with numpy fields."""
    from . import jld2
    out = {}
    for k, v in jld2.load(path).items():
        out[k] = SimpleNamespace(**v) if isinstance(v, dict) else v
    return out


def save_parameters(path, width, depth, parameters, betas=None, best_model_index=None, **extra):
    """The checkpoint the training scripts write (c-peptide/02-conditional.jl:44-50, 07-covariate-inclusion.jl:59-65):
    width, depth, parameters, betas, best_model_index (1-based, as in the reference).  Lists of vectors are stored
    as Vector{Vector{Float64}}, so the reference's `file["parameters"][best_model_index]` reads them unchanged; with
    the reference's content the file is byte-identical to the one JLD2.jl wrote (tests/test_jld2.py)."""
    from . import jld2
    entries = {"width": int(width), "depth": int(depth), "parameters": parameters}
    if betas is not None:
        entries["betas"] = betas
    if best_model_index is not None:
        entries["best_model_index"] = int(best_model_index)
    entries.update(extra)
    jld2.save(path, entries)


def load_parameters(path):
    """Reads a checkpoint written by the reference or by save_parameters (vectors of vectors; a matrix is taken as
    one vector per column):
    namespace with width, depth, parameters (list of vectors, or one vector), betas, best_model_index."""
    from . import jld2
    d = jld2.load(path)

    def as_list(v):
        if isinstance(v, np.ndarray) and v.ndim == 2:
            return [v[:, j].copy() for j in range(v.shape[1])]
        return v
    return SimpleNamespace(width=d.get("width"), depth=d.get("depth"), parameters=as_list(d.get("parameters")),
                           betas=as_list(d.get("betas")), best_model_index=d.get("best_model_index"),
                           extra={k: v for k, v in d.items()
                                  if k not in ("width", "depth", "parameters", "betas", "best_model_index")})
