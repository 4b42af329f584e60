"""Thin object wrapper over the C ABI: one Engine = one cude_ctx = one GPU.

Holds no numerics: every number comes from libcude_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import COND_LOG, COND_RAW, MODEL_CPEP, MODEL_CPEP_SYM, MODEL_SUPP, CudeError, check


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def n_params(nn_in, width, depth):
    return check(_lib.load().cude_n_params(nn_in, width, depth))


_OBJECTIVE = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                         C.c_void_p)


def lbfgs_minimize(fg, x0, maxiters=1000):
    """The library's host-side L-BFGS + BackTracking (cude_lbfgs_minimize; needs no GPU) on a Python objective
    fg(x) -> (f, g).  Returns dict(x, f, iterations, f_calls, converged) like cude.lbfgs.lbfgs."""
    x0 = _f64(x0).reshape(-1)
    n = x0.size

    err = []

    def thunk(xp, nn, fp, gp, _user):
        try:
            f, g = fg(np.ctypeslib.as_array(xp, shape=(nn,)).copy())
            fp[0] = float(f)
            np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
            return 0
        except BaseException as e:      # a ctypes callback must not raise: hand the error back through the status
            err.append(e)
            return -1
    cb = _OBJECTIVE(thunk)
    x = np.empty(n)
    f, it, calls, conv = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = _lib.load().cude_lbfgs_minimize(n, _ptr(x0), int(maxiters), C.cast(cb, C.c_void_p), None, _ptr(x),
                                         C.byref(f), C.byref(it), C.byref(calls), C.byref(conv))
    if err:
        raise err[0]
    check(rc)
    return dict(x=x, f=f.value, iterations=it.value, f_calls=calls.value, converged=bool(conv.value))


_CANDIDATES = C.CFUNCTYPE(C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
_REDUCE = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_double), C.c_int32, C.c_int32, C.c_void_p)


def lbfgs_minimize_sharded(fg, x0, n_shared, reduce, maxiters=1000):
    """cude_lbfgs_minimize_sharded: x = [shared (n_shared, replicated); local]; fg(x) -> (global f, local g);
    reduce(values: ndarray, op) -> ndarray sums (op 0) / maximises (op 1) over the ranks."""
    x0 = _f64(x0).reshape(-1)
    n = x0.size

    err = []

    # A callback that raised would leave this rank on a different L-BFGS path from its peers, which then block in the
    # next collective: the error goes back through the status (rc < 0 -> CUDE_ERR_ARG / CUDE_ERR_COMM) and is re-raised
    # here once the library has returned.
    def thunk(xp, nn, fp, gp, _user):
        try:
            f, g = fg(np.ctypeslib.as_array(xp, shape=(nn,)).copy())
            fp[0] = float(f)
            np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
            return 0
        except BaseException as e:
            err.append(e)
            return -1

    def red(vp, count, op, _user):
        try:
            v = np.ctypeslib.as_array(vp, shape=(count,))
            v[:] = reduce(v.copy(), int(op))
            return 0
        except BaseException as e:
            err.append(e)
            return -1
    cb, rcb = _OBJECTIVE(thunk), _REDUCE(red)
    x = np.empty(n)
    f, it, calls, conv = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
    rc = _lib.load().cude_lbfgs_minimize_sharded(n, int(n_shared), _ptr(x0), int(maxiters), C.cast(cb, C.c_void_p),
                                                 C.cast(rcb, C.c_void_p), None, _ptr(x), C.byref(f), C.byref(it),
                                                 C.byref(calls), C.byref(conv))
    if err:
        raise err[0]
    check(rc)
    return dict(x=x, f=f.value, iterations=it.value, f_calls=calls.value, converged=bool(conv.value))


def device_count():
    n = C.c_int32(0)
    check(_lib.load().cude_device_count(C.byref(n)))
    return n.value


class Engine:
    """model: 'cpep' | 'supp' | 'cpep_sym'.  arch = (nn_in, width, depth); (1, 0, 0) for 'cpep_sym' (the analytic
    production p0*dG/(dG+k): parameters [p0], conditional k or log k according to cond_space = 'raw' | 'log')."""

    def __init__(self, model, arch=(1, 0, 0), n_steps=30, n_state=None, lam=0.0, device=0, cond_space="log"):
        self._lib = _lib.load()
        self.model = {"cpep": MODEL_CPEP, "supp": MODEL_SUPP, "cpep_sym": MODEL_CPEP_SYM}[model]
        # the general form `chain(widths, activation_functions; input_dims, output_activation)`: arch = (nn_in, [widths],
        # [one activation per hidden layer], output activation), names as in ACTIVATIONS (cude_set_network)
        general = None
        if len(arch) >= 2 and isinstance(arch[1], (list, tuple)):
            widths = [int(w) for w in arch[1]]
            hidden = arch[2] if len(arch) > 2 else "tanh"
            hidden = [hidden] * len(widths) if isinstance(hidden, str) else list(hidden)
            if len(hidden) != len(widths):
                raise ValueError("The number of widths must match the number of activation functions.")
            general = (widths, hidden + [arch[3] if len(arch) > 3 else "softplus"])
            arch = (arch[0], max(widths), len(widths))
        self.arch = tuple(int(v) for v in arch[:3])
        if n_state is None:
            n_state = 3 if self.model == MODEL_SUPP else 2
        space = {"log": COND_LOG, "raw": COND_RAW}[cond_space]
        cfg = _lib.Config(self.model, n_state, self.arch[0], self.arch[1], self.arch[2], int(n_steps), int(device),
                          space, float(lam))
        h = C.c_void_p()
        check(self._lib.cude_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.n_state = n_state
        self.lam = float(lam)
        self.N = 0
        self.T = 0
        if general is not None:
            self.set_network(*general)
        self.P, self.fallback_kernel = self.network_info()

    ACTIVATIONS = {"tanh": 0, "relu": 1, "sigmoid": 2, "softplus": 3, "identity": 4}

    def set_network(self, widths, activations):
        """cude_set_network: per-layer widths and activation names (hidden layers, then the output layer's).  Before the
        population is uploaded.  Shapes without a tuned kernel run on the fallback kernel (csrc/cude_generic.hip)."""
        w = (C.c_int32 * len(widths))(*[int(v) for v in widths])
        a = (C.c_int32 * len(activations))(*[self.ACTIVATIONS[str(v)] for v in activations])
        if len(activations) != len(widths) + 1:
            raise ValueError("one activation per hidden layer and one for the output layer")
        check(self._lib.cude_set_network(self._h, len(widths), w, a))
        self.P, self.fallback_kernel = self.network_info()

    def network_info(self):
        """(number of shared parameters, True if the network runs on the fallback kernel)."""
        p, g = C.c_int32(), C.c_int32()
        check(self._lib.cude_network_info(self._h, C.byref(p), C.byref(g)))
        return p.value, bool(g.value)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.cude_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tolerances(self, abstol=1e-6, reltol=1e-3):
        """Tolerances of the adaptive mode (n_steps = 0)."""
        check(self._lib.cude_set_tolerances(self._h, float(abstol), float(reltol)))

    def adaptive_steps(self, subject):
        """(t_n, dt_n) of the steps `subject` accepted in the last gradient evaluation (adaptive mode): `sol.t`."""
        n = C.c_int32(0)
        check(self._lib.cude_adaptive_steps(self._h, int(subject), 0, None, None, C.byref(n)))
        t, dt = np.empty(n.value), np.empty(n.value)
        check(self._lib.cude_adaptive_steps(self._h, int(subject), n.value, _ptr(t), _ptr(dt), C.byref(n)))
        return t, dt

    # -- population
    def set_population_cpep(self, timepoints, glucose, cpeptide, age, t2dm):
        """glucose, cpeptide: (N, T) arrays (any strides are honoured without a host copy when
        they are element-strided float64)."""
        tp = _f64(timepoints)
        g = np.asarray(glucose, dtype=np.float64)
        cp = np.asarray(cpeptide, dtype=np.float64)
        if g.shape != cp.shape or g.ndim != 2 or g.shape[1] != tp.size:
            raise ValueError("glucose/cpeptide must be (N, T) with T = len(timepoints)")
        if g.strides != cp.strides or g.strides[0] % 8 or g.strides[1] % 8:
            g, cp = np.ascontiguousarray(g), np.ascontiguousarray(cp)
        N, T = g.shape
        age = _f64(age)
        t2 = np.ascontiguousarray(t2dm, dtype=np.uint8)
        if age.size != N or t2.size != N:
            raise ValueError("age/t2dm must have N entries")
        check(self._lib.cude_set_population_cpep(self._h, N, T, _ptr(tp), _ptr(g), _ptr(cp), g.strides[0] // 8,
                                                 g.strides[1] // 8, _ptr(age), _ptr(t2)))
        self.N, self.T = N, T

    def set_population_supp(self, timepoints, data):
        """data: (3, T, N) array (Julia individual_data); passed in Julia column-major order."""
        tp = _f64(timepoints)
        d = np.asarray(data, dtype=np.float64)
        if d.ndim != 3 or d.shape[0] != 3 or d.shape[1] != tp.size:
            raise ValueError("data must be (3, T, N)")
        dcol = np.ascontiguousarray(d.transpose(2, 1, 0))
        N, T = d.shape[2], d.shape[1]
        check(self._lib.cude_set_population_supp(self._h, N, T, _ptr(tp), _ptr(dcol)))
        self.N, self.T = N, T

    # -- parameters
    def set_params(self, nn=None, cond=None):
        nn_a = None if nn is None else _f64(nn)
        cd_a = None if cond is None else _f64(cond).reshape(-1)
        if nn_a is not None and nn_a.size != self.P:
            raise ValueError(f"expected {self.P} network parameters, got {nn_a.size}")
        if cd_a is not None and cd_a.size != self.N:
            raise ValueError(f"expected {self.N} conditional parameters, got {cd_a.size}")
        check(self._lib.cude_set_params(self._h, _ptr(nn_a), _ptr(cd_a)))

    def get_params(self):
        nn = np.empty(self.P)
        cond = np.empty(self.N)
        check(self._lib.cude_get_params(self._h, _ptr(nn), _ptr(cond)))
        return nn, cond

    # -- evaluation
    def forward(self, want_sse=False, want_traj=False):
        loss = C.c_double()
        sse = np.empty(self.N) if want_sse else None
        traj = np.empty((self.N, self.T, self.n_state)) if want_traj else None
        check(self._lib.cude_forward(self._h, C.byref(loss), _ptr(sse), _ptr(traj)))
        out = {"loss": loss.value}
        if want_sse:
            out["sse"] = sse
        if want_traj:
            out["traj"] = traj.transpose(2, 1, 0)     # -> (n_state, T, N), the Julia Array(sol) shape
        return out

    def simulate(self, times):
        """States of every subject at arbitrary non-decreasing times inside the population's time span (dense
        output of the same fixed-step solve) -> array (n_state, len(times), N).  c-peptide models only."""
        t = _f64(times).reshape(-1)
        out = np.empty((self.N, t.size, self.n_state))
        check(self._lib.cude_simulate(self._h, t.size, _ptr(t), _ptr(out)))
        return np.ascontiguousarray(out.transpose(2, 1, 0))

    def multistart_forward(self, nn_sets, cond_sets):
        """Losses of K candidate (network, conditional) parameter sets in one launch.
        nn_sets: (K, P); cond_sets: (K, N)."""
        nn = _f64(nn_sets)
        cd = _f64(cond_sets)
        if nn.ndim != 2 or nn.shape[1] != self.P or cd.shape != (nn.shape[0], self.N):
            raise ValueError(f"expected nn_sets (K, {self.P}) and cond_sets (K, {self.N})")
        losses = np.empty(nn.shape[0])
        check(self._lib.cude_multistart_forward(self._h, nn.shape[0], _ptr(nn), _ptr(cd), _ptr(losses)))
        return losses

    def screen_candidates(self, n_candidates, n_keep, gen):
        """Screening with the selection on the device: gen(first, count) -> (nn (count, P), cond (count, N)) is asked
        for one chunk at a time; returns (indices, losses, nn (n_keep, P), cond (n_keep, N)) of the n_keep best
        candidates in increasing order of loss (ties: lower index first, as partialsortperm)."""
        P, N = self.P, self.N

        def thunk(first, count, nn_p, cond_p, _user):
            try:
                nn, cond = gen(int(first), int(count))
                np.ctypeslib.as_array(nn_p, shape=(count, P))[:] = nn
                np.ctypeslib.as_array(cond_p, shape=(count, N))[:] = cond
                return 0
            except Exception:      # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1
        cb = _CANDIDATES(thunk)
        n_keep = min(int(n_keep), int(n_candidates))
        idx, loss = np.empty(n_keep, dtype=np.int64), np.empty(n_keep)
        nn_out, cond_out = np.empty((n_keep, P)), np.empty((n_keep, N))
        check(self._lib.cude_screen_candidates(self._h, int(n_candidates), n_keep, C.cast(cb, C.c_void_p), None,
                                               _ptr(idx), _ptr(loss), _ptr(nn_out), _ptr(cond_out)))
        return idx, loss, nn_out, cond_out

    def multistart_loss_grad(self, nn_sets, cond_sets):
        """Loss and gradient of K parameter sets in one launch (restarts trained side by side).
        nn_sets: (K, P); cond_sets: (K, N) -> losses (K,), g_nn (K, P), g_cond (K, N)."""
        nn = _f64(nn_sets)
        cd = _f64(cond_sets)
        if nn.ndim != 2 or nn.shape[1] != self.P or cd.shape != (nn.shape[0], self.N):
            raise ValueError(f"expected nn_sets (K, {self.P}) and cond_sets (K, {self.N})")
        K = nn.shape[0]
        losses, g_nn, g_cond = np.empty(K), np.empty((K, self.P)), np.empty((K, self.N))
        check(self._lib.cude_multistart_loss_grad(self._h, K, _ptr(nn), _ptr(cd), _ptr(losses), _ptr(g_nn),
                                                  _ptr(g_cond)))
        return losses, g_nn, g_cond

    def mh_estep(self, normals, uniforms, sigma, prior_mean, prior_sd, proposal_std, temperature=1.0, gamma=1.0,
                 n_mc=None):
        """n_mc Metropolis-Hastings steps for every subject on the device (chain state = the context's
        conditional parameters, updated in place).  normals/uniforms: (n_mc, N) host draws, or both None with n_mc
        given: the draws come from the device-side generator (set_rng).  Returns acceptance counts."""
        acc = np.zeros(self.N, dtype=np.int64)
        if normals is None and uniforms is None:
            if n_mc is None:
                raise ValueError("n_mc is required when the draws are generated on the device")
            z = u = None
            steps = int(n_mc)
        else:
            z = _f64(normals)
            u = _f64(uniforms)
            if z.ndim != 2 or z.shape[1] != self.N or u.shape != z.shape:
                raise ValueError(f"expected draws of shape (n_mc, {self.N})")
            steps = z.shape[0]
        check(self._lib.cude_mh_estep(self._h, steps, _ptr(z), _ptr(u), float(sigma), float(prior_mean),
                                      float(prior_sd), float(proposal_std), float(temperature), float(gamma),
                                      _ptr(acc)))
        return acc

    def set_param_mask(self, mask):
        """Freeze shared parameters: network-gradient entries are multiplied by mask (P entries of 0 / 1; None lifts
        it), so frozen entries keep their value under every optimiser of the library."""
        m = None if mask is None else _f64(mask).reshape(-1)
        if m is not None and m.size != self.P:
            raise ValueError(f"expected a mask of {self.P} entries")
        check(self._lib.cude_set_param_mask(self._h, _ptr(m)))

    def set_rng(self, seed, subject_offset=0):
        """Seed / rewind the device-side draws of mh_estep / mh_chain; subject_offset = global index of this
        context's first subject (draws do not depend on the sharding)."""
        check(self._lib.cude_set_rng(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(subject_offset)))

    def rng_draws(self, first_step, n_steps):
        """(normals, uniforms), each (n_steps, N): the draws steps first_step ... of the device stream use."""
        z = np.empty((int(n_steps), self.N))
        u = np.empty((int(n_steps), self.N))
        check(self._lib.cude_rng_draws(self._h, int(first_step), int(n_steps), _ptr(z), _ptr(u)))
        return z, u

    def train_restarts(self, nn_sets, cond_sets, adam_iters, learning_rate, lbfgs_iters, want_trace=False):
        """K restarts trained side by side inside the library (Adam, then L-BFGS + BackTracking in lock step):
        nn_sets (K, P), cond_sets (K, N) -> trained (nn (K, P), cond (K, N), objective (K,); +Inf = dropped)
        [, loss trace (K, adam_iters + lbfgs_iters), NaN after a run stopped]."""
        nn = _f64(nn_sets)
        cd = _f64(cond_sets)
        if nn.ndim != 2 or nn.shape[1] != self.P or cd.shape != (nn.shape[0], self.N):
            raise ValueError(f"expected nn_sets (K, {self.P}) and cond_sets (K, {self.N})")
        K = nn.shape[0]
        nn_out, cond_out, obj = np.empty_like(nn), np.empty_like(cd), np.empty(K)
        trace = np.empty((K, int(adam_iters) + int(lbfgs_iters))) if want_trace else None
        check(self._lib.cude_train_restarts(self._h, K, _ptr(nn), _ptr(cd), int(adam_iters), float(learning_rate),
                                            int(lbfgs_iters), _ptr(nn_out), _ptr(cond_out), _ptr(obj), _ptr(trace)))
        return (nn_out, cond_out, obj, trace) if want_trace else (nn_out, cond_out, obj)

    def profile_conditional(self, values):
        """SSE of every subject at every scan value of the conditional parameter (shared parameters frozen), all in
        one launch: values (K,) -> (K, N)."""
        v = _f64(values).reshape(-1)
        out = np.empty((v.size, self.N))
        check(self._lib.cude_profile_conditional(self._h, v.size, _ptr(v), _ptr(out)))
        return out

    def fit_conditional(self, lower, upper, n_grid=41, n_iters=48, penalty_weight=0.0, penalty_center=0.0):
        """All subjects' 1-D fits of the conditional parameter with the shared parameters frozen, on the device:
        argmin_x SSE_i(x) + penalty_weight (x - penalty_center)^2 over [lower, upper] -> (x[N], objective[N], sse[N])."""
        x, obj, sse = np.empty(self.N), np.empty(self.N), np.empty(self.N)
        check(self._lib.cude_fit_conditional(self._h, float(lower), float(upper), int(n_grid), int(n_iters),
                                             float(penalty_weight), float(penalty_center), _ptr(x), _ptr(obj),
                                             _ptr(sse)))
        return x, obj, sse

    def mh_chain(self, normals, uniforms, sigma, prior_mean, prior_sd, proposal_std, temperature=1.0, gamma=1.0,
                 n_mc=None):
        """mh_estep that also returns every state of the chain: (accepted (N,), samples (n_mc, N)).  normals =
        uniforms = None with n_mc given: device-side draws."""
        if normals is None and uniforms is None:
            if n_mc is None:
                raise ValueError("n_mc is required when the draws are generated on the device")
            z = u = None
            shape = (int(n_mc), self.N)
        else:
            z = _f64(normals)
            u = _f64(uniforms)
            if z.ndim != 2 or z.shape[1] != self.N or u.shape != z.shape:
                raise ValueError(f"expected draws of shape (n_mc, {self.N})")
            shape = z.shape
        acc = np.zeros(self.N, dtype=np.int64)
        samples = np.empty(shape)
        check(self._lib.cude_mh_chain(self._h, shape[0], _ptr(z), _ptr(u), float(sigma), float(prior_mean),
                                      float(prior_sd), float(proposal_std), float(temperature), float(gamma),
                                      _ptr(acc), _ptr(samples)))
        return acc, samples

    def loss_grad(self, want_cond_grad=True):
        loss = C.c_double()
        g_nn = np.empty(self.P)
        g_cond = np.empty(self.N) if want_cond_grad else None
        check(self._lib.cude_loss_grad(self._h, C.byref(loss), _ptr(g_nn), _ptr(g_cond)))
        return loss.value, g_nn, g_cond

    def n_failed(self):
        n = C.c_int64()
        check(self._lib.cude_n_failed(self._h, C.byref(n)))
        return n.value

    # -- optimiser
    def adam_init(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        check(self._lib.cude_adam_init(self._h, lr, beta1, beta2, eps))

    def adam_step(self, want_loss=True):
        if want_loss:
            loss = C.c_double()
            check(self._lib.cude_adam_step(self._h, C.byref(loss)))
            return loss.value
        check(self._lib.cude_adam_step(self._h, None))
        return None

    def adam_run(self, n_iters):
        """n_iters fused optimiser iterations with one synchronisation (hipGraph replay); returns the loss
        before each update."""
        losses = np.empty(int(n_iters))
        check(self._lib.cude_adam_run(self._h, int(n_iters), _ptr(losses)))
        return losses

    # -- bring-your-own collective
    def set_global_subjects(self, n_global, scale=None):
        sc = None if scale is None else _f64(scale)
        check(self._lib.cude_set_global_subjects(self._h, float(n_global), _ptr(sc)))

    def get_scale(self):
        sc = np.empty(3)
        n = C.c_double()
        check(self._lib.cude_get_scale(self._h, _ptr(sc), C.byref(n)))
        return sc, n.value

    def loss_grad_partial(self, want_cond_grad=False):
        """This rank's un-reduced [g_nn; sum sse; n_failed] (and optionally dL/dcond of its subjects)."""
        part = np.empty(self.P + 2)
        g_cond = np.empty(self.N) if want_cond_grad else None
        check(self._lib.cude_loss_grad_partial(self._h, _ptr(part), _ptr(g_cond)))
        return part, g_cond

    def adam_apply(self, reduced):
        r = _f64(reduced)
        loss = C.c_double()
        check(self._lib.cude_adam_apply(self._h, _ptr(r), C.byref(loss)))
        return loss.value

    # -- the same exchange without the host round trip (a collective that reduces device memory in place)
    def partial_buffer(self):
        """(device address, count = P + 2) of the context's [g_nn; sum sse; n_failed] vector."""
        ptr, n = C.c_void_p(), C.c_int32()
        check(self._lib.cude_partial_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def partial_tensor(self, torch, device):
        """A torch tensor ALIASING that vector (no copy): torch.distributed.all_reduce on it reduces in place."""
        ptr, n = self.partial_buffer()

        class _Alias:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3,
                                        "strides": None}
        return torch.as_tensor(_Alias(), device=device)

    def loss_grad_partial_device(self):
        check(self._lib.cude_loss_grad_partial_device(self._h))

    def adam_apply_device(self):
        loss = C.c_double()
        check(self._lib.cude_adam_apply_device(self._h, C.byref(loss)))
        return loss.value

    def synchronize(self):
        check(self._lib.cude_synchronize(self._h))

    def adaptive_regroup(self):
        """Adaptive mode: order the launch by the accepted-step counts of the last gradient evaluation so that a wave's
        lanes finish together.  Returns (mean max - min steps within a wave before, after)."""
        b, a = C.c_int32(), C.c_int32()
        check(self._lib.cude_adaptive_regroup(self._h, C.byref(b), C.byref(a)))
        return b.value, a.value

    def grad_occupancy(self):
        """Resident waves per CU the runtime grants the one-lane gradient kernel of this context."""
        n = C.c_int32()
        check(self._lib.cude_grad_occupancy(self._h, C.byref(n)))
        return n.value

    def set_kernel_timing(self, enabled):
        """True / 1: HIP events around every ensemble launch; n > 1: around every n-th; False / 0: off."""
        check(self._lib.cude_set_kernel_timing(self._h, int(enabled)))

    def kernel_time_ms(self):
        ms = C.c_double()
        n = C.c_int64()
        check(self._lib.cude_kernel_time_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_stats(self):
        """(mean, median, minimum in ms, launches timed) of the dominant kernel since the last query (cude_kernel_time_stats)."""
        ms, med, mn = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int64()
        check(self._lib.cude_kernel_time_stats(self._h, C.byref(ms), C.byref(med), C.byref(mn), C.byref(n)))
        return ms.value, med.value, mn.value, n.value

    # -- communicator
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
        check(_lib.load().cude_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, n_ranks, rank, unique_id):
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        check(self._lib.cude_comm_init(self._h, n_ranks, rank, buf))

    def comm_info(self):
        """(ranks, rank, rccl version) as the attached communicator reports them; (1, 0, 0) without one."""
        n, r, v = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._lib.cude_comm_info(self._h, C.byref(n), C.byref(r), C.byref(v)))
        return n.value, r.value, v.value

    def allreduce_host(self, values):
        v = _f64(values).copy()
        check(self._lib.cude_comm_allreduce_host(self._h, _ptr(v), v.size))
        return v

    # -- the peer-write exchange (cude_xchg_*): multi-GPU without a collective library
    def xchg_export(self, n_ranks, rank):
        """This rank's mailbox as 128 bytes for its peers (cude_xchg_export)."""
        buf = (C.c_uint8 * _lib.XCHG_HANDLE_BYTES)()
        check(self._lib.cude_xchg_export(self._h, int(n_ranks), int(rank), buf))
        return bytes(buf)

    def xchg_attach(self, handles, timeout_s=20.0):
        """handles: the n_ranks exported handles in rank order (bytes each).  Collective: ends with a self-test."""
        blob = b"".join(handles)
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        check(self._lib.cude_xchg_attach(self._h, buf, float(timeout_s)))

    def xchg_detach(self):
        """Releases the exported / attached exchange; the next xchg_export offers the next kind of mailbox memory."""
        check(self._lib.cude_xchg_detach(self._h))

    def xchg_enable(self, enabled=True):
        check(self._lib.cude_xchg_enable(self._h, 1 if enabled else 0))

    def xchg_info(self):
        """(ranks, rank, memory kind of the mailbox: 3 uncached / 1 fine-grained / 0 plain, timed-out waits so far)."""
        n, r, k, t = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(self._lib.cude_xchg_info(self._h, C.byref(n), C.byref(r), C.byref(k), C.byref(t)))
        return n.value, r.value, k.value, t.value

    def set_option(self, name, value):
        """cude_set_option: run-time options of the context (include/cude.h lists them)."""
        check(self._lib.cude_set_option(self._h, str(name).encode(), str(value).encode()))


__all__ = ["Engine", "CudeError", "n_params", "device_count"]
