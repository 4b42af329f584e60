"""ctypes binding of oracle/libcude_oracle.so (TEST INFRASTRUCTURE ONLY; see cude_oracle.c)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _n_params(nin, width, depth):
    p, fan = 0, nin
    for _ in range(depth):
        p += width * fan + width
        fan = width
    return p + fan + 1


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcude_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = C.CDLL(path)
        _LIB.cude_oracle_num_threads.restype = C.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads():
    return lib().cude_oracle_num_threads()


def cpep(timepoints, glucose, cpeptide, age, t2dm, arch, nn, beta, n_steps, n_state=2,
         want_grad=True, want_traj=False, covariate=False, nthreads=0, method="forward"):
    """Returns dict(loss, sse, g_nn, g_beta, traj, n_failed).  method: "forward" = the reference's own AD
    (forward-mode duals, cude_oracle.c) or "reverse" = per-subject discrete adjoint (cude_oracle_rev.c, no traj)."""
    glucose = np.ascontiguousarray(glucose, dtype=np.float64)
    cpeptide = np.ascontiguousarray(cpeptide, dtype=np.float64)
    N, T = glucose.shape
    tp = np.ascontiguousarray(timepoints, dtype=np.float64)
    age = np.ascontiguousarray(age, dtype=np.float64)
    t2 = np.ascontiguousarray(t2dm, dtype=np.uint8)
    nn = np.ascontiguousarray(nn, dtype=np.float64)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    nin, width, depth = arch
    P = _n_params(*arch)
    assert nn.size == P and beta.size == N
    loss = C.c_double(0.0)
    sse = np.zeros(N)
    g_nn = np.zeros(P) if want_grad else None
    g_beta = np.zeros(N) if want_grad else None
    traj = np.zeros((N, T, n_state)) if want_traj else None
    if method == "reverse":
        assert not want_traj
        rc = lib().cude_oracle_cpep_rev(C.c_int(N), C.c_int(T), _p(tp), _p(glucose), _p(cpeptide), _p(age), _p(t2),
                                        C.c_int(int(covariate)), C.c_int(nin), C.c_int(width), C.c_int(depth),
                                        _p(nn), _p(beta), C.c_int(n_steps), C.c_int(n_state), C.c_int(int(want_grad)),
                                        C.c_int(nthreads), C.byref(loss), _p(sse), _p(g_nn), _p(g_beta))
        if rc < 0:
            raise ValueError("cude_oracle_cpep_rev: unsupported size")
        return dict(loss=loss.value, sse=sse, g_nn=g_nn, g_beta=g_beta, traj=None, n_failed=rc)
    rc = lib().cude_oracle_cpep(C.c_int(N), C.c_int(T), _p(tp), _p(glucose), _p(cpeptide), _p(age), _p(t2),
                                C.c_int(int(covariate)), C.c_int(nin), C.c_int(width), C.c_int(depth),
                                _p(nn), _p(beta), C.c_int(n_steps), C.c_int(n_state), C.c_int(int(want_grad)),
                                C.c_int(nthreads), C.byref(loss), _p(sse), _p(g_nn), _p(g_beta), _p(traj))
    if rc < 0:
        raise ValueError("cude_oracle_cpep: unsupported size")
    return dict(loss=loss.value, sse=sse, g_nn=g_nn, g_beta=g_beta, traj=traj, n_failed=rc)


def cpep_adaptive(timepoints, glucose, cpeptide, age, t2dm, arch, nn, cond, out_times, covariate=False,
                  abstol=1e-6, reltol=1e-3, nthreads=0):
    """Plasma c-peptide of N subjects at `out_times` (N x n_out) integrated with the adaptive restatement
    (cude_oracle.solve_adaptive in C).  cond = exp(beta), or k for the symbolic model (arch width 0).  Failed
    subjects give NaN rows."""
    glucose = np.ascontiguousarray(glucose, dtype=np.float64)
    cpeptide = np.ascontiguousarray(cpeptide, dtype=np.float64)
    N, T = glucose.shape
    tp = np.ascontiguousarray(timepoints, dtype=np.float64)
    age = np.ascontiguousarray(age, dtype=np.float64)
    t2 = np.ascontiguousarray(t2dm, dtype=np.uint8)
    nn = np.ascontiguousarray(nn, dtype=np.float64)
    cond = np.ascontiguousarray(cond, dtype=np.float64)
    tout = np.ascontiguousarray(out_times, dtype=np.float64)
    nin, width, depth = arch
    assert cond.size == N and age.size == N and nn.size == (1 if width == 0 else _n_params(*arch))
    out = np.zeros((N, tout.size))
    nthreads = nthreads or default_threads()
    rc = lib().cude_oracle_cpep_adaptive(C.c_int(N), C.c_int(T), _p(tp), _p(glucose), _p(cpeptide), _p(age), _p(t2),
                                         C.c_int(int(covariate)), C.c_int(nin), C.c_int(width), C.c_int(depth),
                                         _p(nn), _p(cond), C.c_int(tout.size), _p(tout), C.c_double(abstol),
                                         C.c_double(reltol), C.c_int(nthreads), _p(out))
    if rc < 0:
        raise ValueError("cude_oracle_cpep_adaptive: unsupported size")
    return out


def supp(timepoints, data, arch, nn, theta, lam, n_steps, want_grad=True, want_traj=False, nthreads=0,
         method="forward"):
    """data: 3 x T x N numpy array (any layout; converted to Julia column-major).  method as in cpep."""
    data = np.asarray(data, dtype=np.float64)
    _, T, N = data.shape
    dcol = np.ascontiguousarray(data.transpose(2, 1, 0))      # memory: i slowest, s fastest
    tp = np.ascontiguousarray(timepoints, dtype=np.float64)
    nn = np.ascontiguousarray(nn, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    nin, width, depth = arch
    assert nin == 4
    P = _n_params(*arch)
    assert nn.size == P and theta.size == N
    loss = C.c_double(0.0)
    sse = np.zeros(N)
    g_nn = np.zeros(P) if want_grad else None
    g_th = np.zeros(N) if want_grad else None
    traj = np.zeros((N, T, 3)) if want_traj else None
    if method == "reverse":
        assert not want_traj
        rc = lib().cude_oracle_supp_rev(C.c_int(N), C.c_int(T), _p(tp), _p(dcol), C.c_int(width), C.c_int(depth),
                                        _p(nn), _p(theta), C.c_double(lam), C.c_int(n_steps), C.c_int(int(want_grad)),
                                        C.c_int(nthreads), C.byref(loss), _p(sse), _p(g_nn), _p(g_th))
        if rc < 0:
            raise ValueError("cude_oracle_supp_rev: unsupported size")
        return dict(loss=loss.value, sse=sse, g_nn=g_nn, g_theta=g_th, traj=None, n_failed=rc)
    rc = lib().cude_oracle_supp(C.c_int(N), C.c_int(T), _p(tp), _p(dcol), C.c_int(width), C.c_int(depth),
                                _p(nn), _p(theta), C.c_double(lam), C.c_int(n_steps), C.c_int(int(want_grad)),
                                C.c_int(nthreads), C.byref(loss), _p(sse), _p(g_nn), _p(g_th), _p(traj))
    if rc < 0:
        raise ValueError("cude_oracle_supp: unsupported size")
    if traj is not None:
        traj = traj.transpose(2, 1, 0)                         # -> 3 x T x N
    return dict(loss=loss.value, sse=sse, g_nn=g_nn, g_theta=g_th, traj=traj, n_failed=rc)


def default_threads():
    """Threads for the many small parallel regions of the parameter scans: the container's CPU quota when one is set
    (a GPU box shows 256 logical CPUs to a 16-core share: 256 OpenMP threads for 37 subjects turn a 0.1 ms region into
    seconds), never more than 16."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return max(1, min(n, 16))


def supp_adaptive(timepoints, data, arch, nn, theta, abstol=1e-6, reltol=1e-3, nthreads=0):
    """Trajectories (3 x T x N) of the suppression model integrated with the adaptive restatement
    (cude_oracle.solve_adaptive + supp_rhs in C): u0 = data[:, 0, :], outputs at `timepoints`.  A failed subject's
    slice is NaN.  The loss is formed by the caller (cude_oracle.supp_scale etc.)."""
    data = np.asarray(data, dtype=np.float64)
    _, T, N = data.shape
    dcol = np.ascontiguousarray(data.transpose(2, 1, 0))      # memory: i slowest, s fastest
    tp = np.ascontiguousarray(timepoints, dtype=np.float64)
    nn = np.ascontiguousarray(nn, dtype=np.float64)
    eth = np.ascontiguousarray(np.exp(np.asarray(theta, dtype=np.float64)))
    nin, width, depth = arch
    assert nin == 4 and nn.size == _n_params(*arch) and eth.size == N
    out = np.zeros((N, T, 3))
    nthreads = nthreads or default_threads()
    rc = lib().cude_oracle_supp_adaptive(C.c_int(N), C.c_int(T), _p(tp), _p(dcol), C.c_int(width), C.c_int(depth),
                                         _p(nn), _p(eth), C.c_double(abstol), C.c_double(reltol), C.c_int(nthreads),
                                         _p(out))
    if rc < 0:
        raise ValueError("cude_oracle_supp_adaptive: unsupported size")
    return np.ascontiguousarray(out.transpose(2, 1, 0))


def supp_adaptive_loss(timepoints, data, arch, nn, theta, lam, abstol=1e-6, reltol=1e-3):
    """suppression_loss (suppression/src/suppression_model.jl:117-130) on the adaptive trajectories: +Inf when a solve
    failed."""
    data = np.asarray(data, dtype=np.float64)
    sims = supp_adaptive(timepoints, data, arch, nn, theta, abstol, reltol)
    if not np.all(np.isfinite(sims)):
        return np.inf
    scale = data.max(axis=1).mean(axis=1)
    r = (sims - data) / scale[:, None, None]
    return float(np.sum(r * r)) / data.shape[2] + lam * float(np.sum(np.asarray(nn) ** 2))
