// libcude_hip.so -- the optimisers of the reference's training loop on the device-resident state: Optimisers.Adam as
// single steps and as captured runs (src/parameter-estimation.jl:176, suppression_model.jl:164, saem.jl:128) and Optim's
// L-BFGS + BackTracking (:179-180, :168) on a caller's objective; the restarts trained side by side: cude_train.hip.
#include "cude_ctx.h"

namespace cude {
namespace api {

void drop_graph(cude_ctx* c) {
    for (int u = 0; u < cude_ctx::kGraphKinds; u++) {
        if (c->graph_exec[u]) { (void)hipGraphExecDestroy(c->graph_exec[u]); c->graph_exec[u] = nullptr; }
        if (c->graph[u]) { (void)hipGraphDestroy(c->graph[u]); c->graph[u] = nullptr; }
    }
}

int32_t ensure_trace(cude_ctx* c, int64_t n) {
    if (n <= c->trace_cap) return CUDE_OK;
    drop_graph(c);                              // the captured kernels hold the old trace pointer
    int64_t cap = std::max<int64_t>(n, 4096);
    HIP_TRY(c->adam_trace.resize((size_t)cap * 2));
    c->trace_cap = cap;
    return CUDE_OK;
}

cude::TailAdvance tail_advance(cude_ctx* c) {
    cude::TailAdvance t;
    t.state = c->adam_state.p; t.b1 = c->b1; t.b2 = c->b2; t.trace = c->adam_trace.p; t.cap = c->trace_cap;
    return t;
}

// queues the Adam update (+ state advance / loss trace unless run_ensemble folded it) behind the gradient already on the stream
int32_t enqueue_adam(cude_ctx* c) {
    if (!c->advance_done) HIP_TRY(cude::launch_adam_advance(tail_advance(c), c->g_nn.p + c->P, c->stream));
    c->advance_done = false;
    cude::AdamArgs a{};
    a.N = c->N; a.P = c->P;
    a.cond = c->cond.p; a.m_cond = c->m_cond.p; a.v_cond = c->v_cond.p; a.g_cond = c->g_cond.p;
    a.nn = c->nn.p; a.m_nn = c->m_nn.p; a.v_nn = c->v_nn.p; a.g_nn = c->g_nn.p;
    a.lr = c->lr; a.b1 = c->b1; a.b2 = c->b2; a.eps = c->eps;
    a.state = c->adam_state.p;
    HIP_TRY(cude::launch_adam(a, c->stream));
    return CUDE_OK;
}

}  // namespace api
}  // namespace cude

using namespace cude::api;

extern "C" {

int32_t cude_adam_init(cude_ctx* c, double lr, double beta1, double beta2, double eps) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!(lr > 0) || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1) || !(eps > 0))
        return fail(CUDE_ERR_ARG, "bad Adam hyper-parameters");
    c->lr = lr; c->b1 = beta1; c->b2 = beta2; c->eps = eps;
    c->adam_t = 0;
    drop_graph(c);                              // hyper-parameters are baked into the captured launches
    HIP_TRY(c->adam_state.resize(4));
    if ((rc = ensure_trace(c, 1))) return rc;
    const double st0[4] = {1.0, 1.0, 0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(c->adam_state.p, st0, sizeof(st0), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // st0 is a stack buffer
    HIP_TRY(hipMemsetAsync(c->m_nn.p, 0, c->P * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->v_nn.p, 0, c->P * sizeof(double), c->stream));
    if (c->have_pop) {
        HIP_TRY(hipMemsetAsync(c->m_cond.p, 0, c->N * sizeof(double), c->stream));
        HIP_TRY(hipMemsetAsync(c->v_cond.p, 0, c->N * sizeof(double), c->stream));
    }
    c->adam_ready = true;
    return CUDE_OK;
}

int32_t cude_adam_step(cude_ctx* c, double* loss) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->adam_ready) return fail(CUDE_ERR_STATE, "call cude_adam_init first");
    c->fold_advance = true;
    c->allow_watch = loss != nullptr;
    rc = run_ensemble(c, true, nullptr);
    c->fold_advance = false;
    if (rc) { c->advance_done = false; return rc; }
    c->adam_t += 1;
    // the update is queued BEFORE the host waits for the loss: the update kernel only reads g_nn (where the loss sum
    // and the failure count live), so the value read back is still the loss of the iterate the gradient was taken
    // at, and the GPU is not left idle while the host turns around
    if ((rc = enqueue_adam(c))) return rc;
    return loss ? finish_loss(c, loss, nullptr) : CUDE_OK;
}

// n_iters optimiser iterations without any host round trip: one iteration (gradient kernels, reductions, L2
// term, Adam, state advance) is captured once into a hipGraph and replayed; the per-iteration losses are
// appended to a device trace and copied back after a single synchronisation.  The peer-write exchange lives inside
// the reduction kernels and is captured with them; with an RCCL communicator as the transport the iterations are
// queued as plain launches (RCCL calls are not captured).
namespace {

// queues `n` iterations behind whatever is on the stream (no synchronisation)
int32_t queue_iterations(cude_ctx* c, int32_t n) {
    int32_t rc = CUDE_OK;
    const bool rccl = c->comm != nullptr && !c->xchg.ready;
    const bool use_graph = !rccl && !c->timing && c->opt.graph;
    const int unroll = std::max(1, c->opt.graph_unroll);
    int u_max = 0;
    while (u_max + 1 < cude_ctx::kGraphKinds && (2 << u_max) <= unroll) u_max++;
    for (int u = 0; u <= u_max && use_graph; u++) {
        const int reps = 1 << u;
        // needed by this run: the largest kind as often as it fits, the smaller ones by the bits of the remainder
        const bool needed = u == u_max ? n >= reps : (((n % (1 << u_max)) >> u) & 1) != 0;
        if (c->graph_exec[u] || !needed) continue;
        c->capturing = true;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            c->capturing = false;
            return fail(CUDE_ERR_HIP, hipGetErrorString(e));
        }
        for (int r = 0; r < reps && !rc; r++) {
            c->fold_advance = true;
            rc = run_ensemble(c, true, nullptr);
            c->fold_advance = false;
            if (!rc) rc = enqueue_adam(c);
            c->advance_done = false;
        }
        hipError_t e2 = hipStreamEndCapture(c->stream, &c->graph[u]);
        if (!rc && e2 == hipSuccess) e2 = hipGraphInstantiate(&c->graph_exec[u], c->graph[u], nullptr, nullptr, 0);
        c->capturing = false;
        if (rc || e2 != hipSuccess) {
            drop_graph(c);
            if (rc) return rc;
            return fail(CUDE_ERR_HIP, hipGetErrorString(e2));
        }
    }
    for (int k = 0; k < n;) {
        if (use_graph) {
            int u = u_max;
            while (u > 0 && (n - k) < (1 << u)) u--;
            HIP_TRY(hipGraphLaunch(c->graph_exec[u], c->stream));
            k += 1 << u;
        } else {
            c->fold_advance = true;
            rc = run_ensemble(c, true, nullptr);
            c->fold_advance = false;
            if (rc) { c->advance_done = false; return rc; }
            if ((rc = enqueue_adam(c))) return rc;
            k++;
        }
    }
    if (adaptive(c)) c->have_tape = c->have_counts = true;     // (a replayed graph writes the tape as a plain launch does)
    return CUDE_OK;
}

}  // namespace

int32_t cude_adam_run(cude_ctx* c, int32_t n_iters, double* losses) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->adam_ready) return fail(CUDE_ERR_STATE, "call cude_adam_init first");
    if (n_iters < 1) return fail(CUDE_ERR_ARG, "n_iters must be >= 1");
    if (!c->have_pop || !c->have_nn || !c->have_cond) return fail(CUDE_ERR_STATE, "population / parameters not set");
    if ((rc = ensure_trace(c, n_iters))) return rc;
    if ((rc = ensure_tape(c))) return rc;
    HIP_TRY(hipMemsetAsync(c->adam_state.p + 3, 0, sizeof(double), c->stream));     // trace position = 0
    // Large adaptive populations: the launch is kept ordered by accepted-step count (cude_adaptive_regroup) -- after the
    // FIRST iteration this entry point has run on the population (its evaluation tells the counts), then after every
    // kRegroupEvery-th.  The schedule counts iterations of cude_adam_run since the population was set, so it does not
    // depend on how a caller cuts its run into calls (the order of the lanes is the summation order of the shared
    // gradient: adam_run(400) and 2 x adam_run(200) give the same bits).  Costs one read-back of N counters and a host
    // sort, ~10 ms at 1e5 subjects.  Option "auto_regroup" = 0 leaves the order to the caller.
    constexpr int64_t kRegroupEvery = 200;
    const bool regroup = adaptive(c) && c->N >= 8192 && c->opt.auto_regroup;
    auto boundary = [&](int64_t k) { return k == 1 || (k > 0 && k % kRegroupEvery == 0); };
    for (int32_t done = 0; done < n_iters;) {
        if (regroup && boundary(c->run_iters) && c->regroup_done_at != c->run_iters && c->have_counts) {
            if ((rc = adaptive_regroup(c, nullptr, nullptr))) return rc;
            c->regroup_done_at = c->run_iters;
        }
        int32_t seg = n_iters - done;
        if (regroup) {
            const int64_t next = c->run_iters == 0 ? 1 : (c->run_iters / kRegroupEvery + 1) * kRegroupEvery;
            seg = (int32_t)std::min<int64_t>(seg, next - c->run_iters);
        }
        if ((rc = queue_iterations(c, seg))) return rc;
        done += seg;
        c->run_iters += seg;
    }
    c->adam_t += n_iters;
    std::vector<double> tr((size_t)n_iters * 2);
    HIP_TRY(hipMemcpyAsync(tr.data(), c->adam_trace.p, tr.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->xchg.ready && (rc = xchg_check(c))) return rc;     // (a wait that gave up in any column of any iteration)
    c->last_failed = (int64_t)std::llround(tr[(size_t)(n_iters - 1) * 2 + 1]);
    if (losses)
        for (int k = 0; k < n_iters; k++)
            losses[k] = (tr[2 * k + 1] > 0.0 || !std::isfinite(tr[2 * k])) ? std::numeric_limits<double>::infinity()
                                                                         : tr[2 * k] / c->n_global;
    return CUDE_OK;
}

int32_t cude_adam_apply_device(cude_ctx* c, double* loss) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->adam_ready) return fail(CUDE_ERR_STATE, "call cude_adam_init first");
    const int P = c->P;
    if (c->cfg.lambda != 0.0)
        HIP_TRY(cude::launch_l2_term(c->nn.p, P, c->cfg.lambda, c->n_global, c->g_nn.p, c->stream, c->param_mask.p));
    c->adam_t += 1;
    if ((rc = finish_loss(c, loss, nullptr))) return rc;
    return enqueue_adam(c);
}

int32_t cude_adam_apply(cude_ctx* c, const double* reduced, double* loss) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->adam_ready) return fail(CUDE_ERR_STATE, "call cude_adam_init first");
    if (!reduced) return fail(CUDE_ERR_ARG, "null input");
    const int P = c->P;
    HIP_TRY(hipMemcpyAsync(c->g_nn.p, reduced, (P + 2) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (c->cfg.lambda != 0.0)
        HIP_TRY(cude::launch_l2_term(c->nn.p, P, c->cfg.lambda, c->n_global, c->g_nn.p, c->stream, c->param_mask.p));
    c->adam_t += 1;
    if ((rc = finish_loss(c, loss, nullptr))) return rc;   // also synchronises: `reduced` may be freed after return
    return enqueue_adam(c);
}

int32_t cude_lbfgs_minimize(int32_t n, const double* x0, int32_t maxiters, cude_objective_fn fn, void* user,
                            double* x_out, double* f_out, int32_t* iterations, int32_t* f_calls, int32_t* converged) {
    if (n < 1 || !x0 || !fn || !x_out || maxiters < 0) return fail(CUDE_ERR_ARG, "bad argument");
    cude::Lbfgs opt(x0, n, maxiters);
    std::vector<double> g(n);
    while (const double* x = opt.pending()) {
        double f = std::numeric_limits<double>::quiet_NaN();
        const int32_t rc = fn(x, n, &f, g.data(), user);
        if (rc < 0) return fail(CUDE_ERR_ARG, "objective callback reported an error");
        opt.feed(f, g.data());
    }
    const cude::Lbfgs::Result r = opt.result();
    std::copy(opt.x().begin(), opt.x().end(), x_out);
    if (f_out) *f_out = r.f;
    if (iterations) *iterations = r.iterations;
    if (f_calls) *f_calls = r.f_calls;
    if (converged) *converged = r.converged ? 1 : 0;
    return CUDE_OK;
}

int32_t cude_lbfgs_minimize_sharded(int32_t n, int32_t n_shared, const double* x0, int32_t maxiters, cude_objective_fn fn,
                                    cude_reduce_fn reduce, void* user, double* x_out, double* f_out,
                                    int32_t* iterations, int32_t* f_calls, int32_t* converged) {
    if (n < 1 || n_shared < 0 || n_shared > n || !x0 || !fn || !reduce || !x_out || maxiters < 0)
        return fail(CUDE_ERR_ARG, "bad argument");
    cude::Lbfgs opt(x0, n, maxiters, 10, 1e-8, n_shared, reduce, user);
    std::vector<double> g(n);
    while (const double* x = opt.pending()) {
        double f = std::numeric_limits<double>::quiet_NaN();
        const int32_t rc = fn(x, n, &f, g.data(), user);
        if (rc < 0) return fail(CUDE_ERR_ARG, "objective callback reported an error");
        opt.feed(f, g.data());
        if (opt.comm_failed()) return fail(CUDE_ERR_COMM, "reduce callback reported an error");
    }
    const cude::Lbfgs::Result r = opt.result();
    std::copy(opt.x().begin(), opt.x().end(), x_out);
    if (f_out) *f_out = r.f;
    if (iterations) *iterations = r.iterations;
    if (f_calls) *f_calls = r.f_calls;
    if (converged) *converged = r.converged ? 1 : 0;
    return CUDE_OK;
}

}  // extern "C"
