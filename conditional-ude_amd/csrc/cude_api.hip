// libcude_hip.so -- host side of the C ABI declared in include/cude.h.
// Owns the device-resident population (subject-major SoA), the solver tables, the HIP stream,
// and (optionally) an RCCL communicator for the one all-reduce per optimiser step.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/cude.h"
#include "cude_kernels.h"
#include "cude_optim.h"

namespace {

thread_local std::string g_err;

int32_t fail(int32_t code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(CUDE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

// ---------------------------------------------------------------------------------- solver tables
const double TA7[6] = {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
                       2.324710524099774};
const double TC[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
const double TR[7][4] = {{1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216},
                         {0.0, 0.13169999999999998, -0.2234, 0.1017},
                         {0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253},
                         {0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902},
                         {0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928},
                         {0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661},
                         {0.0, 1.5, -4.0, 2.5}};

void interp_weights(double th, double* w) {
    if (std::fabs(th - 1.0) < 1e-12) {
        for (int j = 0; j < 6; j++) w[j] = TA7[j];
        w[6] = 0.0;
        return;
    }
    for (int i = 0; i < 7; i++) w[i] = ((TR[i][3] * th + TR[i][2]) * th + TR[i][1]) * th * th + TR[i][0] * th;
}

// observation tau lies in step n with t_n < tau <= t_{n+1}  (tau = t_0 -> step 0, theta 0)
void locate_obs(const std::vector<double>& tp, int S, std::vector<int32_t>& step, std::vector<double>& w) {
    const int T = (int)tp.size();
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    step.resize(T);
    w.resize((size_t)T * 7);
    for (int i = 0; i < T; i++) {
        const double x = (tp[i] - t0) / h;
        int n = (int)std::ceil(x - 1e-9) - 1;
        if (n < 0) n = 0;
        if (n > S - 1) n = S - 1;
        step[i] = n;
        interp_weights((tp[i] - (t0 + n * h)) / h, &w[(size_t)i * 7]);
    }
}

// glucose segment + fraction for the 5 distinct stage times of every step (c2..c5 and 1)
void glucose_tables(const std::vector<double>& tp, int S, std::vector<int32_t>& seg, std::vector<double>& phi) {
    const int T = (int)tp.size();
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    seg.resize((size_t)S * 5);
    phi.resize((size_t)S * 5);
    for (int n = 0; n < S; n++) {
        for (int s = 0; s < 5; s++) {
            const double t = (s < 4) ? (t0 + n * h) + TC[s + 1] * h : t0 + (n + 1) * h;
            int j = 0;
            while (j + 1 < T && tp[j + 1] <= t) j++;
            if (j > T - 2) j = T - 2;
            seg[(size_t)n * 5 + s] = j;
            phi[(size_t)n * 5 + s] = (t - tp[j]) / (tp[j + 1] - tp[j]);
        }
    }
}

// Per-step tables of the layer-1 exponent recurrence (CpepArgs::stepk / stepd): a step is "inside" glucose piece j
// when [t_n, t_n+h] lies within [tp[j], tp[j+1]]; consecutive inside steps of one piece form a run.
void step_tables(const std::vector<double>& tp, int S, std::vector<int32_t>& k, std::vector<double>& d) {
    const int T = (int)tp.size();
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S, tol = 1e-9 * h;
    k.assign((size_t)S * 3, 0);
    d.assign((size_t)S * 3, 0.0);
    std::vector<int> piece(S, -1);
    for (int n = 0; n < S; n++) {
        const double ta = t0 + n * h, tb = t0 + (n + 1) * h;
        int j = 0;
        while (j + 1 < T - 1 && tp[j + 1] <= ta + tol) j++;
        const double len = tp[j + 1] - tp[j];
        if (ta >= tp[j] - tol && tb <= tp[j + 1] + tol) piece[n] = j;
        k[(size_t)n * 3 + 2] = j;
        d[(size_t)n * 3 + 0] = (ta - tp[j]) / len;
        d[(size_t)n * 3 + 1] = (tb - tp[j]) / len;
        d[(size_t)n * 3 + 2] = h / len;
    }
    // a run is re-anchored with fresh exponentials every kReanchor steps so that the rounding of the anchor
    // recurrence (one multiply per step) stays below 256 ulp however many steps a piece holds
    constexpr int kReanchor = 256;
    for (int n = 0, pos = 0; n < S; n++) {
        if (piece[n] < 0) { pos = 0; continue; }
        const bool cont = n > 0 && piece[n - 1] == piece[n] && pos + 1 < kReanchor;
        k[(size_t)n * 3 + 0] = cont ? 2 : 1;
        pos = cont ? pos + 1 : 0;
    }
    for (int n = S - 1, pos = 0; n >= 0; n--) {
        if (piece[n] < 0) { pos = 0; continue; }
        const bool cont = n + 1 < S && piece[n + 1] == piece[n] && pos + 1 < kReanchor;
        k[(size_t)n * 3 + 1] = cont ? 2 : 1;
        pos = cont ? pos + 1 : 0;
    }
}

// ---------------------------------------------------------------------------------- RCCL (dlopen)
typedef struct { char internal[CUDE_UNIQUE_ID_BYTES]; } nccl_uid;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(void**, int, nccl_uid, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    const char* (*GetLastError)(void*) = nullptr;      // NCCL >= 2.13: the library's own description of what went wrong
};
Rccl g_rccl;

int32_t load_rccl() {
    if (g_rccl.handle) return CUDE_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {   // reuse a copy already in the process (e.g. PyTorch's) first
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    if (!h)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
    if (!h) return fail(CUDE_ERR_COMM, std::string("cannot load librccl: ") + dlerror());
    g_rccl.GetUniqueId = (int (*)(nccl_uid*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, nccl_uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce =
        (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    g_rccl.GetVersion = (int (*)(int*))dlsym(h, "ncclGetVersion");
    g_rccl.GetLastError = (const char* (*)(void*))dlsym(h, "ncclGetLastError");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(CUDE_ERR_COMM, "librccl lacks a required symbol");
    g_rccl.handle = h;
    return CUDE_OK;
}

inline std::string rccl_diagnosis(int r) {
    std::string m = g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error";
    if (g_rccl.GetLastError) {
        const char* last = g_rccl.GetLastError(nullptr);
        if (last && last[0]) m += std::string(" [") + last + "]";
    }
    return m;
}
#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        int _r = (expr);                                                                           \
        if (_r != 0) return fail(CUDE_ERR_COMM, std::string(#expr) + ": " + rccl_diagnosis(_r));   \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t resize(size_t count) {
        if (count == n) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    // grow-only variant for scratch that is reused across calls
    hipError_t reserve(size_t count) { return count <= n ? hipSuccess : resize(count); }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

// bit pattern the host puts into every slot of cude_ctx::pinned_pairs before a launch it is going to watch: a quiet NaN
// with a payload no arithmetic produces
constexpr uint64_t kPairSentinel = 0x7ff8dead5eed0001ull;

struct cude_ctx {
    cude_config cfg;
    cude::NetShape net;
    int P = 0;
    hipStream_t stream = nullptr;
    int64_t N = 0;          // local subjects
    double n_global = 0;    // subjects over all ranks
    int T = 0;
    std::vector<double> tp;
    bool have_pop = false, have_nn = false, have_cond = false;
    // population (CPEP)
    DevBuf<double> k0, k1, k2, c0, dG, obs, age;
    // population (SUPP)
    DevBuf<double> data, ckpt;
    double scale[3] = {1, 1, 1};
    // tables
    DevBuf<int32_t> seg, obs_step, stepk;
    DevBuf<double> phi, obs_w, stepd, tp_dev;
    double abstol = 1e-6, reltol = 1e-3;   // adaptive mode (n_steps == 0): OrdinaryDiffEq's defaults
    // parameters / gradients / optimiser
    DevBuf<double> nn, cond, g_nn, g_cond, sse, auc, partials, traj;
    // chunked gradient path (cude_cpep2.hip)
    int chunks = 1;
    int64_t blk0 = 0;       // > 0: mixed gradient launch -- blocks [0, blk0) on the one-lane kernel, the rest time-split
    int64_t slots_one = 0, half_slots = 0;          // resident-wave slots of the one-lane gradient kernel; one per SIMD
    hipStream_t stream2 = nullptr;                  // mixed launch: the time-split remainder runs beside the whole rounds
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf<double> param_mask;                      // frozen shared parameters (cude_set_param_mask); empty = none
    std::vector<double> mask_host;
    DevBuf<int32_t> chunk_start;
    DevBuf<double> hom_M, hom_obs, fsum, res, g_cond_part, partials2;
    // forward-only launches of the time-split path have their own split (chunks_f, 0 = the gradient's): the scan's cost
    // grows with the chunk count and there is no reverse kernel to feed, so fewer, longer chunks win there
    int chunks_f = 0;
    DevBuf<int32_t> chunk_start_f;
    DevBuf<double> hom_M_f, hom_obs_f, fsum_f;
    DevBuf<double> m_nn, v_nn, m_cond, v_cond;
    int64_t nblocks = 0;
    double lr = 1e-3, b1 = 0.9, b2 = 0.999, eps = 1e-8;
    int64_t adam_t = 0;
    bool adam_ready = false;
    DevBuf<double> adam_state, adam_trace;   // device-resident step state and per-iteration loss trace
    int64_t trace_cap = 0;
    // cude_adam_run: captured optimiser iterations -- graph [u] holds 2^u of them back to back (kernels of one graph
    // follow each other without a gap; between two graph launches the GPU idles ~8 us, tools/step_gaps.py); a run of
    // n iterations is its binary decomposition, largest graphs first (8 at most: CUDE_GRAPH_UNROLL)
    static constexpr int kGraphKinds = 4;
    hipGraph_t graph[kGraphKinds] = {nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t graph_exec[kGraphKinds] = {nullptr, nullptr, nullptr, nullptr};
    bool capturing = false;
    int32_t timing_period = 1;      // kernel timing: events around every timing_period-th ensemble launch
    int64_t timing_count = 0;
    // Adam state advance (running powers, step counter, loss trace): folded into the kernel that finishes an iteration's
    // [sum loss, n_failed] when the iteration is run by cude_adam_step / cude_adam_run (fold_advance), otherwise -- and
    // with a communicator but no L2 term, where the pair is final only after the all-reduce -- its own launch
    bool fold_advance = false, advance_done = false;
    int64_t last_failed = 0;
    // comm
    void* comm = nullptr;
    int n_ranks = 1, rank = 0;
    // timing of the dominant kernel
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t rng_seed = 0x243F6A8885A308D3ull;      // device-side draws of the Metropolis steps (cude_set_rng)
    int64_t rng_offset = 0, rng_step = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double host_red[3];
    double* pinned = nullptr;       // page-locked staging of the small result vectors ([g_nn; loss sum; n_failed])
    // set by a caller right before run_ensemble when finish_loss(loss, nullptr) follows at once and is the call's ONLY
    // pending output: then the result slots in page-locked memory may be watched instead of waiting for the stream (a
    // launch whose result nobody fetches must not write there: a later call's watch would take it for its own)
    bool allow_watch = false;
    bool poll_pairs = false;
    bool tail_in_pinned = false;    // the tail reduction of the last launch also wrote [sum loss, n_failed] to pinned[P..P+1]
    double* pinned_pairs = nullptr; // page-locked [nblocks][2]: per-workgroup (sum SSE, failures) of a forward-only launch,
    int64_t pinned_pairs_n = 0;     // written by the scan kernel itself and added up by the host (finish_loss)
    bool loss_in_pinned = false;    // the last forward launch left its result there
    // scratch of cude_multistart_loss_grad (kept between calls: it is called once per optimiser iteration)
    DevBuf<double> ms_nn, ms_cond, ms_part, ms_out, ms_gcond, ms_ckpt, ms_act;
    DevBuf<double> ms_fsum, ms_wts, ms_gcp, ms_p2;   // time-split path with parameter sets (small populations)
    DevBuf<double> act;     // SUPP: kept network activations of the gradient launch (small populations only)
    DevBuf<double> tape, ms_tape;   // adaptive mode: accepted steps of the forward sweep, walked back by the adjoint
    DevBuf<int32_t> tape_n;
    DevBuf<int32_t> perm;                           // adaptive kernels: subject of every launch position (cude_adaptive_regroup)
    std::vector<int32_t> slot_of;                   // its inverse on the host (empty = identity)
    int64_t regroup_age = 0;                        // optimiser iterations since the launch order was last rebuilt
    int tape_cap = 0;
    bool have_tape = false;
    DevBuf<double> red_tmp; // staging of small host vectors reduced through the communicator
    std::vector<double> ms_host;
#ifdef CUDE_WAVE_TIMING
    DevBuf<long long> dbg;
#endif
};

namespace {

// SUPP gradient launches keep the network activations of the forward sweep (instead of recomputing them in the
// reverse sweep) when that buffer is small: the launch is then latency-bound and 2/3 of the reverse sweep's
// instructions are worth 8*(D*W+1) bytes per evaluation; at 1e5 subjects the 2.3 GB each way would cost more than the
// recomputation.  CUDE_SUPP_STORE=0/1 overrides the size rule (A/B runs).
size_t supp_act_doubles(const cude_ctx* c) {
    return (size_t)(6 * c->cfg.n_steps + 1) * (size_t)(c->net.depth * c->net.width + 1) * (size_t)c->N;
}
bool supp_keep_activations(const cude_ctx* c, int64_t n_sets) {
    const char* env = getenv("CUDE_SUPP_STORE");
    if (env && (env[0] == '0' || env[0] == '1')) return env[0] == '1';
    return (double)n_sets * (double)supp_act_doubles(c) * 8.0 <= 256e6;
}

// c-peptide gradient launches (one lane per subject, single parameter set): CUDE_CPEP_KEEP=1 keeps the upper layers'
// activations of the forward sweep in HBM for the reverse sweep (CpepArgs::act; measured slower at the benchmark sizes,
// see cpep_kernel -- off unless asked for)
#ifndef CUDE_CPEP_KEEP_DEFAULT
#define CUDE_CPEP_KEEP_DEFAULT 0
#endif
size_t cpep_act_doubles(const cude_ctx* c) {
    const int nk = cude::cpep_keep_values(c->net);
    if (nk == 0 || c->cfg.n_steps == 0) return 0;
    const size_t nblocks = (size_t)((c->N + cude::kBlock - 1) / cude::kBlock);
    return (size_t)(5 * c->cfg.n_steps + 1) * (size_t)nk * nblocks * cude::kBlock;
}
// CUDE_CPEP_KEEP = 0 (recompute everything) | 1 (keep the output unit's logistic derivative) | 2 (keep the upper layers)
int cpep_keep_mode(const cude_ctx* c) {
    const char* env = getenv("CUDE_CPEP_KEEP");
    const int m = env ? atoi(env) : CUDE_CPEP_KEEP_DEFAULT;
    return (m == 1 || m == 2) && cpep_act_doubles(c) > 0 ? m : 0;
}
bool cpep_keep_activations(const cude_ctx* c) { return cpep_keep_mode(c) != 0; }

// both c-peptide models share the population layout, solver tables and the ensemble kernel
bool is_cpep(const cude_ctx* c) { return c->cfg.model == CUDE_MODEL_CPEP || c->cfg.model == CUDE_MODEL_CPEP_SYM; }

bool adaptive(const cude_ctx* c) { return c->cfg.n_steps == 0; }
double step_size(const cude_ctx* c) { return adaptive(c) ? 0.0 : (c->tp.back() - c->tp.front()) / c->cfg.n_steps; }

// population, tables and solver settings of a c-peptide launch; the caller adds parameters and outputs
cude::CpepArgs cpep_args(const cude_ctx* c) {
    cude::CpepArgs a{};
    a.cond_raw = c->cfg.cond_space == CUDE_COND_RAW;
    a.N = c->N;
    a.k0 = c->k0.p; a.k1 = c->k1.p; a.k2 = c->k2.p; a.c0 = c->c0.p;
    a.dG = c->dG.p; a.obs = c->obs.p; a.age = c->age.p;
    a.seg = c->seg.p; a.phi = c->phi.p; a.obs_step = c->obs_step.p; a.obs_w = c->obs_w.p;
    a.stepk = c->stepk.p; a.stepd = c->stepd.p;
    a.T = c->T; a.S = c->cfg.n_steps; a.h = step_size(c); a.inv_n = 1.0 / c->n_global;
    a.tp = c->tp_dev.p; a.TG = c->T; a.out_times = c->tp_dev.p;
    a.t_begin = c->tp.front(); a.t_end = c->tp.back();
    a.abstol = c->abstol; a.reltol = c->reltol;
    a.tape = c->tape.p; a.tape_cap = c->tape_cap; a.tape_n = c->tape_n.p;
    a.perm = (adaptive(c) && !c->slot_of.empty()) ? c->perm.p : nullptr;
    return a;
}

// CUDE_SUPP_CKPT=steps: gradient launches of the suppression model keep only the step states (744 B per subject at
// S = 30) and re-run the stages in the reverse sweep, instead of keeping every stage input (4.3 KB per subject)
bool supp_steps_only() {
    const char* env = getenv("CUDE_SUPP_CKPT");
    return env && std::strcmp(env, "steps") == 0;
}

cude::SuppArgs supp_args(const cude_ctx* c) {
    cude::SuppArgs a{};
    a.ckpt_steps_only = supp_steps_only() ? 1 : 0;
    a.N = c->N;
    a.data = c->data.p;
    a.obs_step = c->obs_step.p; a.obs_w = c->obs_w.p;
    a.T = c->T; a.S = c->cfg.n_steps; a.h = step_size(c); a.inv_n = 1.0 / c->n_global;
    for (int s = 0; s < 3; s++) a.iscale2[s] = 1.0 / (c->scale[s] * c->scale[s]);
    a.out_times = c->tp_dev.p;
    a.t_begin = c->tp.front(); a.t_end = c->tp.back();
    a.abstol = c->abstol; a.reltol = c->reltol;
    a.tape = c->tape.p; a.tape_cap = c->tape_cap; a.tape_n = c->tape_n.p;
    a.perm = (adaptive(c) && !c->slot_of.empty()) ? c->perm.p : nullptr;
    return a;
}

int32_t bind(cude_ctx* c) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(c->cfg.device));
    return CUDE_OK;
}

// The enumerators of nccl.h this file needs (librccl is dlopen'ed: its header is not compiled against).  They are not
// trusted: cude_comm_init runs comm_self_test(), which fails unless a sum and a max of known doubles come back right.
constexpr int kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2;

// op: 0 = sum, 1 = max
int32_t allreduce_dev(cude_ctx* c, double* buf, size_t count, int op = 0) {
    if (!c->comm) return CUDE_OK;
    RCCL_TRY(g_rccl.AllReduce(buf, buf, count, kNcclFloat64, op == 1 ? kNcclMax : kNcclSum, c->comm, c->stream));
    return CUDE_OK;
}

// sum / max of a small host vector over all ranks through the context's communicator (identity without one)
int32_t comm_reduce_host(cude_ctx* c, double* values, int32_t count, int op) {
    if (!c->comm) return CUDE_OK;
    HIP_TRY(c->red_tmp.reserve((size_t)count));
    HIP_TRY(hipMemcpyAsync(c->red_tmp.p, values, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int32_t rc = allreduce_dev(c, c->red_tmp.p, (size_t)count, op);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(values, c->red_tmp.p, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

// Every rank contributes [1, 2, rank + 1]: the sum must be [n, 2n, n(n+1)/2] and the max [1, 2, n].  A wrong datatype
// or operator enumerator (or a communicator that silently spans fewer ranks) cannot produce both.
int32_t comm_self_test(cude_ctx* c) {
    const double n = (double)c->n_ranks;
    double v[3] = {1.0, 2.0, (double)c->rank + 1.0};
    int32_t rc = comm_reduce_host(c, v, 3, 0);
    if (rc) return rc;
    if (v[0] != n || v[1] != 2.0 * n || v[2] != 0.5 * n * (n + 1.0))
        return fail(CUDE_ERR_COMM, "RCCL self-test: sum all-reduce of doubles returned a wrong result");
    double w[3] = {1.0, 2.0, (double)c->rank + 1.0};
    if ((rc = comm_reduce_host(c, w, 3, 1))) return rc;
    if (w[0] != 1.0 || w[1] != 2.0 || w[2] != n)
        return fail(CUDE_ERR_COMM, "RCCL self-test: max all-reduce of doubles returned a wrong result");
    return CUDE_OK;
}

// cude::ReduceFn over the context's communicator (the L-BFGS stage of cude_train_restarts on a sharded population)
int32_t lbfgs_comm_reduce(double* values, int32_t count, int32_t op, void* user) {
    return comm_reduce_host(static_cast<cude_ctx*>(user), values, count, op);
}

void drop_graph(cude_ctx* c);

// Tape of the adaptive gradient: (2 + NS) doubles per accepted step and subject + T saved outputs.  The reference's
// problems take 10-40 steps at its tolerances; the capacity is what ~4 GB hold, between 64 and 1024 steps
// (CUDE_TAPE_STEPS overrides).  A subject with more accepted steps fails its gradient evaluation (+Inf), not the
// process.  Allocated by the first gradient evaluation (forward-only users of the adaptive mode never pay for it), never
// under stream capture.
int32_t ensure_tape(cude_ctx* c) {
    if (!adaptive(c) || c->tape.p) return CUDE_OK;
    if (c->capturing) return fail(CUDE_ERR_STATE, "adaptive gradient tape not allocated before stream capture");
    const int64_t N = c->N;
    const int rows = cude::adaptive_tape_rows(c->cfg.model == CUDE_MODEL_SUPP ? 3 : 2);
    int64_t cap = (int64_t)(4e9 / (8.0 * rows * (double)N));
    cap = std::max<int64_t>(64, std::min<int64_t>(1024, cap));
    if (const char* env = getenv("CUDE_TAPE_STEPS")) cap = std::max(1, atoi(env));
    c->tape_cap = (int)cap;
    HIP_TRY(c->tape.resize((size_t)cude::adaptive_tape_rows(rows - 2, (int)cap, c->T) * N));
    HIP_TRY(c->tape_n.resize((size_t)N));
    return CUDE_OK;
}

int32_t alloc_common(cude_ctx* c) {
    const int64_t N = c->N;
    drop_graph(c);
    c->nblocks = (N + cude::kBlock - 1) / cude::kBlock;
    HIP_TRY(c->cond.resize(N));
    HIP_TRY(c->g_cond.resize(N));
    HIP_TRY(c->sse.resize(N));
    HIP_TRY(c->partials.resize((size_t)c->nblocks * (c->P + 2)));
    if (c->pinned_pairs_n < c->nblocks && c->nblocks <= 8192) {      // (bigger populations are not launch-bound)
        if (c->pinned_pairs) (void)hipHostFree(c->pinned_pairs);
        c->pinned_pairs = nullptr;
        c->pinned_pairs_n = 0;
        if (hipHostMalloc((void**)&c->pinned_pairs, (size_t)c->nblocks * 2 * sizeof(double), hipHostMallocDefault) ==
            hipSuccess)
            c->pinned_pairs_n = c->nblocks;
        else
            c->pinned_pairs = nullptr;
    }
    HIP_TRY(c->perm.resize(0));          // a new population starts in its own order
    c->slot_of.clear();
    HIP_TRY(c->tape.resize(0));          // adaptive gradient tape: allocated by the first gradient evaluation
    c->tape_cap = 0;
    c->have_tape = false;
    HIP_TRY(c->m_cond.resize(N));
    HIP_TRY(c->v_cond.resize(N));
    HIP_TRY(hipMemsetAsync(c->cond.p, 0, N * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->m_cond.p, 0, N * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->v_cond.p, 0, N * sizeof(double), c->stream));
    c->have_cond = false;
    c->adam_t = 0;
    return CUDE_OK;
}

int32_t upload_tables(cude_ctx* c, bool glucose) {
    HIP_TRY(c->tp_dev.resize(c->tp.size()));
    HIP_TRY(hipMemcpyAsync(c->tp_dev.p, c->tp.data(), c->tp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (adaptive(c)) {                       // no step grid: the kernels locate knots and outputs themselves
        HIP_TRY(hipStreamSynchronize(c->stream));
        return CUDE_OK;
    }
    std::vector<int32_t> step;
    std::vector<double> w;
    locate_obs(c->tp, c->cfg.n_steps, step, w);
    HIP_TRY(c->obs_step.resize(step.size()));
    HIP_TRY(c->obs_w.resize(w.size()));
    HIP_TRY(hipMemcpyAsync(c->obs_step.p, step.data(), step.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->obs_w.p, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (glucose) {
        std::vector<int32_t> seg;
        std::vector<double> phi;
        glucose_tables(c->tp, c->cfg.n_steps, seg, phi);
        HIP_TRY(c->seg.resize(seg.size()));
        HIP_TRY(c->phi.resize(phi.size()));
        HIP_TRY(hipMemcpyAsync(c->seg.p, seg.data(), seg.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->phi.p, phi.data(), phi.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        std::vector<int32_t> sk;
        std::vector<double> sd;
        step_tables(c->tp, c->cfg.n_steps, sk, sd);
        if (getenv("CUDE_NO_EXPTAB"))                      // development switch: direct exponentials everywhere
            for (size_t q = 0; q < sk.size(); q += 3) sk[q] = sk[q + 1] = 0;
        HIP_TRY(c->stepk.resize(sk.size()));
        HIP_TRY(c->stepd.resize(sd.size()));
        HIP_TRY(hipMemcpyAsync(c->stepk.p, sk.data(), sk.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->stepd.p, sd.data(), sd.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));          // sk / sd die at the end of this block
    }
    HIP_TRY(hipStreamSynchronize(c->stream));   // host vectors die here
    return CUDE_OK;
}

int32_t check_times(int32_t n_obs, const double* tp) {
    if (n_obs < 2 || n_obs > cude::kMaxObs) return fail(CUDE_ERR_ARG, "n_obs must be in [2, 32]");
    for (int t = 1; t < n_obs; t++)
        if (!(tp[t] > tp[t - 1])) return fail(CUDE_ERR_ARG, "timepoints must be strictly increasing");
    return CUDE_OK;
}

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

// all_blocks: the time-split kernels for every workgroup (forward-only launches, also when the gradient launch is mixed:
// the chunk tables cover all subjects); otherwise from the mixed launch's first time-split block on
cude::Cpep2Args chunk_args(cude_ctx* c, const cude::CpepArgs& base, bool all_blocks = false, bool forward_only = false) {
    cude::Cpep2Args a2{};
    a2.base = base;
    a2.L = c->chunks;
    a2.chunk_start = c->chunk_start.p;
    a2.hom_M = c->hom_M.p; a2.hom_obs = c->hom_obs.p; a2.fsum = c->fsum.p; a2.wts = c->res.p;
    if (forward_only && all_blocks && c->chunks_f > 1) {     // the forward-only split and its own transfer matrices
        a2.L = c->chunks_f;
        a2.chunk_start = c->chunk_start_f.p;
        a2.hom_M = c->hom_M_f.p; a2.hom_obs = c->hom_obs_f.p; a2.fsum = c->fsum_f.p; a2.wts = nullptr;
    }
    a2.g_cond_part = c->g_cond_part.p; a2.partials2 = c->partials2.p;
    a2.base.blk0 = all_blocks ? 0 : c->blk0; a2.base.blk_count = 0;
    return a2;
}

// Relative cost of one gradient launch when `waves` workgroups of `evals` network evaluations each run on `slots`
// resident-wave slots: full rounds cost one wave length each; a last partial round that leaves at least half of the
// SIMDs with a single wave runs at single-wave speed (measured on MI355X: a wave alone on its SIMD takes 0.69 of the
// time it takes next to a second one; profiles/r02/sweep_chunks.txt).
double launch_cost(double waves, double slots, double evals) {
    const double x = waves / slots;
    const double full = std::floor(x + 1e-9), frac = x - full;
    const double rounds = full + (frac < 1e-9 ? 0.0 : (frac <= 0.5 ? 0.69 : 1.0));
    return rounds * evals;
}

// Chunking of the step range for the time-split gradient path (1 <-> the one-lane-per-subject kernel; forced by
// CUDE_CPEP_PATH=1 / =2:L or an unsupported shape).  Precomputes the kinetics-only chunk transfer matrices.
//
// Choice of L (divisors of S): the launch that minimises launch_cost().  One lane per subject gives nblocks waves of
// 5S+1 evaluations; L chunks give nblocks*L waves of 5S/L+3 (two extra forward and one extra reverse evaluation per
// chunk).  Splitting pays when the one-shot grid would end in a mostly idle round -- 1e5 subjects = 1563 waves on
// 2048 slots took as long as 125 000 -- or leave the chip empty (57 subjects = one wave).  Measured (round 2, 2x6x6x1,
// S = 30): 1e5 subjects 0.644 -> 0.571 ms (L = 5), 8e4 0.637 -> 0.477 (L = 3), 2e5 1.222 -> 1.135 (L = 3), 125 000
// stays at L = 1 (0.654 vs 0.683 for L = 2).
int32_t setup_chunks(cude_ctx* c) {
    c->chunks = 1;
    c->chunks_f = 0;
    c->blk0 = 0;
    c->slots_one = c->half_slots = 0;
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->cfg.device);
    c->half_slots = (int64_t)n_cu * 4;
    if (adaptive(c)) return CUDE_OK;
    c->slots_one = (int64_t)n_cu * std::max(1, cude::cpep_grad_waves_per_cu(c->net, c->cfg.n_state, c->T));
    c->half_slots = (int64_t)n_cu * 4;
    const char* env = getenv("CUDE_CPEP_PATH");
    if (env && env[0] == '1') return CUDE_OK;
    if (!cude::cpep2_shape_supported(c->net, c->cfg.n_state)) return CUDE_OK;
    const int S = c->cfg.n_steps;
    const int occ_rev = std::max(1, cude::cpep2_rev_waves_per_cu(c->net));
    const int occ_one = std::max(1, cude::cpep_grad_waves_per_cu(c->net, c->cfg.n_state, c->T));
    int L = 1;
    double best = launch_cost((double)c->nblocks, (double)n_cu * occ_one, 5.0 * S + 1.0);
    for (int d = 2; d <= S; d++) {
        if (S % d) continue;
        // (x 1.10: time-split launches of a whole population measure 8-10 % above this model, the one-lane launch on it --
        // 300 000 subjects: L = 3 modelled 9 % faster than one lane per subject, measured 7 % slower)
        const double cost = 1.10 * launch_cost((double)c->nblocks * d, (double)n_cu * occ_rev, 5.0 * S / d + 3.0);
        if (cost < best * (1.0 - 1e-3)) { best = cost; L = d; }
    }
    // Mixed launch (more than one machine-fill of subjects): whole rounds of the one-lane kernel, the remainder --
    // which would otherwise be a second, mostly idle round -- time-split on its own.  Cost = the two parts one after
    // the other (they do overlap at the seam; not counted).
    int64_t blk0 = 0;
    const int64_t slots_one = (int64_t)n_cu * occ_one;
    // Only between one and two machine-fills: with two or more whole rounds the one-lane launch's own tail is amortised
    // and the mixed launch measured slower (300 000 subjects 1.461 against 1.366 ms, 1e6 4.415 against 4.183 ms).
    if (c->nblocks > slots_one && c->nblocks < 2 * slots_one && getenv("CUDE_NO_MIXED") == nullptr) {
        const int64_t bulk = (c->nblocks / slots_one) * slots_one, rem = c->nblocks - bulk;
        if (rem > 0) {
            const double cost_bulk = launch_cost((double)bulk, (double)slots_one, 5.0 * S + 1.0);
            // chunks of ~6 steps for the remainder (measured best at 140 000 ... 200 000 subjects: L = 5 or 6 of S = 30)
            int Lm = 0;
            for (int d = 2; d <= S; d++)
                if (S % d == 0 && (Lm == 0 || std::fabs(d - S / 6.0) < std::fabs(Lm - S / 6.0))) Lm = d;
            if (Lm > 0) {
                const double cost = cost_bulk + launch_cost((double)rem * Lm, (double)n_cu * occ_rev, 5.0 * S / Lm + 3.0);
                if (cost < best * (1.0 - 3e-2)) { L = Lm; blk0 = bulk; best = cost; }
            }
        }
    }
    // Mixed launch below one machine-fill (between one and two waves per SIMD, register-limited one-lane kernel): one
    // long wave on every SIMD and the remainder as short waves in the second slot, side by side.  Measured on the
    // headline instance (profiles/r02/mixed_launch.txt): 1e5 subjects 0.549 -> 0.485 ms, 8e4 0.448 -> 0.402 ms, no gain
    // at 125 000 (0.610 vs 0.580) or at <= 65 536.  Model: the longer of the lone long wave (0.69 of its co-resident
    // time) and the whole work at two waves per SIMD, + 6 %; chunks of ~6 steps, ~3 for a small remainder.
    const int64_t half = (int64_t)n_cu * 4;
    if (blk0 == 0 && occ_one == 8 && c->nblocks > half && c->nblocks < slots_one && getenv("CUDE_NO_MIXED") == nullptr) {
        const int64_t rem = c->nblocks - half;
        const double target = S / (rem >= 300 ? 6.0 : 3.0);
        int Lm = 0;
        for (int d = 2; d <= S; d++)
            if (S % d == 0 && (Lm == 0 || std::fabs(d - target) < std::fabs(Lm - target))) Lm = d;
        if (Lm > 0) {
            const double e1 = 5.0 * S + 1.0;
            const double cost = std::max(0.69 * e1, 1.06 * ((double)half * e1 + (double)rem * (5.0 * S + 3.0 * Lm)) / (double)slots_one);
            if (cost <= best) { L = Lm; blk0 = half; best = cost; }
        }
    }
    // Kernels that could hold three waves per SIMD (the reference's 2-4-4-1 / 2-state instance): a third resident wave
    // adds no throughput to the one-lane launch (0.45 ms with one or two waves per SIMD, 0.83 ms with three), which the
    // slot-count model above does not know.  Measured rule (profiles/r02/mixed_launch.txt, second table): from ~1.2
    // waves per SIMD up to two, one long wave per SIMD + the rest in chunks of ~3 steps (1e5 subjects 0.387 -> 0.369 ms);
    // between two and three, two long waves per SIMD + the rest in chunks of ~6 steps (150 000: 0.549 -> 0.494 ms,
    // 196 608: 0.834 -> 0.629 ms).
    if (blk0 == 0 && occ_one >= 12 && getenv("CUDE_NO_MIXED") == nullptr) {
        int64_t bulk = 0;
        double target = 0.0;
        if (c->nblocks >= half + half / 6 && c->nblocks <= half + 3 * half / 4) { bulk = half; target = S / 3.0; }
        else if (c->nblocks > half + 3 * half / 4 && c->nblocks <= 2 * half) L = 1;   // two waves per SIMD, taking turns at
                                                                                     // the issue priority: 0.428 ms up to 131 072
        else if (c->nblocks > 2 * half && c->nblocks <= 3 * half) { bulk = 2 * half; target = S / 6.0; }
        if (bulk > 0) {
            int Lm = 0;
            for (int d = 2; d <= S; d++)
                if (S % d == 0 && (Lm == 0 || std::fabs(d - target) < std::fabs(Lm - target))) Lm = d;
            if (Lm > 0) { L = Lm; blk0 = bulk; }
        }
    }
    if (getenv("CUDE_DEBUG_SELECTOR"))
        fprintf(stderr, "[cude] chunk selector: nblocks=%lld CUs=%d waves/CU one-lane=%d reverse=%d -> L=%d, one-lane blocks %lld\n",
                (long long)c->nblocks, n_cu, occ_one, occ_rev, L, (long long)blk0);
    if (env && env[0] == '2' && env[1] == ':') { L = atoi(env + 2); blk0 = 0; }
    if (env && env[0] == '3' && env[1] == ':') {              // CUDE_CPEP_PATH=3:<one-lane blocks>:<L> (tests)
        blk0 = std::min<int64_t>(std::max<int64_t>(atoll(env + 2), 0), c->nblocks - 1);
        const char* q = std::strchr(env + 2, ':');
        L = q ? atoi(q + 1) : 2;
    }
    if (L > S) L = S;
    if (L < 2) return CUDE_OK;
    std::vector<int32_t> cs(L + 1);
    for (int k = 0; k <= L; k++) cs[k] = (int32_t)((int64_t)k * S / L);
    const int64_t N = c->N;
    const int T = c->T;
    HIP_TRY(c->chunk_start.resize(L + 1));
    HIP_TRY(c->hom_M.resize((size_t)L * 4 * N));
    HIP_TRY(c->hom_obs.resize((size_t)T * 2 * N));
    HIP_TRY(c->fsum.resize((size_t)L * (3 + T) * N));
    HIP_TRY(c->res.resize((size_t)5 * S * N));   // adjoint weights wts[5S][N]
    HIP_TRY(c->g_cond_part.resize((size_t)L * N));
    HIP_TRY(c->partials2.resize((size_t)L * c->nblocks * c->P));
    HIP_TRY(hipMemcpyAsync(c->chunk_start.p, cs.data(), (L + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    c->chunks = L;
    c->blk0 = blk0;
    if (blk0 > 0 && !c->stream2 && getenv("CUDE_MIXED_ONE_STREAM") == nullptr) {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    cude::CpepArgs a = cpep_args(c);
    cude::Cpep2Args a2 = chunk_args(c, a);
    HIP_TRY(cude::launch_cpep2_homog(a2, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // cs (host vector) dies here
    // ---- the forward-only split.  Model (fitted to profiles/r03/forward_chunks.txt): a SIMD that holds w waves of e
    // evaluations each needs e * w / thr(w) evaluation times (thr = 1, 1.33, 1.36, 1.38 ... for 1, 2, 3, 4+ waves: the
    // issue rates of profiles/r02/ubench_fma_latency.txt), e = 5S/L + 2, and the scan adds ~0.6 evaluation times per chunk.
    c->chunks_f = 0;
    if (getenv("CUDE_NO_FWD_SPLIT") == nullptr && !(env && (env[0] == '2' || env[0] == '3'))) {
        const double simds = (double)n_cu * 4.0;
        int Lf = 0;
        double best_f = 0.0;
        for (int d = 2; d <= S; d++) {
            if (S % d) continue;
            const double w = std::ceil((double)c->nblocks * d / simds);
            const double thr = w <= 1.0 ? 1.0 : (w <= 2.0 ? 1.33 : (w <= 3.0 ? 1.36 : 1.38));
            const double cost = (5.0 * S / d + 2.0) * w / thr + 0.6 * d;
            if (Lf == 0 || cost < best_f) { best_f = cost; Lf = d; }
        }
        if (Lf >= 2 && Lf != L) {
            std::vector<int32_t> csf(Lf + 1);
            for (int k = 0; k <= Lf; k++) csf[k] = (int32_t)((int64_t)k * S / Lf);
            HIP_TRY(c->chunk_start_f.resize(Lf + 1));
            HIP_TRY(c->hom_M_f.resize((size_t)Lf * 4 * N));
            HIP_TRY(c->hom_obs_f.resize((size_t)T * 2 * N));
            HIP_TRY(c->fsum_f.resize((size_t)Lf * (3 + T) * N));
            HIP_TRY(hipMemcpyAsync(c->chunk_start_f.p, csf.data(), (Lf + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            c->chunks_f = Lf;
            cude::Cpep2Args af = chunk_args(c, a, /*all_blocks=*/true, /*forward_only=*/true);
            HIP_TRY(cude::launch_cpep2_homog(af, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        if (getenv("CUDE_DEBUG_SELECTOR"))
            fprintf(stderr, "[cude] forward-only split: L_f=%d (gradient L=%d)\n", Lf, L);
    }
    return CUDE_OK;
}

// Alternating issue priority in the one-lane gradient kernel (CpepArgs::prio_shift): for a launch of `blocks` workgroups
// that is a single round with two waves on (some of) the SIMDs.  CUDE_PRIO_SHIFT=k overrides (0 = never).
int prio_shift_for(const cude_ctx* c, int64_t blocks) {
    static const char* env = getenv("CUDE_PRIO_SHIFT");
    if (env) return atoi(env);
    // two waves on (some of) the SIMDs and no third: also the kernels that could hold three (2-4-4-1 / 2 states at
    // 120 000 ... 131 072 subjects: 0.451 -> 0.426 ms); not the one-wave kernels (2-7-7-1: +2 %)
    return (c->slots_one >= 2 * c->half_slots && blocks > c->half_slots && blocks <= 2 * c->half_slots) ? 5 : 0;
}

// launches the ensemble kernel + second-stage reduction (+ all-reduce, + L2 term)
// cond_ov / sse_ov: evaluate at other conditional parameters / write the per-subject SSE elsewhere and stop
// after the ensemble kernels (used by the Metropolis E-step, which needs neither loss nor gradient).
int32_t run_ensemble(cude_ctx* c, bool grad, double* traj_dev, bool local_only = false,
                     const double* cond_ov = nullptr, double* sse_ov = nullptr) {
    const bool watch = c->allow_watch;      // (consumed here, before any early return: it belongs to THIS call only)
    c->allow_watch = false;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (!c->have_nn || (!c->have_cond && !cond_ov)) return fail(CUDE_ERR_STATE, "parameters not set");
    if (grad) { int32_t rc = ensure_tape(c); if (rc) return rc; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool fused_final = false;
    c->loss_in_pinned = false;
    if (c->timing && !c->capturing && (c->timing_count++ % c->timing_period) == 0) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            c->ev_pool.emplace_back(a, b);
        }
        e0 = c->ev_pool[c->ev_used].first;
        e1 = c->ev_pool[c->ev_used].second;
        c->ev_used++;
        HIP_TRY(hipEventRecord(e0, c->stream));
    }
    if (is_cpep(c)) {
        cude::CpepArgs a = cpep_args(c);
#ifdef CUDE_WAVE_TIMING
        if (grad) {
            HIP_TRY(c->dbg.resize((size_t)c->nblocks * 4));
            a.dbg = c->dbg.p;
        }
#endif
        a.cond = cond_ov ? cond_ov : c->cond.p; a.nn = c->nn.p;
        a.sse = sse_ov ? sse_ov : c->sse.p; a.traj = traj_dev; a.auc = c->auc.p;
        a.g_cond = c->g_cond.p; a.partials = c->partials.p;
        // allocated by cude_set_population_cpep (never here: this function also runs under stream capture)
        if (grad && !adaptive(c) && c->act.p && cpep_act_doubles(c) > 0 && c->act.n >= cpep_act_doubles(c)) {
            a.act = c->act.p;
            a.keep_mode = cpep_keep_mode(c);
        }
        if (c->chunks > 1 && c->blk0 > 0 && grad) {
            // whole rounds: one lane per subject; the remainder: time-split, on a second stream so that its short waves
            // fill the SIMDs the long ones leave one by one (fork / join by events: capturable)
            hipStream_t s2 = c->stream2 ? c->stream2 : c->stream;
            if (c->stream2) {
                HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
                HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            }
            a.blk_count = c->blk0;          // (no alternating issue priority here: it starves the short waves -- measured)
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, true, a, c->stream));
            a.blk_count = 0;
            cude::Cpep2Args a2 = chunk_args(c, a);
            HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, true, a2, s2));
            if (c->stream2) {
                HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
                HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            }
        } else if (c->chunks > 1 && (grad || traj_dev == nullptr)) {
            cude::Cpep2Args a2 = chunk_args(c, a, /*all_blocks=*/true, /*forward_only=*/!grad);
            // forward-only on one rank without an L2 term: the scan kernel's workgroups write their (sum SSE, failures)
            // pairs straight into page-locked host memory, which finish_loss adds up (no reduction launch, no copy)
            static const bool no_fuse = getenv("CUDE_NO_FUSED_FINAL") != nullptr;
            if (!grad && !sse_ov && !c->comm && c->cfg.lambda == 0.0 && c->pinned_pairs &&
                c->pinned_pairs_n >= c->nblocks && !c->capturing && !no_fuse) {
                a2.final_host = c->pinned_pairs;
                fused_final = true;
                // the host watches the pairs arrive (finish_loss) instead of going through the runtime's completion wait
                // (forward call at 1e4 subjects 56.9 -> 51.6 us, at 57 subjects 41.3 -> 36.6 us): every slot starts as a
                // NaN no kernel produces.  CUDE_NO_POLL_PINNED=1: plain hipStreamSynchronize.
                static const bool poll = getenv("CUDE_NO_POLL_PINNED") == nullptr;
                c->poll_pairs = poll && watch;
                if (c->poll_pairs) {
                    volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(c->pinned_pairs);
                    for (int64_t q = 0; q < 2 * c->nblocks; q++) w[q] = kPairSentinel;
                }
            }
            HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, grad, a2, c->stream));
        } else {
            if (grad && !adaptive(c)) a.prio_shift = prio_shift_for(c, c->nblocks);
            if (grad && adaptive(c)) {      // adaptive gradient kernel: at most two waves per SIMD (measured -4.8 %)
                static const char* env_ps = getenv("CUDE_PRIO_SHIFT");
                a.prio_shift = env_ps ? atoi(env_ps)
                                      : ((c->nblocks > c->half_slots && c->nblocks <= 2 * c->half_slots) ? 5 : 0);
            }
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, grad, a, c->stream));
        }
    } else {
        cude::SuppArgs a = supp_args(c);
        a.cond = cond_ov ? cond_ov : c->cond.p; a.nn = c->nn.p;
        a.ckpt = c->ckpt.p; a.sse = sse_ov ? sse_ov : c->sse.p; a.traj = traj_dev;
        // allocated by cude_set_population_supp (never here: this function also runs under stream capture)
        if (grad && !a.ckpt_steps_only && c->act.p && c->act.n >= supp_act_doubles(c)) a.act = c->act.p;
        a.g_cond = c->g_cond.p; a.partials = c->partials.p;
        HIP_TRY(cude::launch_supp(c->net, grad, a, c->stream));
    }
    if (e1) HIP_TRY(hipEventRecord(e1, c->stream));
    if (grad && adaptive(c)) c->have_tape = true;
    if (sse_ov) return CUDE_OK;
    const int P = c->P;
    const bool fold = c->fold_advance && grad && !local_only;
    const cude::TailAdvance adv_args = tail_advance(c);
    const cude::TailAdvance* adv_red = (fold && c->comm == nullptr && c->cfg.lambda == 0.0) ? &adv_args : nullptr;
    const cude::TailAdvance* adv_l2 = (fold && c->cfg.lambda != 0.0) ? &adv_args : nullptr;
    c->advance_done = adv_red != nullptr || adv_l2 != nullptr;
    // one rank, no L2 term, not capturing: the reduction that finishes [sum loss, n_failed] writes the pair into the
    // page-locked result buffer too, and finish_loss watches it arrive instead of queueing a copy and waiting for the stream
    double* host_tail = nullptr;
    {
        static const bool poll = getenv("CUDE_NO_POLL_PINNED") == nullptr;
        if (poll && watch && c->pinned && !c->comm && c->cfg.lambda == 0.0 && !c->capturing && !local_only && !fused_final) {
            host_tail = c->pinned + P;
            volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(host_tail);
            w[0] = kPairSentinel; w[1] = kPairSentinel;
        }
        c->tail_in_pinned = host_tail != nullptr;
    }
    if (grad && is_cpep(c) && c->chunks > 1 && c->blk0 > 0) {
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->blk0, P + 2, 0, P, c->g_nn.p, c->stream, 1, c->param_mask.p, P));
        HIP_TRY(cude::launch_reduce_cols(c->partials2.p, (c->nblocks - c->blk0) * c->chunks, P, 0, P, c->g_nn.p, c->stream, 1,
                                         c->param_mask.p, P, 0, /*accumulate=*/true));
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, P, 2, c->g_nn.p, c->stream, 1, nullptr, 0, 0, false,
                                         adv_red, host_tail));
    } else if (grad && is_cpep(c) && c->chunks > 1) {
        HIP_TRY(cude::launch_reduce_cols(c->partials2.p, c->nblocks * c->chunks, P, 0, P, c->g_nn.p, c->stream, 1,
                                         c->param_mask.p, P));
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, P, 2, c->g_nn.p, c->stream, 1, nullptr, 0, 0, false,
                                         adv_red, host_tail));
    } else if (grad) {
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, 0, P + 2, c->g_nn.p, c->stream, 1,
                                         c->param_mask.p, P, 0, false, adv_red, host_tail));
    } else if (fused_final) {
        c->loss_in_pinned = true;
    } else {
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, P, 2, c->g_nn.p, c->stream, 1, nullptr, 0, 0, false,
                                         nullptr, host_tail));
    }
    if (local_only) return CUDE_OK;   // the caller reduces across ranks and applies the L2 term
    if (c->comm) {
        int32_t rc = grad ? allreduce_dev(c, c->g_nn.p, P + 2) : allreduce_dev(c, c->g_nn.p + P, 2);
        if (rc) return rc;
    }
    if (c->cfg.lambda != 0.0) {
        if (!grad) HIP_TRY(hipMemsetAsync(c->g_nn.p, 0, P * sizeof(double), c->stream));
        HIP_TRY(cude::launch_l2_term(c->nn.p, P, c->cfg.lambda, c->n_global, c->g_nn.p, c->stream, c->param_mask.p, adv_l2));
    }
    return CUDE_OK;
}

// copies [loss_sum, n_failed] (and optionally g_nn) back and forms the reference's loss value
int32_t finish_loss(cude_ctx* c, double* loss, double* g_nn_host) {
    const int P = c->P;
    std::vector<double> pageable;
    if (!c->pinned) pageable.resize(P + 2);
    double* const tmp = c->pinned ? c->pinned : pageable.data();      // page-locked: no staging copy behind the sync
    const bool watch_tail = c->tail_in_pinned && !g_nn_host && !c->loss_in_pinned && c->pinned;
    c->tail_in_pinned = false;
    if (g_nn_host) {
        HIP_TRY(hipMemcpyAsync(tmp, c->g_nn.p, (P + 2) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    } else if (!c->loss_in_pinned && !watch_tail) {
        HIP_TRY(hipMemcpyAsync(tmp + P, c->g_nn.p + P, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    bool arrived = false;
    if (watch_tail) {
        volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(c->pinned + P);
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0; !arrived; spin++) {
            arrived = w[0] != kPairSentinel && w[1] != kPairSentinel;
            if (!arrived && (spin & 63) == 63 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!arrived) {     // (a kernel that never got there: let the ordinary path report it / fetch the pair)
            HIP_TRY(hipMemcpyAsync(tmp + P, c->g_nn.p + P, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
    }
    if (c->loss_in_pinned && !g_nn_host && c->poll_pairs) {
        // the pairs are written into page-locked host memory by the last kernel of the call: watching them arrive skips
        // the runtime's completion path; bounded (a faulting kernel never writes them): then the ordinary wait decides
        volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(c->pinned_pairs);
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0; !arrived; spin++) {
            arrived = true;
            for (int64_t q = 2 * c->nblocks - 1; q >= 0; q--)
                if (w[q] == kPairSentinel) { arrived = false; break; }
            if (!arrived && (spin & 63) == 63 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    c->poll_pairs = false;
    if (!arrived) HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->loss_in_pinned && !g_nn_host) {
        // the scan kernel's per-workgroup pairs, added in the order of reduce_partials_kernel (256 strided partial sums,
        // then the halving tree), so that the value does not depend on which of the two ways produced it
        for (int col = 0; col < 2; col++) {
            double part[256];
            for (int t = 0; t < 256; t++) {
                double v = 0.0;
                for (int64_t b = t; b < c->nblocks; b += 256) v += c->pinned_pairs[2 * b + col];
                part[t] = v;
            }
            for (int off = 128; off >= 1; off >>= 1)
                for (int t = 0; t < off; t++) part[t] += part[t + off];
            tmp[P + col] = part[0];
        }
    }
    c->loss_in_pinned = false;
    c->last_failed = (int64_t)std::llround(tmp[P + 1]);
    if (g_nn_host) std::memcpy(g_nn_host, tmp, P * sizeof(double));
    if (loss) *loss = (c->last_failed > 0 || !std::isfinite(tmp[P])) ? std::numeric_limits<double>::infinity()
                                                                     : tmp[P] / c->n_global;
    return CUDE_OK;
}

// (see cude_adaptive_regroup in include/cude.h)
int32_t adaptive_regroup(cude_ctx* c, int32_t* spread_before, int32_t* spread_after) {
    if (!adaptive(c) || !c->have_tape) return fail(CUDE_ERR_STATE, "no adaptive gradient evaluation on this context yet");
    const int64_t N = c->N;
    std::vector<int32_t> n_acc((size_t)N);
    HIP_TRY(hipMemcpyAsync(n_acc.data(), c->tape_n.p, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // mean over the waves of (largest - smallest accepted-step count among the wave's lanes), in the current order
    auto spread = [&](const std::vector<int32_t>& order) {
        int64_t tot = 0, waves = 0;
        for (int64_t b = 0; b < N; b += cude::kBlock) {
            int32_t lo = INT32_MAX, hi = 0;
            for (int64_t k = b; k < std::min<int64_t>(b + cude::kBlock, N); k++) {
                const int32_t v = n_acc[(size_t)(order.empty() ? k : order[(size_t)k])];
                lo = std::min(lo, v); hi = std::max(hi, v);
            }
            tot += hi - lo; waves++;
        }
        return (int32_t)((tot + waves / 2) / std::max<int64_t>(waves, 1));
    };
    std::vector<int32_t> cur;
    if (!c->slot_of.empty()) {
        cur.resize((size_t)N);
        for (int64_t sbj = 0; sbj < N; sbj++) cur[(size_t)c->slot_of[(size_t)sbj]] = (int32_t)sbj;
    }
    if (spread_before) *spread_before = spread(cur);
    std::vector<int32_t> order((size_t)N);
    for (int64_t k = 0; k < N; k++) order[(size_t)k] = (int32_t)k;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return n_acc[(size_t)x] > n_acc[(size_t)y]; });
    if (spread_after) *spread_after = spread(order);
    if (c->slot_of.empty() || c->perm.n != (size_t)N) drop_graph(c);   // the launches' `perm` argument changes (null -> buffer)
    HIP_TRY(c->perm.resize((size_t)N));
    HIP_TRY(hipMemcpyAsync(c->perm.p, order.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->slot_of.assign((size_t)N, 0);
    for (int64_t k = 0; k < N; k++) c->slot_of[(size_t)order[(size_t)k]] = (int32_t)k;
    c->have_tape = false;                           // the tape on the device is in the OLD launch order
    c->regroup_age = 0;
    return CUDE_OK;
}

}  // namespace

// =============================================================================== exported ABI
extern "C" {

const char* cude_last_error(void) { return g_err.c_str(); }

int32_t cude_device_count(int32_t* count) {
    if (!count) return fail(CUDE_ERR_ARG, "null count");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    *count = n;
    return CUDE_OK;
}

int32_t cude_n_params(int32_t nn_in, int32_t nn_width, int32_t nn_depth) {
    if (nn_width == 0 && nn_depth == 0) return 1;   // analytic production model: [p0]
    if (nn_in < 1 || nn_width < 1 || nn_depth < 1) return fail(CUDE_ERR_ARG, "bad network shape");
    cude::NetShape n{nn_in, nn_width, nn_depth};
    return n.n_params();
}

int32_t cude_create(const cude_config* cfg, cude_ctx** out) {
    if (!cfg || !out) return fail(CUDE_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->n_steps < 0 || cfg->n_steps > 100000) return fail(CUDE_ERR_ARG, "n_steps out of range");
    if (cfg->n_steps == 0 && cfg->model != CUDE_MODEL_SUPP && cfg->n_state != 2)
        return fail(CUDE_ERR_UNSUPPORTED, "adaptive mode (n_steps = 0) integrates the reference's 2-state c-peptide model");
    cude::NetShape net{cfg->nn_in, cfg->nn_width, cfg->nn_depth};
    if (cfg->model == CUDE_MODEL_CPEP_SYM) {
        if (cfg->nn_width != 0 || cfg->nn_depth != 0)
            return fail(CUDE_ERR_ARG, "the symbolic model has no network: nn_width and nn_depth must be 0");
        if (cfg->cond_space != CUDE_COND_LOG && cfg->cond_space != CUDE_COND_RAW)
            return fail(CUDE_ERR_ARG, "cond_space must be CUDE_COND_LOG or CUDE_COND_RAW");
        if (cfg->n_state != 2 && cfg->n_state != 3) return fail(CUDE_ERR_UNSUPPORTED, "n_state must be 2 or 3");
        net = cude::NetShape{1, 0, 0};
    } else if (cfg->model == CUDE_MODEL_CPEP || cfg->model == CUDE_MODEL_SUPP) {
        if (cfg->nn_in < 1 || cfg->nn_width < 1 || cfg->nn_depth < 1) return fail(CUDE_ERR_ARG, "bad network shape");
        if (cfg->cond_space != CUDE_COND_LOG)
            return fail(CUDE_ERR_ARG, "cond_space must be CUDE_COND_LOG for the network models");
        if (cfg->model == CUDE_MODEL_CPEP && !cude::cpep_shape_supported(net, cfg->n_state))
            return fail(CUDE_ERR_UNSUPPORTED, "c-peptide kernel not compiled for this (nn_in,width,depth,n_state)");
        if (cfg->model == CUDE_MODEL_SUPP && (cfg->n_state != 3 || !cude::supp_shape_supported(net)))
            return fail(CUDE_ERR_UNSUPPORTED, "suppression kernel not compiled for this (width,depth)");
    } else {
        return fail(CUDE_ERR_ARG, "unknown model id");
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(CUDE_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));
    cude_ctx* c = new (std::nothrow) cude_ctx();
    if (!c) return fail(CUDE_ERR_ARG, "out of host memory");
    c->cfg = *cfg;
    c->net = net;
    c->P = net.n_params();
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(CUDE_ERR_HIP, hipGetErrorString(e)); }
    const int P = c->P;
    if (hipHostMalloc((void**)&c->pinned, (size_t)(P + 2) * sizeof(double), hipHostMallocDefault) != hipSuccess) c->pinned = nullptr;
    if (c->nn.resize(P) || c->g_nn.resize(P + 2) || c->m_nn.resize(P) || c->v_nn.resize(P)) {
        cude_destroy(c);
        return fail(CUDE_ERR_HIP, "hipMalloc failed");
    }
    (void)hipMemsetAsync(c->g_nn.p, 0, (P + 2) * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->m_nn.p, 0, P * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->v_nn.p, 0, P * sizeof(double), c->stream);
    *out = c;
    return CUDE_OK;
}

int32_t cude_destroy(cude_ctx* c) {
    if (!c) return CUDE_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    for (auto& pr : c->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_pairs) (void)hipHostFree(c->pinned_pairs);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CUDE_OK;
}

int32_t cude_set_population_cpep(cude_ctx* c, int64_t N, int32_t n_obs, const double* timepoints,
                                 const double* glucose, const double* cpeptide, int64_t ld_subject, int64_t ld_time,
                                 const double* age, const uint8_t* t2dm) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!is_cpep(c)) return fail(CUDE_ERR_STATE, "context is not a c-peptide model");
    if (N < 1 || !timepoints || !glucose || !cpeptide || !age || !t2dm) return fail(CUDE_ERR_ARG, "null/empty input");
    if ((rc = check_times(n_obs, timepoints))) return rc;
    const int T = n_obs;
    c->have_pop = false;
    c->N = N;
    c->T = T;
    c->tp.assign(timepoints, timepoints + T);
    // stage as [T][N] (subject fastest) so every device access is coalesced
    std::vector<double> g((size_t)T * N), cp((size_t)T * N);
    for (int t = 0; t < T; t++)
        for (int64_t i = 0; i < N; i++) {
            g[(size_t)t * N + i] = glucose[i * ld_subject + t * ld_time];
            cp[(size_t)t * N + i] = cpeptide[i * ld_subject + t * ld_time];
        }
    DevBuf<double> gdev;
    DevBuf<uint8_t> t2dev;
    HIP_TRY(gdev.resize((size_t)T * N));
    HIP_TRY(t2dev.resize(N));
    HIP_TRY(c->obs.resize((size_t)T * N));
    HIP_TRY(c->dG.resize((size_t)T * N));
    HIP_TRY(c->k0.resize(N)); HIP_TRY(c->k1.resize(N)); HIP_TRY(c->k2.resize(N)); HIP_TRY(c->c0.resize(N));
    HIP_TRY(c->age.resize(N));
    HIP_TRY(c->auc.resize(c->cfg.n_state == 3 ? N : 0));
    HIP_TRY(hipMemcpyAsync(gdev.p, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->obs.p, cp.data(), cp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->age.p, age, N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t2dev.p, t2dm, N * sizeof(uint8_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(cude::launch_prepare_cpep(N, T, gdev.p, c->obs.p, c->age.p, t2dev.p, c->k0.p, c->k1.p, c->k2.p, c->c0.p,
                                      c->dG.p, c->stream));
    if ((rc = alloc_common(c))) return rc;
    HIP_TRY(c->act.resize(cpep_keep_activations(c) ? cpep_act_doubles(c) : 0));
    if ((rc = upload_tables(c, true))) return rc;
    if ((rc = setup_chunks(c))) return rc;
    c->n_global = (double)N;
    if (c->comm) {
        double v[1] = {(double)N};
        if ((rc = cude_comm_allreduce_host(c, v, 1))) return rc;
        c->n_global = v[0];
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_pop = true;
    return CUDE_OK;
}

int32_t cude_set_population_supp(cude_ctx* c, int64_t N, int32_t n_obs, const double* timepoints, const double* data) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (c->cfg.model != CUDE_MODEL_SUPP) return fail(CUDE_ERR_STATE, "context is not a suppression model");
    if (N < 1 || !timepoints || !data) return fail(CUDE_ERR_ARG, "null/empty input");
    if ((rc = check_times(n_obs, timepoints))) return rc;
    const int T = n_obs;
    c->have_pop = false;
    c->N = N;
    c->T = T;
    c->tp.assign(timepoints, timepoints + T);
    std::vector<double> d((size_t)3 * T * N);
    double ssum[4] = {0, 0, 0, (double)N};
    for (int64_t i = 0; i < N; i++)
        for (int s = 0; s < 3; s++) {
            double m = -std::numeric_limits<double>::infinity();
            for (int t = 0; t < T; t++) {
                const double v = data[s + 3 * (t + (int64_t)T * i)];
                d[((size_t)s * T + t) * N + i] = v;
                if (v > m) m = v;
            }
            ssum[s] += m;
        }
    if (c->comm && (rc = cude_comm_allreduce_host(c, ssum, 4))) return rc;
    c->n_global = ssum[3];
    for (int s = 0; s < 3; s++) c->scale[s] = ssum[s] / ssum[3];
    HIP_TRY(c->data.resize(d.size()));
    HIP_TRY(c->ckpt.resize((size_t)cude::supp_ckpt_rows(c->cfg.n_steps, c->T) * N));   // every stage input + residuals
    HIP_TRY(c->act.resize(supp_keep_activations(c, 1) ? supp_act_doubles(c) : 0));   // kept activations (small N)
    HIP_TRY(hipMemcpyAsync(c->data.p, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = alloc_common(c))) return rc;
    if ((rc = upload_tables(c, false))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_pop = true;
    return CUDE_OK;
}

int32_t cude_set_params(cude_ctx* c, const double* nn, const double* cond) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (nn) {
        HIP_TRY(hipMemcpyAsync(c->nn.p, nn, c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->have_nn = true;
    }
    if (cond) {
        if (!c->have_pop) return fail(CUDE_ERR_STATE, "set the population before the conditional parameters");
        HIP_TRY(hipMemcpyAsync(c->cond.p, cond, c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->have_cond = true;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_get_params(cude_ctx* c, double* nn, double* cond) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (nn) HIP_TRY(hipMemcpyAsync(nn, c->nn.p, c->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (cond) {
        if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
        HIP_TRY(hipMemcpyAsync(cond, c->cond.p, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_forward(cude_ctx* c, double* loss, double* per_subject_sse, double* traj) {
    int32_t rc = bind(c);
    if (rc) return rc;
    double* traj_dev = nullptr;
    if (traj) {
        if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
        HIP_TRY(c->traj.resize((size_t)c->cfg.n_state * c->T * c->N));
        traj_dev = c->traj.p;
    }
    c->allow_watch = !per_subject_sse && !traj;        // (copies into caller memory pending: the stream must be waited for)
    if ((rc = run_ensemble(c, false, traj_dev))) return rc;
    if (per_subject_sse)
        HIP_TRY(hipMemcpyAsync(per_subject_sse, c->sse.p, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (traj)
        HIP_TRY(hipMemcpyAsync(traj, c->traj.p, c->traj.n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return finish_loss(c, loss, nullptr);
}

int32_t cude_loss_grad(cude_ctx* c, double* loss, double* g_nn, double* g_cond) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if ((rc = run_ensemble(c, true, nullptr))) return rc;
    if (g_cond)
        HIP_TRY(hipMemcpyAsync(g_cond, c->g_cond.p, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    std::vector<double> tmp;
    return finish_loss(c, loss, g_nn);
}

int32_t cude_simulate(cude_ctx* c, int32_t n_times, const double* times, double* traj) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!is_cpep(c)) return fail(CUDE_ERR_UNSUPPORTED, "cude_simulate: c-peptide models only");
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (!c->have_nn || !c->have_cond) return fail(CUDE_ERR_STATE, "parameters not set");
    if (n_times < 1 || !times || !traj) return fail(CUDE_ERR_ARG, "null/empty input");
    const int S = c->cfg.n_steps, NS = c->cfg.n_state;
    const double t0 = c->tp.front(), t1 = c->tp.back(), h = adaptive(c) ? (t1 - t0) : (t1 - t0) / S;
    for (int i = 0; i < n_times; i++) {
        if (!(times[i] >= t0 - 1e-9 * h && times[i] <= t1 + 1e-9 * h))
            return fail(CUDE_ERR_ARG, "output times must lie inside the time span of the population");
        if (i > 0 && !(times[i] >= times[i - 1])) return fail(CUDE_ERR_ARG, "output times must be non-decreasing");
    }
    const int64_t N = c->N;
    // output times per launch: ~1 GB of trajectory scratch at most (every launch integrates from t_0 again)
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_times, (int64_t)(1e9 / (8.0 * NS * (double)N))));
    DevBuf<double> d_traj, d_w, d_times;
    DevBuf<int32_t> d_step;
    HIP_TRY(d_traj.resize((size_t)NS * chunk * N));
    HIP_TRY(d_w.resize((size_t)chunk * 7));
    HIP_TRY(d_step.resize((size_t)chunk));
    HIP_TRY(d_times.resize((size_t)chunk));
    std::vector<int32_t> step(chunk);
    std::vector<double> w((size_t)chunk * 7);
    for (int64_t k0 = 0; k0 < n_times; k0 += chunk) {
        const int64_t kn = std::min<int64_t>(chunk, n_times - k0);
        if (adaptive(c)) {                                  // the kernel interpolates at the times themselves
            HIP_TRY(hipMemcpyAsync(d_times.p, times + k0, kn * sizeof(double), hipMemcpyHostToDevice, c->stream));
        } else {
            for (int64_t i = 0; i < kn; i++) {              // as locate_obs: tau in (t_n, t_{n+1}]
                const double x = (times[k0 + i] - t0) / h;
                int n = (int)std::ceil(x - 1e-9) - 1;
                n = std::min(std::max(n, 0), S - 1);
                step[i] = n;
                interp_weights((times[k0 + i] - (t0 + n * h)) / h, &w[(size_t)i * 7]);
            }
            HIP_TRY(hipMemcpyAsync(d_step.p, step.data(), kn * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_w.p, w.data(), kn * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        cude::CpepArgs a = cpep_args(c);
        a.obs = nullptr;                                    // no residuals: outputs only
        a.cond = c->cond.p; a.nn = c->nn.p;
        a.obs_step = d_step.p; a.obs_w = d_w.p; a.out_times = d_times.p;
        a.T = (int32_t)kn;
        a.traj = d_traj.p; a.partials = c->partials.p;
        HIP_TRY(cude::launch_cpep(c->net, NS, false, a, c->stream));
        // device chunk [NS x kn x N] -> rows k0..k0+kn of the caller's [NS x n_times x N]
        HIP_TRY(hipMemcpy2DAsync(traj + (size_t)NS * k0, (size_t)NS * n_times * sizeof(double), d_traj.p,
                                 (size_t)NS * kn * sizeof(double), (size_t)NS * kn * sizeof(double), (size_t)N,
                                 hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));           // step / w are reused by the next chunk
    }
    return CUDE_OK;
}

int32_t cude_n_failed(cude_ctx* c, int64_t* n_failed) {
    if (!c || !n_failed) return fail(CUDE_ERR_ARG, "null argument");
    *n_failed = c->last_failed;
    return CUDE_OK;
}

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
// appended to a device trace and copied back after a single synchronisation.  With a communicator attached
// the iterations are queued as plain launches (RCCL calls are not captured).
int32_t cude_adam_run(cude_ctx* c, int32_t n_iters, double* losses) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->adam_ready) return fail(CUDE_ERR_STATE, "call cude_adam_init first");
    if (n_iters < 1) return fail(CUDE_ERR_ARG, "n_iters must be >= 1");
    if (!c->have_pop || !c->have_nn || !c->have_cond) return fail(CUDE_ERR_STATE, "population / parameters not set");
    if ((rc = ensure_trace(c, n_iters))) return rc;
    if ((rc = ensure_tape(c))) return rc;
    // Large adaptive populations: keep the launch ordered by accepted-step count (cude_adaptive_regroup) -- once the
    // first evaluation has told the counts, then every 200 iterations (they drift with the parameters).  Costs one
    // read-back of N counters and a host sort, ~10 ms at 1e5 subjects; CUDE_NO_AUTO_REGROUP=1 leaves it to the caller.
    if (adaptive(c) && c->N >= 8192 && c->have_tape && (c->slot_of.empty() || c->regroup_age >= 200) &&
        getenv("CUDE_NO_AUTO_REGROUP") == nullptr) {
        if ((rc = adaptive_regroup(c, nullptr, nullptr))) return rc;
    }
    c->regroup_age += n_iters;
    HIP_TRY(hipMemsetAsync(c->adam_state.p + 3, 0, sizeof(double), c->stream));     // trace position = 0
    const bool use_graph = (c->comm == nullptr) && !c->timing && getenv("CUDE_NO_GRAPH") == nullptr;
    static const int kGraphUnroll = getenv("CUDE_GRAPH_UNROLL") ? std::max(1, atoi(getenv("CUDE_GRAPH_UNROLL"))) : 8;
    int u_max = 0;
    while (u_max + 1 < cude_ctx::kGraphKinds && (2 << u_max) <= kGraphUnroll) u_max++;
    for (int u = 0; u <= u_max && use_graph; u++) {
        const int reps = 1 << u;
        // needed by this run: the largest kind as often as it fits, the smaller ones by the bits of the remainder
        const bool needed = u == u_max ? n_iters >= reps : (((n_iters % (1 << u_max)) >> u) & 1) != 0;
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
    for (int k = 0; k < n_iters;) {
        if (use_graph) {
            int u = u_max;
            while (u > 0 && (n_iters - k) < (1 << u)) u--;
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
    c->adam_t += n_iters;
    std::vector<double> tr((size_t)n_iters * 2);
    HIP_TRY(hipMemcpyAsync(tr.data(), c->adam_trace.p, tr.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->last_failed = (int64_t)std::llround(tr[(size_t)(n_iters - 1) * 2 + 1]);
    if (losses)
        for (int k = 0; k < n_iters; k++)
            losses[k] = (tr[2 * k + 1] > 0.0 || !std::isfinite(tr[2 * k])) ? std::numeric_limits<double>::infinity()
                                                                         : tr[2 * k] / c->n_global;
    return CUDE_OK;
}

int32_t cude_multistart_forward(cude_ctx* c, int32_t n_sets, const double* nn_sets, const double* cond_sets,
                                double* losses) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (n_sets < 1 || !nn_sets || !cond_sets || !losses) return fail(CUDE_ERR_ARG, "null/empty input");
    const int P = c->P;
    const int64_t N = c->N, nb = c->nblocks;
    // chunk so that one launch stays below 2^16-1 grid rows and ~256 MB of partials
    int64_t chunk = std::min<int64_t>(n_sets, 32768);
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, (int64_t)(256e6 / ((double)nb * (P + 2) * 8.0))));
    DevBuf<double> d_nn, d_cond, d_part, d_out;
    HIP_TRY(d_nn.resize((size_t)chunk * P));
    HIP_TRY(d_cond.resize((size_t)chunk * N));
    HIP_TRY(d_part.resize((size_t)chunk * nb * (P + 2)));
    HIP_TRY(d_out.resize((size_t)chunk * 2));
    std::vector<double> h_out((size_t)chunk * 2);
    double reg = 0.0;
    for (int64_t k0 = 0; k0 < n_sets; k0 += chunk) {
        const int64_t kn = std::min<int64_t>(chunk, n_sets - k0);
        HIP_TRY(hipMemcpyAsync(d_nn.p, nn_sets + k0 * P, kn * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_cond.p, cond_sets + k0 * N, kn * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (is_cpep(c)) {
            cude::CpepArgs a = cpep_args(c);
            a.cond = d_cond.p; a.nn = d_nn.p;
            a.partials = d_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = P; a.set_stride_cond = N;
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, false, a, c->stream));
        } else {
            cude::SuppArgs a = supp_args(c);
            a.cond = d_cond.p; a.nn = d_nn.p;
            a.partials = d_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = P; a.set_stride_cond = N;
            HIP_TRY(cude::launch_supp(c->net, false, a, c->stream));
        }
        HIP_TRY(cude::launch_reduce_sets(d_part.p, (int)kn, nb, P + 2, P, d_out.p, c->stream));
        if (c->comm && (rc = allreduce_dev(c, d_out.p, (size_t)kn * 2))) return rc;
        HIP_TRY(hipMemcpyAsync(h_out.data(), d_out.p, kn * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int64_t k = 0; k < kn; k++) {
            reg = 0.0;
            if (c->cfg.lambda != 0.0) {
                const double* w = nn_sets + (k0 + k) * P;
                for (int q = 0; q < P; q++) reg += w[q] * w[q];
            }
            const double sum = h_out[2 * k], nf = h_out[2 * k + 1];
            losses[k0 + k] = (nf > 0.0 || !std::isfinite(sum)) ? std::numeric_limits<double>::infinity()
                                                             : sum / c->n_global + c->cfg.lambda * reg;
        }
    }
    return CUDE_OK;
}

int32_t cude_screen_candidates(cude_ctx* c, int64_t n_candidates, int32_t n_keep, cude_candidate_fn gen, void* user,
                               int64_t* index_out, double* loss_out, double* nn_out, double* cond_out) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (n_candidates < 1 || n_keep < 1 || n_keep > 4096 || !gen || !index_out || !loss_out || !nn_out || !cond_out)
        return fail(CUDE_ERR_ARG, "bad argument (1 <= n_keep <= 4096)");
    if (n_keep > n_candidates) n_keep = (int32_t)n_candidates;
    const int P = c->P;
    const int64_t N = c->N, nb = c->nblocks;
    // candidates per launch: the grid's y dimension and ~256 MB of partial rows; the host holds ONE chunk at a time
    int64_t chunk = std::min<int64_t>(n_candidates, 32768);
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, (int64_t)(256e6 / ((double)nb * (P + 2) * 8.0))));
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, (int64_t)(512e6 / ((double)(P + N) * 8.0))));
    DevBuf<double> d_nn, d_cond, d_part, d_sums, d_loss, d_work, b_loss[2], b_nn[2], b_cond[2];
    DevBuf<long long> b_idx[2];
    DevBuf<int> d_sel;
    HIP_TRY(d_nn.resize((size_t)chunk * P));
    HIP_TRY(d_cond.resize((size_t)chunk * N));
    HIP_TRY(d_part.resize((size_t)chunk * nb * (P + 2)));
    HIP_TRY(d_sums.resize((size_t)chunk * 2));
    HIP_TRY(d_loss.resize((size_t)chunk));
    HIP_TRY(d_work.resize((size_t)chunk + n_keep));
    HIP_TRY(d_sel.resize((size_t)n_keep));
    for (int b = 0; b < 2; b++) {
        HIP_TRY(b_loss[b].resize(n_keep));
        HIP_TRY(b_idx[b].resize(n_keep));
        HIP_TRY(b_nn[b].resize((size_t)n_keep * P));
        HIP_TRY(b_cond[b].resize((size_t)n_keep * N));
    }
    std::vector<double> h_nn((size_t)chunk * P), h_cond((size_t)chunk * N);
    int have = 0, cur = 0;
    for (int64_t k0 = 0; k0 < n_candidates; k0 += chunk) {
        const int64_t kn = std::min<int64_t>(chunk, n_candidates - k0);
        if (gen(k0, (int32_t)kn, h_nn.data(), h_cond.data(), user) < 0)
            return fail(CUDE_ERR_ARG, "candidate generator reported an error");
        HIP_TRY(hipMemcpyAsync(d_nn.p, h_nn.data(), kn * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_cond.p, h_cond.data(), kn * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (is_cpep(c)) {
            cude::CpepArgs a = cpep_args(c);
            a.cond = d_cond.p; a.nn = d_nn.p;
            a.partials = d_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = P; a.set_stride_cond = N;
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, false, a, c->stream));
        } else {
            cude::SuppArgs a = supp_args(c);
            a.cond = d_cond.p; a.nn = d_nn.p;
            a.partials = d_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = P; a.set_stride_cond = N;
            HIP_TRY(cude::launch_supp(c->net, false, a, c->stream));
        }
        HIP_TRY(cude::launch_reduce_sets(d_part.p, (int)kn, nb, P + 2, P, d_sums.p, c->stream));
        if (c->comm && (rc = allreduce_dev(c, d_sums.p, (size_t)kn * 2))) return rc;
        HIP_TRY(cude::launch_set_losses((int)kn, d_sums.p, d_nn.p, P, c->cfg.lambda, c->n_global, d_loss.p, c->stream));
        cude::TopkArgs t{};
        t.n_keep = n_keep; t.n_have = have; t.n_new = (int)kn; t.first = k0;
        t.best_loss = b_loss[cur].p; t.best_idx = b_idx[cur].p; t.chunk_loss = d_loss.p; t.work = d_work.p;
        t.new_loss = b_loss[1 - cur].p; t.new_idx = b_idx[1 - cur].p; t.sel_src = d_sel.p;
        HIP_TRY(cude::launch_topk_merge(t, P, N, b_nn[cur].p, b_cond[cur].p, d_nn.p, d_cond.p, b_nn[1 - cur].p,
                                        b_cond[1 - cur].p, c->stream));
        have = (int)std::min<int64_t>(n_keep, have + kn);
        cur = 1 - cur;
        HIP_TRY(hipStreamSynchronize(c->stream));            // the host chunk buffers are refilled by the next gen()
    }
    std::vector<long long> idx(n_keep);
    HIP_TRY(hipMemcpyAsync(idx.data(), b_idx[cur].p, n_keep * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(loss_out, b_loss[cur].p, n_keep * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(nn_out, b_nn[cur].p, (size_t)n_keep * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(cond_out, b_cond[cur].p, (size_t)n_keep * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int r = 0; r < n_keep; r++) index_out[r] = idx[r];
    return CUDE_OK;
}

int32_t cude_multistart_loss_grad(cude_ctx* c, int32_t n_sets, const double* nn_sets, const double* cond_sets,
                                  double* losses, double* g_nn_sets, double* g_cond_sets) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (n_sets < 1 || !nn_sets || !cond_sets || !losses || !g_nn_sets || !g_cond_sets)
        return fail(CUDE_ERR_ARG, "null/empty input");
    const int P = c->P, S = c->cfg.n_steps;
    const int64_t N = c->N, nb = c->nblocks;
    const bool supp = c->cfg.model == CUDE_MODEL_SUPP;
    // Small populations: K restarts of a few dozen subjects are K single-wave chains on the one-lane kernel (25 waves on
    // 1024 SIMDs, each paying the full single-wave latency).  When the population itself runs time-split (chunks > 1) and
    // the sets do not fill the chip either, the time-split kernels take the set index as a third grid dimension: K x L
    // short waves instead.  Same kernels as cude_loss_grad on this context, so a set's result is bit-identical to it.
    const int L = c->chunks;
    const bool split = !supp && L > 1 && c->blk0 == 0 && nb * (int64_t)std::min<int64_t>(n_sets, 64) <= 512 &&
                       getenv("CUDE_NO_MS_SPLIT") == nullptr;
    // sets per launch: bounded by the grid's y / z dimension and ~512 MB of scratch
    if ((rc = ensure_tape(c))) return rc;                 // (fixes the capacity the per-set tapes share)
    const int64_t tape_rows = adaptive(c) ? cude::adaptive_tape_rows(supp ? 3 : 2, c->tape_cap, c->T) : 0;
    const double per_set = 8.0 * ((double)nb * (P + 2) + 2.0 * N + P + (double)tape_rows * N +
                                  (supp && !adaptive(c) ? (double)cude::supp_ckpt_rows(S, c->T) * N : 0.0) +
                                  (split ? (double)L * (3 + c->T) * N + 5.0 * S * N + (double)L * N + (double)L * nb * P : 0.0));
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_sets, split ? 16384 : 32768), (int64_t)(512e6 / per_set)));
    if (split) {
        HIP_TRY(c->ms_fsum.reserve((size_t)chunk * L * (3 + c->T) * N));
        HIP_TRY(c->ms_wts.reserve((size_t)chunk * 5 * S * N));
        HIP_TRY(c->ms_gcp.reserve((size_t)chunk * L * N));
        HIP_TRY(c->ms_p2.reserve((size_t)chunk * L * nb * P));
    }
    HIP_TRY(c->ms_nn.reserve((size_t)chunk * P));
    HIP_TRY(c->ms_cond.reserve((size_t)chunk * N));
    HIP_TRY(c->ms_gcond.reserve((size_t)chunk * N));
    HIP_TRY(c->ms_part.reserve((size_t)chunk * nb * (P + 2)));
    HIP_TRY(c->ms_out.reserve((size_t)chunk * (P + 2)));
    if (supp && !adaptive(c)) HIP_TRY(c->ms_ckpt.reserve((size_t)chunk * cude::supp_ckpt_rows(S, c->T) * N));
    if (adaptive(c)) HIP_TRY(c->ms_tape.reserve((size_t)chunk * tape_rows * N));
    c->ms_host.resize((size_t)chunk * (P + 2));
    for (int64_t k0 = 0; k0 < n_sets; k0 += chunk) {
        const int64_t kn = std::min<int64_t>(chunk, n_sets - k0);
        HIP_TRY(hipMemcpyAsync(c->ms_nn.p, nn_sets + k0 * P, kn * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->ms_cond.p, cond_sets + k0 * N, kn * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (split) {
            cude::CpepArgs a = cpep_args(c);
            a.cond = c->ms_cond.p; a.nn = c->ms_nn.p;
            a.g_cond = c->ms_gcond.p; a.partials = c->ms_part.p;
            a.sse = nullptr; a.traj = nullptr; a.auc = nullptr;
            a.n_sets = (int32_t)kn; a.set_stride_nn = P; a.set_stride_cond = N;
            cude::Cpep2Args a2 = chunk_args(c, a);
            a2.fsum = c->ms_fsum.p; a2.wts = c->ms_wts.p; a2.g_cond_part = c->ms_gcp.p; a2.partials2 = c->ms_p2.p;
            HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, true, a2, c->stream));
            // network gradient: the reverse chunks' partial rows; loss / failure columns: the scan's
            HIP_TRY(cude::launch_reduce_cols(c->ms_p2.p, nb * L, P, 0, P, c->ms_out.p, c->stream, (int)kn, c->param_mask.p, P,
                                             P + 2));
            HIP_TRY(cude::launch_reduce_cols(c->ms_part.p, nb, P + 2, P, 2, c->ms_out.p, c->stream, (int)kn));
        } else if (!supp) {
            cude::CpepArgs a = cpep_args(c);
            a.cond = c->ms_cond.p; a.nn = c->ms_nn.p;
            a.g_cond = c->ms_gcond.p; a.partials = c->ms_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = P; a.set_stride_cond = N;
            if (adaptive(c)) { a.tape = c->ms_tape.p; a.tape_n = nullptr; }
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, true, a, c->stream));     // one-lane kernel: the sets fill the chip
        } else {
            cude::SuppArgs a = supp_args(c);
            a.cond = c->ms_cond.p; a.nn = c->ms_nn.p;
            a.ckpt = c->ms_ckpt.p; a.g_cond = c->ms_gcond.p; a.partials = c->ms_part.p;
            if (adaptive(c)) { a.tape = c->ms_tape.p; a.tape_n = nullptr; }
            if (!adaptive(c) && !a.ckpt_steps_only && supp_keep_activations(c, kn)) {
                HIP_TRY(c->ms_act.reserve((size_t)kn * supp_act_doubles(c)));
                a.act = c->ms_act.p;
            }
            a.n_sets = (int32_t)kn; a.set_stride_nn = P; a.set_stride_cond = N;
            HIP_TRY(cude::launch_supp(c->net, true, a, c->stream));
        }
        if (!split)
            HIP_TRY(cude::launch_reduce_cols(c->ms_part.p, nb, P + 2, 0, P + 2, c->ms_out.p, c->stream, (int)kn,
                                             c->param_mask.p, P));
        if (c->comm && (rc = allreduce_dev(c, c->ms_out.p, (size_t)kn * (P + 2)))) return rc;
        HIP_TRY(hipMemcpyAsync(c->ms_host.data(), c->ms_out.p, kn * (P + 2) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(g_cond_sets + k0 * N, c->ms_gcond.p, kn * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int64_t k = 0; k < kn; k++) {
            const double* r = c->ms_host.data() + k * (P + 2);
            const double* w = nn_sets + (k0 + k) * P;
            double* g = g_nn_sets + (k0 + k) * P;
            // L2 term in the arithmetic of l2_term_kernel (64 strided partial sums, xor-butterfly, fma), so that a
            // set's loss and gradient are bit-identical to cude_loss_grad at the same parameters
            double sum = r[P];
            if (c->cfg.lambda != 0.0) {
                double part[64], tmp[64];
                for (int l = 0; l < 64; l++) {
                    part[l] = 0.0;
                    for (int q = l; q < P; q += 64) part[l] = std::fma(w[q], w[q], part[l]);
                }
                for (int off = 32; off >= 1; off >>= 1) {
                    for (int l = 0; l < 64; l++) tmp[l] = part[l] + part[l ^ off];
                    std::memcpy(part, tmp, sizeof(part));
                }
                sum = std::fma(c->cfg.lambda * c->n_global, part[0], sum);
                for (int q = 0; q < P; q++)
                    g[q] = std::fma(2.0 * c->cfg.lambda * (c->mask_host.empty() ? 1.0 : c->mask_host[q]), w[q], r[q]);
            } else {
                for (int q = 0; q < P; q++) g[q] = r[q];
            }
            losses[k0 + k] = (r[P + 1] > 0.0 || !std::isfinite(sum)) ? std::numeric_limits<double>::infinity()
                                                                    : sum / c->n_global;
        }
    }
    return CUDE_OK;
}

int32_t cude_adaptive_regroup(cude_ctx* c, int32_t* spread_before, int32_t* spread_after) {
    int32_t rc = bind(c);
    if (rc) return rc;
    return adaptive_regroup(c, spread_before, spread_after);
}

int32_t cude_adaptive_steps(cude_ctx* c, int64_t subject, int32_t cap, double* t_out, double* dt_out, int32_t* n_steps) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!n_steps || cap < 0) return fail(CUDE_ERR_ARG, "null/negative argument");
    if (!adaptive(c) || !c->have_tape) return fail(CUDE_ERR_STATE, "no adaptive gradient evaluation on this context yet");
    if (subject < 0 || subject >= c->N) return fail(CUDE_ERR_ARG, "subject out of range");
    int32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, c->tape_n.p + subject, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *n_steps = n;
    const int rows = cude::adaptive_tape_rows(c->cfg.model == CUDE_MODEL_SUPP ? 3 : 2);
    const int m = std::min(std::min(n, cap), c->tape_cap);
    const int64_t slot = c->slot_of.empty() ? subject : c->slot_of[(size_t)subject];      // the tape is in launch order
    for (int r = 0; r < 2 && m > 0; r++) {
        double* dst = r == 0 ? t_out : dt_out;
        if (!dst) continue;
        // entry k, row r of the tape: one double every rows * N
        HIP_TRY(hipMemcpy2DAsync(dst, sizeof(double), c->tape.p + (size_t)r * c->N + slot, (size_t)rows * c->N * sizeof(double),
                                 sizeof(double), (size_t)m, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_profile_conditional(cude_ctx* c, int32_t n_points, const double* values, double* sse_out) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (!c->have_nn) return fail(CUDE_ERR_STATE, "shared parameters not set");
    if (n_points < 1 || !values || !sse_out) return fail(CUDE_ERR_ARG, "null/empty input");
    const int P = c->P;
    const int64_t N = c->N, nb = c->nblocks;
    // grid points per launch: the grid's y dimension and ~512 MB of scratch (conditional sets, SSEs, partial rows)
    const double per_point = 8.0 * (2.0 * N + (double)nb * (P + 2));
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_points, 32768), (int64_t)(512e6 / per_point)));
    DevBuf<double> d_val, d_cond, d_sse, d_part;
    HIP_TRY(d_val.resize((size_t)chunk));
    HIP_TRY(d_cond.resize((size_t)chunk * N));
    HIP_TRY(d_sse.resize((size_t)chunk * N));
    HIP_TRY(d_part.resize((size_t)chunk * nb * (P + 2)));
    for (int64_t k0 = 0; k0 < n_points; k0 += chunk) {
        const int64_t kn = std::min<int64_t>(chunk, n_points - k0);
        HIP_TRY(hipMemcpyAsync(d_val.p, values + k0, kn * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(cude::launch_fill_rows(N, (int)kn, d_val.p, d_cond.p, c->stream));
        if (is_cpep(c)) {
            cude::CpepArgs a = cpep_args(c);
            a.cond = d_cond.p; a.nn = c->nn.p;
            a.sse = d_sse.p; a.partials = d_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = 0; a.set_stride_cond = N;      // one network, kn grid values
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, false, a, c->stream));
        } else {
            cude::SuppArgs a = supp_args(c);
            a.cond = d_cond.p; a.nn = c->nn.p;
            a.sse = d_sse.p; a.partials = d_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = 0; a.set_stride_cond = N;
            HIP_TRY(cude::launch_supp(c->net, false, a, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(sse_out + k0 * N, d_sse.p, kn * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CUDE_OK;
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

int32_t cude_train_restarts(cude_ctx* c, int32_t n_sets, const double* nn_sets, const double* cond_sets,
                            int32_t adam_iters, double learning_rate, int32_t lbfgs_iters, double* nn_out,
                            double* cond_out, double* objective_out, double* loss_trace) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (n_sets < 1 || !nn_sets || !cond_sets || !nn_out || !cond_out || !objective_out || adam_iters < 0 ||
        lbfgs_iters < 0 || !(learning_rate > 0))
        return fail(CUDE_ERR_ARG, "bad argument");
    // Adam is element-wise and shards with the subjects; L-BFGS takes inner products over [neural; conditional]: on a
    // sharded population the conditional part of every inner product / max-norm is reduced over the ranks (a few
    // doubles per iteration, cude::Lbfgs reducer), so every rank follows the same iterates
    const int K = n_sets, P = c->P;
    const int64_t N = c->N, n = P + N;
    // working copies in the ABI's [K][P] / [K][N] layout
    std::vector<double> nn(nn_sets, nn_sets + (size_t)K * P), cond(cond_sets, cond_sets + (size_t)K * N);
    std::vector<double> f(K), g_nn((size_t)K * P), g_cond((size_t)K * N);
    std::vector<char> alive(K, 1);
    const int64_t trace_len = (int64_t)adam_iters + lbfgs_iters;
    if (loss_trace)
        for (int64_t q = 0; q < (int64_t)K * trace_len; q++) loss_trace[q] = std::numeric_limits<double>::quiet_NaN();
    // ---- Adam, vectorised over the restarts; a restart whose loss becomes non-finite is dropped
    {
        std::vector<double> m_nn((size_t)K * P, 0.0), v_nn((size_t)K * P, 0.0), m_c((size_t)K * N, 0.0), v_c((size_t)K * N, 0.0);
        for (int t = 1; t <= adam_iters; t++) {
            if ((rc = cude_multistart_loss_grad(c, K, nn.data(), cond.data(), f.data(), g_nn.data(), g_cond.data()))) return rc;
            for (int k = 0; k < K; k++) {
                if (!std::isfinite(f[k])) alive[k] = 0;
                if (!alive[k]) continue;
                if (loss_trace) loss_trace[(int64_t)k * trace_len + (t - 1)] = f[k];
                cude::adam_update(nn.data() + (size_t)k * P, g_nn.data() + (size_t)k * P, m_nn.data() + (size_t)k * P,
                                  v_nn.data() + (size_t)k * P, P, t, learning_rate);
                cude::adam_update(cond.data() + (size_t)k * N, g_cond.data() + (size_t)k * N, m_c.data() + (size_t)k * N,
                                  v_c.data() + (size_t)k * N, N, t, learning_rate);
            }
        }
    }
    // ---- L-BFGS, one resumable state machine per surviving restart, advanced in lock step
    std::vector<cude::Lbfgs> opt;
    std::vector<int> owner;                               // restart index of each machine
    std::vector<double> x0(n);
    for (int k = 0; k < K; k++) {
        if (!alive[k]) continue;
        std::copy(nn.begin() + (size_t)k * P, nn.begin() + (size_t)(k + 1) * P, x0.begin());
        std::copy(cond.begin() + (size_t)k * N, cond.begin() + (size_t)(k + 1) * N, x0.begin() + P);
        if (c->comm) opt.emplace_back(x0.data(), (int)n, lbfgs_iters, 10, 1e-8, P, lbfgs_comm_reduce, c);
        else opt.emplace_back(x0.data(), (int)n, lbfgs_iters);
        owner.push_back(k);
    }
    std::vector<double> b_nn, b_cond, b_f, b_gnn, b_gcond, gfull(n);
    std::vector<int> active;
    while (true) {
        active.clear();
        for (size_t q = 0; q < opt.size(); q++)
            if (!opt[q].done()) active.push_back((int)q);
        if (active.empty()) break;
        const int A = (int)active.size();
        b_nn.resize((size_t)A * P); b_cond.resize((size_t)A * N); b_f.resize(A);
        b_gnn.resize((size_t)A * P); b_gcond.resize((size_t)A * N);
        for (int a = 0; a < A; a++) {
            const double* x = opt[active[a]].pending();
            std::copy(x, x + P, b_nn.begin() + (size_t)a * P);
            std::copy(x + P, x + n, b_cond.begin() + (size_t)a * N);
        }
        if ((rc = cude_multistart_loss_grad(c, A, b_nn.data(), b_cond.data(), b_f.data(), b_gnn.data(), b_gcond.data()))) return rc;
        for (int a = 0; a < A; a++) {
            std::copy(b_gnn.begin() + (size_t)a * P, b_gnn.begin() + (size_t)(a + 1) * P, gfull.begin());
            std::copy(b_gcond.begin() + (size_t)a * N, b_gcond.begin() + (size_t)(a + 1) * N, gfull.begin() + P);
            cude::Lbfgs& o = opt[active[a]];
            const int before = o.accepted_steps();
            o.feed(b_f[a], gfull.data());
            if (loss_trace && o.accepted_steps() > before && before < lbfgs_iters)     // where Optim's callback fires
                loss_trace[(int64_t)owner[active[a]] * trace_len + adam_iters + before] = o.current_f();
            if (o.comm_failed()) return CUDE_ERR_COMM;     // message already set by the reducer
        }
    }
    for (int k = 0; k < K; k++) objective_out[k] = std::numeric_limits<double>::infinity();
    for (size_t q = 0; q < opt.size(); q++) {
        const int k = owner[q];
        const std::vector<double>& x = opt[q].x();
        std::copy(x.begin(), x.begin() + P, nn.begin() + (size_t)k * P);
        std::copy(x.begin() + P, x.end(), cond.begin() + (size_t)k * N);
        objective_out[k] = opt[q].result().f;
    }
    std::copy(nn.begin(), nn.end(), nn_out);
    std::copy(cond.begin(), cond.end(), cond_out);
    return CUDE_OK;
}

int32_t cude_fit_conditional(cude_ctx* c, double lower, double upper, int32_t n_grid, int32_t n_iters,
                             double penalty_weight, double penalty_center, double* cond_out, double* objective_out,
                             double* sse_out) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (!c->have_nn) return fail(CUDE_ERR_STATE, "shared parameters not set");
    if (!(lower < upper) || !std::isfinite(lower) || !std::isfinite(upper)) return fail(CUDE_ERR_ARG, "need finite lower < upper");
    if (n_grid < 3 || n_iters < 1 || !(penalty_weight >= 0) || !std::isfinite(penalty_center))
        return fail(CUDE_ERR_ARG, "need n_grid >= 3, n_iters >= 1, penalty_weight >= 0");
    if (!cond_out) return fail(CUDE_ERR_ARG, "null output");
    const int64_t N = c->N;
    DevBuf<double> buf;                                   // a, b, c, d, fc, best, sse_c, sse_d
    HIP_TRY(buf.resize((size_t)8 * N));
    cude::FitArgs f{};
    f.N = N;
    f.a = buf.p; f.b = buf.p + N; f.c = buf.p + 2 * N; f.d = buf.p + 3 * N;
    f.fc = buf.p + 4 * N; f.best = buf.p + 5 * N;
    double* sse_c = buf.p + 6 * N;
    double* sse_d = buf.p + 7 * N;
    f.sse_c = sse_c; f.sse_d = sse_d;
    f.w = penalty_weight; f.mu = penalty_center;
    f.lower = lower; f.step = (upper - lower) / (n_grid - 1); f.gr = (std::sqrt(5.0) - 1.0) / 2.0;
    f.n_grid = n_grid;
    // everything below is queued on the stream; the only synchronisation is the copy-back at the end
    for (int k = 0; k < n_grid; k++) {                    // coarse scan of the box
        const double x = (k == n_grid - 1) ? upper : std::fma((double)k, f.step, lower);
        HIP_TRY(cude::launch_fill(N, x, f.c, c->stream));
        if ((rc = run_ensemble(c, false, nullptr, true, f.c, sse_c))) return rc;
        HIP_TRY(cude::launch_fit(0, f, k, x, c->stream));
    }
    HIP_TRY(cude::launch_fit(1, f, 0, 0.0, c->stream));
    for (int it = 0; it < n_iters; it++) {                // golden section inside the bracket
        if ((rc = run_ensemble(c, false, nullptr, true, f.c, sse_c))) return rc;
        if ((rc = run_ensemble(c, false, nullptr, true, f.d, sse_d))) return rc;
        HIP_TRY(cude::launch_fit(2, f, it == n_iters - 1 ? 1 : 0, 0.0, c->stream));
    }
    if ((rc = run_ensemble(c, false, nullptr, true, f.c, sse_c))) return rc;       // at the returned midpoint
    HIP_TRY(cude::launch_fit(3, f, 0, 0.0, c->stream));
    HIP_TRY(hipMemcpyAsync(cond_out, f.c, N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (objective_out) HIP_TRY(hipMemcpyAsync(objective_out, f.fc, N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (sse_out) HIP_TRY(hipMemcpyAsync(sse_out, sse_c, N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_mh_estep(cude_ctx* c, int32_t n_mc, const double* normals, const double* uniforms, double sigma,
                      double prior_mean, double prior_sd, double proposal_std, double temperature, double gamma,
                      int64_t* accepted) {
    return cude_mh_chain(c, n_mc, normals, uniforms, sigma, prior_mean, prior_sd, proposal_std, temperature, gamma,
                         accepted, nullptr);
}

int32_t cude_mh_chain(cude_ctx* c, int32_t n_mc, const double* normals, const double* uniforms, double sigma,
                      double prior_mean, double prior_sd, double proposal_std, double temperature, double gamma,
                      int64_t* accepted, double* samples) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop || !c->have_nn || !c->have_cond) return fail(CUDE_ERR_STATE, "population / parameters not set");
    if (n_mc < 1) return fail(CUDE_ERR_ARG, "n_mc must be >= 1");
    if ((normals == nullptr) != (uniforms == nullptr)) return fail(CUDE_ERR_ARG, "pass both draw arrays or neither");
    const bool device_rng = normals == nullptr;     // draws from the context's counter-based generator (cude_set_rng)
    if (!(sigma > 0) || !(prior_sd > 0) || !(temperature > 0)) return fail(CUDE_ERR_ARG, "sigma, prior_sd, temperature must be > 0");
    const int64_t N = c->N;
    DevBuf<double> d_z, d_u, d_prop, d_sn, d_sc;
    DevBuf<int64_t> d_acc;
    if (!device_rng || samples) HIP_TRY(d_z.resize((size_t)n_mc * N));     // draws, then (samples) the chain states
    if (!device_rng) HIP_TRY(d_u.resize((size_t)n_mc * N));
    HIP_TRY(d_prop.resize(N)); HIP_TRY(d_sn.resize(N)); HIP_TRY(d_sc.resize(N)); HIP_TRY(d_acc.resize(N));
    if (!device_rng) {
        HIP_TRY(hipMemcpyAsync(d_z.p, normals, (size_t)n_mc * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_u.p, uniforms, (size_t)n_mc * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipMemsetAsync(d_acc.p, 0, N * sizeof(int64_t), c->stream));
    cude::MhArgs m{};
    m.N = N; m.p = c->cond.p; m.prop = d_prop.p; m.sse_new = d_sn.p; m.sse_cur = d_sc.p; m.accepted = d_acc.p;
    m.prior_mean = prior_mean; m.prior_sd = prior_sd;
    m.ll_const = -(c->T / 2.0) * std::log(sigma * sigma);
    m.inv_2s2 = 1.0 / (2.0 * sigma * sigma);
    m.temperature = temperature; m.gamma = gamma;
    // The reference re-evaluates the likelihood of the current state in every step (saem.jl:96-97).  With gamma == 1
    // (its burn-in phase, and posterior sampling) the next state is exactly the accepted proposal or the unchanged
    // current one, and the solve is deterministic, so that value is already known: it is carried over instead of
    // recomputed -- the same bits, half the forward launches.  With gamma < 1 the state is a blend and is re-evaluated.
    m.carry_sse = gamma == 1.0 ? 1 : 0;
    if (m.carry_sse && (rc = run_ensemble(c, false, nullptr, true, c->cond.p, d_sc.p))) return rc;
    // Time-split forward path + carried SSE: the proposal is formed inside the forward chunks and accepted inside the
    // scan (Cpep2Args::mh_fused) -- two launches per Metropolis step instead of four, same bits.
    const bool fused = m.carry_sse && is_cpep(c) && !adaptive(c) && c->chunks > 1 && getenv("CUDE_NO_MH_FUSE") == nullptr;
    for (int k = 0; k < n_mc; k++) {          // everything is queued on the stream; one sync at the end
        m.key = cude::RngKey{c->rng_seed, c->rng_offset, c->rng_step + k};
        if (fused) {
            m.u = device_rng ? nullptr : d_u.p + (size_t)k * N;
            m.prop = nullptr; m.sse_new = nullptr;
            cude::CpepArgs a = cpep_args(c);
            a.cond = c->cond.p; a.nn = c->nn.p; a.sse = d_sn.p; a.traj = nullptr; a.auc = c->auc.p;
            a.g_cond = c->g_cond.p; a.partials = c->partials.p;
            cude::Cpep2Args a2 = chunk_args(c, a, /*all_blocks=*/true, /*forward_only=*/true);
            a2.mh_fused = 1;
            a2.mh_z = device_rng ? nullptr : d_z.p + (size_t)k * N;
            a2.mh_std = proposal_std;
            a2.mh = m;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            // kernel timing: a pair of events costs ~4.5 us of stream time, 10 % of a Metropolis step at 1e4 subjects, so
            // inside this loop of identical launches every 8th step is timed (cude_kernel_time_ms averages those)
            if (c->timing && k % 8 == 0) {
                if (c->ev_used == c->ev_pool.size()) {
                    hipEvent_t ea, eb;
                    HIP_TRY(hipEventCreate(&ea));
                    HIP_TRY(hipEventCreate(&eb));
                    c->ev_pool.emplace_back(ea, eb);
                }
                e0 = c->ev_pool[c->ev_used].first; e1 = c->ev_pool[c->ev_used].second;
                c->ev_used++;
                HIP_TRY(hipEventRecord(e0, c->stream));
            }
            HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, false, a2, c->stream));
            if (e1) HIP_TRY(hipEventRecord(e1, c->stream));
            if (samples)
                HIP_TRY(hipMemcpyAsync(d_z.p + (size_t)k * N, c->cond.p, N * sizeof(double), hipMemcpyDeviceToDevice,
                                       c->stream));
            continue;
        }
        HIP_TRY(cude::launch_mh_propose(N, c->cond.p, device_rng ? nullptr : d_z.p + (size_t)k * N, m.key, proposal_std,
                                        d_prop.p, c->stream));
        if ((rc = run_ensemble(c, false, nullptr, true, d_prop.p, d_sn.p))) return rc;
        if (!m.carry_sse && (rc = run_ensemble(c, false, nullptr, true, c->cond.p, d_sc.p))) return rc;
        m.u = device_rng ? nullptr : d_u.p + (size_t)k * N;
        HIP_TRY(cude::launch_mh_accept(m, c->stream));
        if (samples)       // chain state after step k (the draws of this step are no longer needed: reuse their row)
            HIP_TRY(hipMemcpyAsync(d_z.p + (size_t)k * N, c->cond.p, N * sizeof(double), hipMemcpyDeviceToDevice,
                                   c->stream));
    }
    if (samples)
        HIP_TRY(hipMemcpyAsync(samples, d_z.p, (size_t)n_mc * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (accepted) HIP_TRY(hipMemcpyAsync(accepted, d_acc.p, N * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (device_rng) c->rng_step += n_mc;            // the next call continues the stream
    return CUDE_OK;
}

int32_t cude_set_param_mask(cude_ctx* c, const double* mask) {
    int32_t rc = bind(c);
    if (rc) return rc;
    drop_graph(c);                                  // the mask pointer is baked into captured launches
    if (!mask) {
        HIP_TRY(c->param_mask.resize(0));
        c->mask_host.clear();
        return CUDE_OK;
    }
    for (int q = 0; q < c->P; q++)
        if (!std::isfinite(mask[q])) return fail(CUDE_ERR_ARG, "mask entries must be finite");
    c->mask_host.assign(mask, mask + c->P);
    HIP_TRY(c->param_mask.resize((size_t)c->P));
    HIP_TRY(hipMemcpyAsync(c->param_mask.p, c->mask_host.data(), c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // Adam moments gathered before the mask was set would keep moving a frozen entry (lr * m_hat / (sqrt(v_hat) + eps)
    // while m decays): they are multiplied by the mask as well
    if (c->adam_ready && c->m_nn.p && c->v_nn.p) {
        std::vector<double> mv(2 * (size_t)c->P);
        HIP_TRY(hipMemcpyAsync(mv.data(), c->m_nn.p, c->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(mv.data() + c->P, c->v_nn.p, c->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int q = 0; q < c->P; q++) { mv[q] *= mask[q]; mv[c->P + q] *= mask[q] * mask[q]; }
        HIP_TRY(hipMemcpyAsync(c->m_nn.p, mv.data(), c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->v_nn.p, mv.data() + c->P, c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_set_rng(cude_ctx* c, uint64_t seed, int64_t subject_offset) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    if (subject_offset < 0) return fail(CUDE_ERR_ARG, "subject_offset must be >= 0");
    c->rng_seed = seed;
    c->rng_offset = subject_offset;
    c->rng_step = 0;
    return CUDE_OK;
}

int32_t cude_rng_draws(cude_ctx* c, int64_t first_step, int32_t n_steps, double* normals, double* uniforms) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (first_step < 0 || n_steps < 1 || (!normals && !uniforms)) return fail(CUDE_ERR_ARG, "bad argument");
    const int64_t N = c->N;
    DevBuf<double> d_z, d_u;
    if (normals) HIP_TRY(d_z.resize((size_t)n_steps * N));
    if (uniforms) HIP_TRY(d_u.resize((size_t)n_steps * N));
    for (int k = 0; k < n_steps; k++)
        HIP_TRY(cude::launch_rng_draws(N, cude::RngKey{c->rng_seed, c->rng_offset, first_step + k},
                                       normals ? d_z.p + (size_t)k * N : nullptr, uniforms ? d_u.p + (size_t)k * N : nullptr,
                                       c->stream));
    if (normals) HIP_TRY(hipMemcpyAsync(normals, d_z.p, (size_t)n_steps * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (uniforms) HIP_TRY(hipMemcpyAsync(uniforms, d_u.p, (size_t)n_steps * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_set_global_subjects(cude_ctx* c, double n_global, const double* scale3) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (!(n_global >= (double)c->N)) return fail(CUDE_ERR_ARG, "global subject count smaller than the local one");
    drop_graph(c);                              // 1/n_global and the scale are baked into the captured launches
    c->n_global = n_global;
    if (scale3) {
        for (int s = 0; s < 3; s++) {
            if (!(scale3[s] > 0)) return fail(CUDE_ERR_ARG, "scale must be positive");
            c->scale[s] = scale3[s];
        }
    }
    return CUDE_OK;
}

int32_t cude_get_scale(cude_ctx* c, double* scale3, double* n_global) {
    if (!c || !scale3 || !n_global) return fail(CUDE_ERR_ARG, "null argument");
    for (int s = 0; s < 3; s++) scale3[s] = c->scale[s];
    *n_global = c->n_global;
    return CUDE_OK;
}

int32_t cude_loss_grad_partial(cude_ctx* c, double* partial, double* g_cond) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!partial) return fail(CUDE_ERR_ARG, "null output");
    if ((rc = run_ensemble(c, true, nullptr, /*local_only=*/true))) return rc;
    if (g_cond)
        HIP_TRY(hipMemcpyAsync(g_cond, c->g_cond.p, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(partial, c->g_nn.p, (c->P + 2) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_partial_buffer(cude_ctx* c, double** device_ptr, int32_t* count) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!device_ptr || !count) return fail(CUDE_ERR_ARG, "null output");
    if (!c->g_nn.p) return fail(CUDE_ERR_STATE, "context has no parameters yet");
    *device_ptr = c->g_nn.p;
    *count = c->P + 2;
    return CUDE_OK;
}

int32_t cude_loss_grad_partial_device(cude_ctx* c) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if ((rc = run_ensemble(c, true, nullptr, /*local_only=*/true))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));     // the caller's collective runs on a stream this library does not know
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

#ifdef CUDE_WAVE_TIMING
// development builds only: per-wave {start, end of forward sweep, end, hw id} of the last gradient launch
int32_t cude_debug_wave_timing(cude_ctx* c, long long* out, int64_t n_waves) {
    int32_t rc = bind(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->dbg.p, (size_t)std::min<int64_t>(n_waves, c->nblocks) * 4 * sizeof(long long),
                      hipMemcpyDeviceToHost));
    return CUDE_OK;
}
#endif

int32_t cude_set_tolerances(cude_ctx* c, double abstol, double reltol) {
    if (!(abstol > 0) || !(reltol > 0) || !std::isfinite(abstol) || !std::isfinite(reltol))
        return fail(CUDE_ERR_ARG, "tolerances must be positive");
    int32_t rc = bind(c);
    if (rc) return rc;
    drop_graph(c);      // a captured optimiser iteration carries the tolerances by value in its kernel arguments
    c->abstol = abstol;
    c->reltol = reltol;
    return CUDE_OK;
}

int32_t cude_grad_occupancy(cude_ctx* c, int32_t* waves_per_cu) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!waves_per_cu) return fail(CUDE_ERR_ARG, "null output");
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    *waves_per_cu = is_cpep(c) ? cude::cpep_grad_waves_per_cu(c->net, c->cfg.n_state, c->T)
                               : cude::supp_grad_waves_per_cu(c->net);
    return CUDE_OK;
}

int32_t cude_synchronize(cude_ctx* c) {
    int32_t rc = bind(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_set_kernel_timing(cude_ctx* c, int32_t enabled) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    c->timing = enabled != 0;
    c->timing_period = enabled > 1 ? enabled : 1;
    c->timing_count = 0;
    c->ev_used = 0;
    return CUDE_OK;
}

int32_t cude_kernel_time_ms(cude_ctx* c, double* avg_ms, int64_t* launches) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!avg_ms) return fail(CUDE_ERR_ARG, "null output");
    HIP_TRY(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (size_t k = 0; k < c->ev_used; k++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[k].first, c->ev_pool[k].second));
        tot += ms;
    }
    *avg_ms = c->ev_used ? tot / (double)c->ev_used : 0.0;
    if (launches) *launches = (int64_t)c->ev_used;
    c->ev_used = 0;
    c->timing_count = 0;            // (the first launch after a query is a timed one, whatever the period)
    return CUDE_OK;
}

int32_t cude_comm_unique_id(uint8_t id[CUDE_UNIQUE_ID_BYTES]) {
    if (!id) return fail(CUDE_ERR_ARG, "null id");
    int32_t rc = load_rccl();
    if (rc) return rc;
    nccl_uid u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, CUDE_UNIQUE_ID_BYTES);
    return CUDE_OK;
}

int32_t cude_comm_init(cude_ctx* c, int32_t n_ranks, int32_t rank, const uint8_t id[CUDE_UNIQUE_ID_BYTES]) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !id) return fail(CUDE_ERR_ARG, "bad communicator arguments");
    if (c->comm) return fail(CUDE_ERR_STATE, "communicator already attached");
    // Multi-process RCCL needs dmabuf IPC on hosts whose driver has no legacy IPC: without
    // HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the first HIP call, ncclCommInitRank dies much later
    // with "hipIpcGetMemHandle: invalid argument".  Too late to set it here, so say so now.
    if (n_ranks > 1) {
        const char* ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
        if ((!ipc || std::strcmp(ipc, "0") != 0) && !getenv("CUDE_ALLOW_LEGACY_IPC"))
            return fail(CUDE_ERR_COMM, "export HSA_ENABLE_IPC_MODE_LEGACY=0 before the process touches the GPU "
                                       "(dmabuf IPC for RCCL); set CUDE_ALLOW_LEGACY_IPC=1 to skip this check");
    }
    if ((rc = load_rccl())) return rc;
    nccl_uid u;
    std::memcpy(u.internal, id, CUDE_UNIQUE_ID_BYTES);
    RCCL_TRY(g_rccl.CommInitRank(&c->comm, n_ranks, u, rank));
    c->n_ranks = n_ranks;
    c->rank = rank;
    if ((rc = comm_self_test(c))) {          // (collective: every rank runs it, every rank sees the same verdict)
        (void)g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
        c->n_ranks = 1;
        c->rank = 0;
        return rc;
    }
    return CUDE_OK;
}

int32_t cude_comm_allreduce_host(cude_ctx* c, double* values, int32_t count) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!values || count < 1) return fail(CUDE_ERR_ARG, "bad buffer");
    return comm_reduce_host(c, values, count, 0);   // single rank: identity
}

int32_t cude_comm_info(cude_ctx* c, int32_t* n_ranks, int32_t* rank, int32_t* version) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!n_ranks || !rank || !version) return fail(CUDE_ERR_ARG, "null output");
    *n_ranks = 1; *rank = 0; *version = 0;
    if (!c->comm) return CUDE_OK;
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(CUDE_ERR_COMM, "librccl lacks ncclCommCount/ncclCommUserRank");
    int n = 0, r = 0, v = 0;
    RCCL_TRY(g_rccl.CommCount(c->comm, &n));
    RCCL_TRY(g_rccl.CommUserRank(c->comm, &r));
    if (g_rccl.GetVersion) RCCL_TRY(g_rccl.GetVersion(&v));
    *n_ranks = n; *rank = r; *version = v;
    return CUDE_OK;
}

}  // extern "C"
