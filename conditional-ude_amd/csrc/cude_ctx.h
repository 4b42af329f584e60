// Internal header of libcude_hip.so's host side: the context, its helpers and what the translation units share.
//   cude_context.hip   contexts, populations, parameters, solver tables, run-time options
//   cude_launch.hip    launch selection (one-lane / time-split / mixed), ensemble launches and every entry point built on them
//   cude_optimise.hip  Adam (single steps and captured runs), L-BFGS on host vectors
//   cude_train.hip     restarts trained side by side with their optimiser state on the device (cude_train_restarts)
//   cude_comm.hip      multi-GPU: RCCL (dlopen) and the peer-write exchange over IPC-mapped mailboxes
#pragma once
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

namespace cude {
namespace api {

// thread-local text of the last error (cude_last_error); returns `code`
int32_t fail(int32_t code, const std::string& msg);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return ::cude::api::fail(CUDE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
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

// Run-time options of a context.  cude_set_option(ctx, name, value) sets one; at cude_create every option whose
// environment variable is set (table in cude_context.hip) takes that value.  The second group exists for A/B
// measurements only: the shipping library reads their environment variables in -DCUDE_ABLATION builds alone.
struct Options {
    int cpep_path = 0;          // "cpep_path" / CUDE_CPEP_PATH: "" = the cost model chooses; "1" = one lane per subject;
    int path_chunks = 0;        //   "2:L" = time-split into L chunks; "3:B:L" = mixed, blocks [0, B) one-lane, rest in L chunks
    int64_t path_blk0 = 0;
    int cpep_keep = 0;          // "cpep_keep" / CUDE_CPEP_KEEP: 0 recompute, 1 keep the logistic derivative, 2 keep the upper layers
    int supp_store = -1;        // "supp_store" / CUDE_SUPP_STORE: -1 = size rule, 0 / 1 = never / always keep activations
    int supp_ckpt_steps = 0;    // "supp_ckpt" / CUDE_SUPP_CKPT: "steps" = keep step states only, re-run the stages in reverse
    int tape_steps = 0;         // "tape_steps" / CUDE_TAPE_STEPS: capacity of the adaptive gradient tape (0 = sized to ~4 GB)
    int exp_table = 1;          // "exp_table" / CUDE_NO_EXPTAB: layer-1 exponent recurrence along glucose pieces
    int ms_split = 1;           // "ms_split" / CUDE_NO_MS_SPLIT: restarts of a small population on the time-split kernels
    int train_host = 0;         // "train_host" / CUDE_TRAIN_HOST: 1 = cude_train_restarts keeps the L-BFGS vectors on the host
                                //   (what a sharded population always does), 2 = the Adam stage on the host as well
    int mh_spec = -1;           // "mh_spec" / CUDE_MH_SPEC: speculative Metropolis steps per launch chain (cude_mh_estep, gamma == 1,
                                //   time-split forward path): 0 = off, 2 ... 4 = that many, -1 = by population size (mh_spec_depth)
    int fit_spec = -1;          // "fit_spec" / CUDE_FIT_SPEC: cude_fit_conditional with several probes per forward launch (the grid scan
                                //   as parameter sets, the next d golden-section steps as a heap of brackets): 0 = one probe per launch,
                                //   1 ... 4 = that depth, -1 = by population size
    int adaptive_team = 1;      // "adaptive_team" / CUDE_NO_ADAPTIVE_TEAM: adaptive launches of small c-peptide populations put a
                                //   step's five network evaluations on five waves (cude_adaptive_team.hip)
    int auto_regroup = 1;       // "auto_regroup" / CUDE_NO_AUTO_REGROUP: adaptive launches re-ordered by accepted-step count
    int poll_pinned = 1;        // "poll_pinned" / CUDE_NO_POLL_PINNED: watch page-locked result slots instead of the stream wait
    int force_fallback = 0;     // "force_fallback" / CUDE_FORCE_FALLBACK (tests): cude_set_network takes the fallback kernel also
                                //   for shapes a tuned kernel is compiled for
    int debug_selector = 0;     // "debug_selector" / CUDE_DEBUG_SELECTOR: print the launch-path decision
    int xchg_allow_plain = 0;   // "xchg_allow_plain" / CUDE_ALLOW_PLAIN_MAILBOX: ordinary device memory as a mailbox although
                                //   peers sit on other devices (the owner's polls may then be served from its L2)
    int xchg_fail_kinds = 0;    // "xchg_fail_kinds" / CUDE_XCHG_FAIL_KINDS (tests): bit k set = this rank's cude_xchg_attach
                                //   reports a failure when its mailbox is of the k-th memory kind (0 uncached, 1 fine-grained, 2 plain)
    // ("hidden_activation" = tanh | relu | sigmoid, "output_activation" = softplus | identity: kept in cude_ctx::net)
    // ---- ablation
    int mixed = 1;              // CUDE_NO_MIXED
    int mixed_one_stream = 0;   // CUDE_MIXED_ONE_STREAM
    int fwd_split = 1;          // CUDE_NO_FWD_SPLIT
    int fused_final = 1;        // CUDE_NO_FUSED_FINAL
    int fused_tail = 1;         // CUDE_NO_FUSED_TAIL: tail of a time-split gradient evaluation in one launch (round 5)
    int scan_bulk = 1;          // CUDE_NO_SCAN_BULK: small scan launches fetch every row through LDS up front (round 5)
    int scan_map = 1;           // CUDE_NO_SCAN_MAP: the scan's adjoint recursion as a per-subject linear map (round 5)
    int mh_fuse = 1;            // CUDE_NO_MH_FUSE
    int mh_pair = 1;            // CUDE_NO_MH_PAIR: gamma < 1 -- the proposal and both possible next states in one launch (round 5)
    int graph = 1;              // CUDE_NO_GRAPH
    int graph_unroll = 8;       // CUDE_GRAPH_UNROLL
    int prio_shift = -1;        // CUDE_PRIO_SHIFT (-1 = the rule in prio_shift_for)
};

// Peer-write exchange (cude_comm.hip): this rank's mailbox, every rank's mailbox as mapped here, per-column sequence
// counters and the status word on the device.  ready: attached, self-tested and switched on.
struct Exchange {
    bool ready = false;         // the reductions of run_ensemble go through it
    bool attached = false;
    int kind = 0;               // how the mailbox was allocated: 3 uncached, 1 fine-grained, 0 plain device memory
    int kind_index = 0;         // position of `kind` in the order of preference (0 uncached, 1 fine-grained, 2 plain)
    uint64_t* box = nullptr;    // own mailbox [2][n_ranks][cols][2] words
    size_t box_words = 0;
    int cols = 0;
    uint64_t* peers[CUDE_XCHG_MAX_RANKS] = {};
    bool opened[CUDE_XCHG_MAX_RANKS] = {};      // mapped through hipIpcOpenMemHandle (to be closed)
    uint32_t* seq = nullptr;    // [cols]
    int32_t* status = nullptr;  // set by a wait that ran out of time: the device's address of status_host
    int32_t* status_host = nullptr;     // page-locked, coherent: the host reads it after any synchronisation, no copy
    double timeout_s = 20.0;
};

}  // namespace api
}  // namespace cude

// bit pattern the host puts into every slot of cude_ctx::pinned_pairs before a launch it is going to watch: a quiet NaN
// with a payload no arithmetic produces
constexpr uint64_t kPairSentinel = 0x7ff8dead5eed0001ull;

struct cude_ctx {
    cude_config cfg;
    cude::api::Options opt;
    cude::NetShape net;
    int P = 0;
    hipStream_t stream = nullptr;
    int64_t N = 0;          // local subjects
    double n_global = 0;    // subjects over all ranks
    int T = 0;
    std::vector<double> tp;
    bool have_pop = false, have_nn = false, have_cond = false;
    // population (CPEP)
    cude::api::DevBuf<double> k0, k1, k2, c0, dG, obs, age;
    // population (SUPP)
    cude::api::DevBuf<double> data, ckpt;
    double scale[3] = {1, 1, 1};
    // tables
    cude::api::DevBuf<int32_t> seg, obs_step, stepk;
    cude::api::DevBuf<double> phi, obs_w, stepd, tp_dev;
    cude::api::DevBuf<double> supp_rho, supp_obs_rho;   // SUPP, fixed step: state 1 per evaluation / observation relative to u1(t_0)
    double abstol = 1e-6, reltol = 1e-3;   // adaptive mode (n_steps == 0): OrdinaryDiffEq's defaults
    // parameters / gradients / optimiser
    cude::api::DevBuf<double> nn, cond, g_nn, g_cond, sse, auc, partials, traj;
    // chunked gradient path (cude_cpep2.hip)
    int chunks = 1;
    int n_cu = 256;         // compute units of the device (setup_chunks)
    int64_t blk0 = 0;       // > 0: mixed gradient launch -- blocks [0, blk0) on the one-lane kernel, the rest time-split
    int64_t slots_one = 0, half_slots = 0;          // resident-wave slots of the one-lane gradient kernel; one per SIMD
    hipStream_t stream2 = nullptr;                  // mixed launch: the time-split remainder runs beside the whole rounds
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    cude::api::DevBuf<double> param_mask;                      // frozen shared parameters (cude_set_param_mask); empty = none
    std::vector<double> mask_host;
    cude::api::DevBuf<int32_t> chunk_start;
    cude::api::DevBuf<double> hom_M, hom_obs, fsum, res, g_cond_part, partials2;
    // forward-only launches of the time-split path have their own split (chunks_f, 0 = the gradient's): the scan's cost
    // grows with the chunk count and there is no reverse kernel to feed, so fewer, longer chunks win there
    int chunks_f = 0;
    cude::api::DevBuf<int32_t> chunk_start_f;
    cude::api::DevBuf<double> hom_M_f, hom_obs_f, fsum_f;
    cude::api::DevBuf<double> m_nn, v_nn, m_cond, v_cond;
    int64_t nblocks = 0;
    double lr = 1e-3, b1 = 0.9, b2 = 0.999, eps = 1e-8;
    int64_t adam_t = 0;
    bool adam_ready = false;
    cude::api::DevBuf<double> adam_state, adam_trace;   // device-resident step state and per-iteration loss trace
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
    // comm: an RCCL communicator and / or the peer-write exchange (cude_comm.hip); either makes the context "distributed"
    void* comm = nullptr;
    int n_ranks = 1, rank = 0;
    cude::api::Exchange xchg;
    int32_t xchg_timeouts = 0;      // device-side waits of the exchange that gave up so far (cude_xchg_info)
    int xchg_next_kind = 0;         // first memory kind the next cude_xchg_export tries (advanced by a failed attach / a detach)
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
    bool poll_ok = true;            // cleared by the first watch that timed out: host memory is not coherent here
    bool tail_in_pinned = false;    // the tail reduction of the last launch also wrote [sum loss, n_failed] to pinned[P..P+1]
    double* pinned_pairs = nullptr; // page-locked [nblocks][2]: per-workgroup (sum SSE, failures) of a forward-only launch,
    int64_t pinned_pairs_n = 0;     // written by the scan kernel itself and added up by the host (finish_loss)
    bool loss_in_pinned = false;    // the last forward launch left its result there
    // scratch of cude_multistart_loss_grad (kept between calls: it is called once per optimiser iteration)
    cude::api::DevBuf<double> ms_nn, ms_cond, ms_part, ms_out, ms_gcond, ms_ckpt, ms_act;
    cude::api::DevBuf<double> ms_fsum, ms_wts, ms_gcp, ms_p2;   // time-split path with parameter sets (small populations)
    cude::api::DevBuf<double> act;     // SUPP: kept network activations of the gradient launch (small populations only)
    cude::api::DevBuf<double> tape, ms_tape;   // adaptive mode: accepted steps of the forward sweep, walked back by the adjoint
    cude::api::DevBuf<double> adj_map;             // time-split path: Cpep2Args::adj_map of the population
    cude::api::DevBuf<double> gen_acc, ms_gacc;    // general network (cude_generic.hip): [P][N] gradient accumulators per set
    cude::api::DevBuf<int32_t> tape_n;
    cude::api::DevBuf<int32_t> perm;                           // adaptive kernels: subject of every launch position (cude_adaptive_regroup)
    std::vector<int32_t> slot_of;                   // its inverse on the host (empty = identity)
    int64_t run_iters = 0;                          // iterations cude_adam_run has made on this population
    int64_t regroup_done_at = -1;                   // ... and the count at which it last re-ordered the launch by itself
    int tape_cap = 0;
    bool have_tape = false;         // the tape of the last gradient evaluation is readable (it is in the current launch order)
    bool have_counts = false;       // tape_n holds every subject's accepted-step count of some adaptive evaluation
    int64_t evals_since_regroup = 0;
    cude::api::DevBuf<double> red_tmp; // staging of small host vectors reduced through the communicator
    cude::api::DevBuf<double> ms_f;    // per-set loss values of a multi-set evaluation
    double scratch_budget = 0.0;       // bytes of scratch one multi-set launch may use (sets_scratch_budget)
    // cude_train_restarts (cude_train.hip): the restarts' optimiser state, resident between iterations
    cude::api::DevBuf<double> tr_m_nn, tr_v_nn, tr_m_cond, tr_v_cond, tr_trace;     // Adam moments [K][P] / [K][N], loss trace
    cude::api::DevBuf<int32_t> tr_alive, tr_act;
    cude::api::DevBuf<double> tr_x, tr_g, tr_d, tr_s, tr_y, tr_trial, tr_gtrial;    // L-BFGS vectors [R][N + P], history [R][m][N + P]
    cude::api::DevBuf<unsigned char> tr_state;
    void* tr_pinned = nullptr;         // page-locked copy of the L-BFGS states / alive flags the host reads once per round
    size_t tr_pinned_bytes = 0;
#ifdef CUDE_WAVE_TIMING
    cude::api::DevBuf<long long> dbg;
#endif
};

namespace cude {
namespace api {

// both c-peptide models share the population layout, solver tables and the ensemble kernel
inline bool is_cpep(const cude_ctx* c) { return c->cfg.model == CUDE_MODEL_CPEP || c->cfg.model == CUDE_MODEL_CPEP_SYM; }

inline bool adaptive(const cude_ctx* c) { return c->cfg.n_steps == 0; }
inline double step_size(const cude_ctx* c) { return adaptive(c) ? 0.0 : (c->tp.back() - c->tp.front()) / c->cfg.n_steps; }
inline bool distributed(const cude_ctx* c) { return c->comm != nullptr || c->xchg.ready; }

inline int32_t bind(cude_ctx* c) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(c->cfg.device));
    return CUDE_OK;
}

// ---- cude_context.hip
void interp_weights(double theta, double* w7);      // h-free dense-output weights b_i(theta) of Tsit5
int32_t apply_option(cude_ctx* c, const char* name, const char* value);
// ---- cude_launch.hip
cude::CpepArgs cpep_args(const cude_ctx* c);
cude::SuppArgs supp_args(const cude_ctx* c);
size_t supp_act_doubles(const cude_ctx* c);
bool supp_keep_activations(const cude_ctx* c, int64_t n_sets);
size_t cpep_act_doubles(const cude_ctx* c);
bool cpep_keep_activations(const cude_ctx* c);
int32_t ensure_tape(cude_ctx* c);
int32_t setup_chunks(cude_ctx* c);
int32_t run_ensemble(cude_ctx* c, bool grad, double* traj_dev, bool local_only = false, const double* cond_ov = nullptr,
                     double* sse_ov = nullptr);
int32_t finish_loss(cude_ctx* c, double* loss, double* g_nn_host);
int32_t adaptive_regroup(cude_ctx* c, int32_t* spread_before, int32_t* spread_after);
int32_t maybe_regroup(cude_ctx* c);
int32_t eval_sets_device(cude_ctx* c, int64_t n_sets, const double* nn, int64_t stride_nn, const double* cond,
                         int64_t stride_cond, double* g_cond, double* out);
// ---- cude_optimise.hip
void drop_graph(cude_ctx* c);
int32_t ensure_trace(cude_ctx* c, int64_t n);
cude::TailAdvance tail_advance(cude_ctx* c);
int32_t enqueue_adam(cude_ctx* c);
// ---- cude_comm.hip
// sum (op 0) / max (op 1) of a device vector over the ranks, in place, on the context's stream (identity on one rank)
int32_t allreduce_dev(cude_ctx* c, double* buf, size_t count, int op = 0);
int32_t comm_reduce_host(cude_ctx* c, double* values, int32_t count, int op);
int32_t lbfgs_comm_reduce(double* values, int32_t count, int32_t op, void* user);
cude::XchgArgs xchg_args(const cude_ctx* c);
int32_t xchg_check(cude_ctx* c);           // after a synchronisation: CUDE_ERR_COMM if a device-side wait ran out of time
void xchg_release(cude_ctx* c);
void comm_release(cude_ctx* c);

}  // namespace api
}  // namespace cude
