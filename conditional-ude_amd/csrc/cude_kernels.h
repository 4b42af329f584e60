// Kernel argument blocks and host-side launch entry points (implemented in the .hip files).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cude_xchg.h"

namespace cude {

constexpr int kBlock = 64;     // one wave per workgroup: no cross-wave barrier on the path
constexpr int kMaxObs = 32;

// activation functions a network can be built with (`chain(widths, activations; output_activation)`,
// src/neural-network.jl:42-58); tanh / softplus are what every script of the reference uses
constexpr int kActHiddenTanh = 0, kActHiddenRelu = 1, kActHiddenSigmoid = 2, kActHiddenIdentity = 3;
constexpr int kActOutSoftplus = 0, kActOutIdentity = 1;

// The general network (cude_generic.hip): per-layer widths and activation functions, evaluated by the fallback kernel
// when no tuned kernel is compiled for the shape.  Activation codes = CUDE_ACT_* of include/cude.h.
constexpr int kGenMaxLayers = 9;                 // hidden layers + the output layer
constexpr int kGenActTanh = 0, kGenActRelu = 1, kGenActSigmoid = 2, kGenActSoftplus = 3, kGenActIdentity = 4;
constexpr size_t kGenMaxLds = 160 * 1024;        // a workgroup's LDS on gfx950
struct GenNet {
    int32_t n_layers = 0;                        // 0 = not a general network
    int32_t nin = 0;
    int32_t width[kGenMaxLayers] = {};           // units of layer l (the last layer: 1)
    int32_t act[kGenMaxLayers] = {};
    __host__ __device__ int n_params() const {
        int p = 0, fan = nin;
        for (int l = 0; l < n_layers; l++) { p += width[l] * fan + width[l]; fan = width[l]; }
        return p;
    }
    __host__ __device__ int n_units() const {
        int u = 0;
        for (int l = 0; l < n_layers; l++) u += width[l];
        return u;
    }
    __host__ __device__ int max_width() const {
        int m = 0;
        for (int l = 0; l < n_layers; l++) m = width[l] > m ? width[l] : m;
        return m;
    }
};
size_t gen_lds_bytes(const GenNet& net);         // LDS of one workgroup of the fallback kernel

struct NetShape {
    int nin, width, depth;
    int hact = kActHiddenTanh, oact = kActOutSoftplus;
    GenNet gen;                                  // gen.n_layers > 0: the fallback kernel evaluates THIS network
    bool generic() const { return gen.n_layers > 0; }
    bool general() const { return hact != kActHiddenTanh || oact != kActOutSoftplus; }
    bool symbolic() const { return width == 0 && !generic(); }   // analytic production p0*dG/(dG+k): P = 1
    int n_params() const {
        if (generic()) return gen.n_params();
        if (symbolic()) return 1;
        int p = 0, fan = nin;
        for (int l = 0; l < depth; l++) { p += width * fan + width; fan = width; }
        return p + fan + 1;
    }
};

// c-peptide cUDE (linear kinetics + state-independent NN forcing)
struct CpepArgs {
    int64_t N;
    const double* k0; const double* k1; const double* k2; const double* c0;   // [N]
    const double* dG;        // [T][N]  glucose(t_j) - glucose(t_0)
    const double* obs;       // [T][N]
    const double* age;       // [N] (covariate model) or nullptr
    const double* cond;      // [N]
    const double* nn;        // [P]
    const int32_t* seg;      // [S*5]  glucose segment of each distinct stage time
    const double* phi;       // [S*5]  fraction inside the segment
    const int32_t* obs_step; // [T]
    const double* obs_w;     // [T][7] h-free dense-output weights b_i(theta)
    // layer-1 exponent table of the network kernels: per step {kind in the forward sweep, kind in the reverse
    // sweep, glucose piece of the step} with kind 0 = the step straddles a knot (direct exponentials), 1 = first
    // step of a run inside one piece (build table + anchor), 2 = continues the run; and per step {fraction of
    // t_n and of t_{n+1} inside the piece, h / piece length}
    const int32_t* stepk;    // [S][3]
    const double* stepd;     // [S][3]
    int32_t T, S;
    double h, inv_n;
    double* sse;             // [N] or nullptr
    double* traj;            // [NS x T x N] column-major or nullptr
    double* auc;             // [N] or nullptr (NS == 3: cumulative secretion at t_end)
    double* g_cond;          // [n_sets][N] (grad)
    double* partials;        // [n_sets][nblocks][P+2]
    double* act;             // one-lane gradient kernel: [nblocks][5S+1][(D-1)W+1][64] kept activations of the forward
                             // sweep (tanh outputs of the hidden layers 2..D + the output unit's logistic derivative per
                             // evaluation), read back by the reverse sweep instead of re-evaluating those layers;
                             // nullptr = recompute.
    int32_t keep_mode;       // with act: 1 = only the logistic derivative is kept ([nblocks][5S+1][64]), 2 = all of the above
    // multi-start evaluation: n_sets parameter sets in the grid's y dimension, set k reads nn + k*set_stride_nn and
    // cond + k*set_stride_cond (and writes sse / g_cond + k*set_stride_cond); 0/0/0 for the single-set path
    int32_t n_sets;
    int64_t set_stride_nn, set_stride_cond;
    int32_t cond_raw;        // symbolic model only: 1 = k is the conditional itself, 0 = k = exp(conditional)
    // adaptive mode (S == 0, cude_adaptive.hip): glucose knot times, output times (T of them; obs[T][N] when the
    // outputs are the observations) and the solver tolerances
    const double* tp;        // [TG] knot times = the population's timepoints
    int32_t TG;
    const double* out_times; // [T]
    double t_begin, t_end;   // integration span = the population's time span
    double abstol, reltol;
    // mixed launch of a large population (single parameter set): the one-lane kernel takes blocks [0, blk_count)
    // (0 = all of them), the time-split kernels blocks [blk0, nblocks)
    int64_t blk0, blk_count;
    int32_t prio_shift;      // one-lane gradient kernel: co-resident waves alternate issue priority every 2^prio_shift
                             // evaluations (0 = off); set by the host for single-round launches with two waves per SIMD
    double* tape;            // adaptive gradient: [n_sets][tape_cap][N] step sizes dt_n of the accepted steps (+ T saved outputs)
    int32_t tape_cap;
    int32_t* tape_n;         // [N] accepted steps per subject of parameter set 0 (forward and gradient launches), or nullptr
    int32_t team;            // adaptive mode, small launches: 0 = a step's five evaluations on five waves (cude_adaptive_team.hip)
                             // where that kernel applies, -1 = never (option "adaptive_team" = 0)
    double* gen_acc;         // fallback kernel (cude_generic.hip), gradient: [n_sets][P][N] per-lane accumulators; its tape
                             // (CpepArgs::tape) is [n_sets][steps][2 + n_state][N] with steps = S, or tape_cap when S == 0
    const int32_t* perm;     // adaptive kernels: lane `gid` works on subject perm[gid] (nullptr = identity).  Lanes of a wave
                             // run as long as the slowest of them: cude_adaptive_regroup orders the subjects by their
                             // accepted-step counts so that a wave's lanes finish together.  The tape is kept in LANE order.
#ifdef CUDE_WAVE_TIMING
    long long* dbg;          // development builds only: [nblocks][4] = {start, end of forward, end, hw id} per wave
#endif
};

// SAEM E-step (Metropolis-Hastings)
// z == nullptr / MhArgs::u == nullptr: the draw comes from the counter-based generator (RngKey) instead of a host row
struct RngKey {
    uint64_t seed;             // Philox key
    int64_t subject_offset;    // global index of local subject 0 (draws do not depend on how subjects are sharded)
    int64_t step;              // index of the Metropolis step since cude_set_rng
};
struct MhArgs {
    int64_t N;
    double* p;                 // chain state (conditional parameters), updated in place
    const double* prop;        // proposals
    const double* u;           // uniform draws of this step
    const double* sse_new; double* sse_cur;
    int32_t carry_sse;         // gamma == 1: the accepted proposal IS the next state, so its SSE is carried over
    int64_t* accepted;         // per-subject acceptance counter
    double prior_mean, prior_sd, ll_const, inv_2s2, temperature, gamma;
    RngKey key;                // used when u == nullptr
};

// Speculative Metropolis (gamma == 1, carried SSE): the chain state after a step is one of two values and the draws are
// known in advance (counter-based on (seed, subject, step), or the caller's rows), so the 2^d - 1 proposals that d
// consecutive steps CAN make are evaluated in ONE ensemble launch (candidate = parameter set) and the accept / reject
// decisions resolved afterwards: the same chain bit for bit, one dependent launch chain per d steps instead of per step.
// Candidates form a binary heap: node v (1-based; level l = floor(log2 v)) proposes q(v) = s(v) + std * z_l from its
// state s(v); s(1) = the chain state, s(2v) = s(v) (rejected), s(2v + 1) = q(v) (accepted).  cand / sse_sets: [v - 1][N].
constexpr int kMhSpecMaxDepth = 4;
constexpr int kMhSpecMaxDepthBlend = 3;     // gamma < 1 (MhSpecArgs::blend): two parameter sets per node
struct MhSpecArgs {
    MhArgs mh;                 // p, sse_cur, accepted, the prior and likelihood constants; key.step is NOT used (steps below)
    int32_t depth_resolve;     // levels of `cand` / `sse_sets` to resolve into the chain now (0: none -- the first call)
    int32_t depth_next;        // levels of candidates to write for the next round (0: none -- the last call)
    int64_t step_resolve, step_next;   // Metropolis step index of level 0 of either (device draws: the counter's step)
    const double* sse_sets;    // SSE of the candidates being resolved
    double* cand;              // read (resolve), then overwritten with the next round's
    const double* z_rows;      // caller's normals of the NEXT round's steps, row l at z_rows + l * N; nullptr = device draws
    const double* u_rows;      // caller's uniforms of the steps being RESOLVED, likewise
    double proposal_std;
    double* samples;           // chain state after every resolved step, row l at samples + l * N; or nullptr
    int32_t blend;             // gamma < 1: every node carries its state AND its proposal (mh_spec_resolve_blend, cude_rng.h):
                               //   cand / sse_sets rows [v - 1] = state of node v, [nodes + v - 1] = its proposal; depth <= 3
};
hipError_t launch_mh_spec(const MhSpecArgs& a, hipStream_t s);

// chunked loss+gradient path (cude_cpep2.hip): the S steps of every subject are split into L chunks
struct Cpep2Args {
    CpepArgs base;
    int32_t L;
    const int32_t* chunk_start;  // [L+1] step boundaries
    double* hom_M;               // [L][4][N]   chunk transfer matrix d y_out / d y_in (static per population)
    double* hom_obs;             // [T][2][N]   d y1(tau) / d y_in(chunk of tau)            (static)
    double* fsum;                // [L][3+T][N] forced chunk response: v1, v2, quadrature, obs parts
    double* wts;                 // [5S][N]     adjoint weight of every network evaluation (gradient only)
    double* g_cond_part;         // [L][N]
    double* partials2;           // [L][nblocks][P]
    // fused Metropolis step (forward-only launches): the forward chunks evaluate at the PROPOSAL state + std * draw
    // instead of base.cond, and the scan accepts / rejects it right where the SSE is formed -- two launches per
    // Metropolis step instead of four (propose, forward, scan, accept)
    // forward-only calls on one parameter set and one rank: every scan workgroup also writes its (sum SSE, failures)
    // pair straight into page-locked host memory mapped into the device, final_host[2 * workgroup + {0, 1}], and the
    // host adds the pairs up in workgroup order behind the synchronisation it performs anyway: no reduction launch and
    // no copy kernel behind the scan (two dispatches per forward call instead of four).  nullptr = off.
    double* final_host;
    int32_t mh_fused;
    const double* mh_z;          // [N] normals of this step, or nullptr = device stream (mh.key)
    double mh_std;
    MhArgs mh;
    // speculative Metropolis round (MhSpecArgs): spec_slots = 2^d > 0 makes the scan launch hold ALL candidate sets of a
    // subject in one workgroup (lane = slot * (64 / spec_slots) + local subject; slot = candidate set, the last slot
    // idles) and resolve the round right behind the SSEs -- no reduction, no resolver launch
    int32_t spec_slots;
    MhSpecArgs spec;
    int32_t defer_chunk_sum;     // gradient launches: leave g_cond_part to the caller's launch_chunked_tail (no sum_chunks launch)
    // The scan's adjoint recursion as a per-subject LINEAR MAP (round 5): the stage-adjoint algebra of a step (J_f = A, fixed
    // h) maps (lam1, lam2, kap1, kap2) of step n + 1 and the residual seeds of step n to those of step n and to the step's
    // five network weights with coefficients that depend on the subject's kinetics alone.  Rows of N doubles, computed once
    // per population by running the algebra on unit inputs (cpep2_adjmap_kernel): [0, 16) Phi[out][in], [16, 36) W[weight][in],
    // [36 + 9 oi, 36 + 9 oi + 9) response of (lam1, lam2, kap1, kap2, w0 ... w4) to a unit seed of observation oi.
    // 36 multiply-adds of depth 4 per step instead of ~100 of depth ~25; nullptr = the stage-by-stage algebra.
    const double* adj_map;
    // > 0: a scan launch of at most this many workgroups requests every row it will read up front, through LDS
    // (cpep2_scan_bulk_kernel: one memory round trip instead of a dozen dependent ones); 0 = never
    int64_t scan_bulk_blocks;
};
constexpr int kAdjMapRows = 36;
inline int64_t adj_map_rows(int T) { return kAdjMapRows + 9 * (int64_t)T; }
hipError_t launch_cpep2_adjmap(const Cpep2Args& a, double* adj_map, hipStream_t s);
hipError_t cpep2_prepare();
bool cpep2_shape_supported(const NetShape& net, int n_state);
int cpep2_rev_waves_per_cu(const NetShape& net);
int cpep_grad_waves_per_cu(const NetShape& net, int n_state, int T);
int cpep_keep_values(const NetShape& net);
hipError_t launch_cpep2_homog(const Cpep2Args& a, hipStream_t s);
// forward (+ scan) only when !grad: per-subject SSE and the loss partials, no trajectory output
hipError_t launch_cpep2(const NetShape& net, int n_state, bool grad, const Cpep2Args& a, hipStream_t s);

// suppression cUDE (nonlinear: NN input is the state)
// rows of N doubles in one parameter set's gradient scratch
// (states 2 and 3 only: state 1 is parameter-independent and never stored, cude_supp.hip)
inline __host__ __device__ int64_t supp_ckpt_rows(int S, int T) { return (int64_t)(6 * S + 1) * 2 + 2 * (int64_t)T; }

struct SuppArgs {
    int64_t N;
    const double* data;      // [3][T][N]
    const double* cond;      // [N]
    const double* nn;        // [P]
    const int32_t* obs_step; // [T]
    const double* obs_w;     // [T][7]
    int32_t T, S;
    double h, inv_n;
    double iscale2[3];       // 1/scale_s^2
    double* ckpt;            // [n_sets][supp_ckpt_rows][N]: [6S+1][2] stage inputs of states 2, 3 (linearisation points of
                             // the reverse sweep), then [T][2] their residuals
    const double* rho;       // [6S+1] fixed-step mode: state 1 at evaluation e = u1(t_0) * rho[e] (Tsit5 on du1 = -0.4 u1:
                             // a constant of the step size alone), wave-uniform
    const double* obs_rho;   // [T] ... and at observation oi = u1(t_0) * obs_rho[oi] (its dense output)
    double* act;             // [n_sets][6S+1][D*W+1][N] kept network activations, or nullptr = recompute them
    int32_t ckpt_steps_only; // 1: ckpt holds only the step states [S+1][2][N]; the reverse sweep re-runs the stages
    double* sse;             // [N] or nullptr (already divided by scale^2)
    double* traj;            // [3 x T x N] column-major or nullptr
    double* g_cond;          // [n_sets][N]
    double* partials;        // [n_sets][nblocks][P+2]
    int32_t n_sets;          // multi-start evaluation, as CpepArgs
    int64_t set_stride_nn, set_stride_cond;
    const double* out_times; // adaptive mode (S == 0): [T] observation times
    double t_begin, t_end;
    double abstol, reltol;
    double* tape;            // adaptive gradient, as CpepArgs
    int32_t tape_cap;
    int32_t* tape_n;
    double* gen_acc;         // as CpepArgs
    const int32_t* perm;     // as CpepArgs
};

// the fallback kernel for networks no tuned kernel is compiled for (net.generic(); cude_generic.hip)
hipError_t launch_cpep_generic(const NetShape& net, int n_state, bool grad, const CpepArgs& a, hipStream_t s);
hipError_t launch_supp_generic(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s);
// returns hipSuccess, or hipErrorInvalidValue when the shape is not compiled in
hipError_t launch_cpep(const NetShape& net, int n_state, bool grad, const CpepArgs& a, hipStream_t s);
hipError_t launch_supp(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s);
bool cpep_shape_supported(const NetShape& net, int n_state);
bool supp_shape_supported(const NetShape& net);
int supp_grad_waves_per_cu(const NetShape& net);
// adaptive Tsit5 (grad: + the adjoint of the accepted step sequence, needs args.tape); launch_cpep / launch_supp
// route here when args.S == 0
hipError_t launch_cpep_adaptive(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s);
hipError_t launch_supp_adaptive(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s);
// rows of N doubles per accepted step on the adaptive gradient's tape.  Suppression model (3 states): (t_n, dt_n, y_n) and
// the inputs of stages 2..7 (states 2 and 3; state 1's follow from y_n in closed arithmetic) -- the linearisation points
// of the reverse sweep, which then needs no network evaluation of its own (the one-body kernel of cude_adaptive.hip
// re-runs the stages from y_n and leaves those rows unused).  C-peptide models (2 states, constant Jacobian): dt_n
// alone -- the reverse sweep needs the stage times only and steps back from the final time.
constexpr int kSuppTapeHead = 5;                   // t_n, dt_n, y_n[3]
constexpr int kSuppTapeRows = kSuppTapeHead + 6 * 2;
inline __host__ __device__ int adaptive_tape_rows(int n_state) { return n_state == 3 ? kSuppTapeRows : 1; }
// ... and of one parameter set's tape: the steps, then per observation the saved output (c-peptide: state 1) or the
// residual's derivative with respect to states 2 and 3 (suppression)
inline __host__ __device__ int64_t adaptive_tape_rows(int n_state, int cap, int T) {
    return (int64_t)cap * adaptive_tape_rows(n_state) + (n_state == 3 ? 2 : 1) * T;
}

// common kernels
// out[q] = sum_b partials[b][stride*b + q] (fixed order, deterministic) for q in [col0, col0+ncol); with n_sets > 1
// the same for every set k: partials + k*nblocks*stride -> out + k*stride
// Adam state advance + loss-trace entry, folded into the kernel that produces an iteration's final [sum loss, n_failed]
// (the reduction of the two tail columns, or the L2 term behind it) or run on its own before the update kernel.
struct TailAdvance {
    double* state = nullptr;   // {b1^t, b2^t, steps done, trace position}; nullptr: nothing to advance
    double b1 = 0.0, b2 = 0.0;
    double* trace = nullptr;   // [cap][2] = (sum loss, n_failed) per iteration
    int64_t cap = 0;
};
// adv (optional; the launch must cover the tail columns stride-2, stride-1 and not accumulate): the workgroup of the
// failure-count column also sums the loss column and advances the optimiser state with the pair
// xchg (optional; the launch's columns must be final, i.e. not be accumulated into by a later launch): every column's sum
// goes through the exchange and `out` receives the sum over all ranks; with adv the tail workgroup exchanges both tail
// columns itself (the loss column's workgroup leaves them to it)
hipError_t launch_reduce_cols(const double* partials, int64_t nblocks, int stride, int col0, int ncol, double* out,
                              hipStream_t s, int n_sets = 1, const double* mask = nullptr, int n_mask = 0,
                              int out_stride = 0 /* doubles between the sets' output rows; 0 = stride */,
                              bool accumulate = false /* add to out instead of overwriting it */,
                              const TailAdvance* adv = nullptr,
                              double* host_tail = nullptr /* page-locked [2]: the sums of the last two columns as well */,
                              const XchgArgs* xchg = nullptr);
// The whole tail of a time-split gradient evaluation as ONE launch (chunked_tail_kernel, cude_common.hip): the network
// gradient from the reverse chunks' rows, the loss / failure columns from the scan's rows (with the tail work of
// launch_reduce_cols), and the chunks' shares of d loss / d conditional added up (g_cond_part == nullptr: not that part).
struct ChunkedTailArgs {
    const double* partials2; int64_t rows2;    // [n_sets][rows2][P]
    const double* partials; int64_t rows;      // [n_sets][rows][P + 2]
    int P;
    double* out; int out_stride;               // [n_sets][out_stride >= P + 2]
    const double* mask; int n_mask;
    TailAdvance adv; double* host_tail; XchgArgs xchg;
    const double* g_cond_part; int L; int64_t N; double* g_cond; int64_t g_cond_set_stride;
};
hipError_t launch_chunked_tail(const ChunkedTailArgs& a, int n_sets, hipStream_t s);
// buf[0..count) <- sum (op 0) / max (op 1) over the ranks, in place, through the exchange (any count: columns in turn)
hipError_t launch_xchg_allreduce(const XchgArgs& x, double* buf, int64_t count, int op, hipStream_t s);
hipError_t launch_adam_advance(const TailAdvance& adv, const double* g_tail, hipStream_t s);
// out[2k], out[2k+1] = sum_b partials[k][b][col0], [col0+1]  for k < n_sets (multi-start screening)
hipError_t launch_reduce_sets(const double* partials, int n_sets, int64_t nblocks, int stride, int col0, double* out,
                              hipStream_t s);
// screening: per-set losses and the running top-k of `partialsortperm(losses, 1:n_keep)`
hipError_t launch_set_losses(int n_sets, const double* sums, const double* nn, int P, double lambda, double n_global,
                             double* loss, hipStream_t s);
struct TopkArgs {
    int n_keep, n_have, n_new;
    long long first;                 // global index of the chunk's first candidate
    const double* best_loss; const long long* best_idx;     // [n_have]
    const double* chunk_loss;                                // [n_new]
    double* work;                                            // [n_have + n_new]
    double* new_loss; long long* new_idx; int* sel_src;      // [n_keep]
};
hipError_t launch_topk_merge(const TopkArgs& a, int P, int64_t N, const double* old_nn, const double* old_cond,
                             const double* chunk_nn, const double* chunk_cond, double* new_nn, double* new_cond,
                             hipStream_t s);
// g_nn[q] += 2*lambda*nn[q];  out[P] += lambda*sum(nn^2)*n_global   (so that loss = out[P]/n_global)
hipError_t launch_l2_term(const double* nn, int P, double lambda, double n_global, double* out, hipStream_t s,
                          const double* mask = nullptr, const TailAdvance* adv = nullptr);
struct AdamArgs {
    int64_t N; int P;
    double* cond; double* m_cond; double* v_cond; const double* g_cond;
    double* nn; double* m_nn; double* v_nn; const double* g_nn;   // g_nn[P+1] = n_failed
    double lr, b1, b2, eps, c1, c2;   // c1 = 1-b1^t, c2 = 1-b2^t (filled on the device from `state`)
    double* state;                    // device: {b1^t, b2^t, steps done, trace position}, ALREADY advanced to this step
};
hipError_t launch_adam(const AdamArgs& a, hipStream_t s);

// ---- restarts trained side by side with their optimiser state on the device (cude_train.hip)
// Behind a multi-set gradient evaluation (eval_sets_device): per set k the L2 term in l2_term_kernel's arithmetic,
// the loss value, and -- for the Adam stage -- the set's liveness and its loss-trace entry.
struct FinishSetsArgs {
    int P;
    double* out;             // [K][P+2]: in [masked g_nn; sum SSE; failures]; g_nn gets its L2 term in place (g_dst == nullptr)
    const double* nn;        // set k's network parameters at nn + k * stride_nn
    int64_t stride_nn;
    double lambda, n_global;
    const double* mask;      // [P] or nullptr
    double* f;               // [K] loss values, +Inf for a set with a failed subject
    double* g_dst;           // optional: the finished network gradient goes to g_dst + k * g_stride instead
    int64_t g_stride;
    int32_t* alive;          // optional [K]: cleared for a set whose loss is not finite (the reference drops that restart)
    double* trace;           // optional [K][trace_len]: trace[k][trace_pos] = loss of a live set
    int64_t trace_len, trace_pos;
};
hipError_t launch_finish_sets(const FinishSetsArgs& a, int n_sets, hipStream_t s);
// Optimisers.Adam for K sets at once, in the arithmetic of cude::adam_update (cude_optim.h; no contraction): dead sets skipped
struct AdamSetsArgs {
    int64_t N; int P;
    double* cond; double* nn;                       // [K][N], [K][P]
    double* m_cond; double* v_cond; double* m_nn; double* v_nn;
    const double* g_cond;                           // [K][N]
    const double* out;                              // [K][P+2] (network gradients)
    const int32_t* alive;                           // [K]
    double lr, b1, b2, eps, c1, c2;                 // c = 1 - b^t of this step (host: std::pow)
};
hipError_t launch_adam_sets(const AdamSetsArgs& a, int n_sets, hipStream_t s);
// Optim's L-BFGS + BackTracking (cude_optim.h) with the vectors of R restarts on the device: one state per restart,
// one workgroup per restart and round.  Vector layout [conditional (N); network (P)].
constexpr int kLbfgsM = 10;
constexpr int kLbfgsFirst = 0, kLbfgsFinite = 1, kLbfgsArmijo = 2, kLbfgsDone = 3;
struct LbfgsState {
    int32_t phase, pseudo, it, calls, accepted, f_flat, converged, ls_it, n_eval, maxiters, line_search_failed, pad_;
    double f, f0, dphi0, a1, a2, phi1, g_tol;
    double rho[kLbfgsM];
};
struct LbfgsArgs {
    int64_t n;               // N + P
    LbfgsState* state;       // [R]
    const int32_t* act;      // [A]: restart of every active slot
    double* X; double* G; double* D;        // [R][n]: iterate, its gradient, search direction
    double* S; double* Y;                   // [R][kLbfgsM][n]: history ring
    double* trial;           // [A][n]: the points the next evaluation is asked for
    const double* g_trial;   // [A][n]: gradient at the trial points
    const double* f_trial;   // [A]
    double* trace;           // optional [.][trace_len]: row owner[r], entry trace_off + (accepted iterations before) = f
    const int32_t* owner;    // [R]
    int64_t trace_len, trace_off;
    int64_t lds_doubles;     // set by launch_lbfgs_feed: dynamic LDS of the launch, in doubles
};
hipError_t launch_lbfgs_trial(const LbfgsArgs& a, int n_active, hipStream_t s);
hipError_t launch_lbfgs_feed(const LbfgsArgs& a, int n_active, hipStream_t s);
// SAEM E-step (Metropolis-Hastings) helper kernels
hipError_t launch_mh_propose(int64_t N, const double* p, const double* z, RngKey key, double proposal_std, double* prop,
                             hipStream_t s);
// the draws themselves (normals[N], uniforms[N] of one step), for reproducing a device-generated chain elsewhere
hipError_t launch_rng_draws(int64_t N, RngKey key, double* normals, double* uniforms, hipStream_t s);
hipError_t launch_mh_accept(const MhArgs& a, hipStream_t s);
// gamma < 1: proposal and the two possible next states as three parameter sets [3][N]; the decision behind their solves
hipError_t launch_mh_blend_candidates(int64_t N, const double* p, const double* z, RngKey key, double proposal_std, double gamma,
                                      double* cand, hipStream_t s);
hipError_t launch_mh_accept_blend(const MhArgs& a, const double* cand, const double* sse, hipStream_t s);
// per-subject 1-D fits (cude_fit_conditional): device-resident search state, all arrays [N]
struct FitArgs {
    int64_t N;
    double* a; double* b; double* c; double* d;   // bracket and the two probes
    double* fc; double* best;                     // best grid value / index (scan), objective at the result (finish)
    const double* sse_c; const double* sse_d;     // per-subject SSE at the probes c and d
    double w, mu;                                 // penalty w (x - mu)^2
    double lower, step, gr;                       // grid = lower + k*step; golden ratio (sqrt(5)-1)/2
    int32_t n_grid;
};
// phase 0: grid scan update (k = grid index, x = its value); 1: bracket; 2: golden step (k = 1 on the last one);
// 3: objective at the result
hipError_t launch_fit(int phase, const FitArgs& a, int k, double x, hipStream_t s);
// several probes per forward launch (cude_common.hip): the grid scan's update over sets [k0, k0 + kn) of one launch; the
// golden section's next `depth` steps as a heap of brackets, 2 (2^depth - 1) probes in cand / sse_sets [set][N]
constexpr int kFitSpecMaxDepth = 4;
hipError_t launch_fit_grid_all(const FitArgs& a, int k0, int kn, const double* values, const double* sse_sets, hipStream_t s);
hipError_t launch_fit_tree(const FitArgs& a, int depth, int resolve, int final, double* cand, const double* sse_sets, hipStream_t s);
hipError_t launch_fill(int64_t N, double v, double* out, hipStream_t s);
hipError_t launch_fill_rows(int64_t N, int n_rows, const double* values, double* out, hipStream_t s);
// population preparation
hipError_t launch_prepare_cpep(int64_t N, int T, const double* glucose_tn, const double* cpep_tn, const double* age,
                               const uint8_t* t2dm, double* k0, double* k1, double* k2, double* c0, double* dG,
                               hipStream_t s);

}  // namespace cude
