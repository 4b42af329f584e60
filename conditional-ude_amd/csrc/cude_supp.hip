// Suppression conditional-UDE ensemble kernels for gfx950 (3-state nonlinear ODE whose NN input
// is the state): fixed-step Tsit5 forward + discrete adjoint with step-state checkpoints.
//
// Replaces (reference repo paths):
//   ude_lsup!                      suppression/src/suppression_model.jl:88-95
//   get_prob_func / simul          suppression/src/suppression_model.jl:97-115 (EnsembleThreads)
//   suppression_loss (+ its ForwardDiffSensitivity gradient)   :117-130, :155
//
// One lane = one subject.  The forward sweep stores the input of every stage evaluation (the
// linearisation points Y_i, 6 per step) to an HBM scratch laid out [evaluation][state][subject]
// (coalesced); the reverse sweep reloads them and applies the stage VJPs in reverse order
// without recomputing the forward stages: scratch traffic buys a third of the network evaluations
// (measured 2.32 -> 1.78 ms at 1e5 subjects).
//
// State 1 obeys du1 = -0.4 u1 (suppression_model.jl:91): it depends on no parameter and on no other state.
// Under fixed-step Tsit5 every stage input of it is u1(t_0) times a constant of the step size alone
// (Y_st[1] = u1_n rho_st(-0.4 h), u1_{n+1} = u1_n R(-0.4 h): the tableau applied to a scalar linear equation), so
//   * it is not carried through the Runge-Kutta sums: the stage input comes from a wave-uniform table
//     SuppArgs::rho[evaluation] (one scalar load), the observation from SuppArgs::obs_rho[observation];
//   * it has no adjoint at all: nothing that reaches d/d(network) or d/d(theta) passes through it (the VJP weight is
//     kb[3] - kb[2]; the first-layer column of input 1 would only feed u1's own adjoint chain);
//   * it is neither written to nor read from the scratch: 2 rows per evaluation instead of 3.
// It is still an INPUT of the network (value and first-layer weight gradient use it) and still part of the loss.
#include "cude_device.h"
#include "cude_kernels.h"

namespace cude {

// HA / OA: activation functions other than tanh / softplus (cude_device.h CUDE_GENERAL_ACTS) select the general network
template <int W, int D, int HA, int OA>
struct SuppNetSel { using type = SuppNetG<W, D, HA, OA>; };
template <int W, int D>
struct SuppNetSel<W, D, kActHiddenTanh, kActOutSoftplus> { using type = SuppNet<W, D>; };

template <int W, int D, int HA = kActHiddenTanh, int OA = kActOutSoftplus>
struct SuppRhs {
    using Net = typename SuppNetSel<W, D, HA, OA>::type;
    // derivatives of states 2 and 3 (du[0], du[1]); u = (u1, u2, u3)
    __device__ static __forceinline__ void f(cptr_t p, const double (&c)[W], const double (&u)[3], double (&du)[2]) {
        const double uh = Net::eval(p, c, u);
        du[0] = fma(0.4, u[0], -uh);
        du[1] = fma(-0.3, u[2], uh);
    }
    // (ub2, ub3) += rows 2, 3 of J_f(u)^T kb ; acc += (df/dparams)^T kb      [kb = (kb2, kb3): state 1 has no adjoint]
    template <class A>
    __device__ static __forceinline__ void vjp(cptr_t p, const double (&c)[W], const double (&u)[3],
                                               const double (&kb)[2], double (&ub)[2], A& acc) {
        const double wgt = kb[1] - kb[0];
        double dx[3] = {0.0, 0.0, 0.0};
        Net::template eval_grad<true, A, Net::kPinLayers, 1>(p, c, u, wgt, acc, dx);
        ub[0] += dx[1];
        ub[1] += fma(-0.3, kb[1], dx[2]);
    }
    // the same pair with the network activations kept by the forward sweep instead of recomputed in the reverse one
    __device__ static __forceinline__ void f_keep(cptr_t p, const double (&c)[W], const double (&u)[3], double (&du)[2],
                                                  double (&h)[D][W], double* sig) {
        const double uh = Net::eval_keep(p, c, u, h, sig);
        du[0] = fma(0.4, u[0], -uh);
        du[1] = fma(-0.3, u[2], uh);
    }
    template <class A>
    __device__ static __forceinline__ void vjp_kept(cptr_t p, const double (&u)[3], const double (&h)[D][W], double sig,
                                                    const double (&kb)[2], double (&ub)[2], A& acc) {
        const double wgt = kb[1] - kb[0];
        double dx[3] = {0.0, 0.0, 0.0};
        Net::template backward<true, A, Net::kPinLayers, 1>(launder(p), u, h, sig, wgt, acc, dx);
        ub[0] += dx[1];
        ub[1] += fma(-0.3, kb[1], dx[2]);
    }
};

// reverse sweep: the adjoint ub of stage input Y_ST = y_n + h sum_{j<ST} a(ST,j) k_j into the stage adjoints (rows j < ST
// of s_K, [stage][state][lane]); the same fma per term as a rolled loop over j (aj = h * a(ST,j) formed the same way), but
// the ST tableau entries arrive in one scalar round trip and the ST rows in one LDS round trip.  (The entries come from
// the constant table, not from literals: a 64-bit literal needs a VGPR pair, 21 of them are hoisted out of the time loop
// and the kernel -- 128 VGPRs of accumulators -- spills.)
template <int ST>
__device__ __forceinline__ void supp_propagate(double* s_K, int lane, double h, const double (&ub)[2]) {
    double kk[ST][2], aj[ST];
    cptr_t row = launder(as_const(&TS_A[ST][0]));      // (laundered: known entries would be folded back into literals)
#pragma unroll
    for (int j = 0; j < ST; j++) {
        aj[j] = row[j];
#pragma unroll
        for (int s = 0; s < 2; s++) kk[j][s] = s_K[(j * 2 + s) * kBlock + lane];
    }
#pragma unroll
    for (int j = 0; j < ST; j++) {
        const double haj = h * aj[j];
#pragma unroll
        for (int s = 0; s < 2; s++) s_K[(j * 2 + s) * kBlock + lane] = fma(haj, ub[s], kk[j][s]);
    }
}

// LDS rows of kBlock doubles (one per lane), states 2 and 3 only:
//   s_K [7][2]  stage derivatives k_i (forward) / their adjoints (reverse); after the time loops: reduction scratch
//   s_Y [7][2]  stage inputs Y_i of the step being reversed (YONLY only)
// 7 KB per wave (14 KB in the step-state mode; never less than the kRedRows the final reduction needs).  The residuals
// kept for the reverse sweep (2T doubles per subject) live behind the stage inputs in the HBM scratch, not in LDS.
constexpr int kSuppRowsK = 14 > kRedRows ? 14 : kRedRows;

// accumulator container per network shape: registers while they fit next to two resident waves, split otherwise
// (round 3: with the accumulator updates pinned behind their FMAs -- AccPin, cude_device.h -- the reference's 4-3x5-1
// network keeps all 64 accumulators in registers at 229-256 VGPRs = two waves per SIMD; the split container is for
// larger networks only)
#ifndef CUDE_SUPP_SPLIT_ABOVE
#define CUDE_SUPP_SPLIT_ABOVE 64
#endif
template <class Net, bool SPLIT = (Net::NACC > CUDE_SUPP_SPLIT_ABOVE)>
struct SuppAcc {
    struct type {
        double a[Net::NACC];
        __device__ __forceinline__ double& operator[](int q) { return a[q]; }
        __device__ __forceinline__ const double& operator[](int q) const { return a[q]; }
        __device__ __forceinline__ void pin(int q) { AccPin<double[Net::NACC]>::pin(a, q); }
    };
    static constexpr int rows = 0;
    __device__ static __forceinline__ void init(type&, double*) {}
};
template <class Net>
struct SuppAcc<Net, true> {
    using type = SplitAcc<Net, 0>;
    static constexpr int rows = type::NLDS;
    __device__ static __forceinline__ void init(type& acc, double* lds) { acc.lds = lds; }
};

// STORE (gradient only): the forward sweep also writes the D*W tanh outputs and the output unit's logistic derivative
// of every evaluation to HBM ([evaluation][value][subject]) and the reverse sweep reads them back instead of
// re-evaluating the network: 2/3 of the reverse sweep's instructions for 8*(D*W+1) bytes per evaluation and subject.
// Pays when the launch is latency-bound (few waves, e.g. the reference's 37 subjects); the host enables it when the
// buffer is small.
// YONLY (gradient only, SuppArgs::ckpt_steps_only): the forward sweep keeps only the step states y_0 ... y_S (744 B per
// subject at S = 30 instead of 4.3 KB of stage inputs) and the reverse sweep re-runs the six stage evaluations of the
// step it is reversing: the low-traffic / high-arithmetic end of the trade (profiles/r02/supp_scratch_tradeoff.txt).
// resident waves per SIMD the register allocator aims at: two for the gradient kernels of networks with up to 64
// accumulators (4-3x5-1: 256 VGPRs, 5 dwords spilled outside the time loops), otherwise no constraint
template <int W, int D, bool GRAD>
constexpr int supp_waves() {
#ifdef CUDE_SUPP_WAVES
    return CUDE_SUPP_WAVES;
#else
    return (GRAD && SuppNet<W, D>::NACC <= 64) ? 2 : 1;
#endif
}
template <int W, int D, bool GRAD, bool STORE, bool YONLY, int HA = kActHiddenTanh, int OA = kActOutSoftplus>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(supp_waves<W, D, GRAD>()))) void supp_kernel(SuppArgs a) {
    using R = SuppRhs<W, D, HA, OA>;
    using Net = typename R::Net;
    constexpr int P = Net::P;
    extern __shared__ double smem[];
    double* s_K = smem;
    double* s_Y = smem + kSuppRowsK * kBlock;
    double* s_S = smem + kSuppRowsK * (YONLY ? 2 : 1) * kBlock;     // [6] cold adjoint state of the reverse sweep
    double* s_red = s_K;

    const int lane = threadIdx.x;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < a.N;
    const int64_t i = active ? gid : a.N - 1;
    const int64_t N = a.N;
    const int64_t set = blockIdx.y;      // multi-start screening: one parameter set per grid row
    cptr_t p = as_const(a.nn + set * a.set_stride_nn);
    cptr_t obs_w = as_const(a.obs_w);
    cptr_t rho = as_const(a.rho);
    cptr_t obs_rho = as_const(a.obs_rho);
    ciptr_t obs_step = as_const(a.obs_step);
    const int S = a.S, T = a.T;
    const double h = a.h;
    double* ckpt = GRAD ? a.ckpt + set * (supp_ckpt_rows(S, T) * N) : nullptr;        // one scratch per parameter set
    double* res = GRAD ? ckpt + (int64_t)(6 * S + 1) * 2 * N : nullptr;               // [T][2][N] residuals of states 2, 3
    constexpr int NACT = D * W + 1;
    double* act = STORE ? a.act + set * ((int64_t)(6 * S + 1) * NACT * N) : nullptr;
    // rows of states 2 and 3 (s = 0, 1 below): state 1 is a table lookup (header comment)
#define KROW(j, s) s_K[((j) * 2 + (s)) * kBlock + lane]
#define YROW(j, s) s_Y[((j) * 2 + (s)) * kBlock + lane]
#define LAM(s) s_S[(s) * kBlock + lane]           /* adjoint of y_{n+1} */
#define KAP(s) s_S[(2 + (s)) * kBlock + lane]     /* adjoint of k_7 of step n from step n+1's use as k_1 */
#define YB(s) s_S[(4 + (s)) * kBlock + lane]      /* adjoint of y_n being assembled */

    double cst[1] = {exp(a.cond[set * a.set_stride_cond + i])};
    double c[W];
    Net::first_layer_offset(p, cst, c);
    // the stage rows are read six at a time with the tableau row zero-padded (below): they must hold finite numbers
#pragma unroll
    for (int j = 0; j < 7; j++)
#pragma unroll
        for (int s = 0; s < 2; s++) KROW(j, s) = 0.0;

    const double u10 = a.data[((int64_t)0 * T + 0) * N + i];     // u1(t_0): every later u1 is this times a table entry
    double y[2];
#pragma unroll
    for (int s = 0; s < 2; s++) y[s] = a.data[((int64_t)(s + 1) * T + 0) * N + i];

    // ------------------------------------------------------------------ forward
    // One network call site: evaluation e = 0 is k_1 of step 0; e = 6n+st (st = 1..6) is stage
    // st+1 of step n (st = 6: k_7 = f(y_{n+1}), reused as k_1 of the next step).
    double sse = fma(cst[0] + u10, 0.0, Net::param_check(p));   // NaN iff a parameter / theta / u1(t_0) is non-finite
    int oi = 0, n = 0, st = 0;
    if constexpr (Net::USES_TANH) tanh_tab_init(lane, !Net::LDS_BIAS);    // (here: its global read travels with the subject's own loads)
    Net::bias_init(a.nn + set * a.set_stride_nn, lane);
#pragma unroll 1
    for (int e = 0; e <= 6 * S; e++) {
        double u[3];
        u[0] = u10 * rho[e];
        if (st == 0) {
#pragma unroll
            for (int s = 0; s < 2; s++) u[1 + s] = y[s];
        } else {
            // all six stage rows and the (zero-padded) tableau row at once: one LDS and one scalar round trip per
            // evaluation instead of one pair per earlier stage
            double kk[6][2], aj[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                aj[j] = TS_A[st][j];
#pragma unroll
                for (int s = 0; s < 2; s++) kk[j][s] = KROW(j, s);
            }
            double t[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 6; j++)
#pragma unroll
                for (int s = 0; s < 2; s++) t[s] = fma(aj[j], kk[j][s], t[s]);
#pragma unroll
            for (int s = 0; s < 2; s++) u[1 + s] = fma(h, t[s], y[s]);
        }
        if (GRAD && !YONLY) {      // linearisation point of evaluation e = 6n+st, reloaded by the reverse sweep
#pragma unroll
            for (int s = 0; s < 2; s++) ckpt[((int64_t)e * 2 + s) * N + i] = u[1 + s];
        }
        if (GRAD && YONLY && (e == 0 || st == 6)) {     // step states only: y_0, then y_{n+1} at the end of step n
#pragma unroll
            for (int s = 0; s < 2; s++) ckpt[((int64_t)(e == 0 ? 0 : n + 1) * 2 + s) * N + i] = u[1 + s];
        }
        double du[2];
        if (STORE) {
            double hk[D][W], sg;
            R::f_keep(p, c, u, du, hk, &sg);
            double* dst = act + (int64_t)e * NACT * N + i;
#pragma unroll
            for (int l = 0; l < D; l++)
#pragma unroll
                for (int j = 0; j < W; j++) dst[(int64_t)(l * W + j) * N] = hk[l][j];
            dst[(int64_t)(D * W) * N] = sg;
        } else {
            R::f(p, c, u, du);
        }
#pragma unroll
        for (int s = 0; s < 2; s++) KROW(st, s) = du[s];
        if (e == 0) { st = 1; continue; }
        if (st < 6) { st++; continue; }
        // ---- end of step n: u = y_{n+1}, KROW(6) = k_7
        while (oi < T && obs_step[oi] == n) {
            double o[2] = {0.0, 0.0};
#pragma unroll 1
            for (int j = 0; j < 7; j++) {
                const double w = obs_w[oi * 7 + j];
#pragma unroll
                for (int s = 0; s < 2; s++) o[s] = fma(w, KROW(j, s), o[s]);
            }
            {   // state 1 at the observation: the dense output of the scalar linear equation, from the table
                const double ov = u10 * obs_rho[oi];
                const double r = ov - a.data[((int64_t)0 * T + oi) * N + i];
                sse = fma(r * a.iscale2[0], r, sse);
                if (a.traj != nullptr && active) a.traj[0 + 3 * (oi + (int64_t)T * i)] = ov;
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const double ov = fma(h, o[s], y[s]);
                const double r = ov - a.data[((int64_t)(s + 1) * T + oi) * N + i];
                sse = fma(r * a.iscale2[s + 1], r, sse);
                if (GRAD) res[(int64_t)(oi * 2 + s) * N + i] = r;
                if (a.traj != nullptr && active) a.traj[(s + 1) + 3 * (oi + (int64_t)T * i)] = ov;
            }
            oi++;
        }
#pragma unroll
        for (int s = 0; s < 2; s++) { y[s] = u[1 + s]; KROW(0, s) = du[s]; }
        st = 1;
        n++;
    }
    const bool failed = !(fabs(sse) <= 1.79769313486231570815e308);
    if (active && a.sse != nullptr) a.sse[set * a.set_stride_cond + i] = sse;
    const double red_loss = active ? sse : 0.0;
    const double red_fail = (active && failed) ? 1.0 : 0.0;
    double* out = a.partials + ((int64_t)set * gridDim.x + blockIdx.x) * (P + 2);

    if (!GRAD) {
        const double v2[2] = {red_loss, red_fail};
        block_reduce_store<2>(v2, s_red, out + P, lane);
        return;
    } else {
        // -------------------------------------------------------------- reverse sweep
        // first- and output-layer accumulators in LDS rows behind the adjoint state where the register file cannot hold
        // them all (measured at 1e5 subjects, 4-3x5-1: 1.563 -> 1.506 ms; also moving the last hidden layer's, which are
        // updated while the next weight group is in flight, costs more than it frees: 1.85 ms)
        typename SuppAcc<Net>::type acc;
        SuppAcc<Net>::init(acc, s_S + 6 * kBlock + lane);
#pragma unroll
        for (int q = 0; q < Net::NACC; q++) acc[q] = 0.0;
        // lam / kap / yb are touched once per evaluation or per step: they live in LDS (every VGPR saved is one less
        // accumulator shuttled through the AGPRs)
#pragma unroll
        for (int s = 0; s < 2; s++) { LAM(s) = 0.0; KAP(s) = 0.0; YB(s) = 0.0; }
        const double gs = 2.0 * a.inv_n;
        oi = T - 1;
        n = S - 1;
        st = 6;
        // VJP evaluations in reverse order: idx = 6n+st (st = 6..1) is stage st+1 of step n;
        // idx = 0 is k_1 of step 0 = f(y_0).
        // stage inputs of the NEXT evaluation are requested one evaluation ahead: each is an HBM round trip that the
        // evaluation would otherwise start with (evaluation 0's input y_0 is row 0 of the same scratch)
        constexpr bool kAhead = !YONLY && !STORE;       // (with kept activations the extra live registers cost more)
        double un[2] = {0.0, 0.0};
        if (kAhead) {
#pragma unroll
            for (int s = 0; s < 2; s++) un[s] = ckpt[((int64_t)(6 * S) * 2 + s) * N + i];
        }
#pragma unroll 1
        for (int idx = 6 * S; idx >= 0; idx--) {
            if (YONLY && idx > 0 && st == 6) {
                // ---- re-run stages 1..6 of step n from y_n: their inputs Y_0..Y_5 and Y_6 = y_{n+1} go to s_Y
                double yn[2];
#pragma unroll
                for (int s = 0; s < 2; s++) yn[s] = ckpt[((int64_t)n * 2 + s) * N + i];
#pragma unroll 1
                for (int sq = 0; sq <= 6; sq++) {
                    double uu[3];
                    uu[0] = u10 * rho[6 * n + sq];      // (sq = 0: y_n, the input of evaluation 6n)
                    if (sq == 0) {
#pragma unroll
                        for (int s = 0; s < 2; s++) uu[1 + s] = yn[s];
                    } else {
                        double t[2] = {0.0, 0.0};
#pragma unroll 1
                        for (int j = 0; j < sq; j++) {
                            const double aj = TS_A[sq][j];
#pragma unroll
                            for (int s = 0; s < 2; s++) t[s] = fma(aj, KROW(j, s), t[s]);
                        }
#pragma unroll
                        for (int s = 0; s < 2; s++) uu[1 + s] = fma(h, t[s], yn[s]);
                    }
#pragma unroll
                    for (int s = 0; s < 2; s++) YROW(sq, s) = uu[1 + s];
                    if (sq == 6) break;
                    double dd[2];
                    R::f(p, c, uu, dd);
#pragma unroll
                    for (int s = 0; s < 2; s++) KROW(sq, s) = dd[s];
                }
            }
            if (idx > 0 && st == 6) {
                // ---- seed the stage adjoints (stored over the stage derivatives)
#pragma unroll 1
                for (int j = 0; j < 6; j++) {
#pragma unroll
                    for (int s = 0; s < 2; s++) KROW(j, s) = 0.0;
                }
#pragma unroll
                for (int s = 0; s < 2; s++) { KROW(6, s) = KAP(s); YB(s) = 0.0; }
                while (oi >= 0 && obs_step[oi] == n) {
                    double hg[2];
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const double g = gs * a.iscale2[s + 1] * res[(int64_t)(oi * 2 + s) * N + i];
                        YB(s) += g;
                        hg[s] = h * g;
                    }
#ifdef CUDE_SUPP_ROLLED_REVERSE
#pragma unroll 1
                    for (int j = 0; j < 7; j++) {
                        const double w = obs_w[oi * 7 + j];
#pragma unroll
                        for (int s = 0; s < 2; s++) KROW(j, s) = fma(w, hg[s], KROW(j, s));
                    }
#else
                    {   // the seven interpolation weights in one scalar round trip, the rows in one LDS round trip
                        double w[7], kk[7][2];
#pragma unroll
                        for (int j = 0; j < 7; j++) {
                            w[j] = obs_w[oi * 7 + j];
#pragma unroll
                            for (int s = 0; s < 2; s++) kk[j][s] = KROW(j, s);
                        }
#pragma unroll
                        for (int j = 0; j < 7; j++)
#pragma unroll
                            for (int s = 0; s < 2; s++) KROW(j, s) = fma(w[j], hg[s], kk[j][s]);
                    }
#endif
                    oi--;
                }
            }
            double u[3], kb[2], ub[2];
            u[0] = u10 * rho[idx];
            if (idx > 0) {
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    u[1 + s] = YONLY ? YROW(st, s) : (kAhead ? un[s] : ckpt[((int64_t)idx * 2 + s) * N + i]);
                    kb[s] = KROW(st, s);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    u[1 + s] = kAhead ? un[s] : a.data[((int64_t)(s + 1) * T + 0) * N + i];
                    kb[s] = KAP(s);
                }
            }
            if (kAhead && idx > 0) {
#pragma unroll
                for (int s = 0; s < 2; s++) un[s] = ckpt[((int64_t)(idx - 1) * 2 + s) * N + i];
            }
            if (idx > 0 && st == 6) {
#pragma unroll
                for (int s = 0; s < 2; s++) ub[s] = LAM(s);     // stage 7 adds into the adjoint of y_{n+1}
            } else {
#pragma unroll
                for (int s = 0; s < 2; s++) ub[s] = 0.0;
            }
            if (STORE) {
                double hk[D][W];
                const double* src = act + (int64_t)idx * NACT * N + i;
#pragma unroll
                for (int l = 0; l < D; l++)
#pragma unroll
                    for (int j = 0; j < W; j++) hk[l][j] = src[(int64_t)(l * W + j) * N];
                R::vjp_kept(p, u, hk, src[(int64_t)(D * W) * N], kb, ub, acc);
            } else {
                R::vjp(p, c, u, kb, ub, acc);
            }
            if (idx == 0) break;
            // propagate ub through  Y_st = y_n + h sum_{j<st} a(st,j) k_j
#pragma unroll
            for (int s = 0; s < 2; s++) YB(s) += ub[s];
#ifdef CUDE_SUPP_ROLLED_REVERSE
            const int nj = st < 6 ? st : 6;
#pragma unroll 1
            for (int j = 0; j < nj; j++) {
                const double aj = h * TS_A[st][j];
#pragma unroll
                for (int s = 0; s < 2; s++) KROW(j, s) = fma(aj, ub[s], KROW(j, s));
            }
#else
            // one body per stage (st is wave-uniform): the st rows read together, the tableau row as literals, the rows
            // written together -- one LDS round trip and no scalar load per evaluation instead of one pair per term
            switch (st) {
                case 1: supp_propagate<1>(s_K, lane, h, ub); break;
                case 2: supp_propagate<2>(s_K, lane, h, ub); break;
                case 3: supp_propagate<3>(s_K, lane, h, ub); break;
                case 4: supp_propagate<4>(s_K, lane, h, ub); break;
                case 5: supp_propagate<5>(s_K, lane, h, ub); break;
                default: supp_propagate<6>(s_K, lane, h, ub); break;
            }
#endif
            if (st > 1) {
                st--;
            } else {
#pragma unroll
                for (int s = 0; s < 2; s++) { LAM(s) = YB(s); KAP(s) = KROW(0, s); }
                st = 6;
                n--;
            }
        }
        __syncthreads();                   // s_red aliases s_K, which the reverse sweep has just been using
        const double cst_end[1] = {exp(a.cond[set * a.set_stride_cond + i])};      // not kept live across the sweeps
        if (active) a.g_cond[set * a.set_stride_cond + i] = Net::grad_cond(p, acc, cst_end);
        block_reduce_expand<Net, 1>(acc, cst_end, active ? 1.0 : 0.0, red_loss, red_fail, s_red, out, lane);
    }
#undef KROW
#undef YROW
#undef LAM
#undef KAP
#undef YB
}

template <int W, int D, bool GRAD, bool STORE, bool YONLY = false, int HA = kActHiddenTanh, int OA = kActOutSoftplus>
static hipError_t launch_one(const SuppArgs& a, hipStream_t s) {
    const int64_t nblocks = (a.N + kBlock - 1) / kBlock;
    const size_t lds = sizeof(double) * (size_t)(kSuppRowsK * (YONLY ? 2 : 1) +
                                                 (GRAD ? 6 + SuppAcc<typename SuppRhs<W, D, HA, OA>::Net>::rows : 0)) * kBlock;
    if (a.rho == nullptr || a.obs_rho == nullptr) return hipErrorInvalidValue;
    const unsigned n_sets = a.n_sets > 0 ? (unsigned)a.n_sets : 1u;
    hipLaunchKernelGGL((supp_kernel<W, D, GRAD, STORE, YONLY, HA, OA>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

// the shape of the reference's experiment with the other activation functions (stage-input mode only)
#define CUDE_SUPP_GENERAL_SHAPES(X) X(3, 5) X(3, 3)
template <int W, int D>
static hipError_t launch_general(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s) {
#define Y(HA, OA)                                                                                 \
    if (net.hact == HA && net.oact == OA)                                                         \
        return grad ? launch_one<W, D, true, false, false, HA, OA>(a, s) : launch_one<W, D, false, false, false, HA, OA>(a, s);
    CUDE_GENERAL_ACTS(Y)
#undef Y
    return hipErrorInvalidValue;
}

#define CUDE_SUPP_SHAPES(X) X(3, 5) X(3, 2) X(4, 2) X(6, 2) X(5, 2) X(3, 3) X(8, 2) X(3, 4) X(4, 3) X(4, 4) X(5, 3) X(6, 3) X(3, 1) X(4, 1) X(6, 1) X(8, 1)

// resident waves per CU of the gradient kernel (stage-input mode; 0 = unknown): what the launch actually gets
template <int W, int D>
static int supp_grad_occupancy() {
    int n = 0;
    const size_t lds = sizeof(double) * (size_t)(kSuppRowsK + 6 + SuppAcc<SuppNet<W, D>>::rows) * kBlock;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, supp_kernel<W, D, true, false, false>, kBlock, lds) != hipSuccess)
        return 0;
    return n;
}
int supp_grad_waves_per_cu(const NetShape& net) {
    if (net.generic()) return 1;
    if (net.general()) return 4;
#define X(W, D) if (net.width == W && net.depth == D) return supp_grad_occupancy<W, D>();
    CUDE_SUPP_SHAPES(X)
#undef X
    return 0;
}

bool supp_shape_supported(const NetShape& net) {
    if (net.nin != 4 || net.generic()) return false;
    if (net.general()) {
        if (!general_acts_compiled(net.hact, net.oact)) return false;
#define X(W, D) if (net.width == W && net.depth == D) return true;
        CUDE_SUPP_GENERAL_SHAPES(X)
#undef X
        return false;
    }
#define X(W, D) if (net.width == W && net.depth == D) return true;
    CUDE_SUPP_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_supp(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s) {
    if (net.generic()) return launch_supp_generic(net, grad, a, s);                 // fixed-step and adaptive alike
    if (net.nin != 4) return hipErrorInvalidValue;
    if (a.S == 0) return launch_supp_adaptive(net, grad, a, s);
    if (net.general()) {
        if (a.ckpt_steps_only || a.act != nullptr) return hipErrorInvalidValue;
#define X(W, D) if (net.width == W && net.depth == D) return launch_general<W, D>(net, grad, a, s);
        CUDE_SUPP_GENERAL_SHAPES(X)
#undef X
        return hipErrorInvalidValue;
    }
#define X(W, D)                                                                                   \
    if (net.width == W && net.depth == D)                                                         \
        return !grad ? launch_one<W, D, false, false>(a, s)                                       \
                     : (a.ckpt_steps_only ? launch_one<W, D, true, false, true>(a, s)             \
                        : (a.act != nullptr ? launch_one<W, D, true, true>(a, s) : launch_one<W, D, true, false>(a, s)));
    CUDE_SUPP_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cude
