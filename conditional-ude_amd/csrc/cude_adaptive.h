// Adaptive Tsit5 on the device: tableau rows, tolerances and the model policies shared by the two integrators
// (cude_adaptive.hip: one network body walked through every phase; cude_adaptive_unrolled.hip: the stages of a step
// unrolled, for the constant-Jacobian models on short sampling grids).
#pragma once
#include "cude_device.h"
#include "cude_kernels.h"

namespace cude {

__device__ __constant__ const double TS_C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
__device__ __constant__ const double TS_BT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                                                 -0.1447110071732629, 0.5823571654525552, -0.45808210592918697,
                                                 0.015151515151515152};
__device__ __constant__ const double TS_R[7][4] = {{1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216},
                                                   {0.0, 0.13169999999999998, -0.2234, 0.1017},
                                                   {0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253},
                                                   {0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902},
                                                   {0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928},
                                                   {0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661},
                                                   {0.0, 1.5, -4.0, 2.5}};

constexpr int kAdaptiveMaxSteps = 100000;     // OrdinaryDiffEq's default maxiters
#ifndef CUDE_ADAPT_2W_NACC
#define CUDE_ADAPT_2W_NACC 64
#endif

// ---------------------------------------------------------------------------------- model policies
// c-peptide cUDE / symbolic model: f(t, u) = A u + [k0 c0 + q(t); 0],  q(t) = P(dG(t)) - P(0)
// accumulators pinned behind their updates in every network evaluation of the replay loop (Mlp::eval_grad, PIN)
#ifndef CUDE_ADAPTIVE_PIN
#define CUDE_ADAPTIVE_PIN 1
#endif
constexpr bool kAdaptivePin = CUDE_ADAPTIVE_PIN != 0;

template <class Net>
struct CpepAd {
    static constexpr int NS = 2;
    static constexpr int P = Net::P;
    using Args = CpepArgs;
    double a11, a12, a21, a22, f0, base;
    double c[Net::NCST];
    double cst0;
    const double* s_G;             // [TG][kBlock] glucose increments at the knots (LDS)
    cptr_t tp;
    int TG, lane;
    cptr_t p;
    static __device__ __forceinline__ int lds_rows(const Args& a) { return a.TG; }
    __device__ __forceinline__ double init(const Args& a, double* s_extra, int lane_, int64_t i, int64_t set, double (&y)[NS]) {
        constexpr int NC = Net::NC;
        lane = lane_;
        p = as_const(a.nn + set * a.set_stride_nn);
        tp = as_const(a.tp);
        TG = a.TG;
        const double k0 = a.k0[i], k1 = a.k1[i], k2 = a.k2[i], c0 = a.c0[i];
        a11 = -(k0 + k2); a12 = k1; a21 = k2; a22 = -k1; f0 = k0 * c0;
        double cst[NC];
        cst[0] = Net::cond_input(a.cond[set * a.set_stride_cond + i]);
        if (NC > 1) cst[1] = a.age[i];
        cst0 = cst[0];
        Net::first_layer_offset(p, cst, c);
        double* g = s_extra;
        double chk = fma(cst[0], 0.0, Net::param_check(p));
        if (NC > 1) chk = fma(cst[1], 0.0, chk);
        for (int m = 0; m < TG; m++) {
            const double v = a.dG[(int64_t)m * a.N + i];
            g[m * kBlock + lane] = v;
            chk = fma(v, 0.0, chk);
        }
        s_G = g;
        y[0] = c0;
        y[1] = (k2 / k1) * c0;
        base = 0.0;
        return chk;                               // NaN iff an input of this subject is non-finite
    }
    // network input at time t: glucose(t) - glucose(t_0), linear between the knots (DataInterpolations.LinearInterpolation)
    __device__ __forceinline__ double forcing_input(double t) const {
        int j = 0;
        double tlo = tp[0], thi = tp[1];
        for (int m = 1; m < TG - 1; m++) {
            const double tm = tp[m];
            if (tm <= t) { j = m; tlo = tm; thi = tp[m + 1]; }
        }
        const double glo = s_G[j * kBlock + lane], ghi = s_G[(j + 1) * kBlock + lane];
        return fma(t - tlo, (ghi - glo) / (thi - tlo), glo);
    }
    // production P(x) (the only network call site of the kernel goes through here)
    __device__ __forceinline__ double production(double x) const {
        const double xx[1] = {x};
        return Net::eval(p, c, xx);
    }
    __device__ __forceinline__ void finish_rhs(double prod, const double (&u)[NS], double (&du)[NS]) const {
        du[0] = fma(a11, u[0], fma(a12, u[1], f0 + (prod - base)));
        du[1] = fma(a21, u[0], a22 * u[1]);
    }
    __device__ __forceinline__ double residual2(const Args& a, int oi, int64_t i, const double (&o)[NS], bool active) const {
        if (a.traj != nullptr && active) {
            double* tr = a.traj + (int64_t)NS * (oi + (int64_t)a.T * i);
            tr[0] = o[0];
            tr[1] = o[1];
        }
        if (a.obs == nullptr) return 0.0;
        const double r = o[0] - a.obs[(int64_t)oi * a.N + i];
        return r * r;
    }
    // ---- gradient
    static constexpr int A0 = 0;                   // first state that has an adjoint
    static constexpr bool NEED_Y = false;          // J_f = A: the stage inputs are not linearisation points
    using NetT = Net;
    static constexpr int NCST = Net::NC;
    // d residual2 / d o
    __device__ __forceinline__ void residual_bar(const Args& a, int oi, int64_t i, const double (&o)[NS], double (&ob)[NS]) const {
        ob[0] = 2.0 * (o[0] - a.obs[(int64_t)oi * a.N + i]);
        ob[1] = 0.0;
    }
    // acc += wgt * d production(te) / d params  (the baseline term is collected by the caller)
    template <class A>
    __device__ __forceinline__ void vjp_net(double te, double wgt, A& acc) const {
        const double xx[1] = {forcing_input(te)};
        double dx[1] = {0.0};
        Net::template eval_grad<false, A, kAdaptivePin>(p, c, xx, wgt, acc, dx);
    }
    __device__ __forceinline__ void vjp_linear(const double (&kb)[NS], double (&ub)[NS]) const {
        ub[0] += fma(a11, kb[0], a21 * kb[1]);
        ub[1] += fma(a12, kb[0], a22 * kb[1]);
    }
    template <class A>
    __device__ __forceinline__ void finish_grad(const Args& a, int64_t i, int64_t set, A& acc, double wsum, double carry,
                                                double (&cst)[NCST]) const {
        double dx[1] = {0.0};
        // one call site for the two closing evaluations: k_1 of the first step (time t_0, weight carry), then the
        // baseline term  - sum(kb) * d NN([0; e^beta]) / d params
#pragma unroll 1
        for (int r = 0; r < 2; r++) {
            const double xx[1] = {r == 0 ? forcing_input(a.t_begin) : 0.0};
            Net::template eval_grad<false, A, kAdaptivePin>(p, c, xx, r == 0 ? carry : -wsum, acc, dx);
        }
        cst[0] = Net::cond_input(a.cond[set * a.set_stride_cond + i]);
        if (NCST > 1) cst[NCST - 1] = a.age[i];
    }
};

// suppression cUDE: f(u) = [-0.4 u1, 0.4 u1 - NN(u, e^theta), NN(u, e^theta) - 0.3 u3]
template <int W, int D, int HA = kActHiddenTanh, int OA = kActOutSoftplus>
struct SuppAd {
    static constexpr int NS = 3;
    using Net = Mlp<4, W, D, 3, false, false, HA, OA>;
    static constexpr int P = Net::P;
    using Args = SuppArgs;
    double c[W];
    cptr_t p;
    static __device__ __forceinline__ int lds_rows(const Args&) { return 0; }
    __device__ __forceinline__ double init(const Args& a, double*, int, int64_t i, int64_t set, double (&y)[NS]) {
        p = as_const(a.nn + set * a.set_stride_nn);
        double cst[1] = {exp(a.cond[set * a.set_stride_cond + i])};
        Net::first_layer_offset(p, cst, c);
        double chk = fma(cst[0], 0.0, Net::param_check(p));
#pragma unroll
        for (int s = 0; s < 3; s++) {
            y[s] = a.data[((int64_t)s * a.T + 0) * a.N + i];
            chk = fma(y[s], 0.0, chk);
        }
        return chk;
    }
    __device__ __forceinline__ double residual2(const Args& a, int oi, int64_t i, const double (&o)[NS], bool active) const {
        double s2 = 0.0;
#pragma unroll
        for (int s = 0; s < 3; s++) {
            if (a.traj != nullptr && active) a.traj[s + 3 * (oi + (int64_t)a.T * i)] = o[s];
            const double r = o[s] - a.data[((int64_t)s * a.T + oi) * a.N + i];
            s2 = fma(r * a.iscale2[s], r, s2);
        }
        return s2;
    }
    // ---- gradient
    // State 1 (du1 = -0.4 u1, suppression_model.jl:91) depends on no parameter and on no other state: nothing that
    // reaches d/d(network) or d/d(theta) passes through its adjoint (the VJP weight is kb[2] - kb[1]).  The forward
    // sweep integrates it like the others -- the step-size controller's error norm weighs it -- and the tape keeps it
    // (it is a network input); the reverse sweep carries adjoints for states 2 and 3 only.
    static constexpr int A0 = 1;
    static constexpr bool NEED_Y = true;
    using NetT = Net;
    static constexpr int NCST = 1;
    __device__ __forceinline__ void residual_bar(const Args& a, int oi, int64_t i, const double (&o)[NS], double (&ob)[NS]) const {
        ob[0] = 0.0;
#pragma unroll
        for (int s = 1; s < 3; s++) ob[s] = 2.0 * a.iscale2[s] * (o[s] - a.data[((int64_t)s * a.T + oi) * a.N + i]);
    }
    template <class A>
    __device__ __forceinline__ void vjp(double, const double (&u)[NS], const double (&kb)[NS], double (&ub)[NS], A& acc,
                                        double&) const {
        const double wgt = kb[2] - kb[1];
        double dx[3] = {0.0, 0.0, 0.0};
        Net::template eval_grad<true, A, kAdaptivePin, 1>(p, c, u, wgt, acc, dx);
        ub[1] += dx[1];
        ub[2] += fma(-0.3, kb[2], dx[2]);
    }
    template <class A>
    __device__ __forceinline__ void finish_grad(const Args& a, int64_t i, int64_t set, A&, double, double,
                                                double (&cst)[NCST]) const {
        cst[0] = exp(a.cond[set * a.set_stride_cond + i]);
    }

};

__device__ __forceinline__ double rms(const double* v, int n) {
    double s = 0.0;
    for (int k = 0; k < n; k++) s = fma(v[k], v[k], s);
    return sqrt(s / n);
}

// waves per SIMD the register allocation aims at: the solve is a latency chain, so a second resident wave is worth more
// than unrolling room, as long as the gradient accumulators (2 VGPRs each) leave space for it
template <class M, bool GRAD>
constexpr int adaptive_waves() {
    return !GRAD ? 2 : (M::NetT::NACC <= (M::NEED_Y ? 24 : CUDE_ADAPT_2W_NACC) ? 2 : 1);
}

// ---------------------------------------------------------------------------------- shared by the integrators
// weight of stage derivative j in the `saveat` output at theta = (t_out - t_n) / dt_n: Tsit5's free 4th-order interpolant,
// or -- at the end of the step -- the last tableau row (y_{n+1} itself)
__device__ __forceinline__ double saveat_weight(int j, double th, bool at_end) {
    return at_end ? (j < 6 ? TS_A[6][j < 6 ? j : 0] : 0.0)
                  : fma(fma(fma(TS_R[j][3], th, TS_R[j][2]), th, TS_R[j][1]) * th, th, TS_R[j][0] * th);
}

// OrdinaryDiffEq's PI controller (beta1 = 7/50, beta2 = 2/25, gamma = 0.9, qmin = 0.2, qmax = 10) on the scaled error
// estimate of a trial step.  q = est^(7/50) / qold^(2/25) with qold = max(previous est, 1e-4): both powers come from ONE
// logarithm per trial step -- est^(7/50) = exp(0.14 log est) now, and, when the step is accepted, qold^(2/25) of the
// NEXT step = exp(0.08 max(log est, log 1e-4)), carried in qold_pow (the two pow() calls per trial step were ~10 % of the
// forward sweep's instructions in round 2)
struct StepController {
    double qold_pow = 0.47863009232263831;         // (1e-4)^(2/25)
    double log_est = 0.0, q11 = 0.0;
    __device__ __forceinline__ bool judge(double est) {            // true: the step is accepted
        log_est = est > 0.0 ? log(est) : -1e3;                      // (est = 0: qold = 1e-4 below)
        q11 = est > 0.0 ? exp((7.0 / 50.0) * log_est) : 1e-12;
        return est <= 1.0;
    }
    __device__ __forceinline__ double after_accept(double dt) {    // the next step's size
        double q = q11 / qold_pow;
        q = fmax(1.0 / 10.0, fmin(1.0 / 0.2, q / 0.9));
        qold_pow = exp((2.0 / 25.0) * fmax(log_est, -9.21034037197618273607));     // log 1e-4
        return dt / q;
    }
    __device__ __forceinline__ double after_reject(double dt) const { return dt / fmin(1.0 / 0.2, q11 / 0.9); }
};

// a step's seven stage rows (derivatives, inputs or adjoints) in the unrolled kernels: registers (indices are literals
// after unrolling), or one LDS row each at a constant offset
template <int NS, bool IN_LDS>
struct StageRows {
    double v[7][NS];
    __device__ __forceinline__ StageRows(double*, int) {}
    __device__ __forceinline__ double get(int j, int s) const { return v[j][s]; }
    __device__ __forceinline__ void set(int j, int s, double x) { v[j][s] = x; }
};
template <int NS>
struct StageRows<NS, true> {
    double* row;
    __device__ __forceinline__ StageRows(double* rows, int lane) : row(rows + lane) {}
    __device__ __forceinline__ double get(int j, int s) const { return row[(j * NS + s) * kBlock]; }
    __device__ __forceinline__ void set(int j, int s, double x) { row[(j * NS + s) * kBlock] = x; }
};

constexpr int kUnrolledKnots = 5;                  // the reference's five sampling times (c-peptide/02-conditional.jl)
// (three groups: the two integrators compile them as separate translation units, cude_adaptive*.hip + -DCUDE_AD_PART=k)
#define CUDE_CPEP_AD_SHAPES_0(X) X(2, 4, 2) X(2, 6, 2) X(3, 4, 2) X(2, 8, 2) X(2, 4, 3) X(2, 3, 2) X(2, 5, 2) X(2, 7, 2)
#define CUDE_CPEP_AD_SHAPES_1(X) X(3, 6, 2) X(2, 4, 1) X(2, 6, 1) X(2, 6, 3) X(2, 8, 1) X(2, 8, 3) X(3, 8, 2) X(2, 3, 1)
#define CUDE_CPEP_AD_SHAPES_2(X) X(2, 5, 1) X(2, 7, 1) X(2, 3, 3) X(2, 5, 3) X(2, 7, 3) X(3, 4, 1) X(3, 6, 1) X(3, 4, 3)
#define CUDE_CPEP_AD_SHAPES(X) CUDE_CPEP_AD_SHAPES_0(X) CUDE_CPEP_AD_SHAPES_1(X) CUDE_CPEP_AD_SHAPES_2(X)
// cude_adaptive_unrolled.hip: the c-peptide MLP shapes above and the symbolic model on grids of at most kUnrolledKnots times; hipErrorNotSupported
// for any other shape (the caller then runs the phase-machine kernel)
hipError_t launch_cpep_adaptive_unrolled(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s);
// cude_adaptive_team.hip: the same solve with a step's five network evaluations on five waves, for launches of at most
// kTeamMaxWaves single-wave workgroups (the reference's own population sizes); hipErrorNotSupported otherwise
constexpr int kTeamMaxWaves = 256;
hipError_t launch_cpep_adaptive_team(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s);
// shape groups 1 and 2 of either integrator (their own translation units); hipErrorNotSupported = not in this group
hipError_t launch_cpep_adaptive_unrolled_part1(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s);
hipError_t launch_cpep_adaptive_unrolled_part2(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s);
hipError_t launch_cpep_adaptive_part1(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s);
hipError_t launch_cpep_adaptive_part2(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s);
// cude_adaptive_unrolled_supp.hip: the suppression model, shapes of the reference's experiments
#define CUDE_SUPP_AD_UNROLLED(X) X(3, 5) X(3, 3) X(3, 4) X(3, 2)
hipError_t launch_supp_adaptive_unrolled(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s);

}  // namespace cude
