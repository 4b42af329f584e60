// The general network of `chain(widths, activation_functions; input_dims, output_activation)` for gfx950.
//
// Replaces (reference repo paths):
//   chain(widths::AbstractVector{Int}, activation_functions::AbstractVector{<:Function}; input_dims, output_dims = 1,
//         output_activation = softplus)                                       src/neural-network.jl:42-58
//   ... and its two convenience methods                                      src/neural-network.jl:85-87,105-107
//   every loss / gradient built on such a network: c-peptide cUDE            src/c-peptide-models.jl:86-104,170-194,
//                                                                             src/parameter-estimation.jl:56-68,126-140
//                                                  suppression cUDE          suppression/src/suppression_model.jl:88-130
// for the shapes the tuned kernels are not compiled for: any widths (e.g. the docstring's [10, 20, 30]), any of tanh /
// relu / sigmoid / softplus / identity PER LAYER, any of them at the output.  Parameter vector as SimpleChains lays it
// out: per layer [vec_colmajor(W) (out x in); b].
//
// This is the fallback, written for generality, not for speed (the tuned kernels keep weights in SGPRs and 37-67
// gradient accumulators in registers; a 4-10-20-30-1 network has 931 parameters):
//   * one lane = one subject, fixed-step AND adaptive Tsit5 in one body (S > 0: constant step, the observation tables
//     of the fixed-step kernels; S == 0: OrdinaryDiffEq's controller, initial-step heuristic and `saveat` interpolant
//     as in cude_adaptive.hip);
//   * the shared weights are staged ONCE per workgroup in LDS (every lane reads the same address: a broadcast); a
//     lane's activations live in its own LDS column (inputs, every layer's outputs, two delta buffers);
//   * GRAD: the forward sweep writes (t_n, dt_n, y_n) of every (accepted) step to a tape in HBM; the reverse sweep
//     walks it backwards, re-runs the seven stages of the step from y_n and applies their VJPs in reverse order.  A
//     lane's P gradient accumulators live in HBM, [parameter][subject] (coalesced read-modify-write per VJP), and are
//     summed over the workgroup at the end into the same [nblocks][P + 2] partial rows as every other kernel's.
// Activation derivatives are functions of the layer OUTPUT (tanh: 1 - h^2, relu: h > 0, sigmoid: h (1 - h), softplus:
// 1 - exp(-h), identity: 1), so nothing but the outputs is kept.
#include "cude_adaptive.h"

namespace cude {

namespace {

// ---------------------------------------------------------------------------------- the network on LDS columns
struct GenLds {
    const double* w;      // [P] staged weights
    double* col;          // this lane's column: entry e at col[e * kBlock]
    int d0, d1, dx;       // first entries of the two delta buffers and of the input gradient
};

__device__ __forceinline__ double gen_act(int kind, double z) {
    switch (kind) {
        case kGenActTanh: return m_tanh(z);
        case kGenActRelu: return fmax(z, 0.0);
        case kGenActSigmoid: {
            const double e = m_exp(-fabs(z));
            const double r = 1.0 / (1.0 + e);
            return z >= 0.0 ? r : e * r;
        }
        case kGenActSoftplus: return m_softplus_val(z);
        default: return z;
    }
}
__device__ __forceinline__ double gen_act_deriv(int kind, double h) {
    switch (kind) {
        case kGenActTanh: return fma(-h, h, 1.0);
        case kGenActRelu: return h > 0.0 ? 1.0 : 0.0;                    // (0 at the kink, as ForwardDiff's max)
        case kGenActSigmoid: return h * (1.0 - h);
        case kGenActSoftplus: return -expm1(-h);                          // logistic(z) = 1 - exp(-softplus(z))
        default: return 1.0;
    }
}

// inputs at col[0 .. nin); layer l's outputs behind its inputs; returns the (single) output
__device__ __forceinline__ double gen_forward(const GenNet& n, const GenLds& s) {
    int in_off = 0, K = n.nin, poff = 0;
    for (int l = 0; l < n.n_layers; l++) {
        const int J = n.width[l], out_off = in_off + K, kind = n.act[l];
        for (int j = 0; j < J; j++) {
            double z = s.w[poff + K * J + j];
            for (int k = 0; k < K; k++) z = fma(s.w[poff + k * J + j], s.col[(in_off + k) * kBlock], z);
            s.col[(out_off + j) * kBlock] = gen_act(kind, z);
        }
        poff += K * J + J;
        in_off = out_off;
        K = J;
    }
    return s.col[in_off * kBlock];
}

// Reverse pass behind gen_forward (activations still in the column): acc[q] += wgt * d out / d param_q, input gradient
// (times wgt) left at col[dx .. dx + nin).  acc: this lane's accumulators, entry q at acc[q * N]; nullptr = no
// parameter gradient wanted (inactive lanes: several of them stand in for the same subject).
__device__ __forceinline__ void gen_backward(const GenNet& n, const GenLds& s, double wgt, double* acc, int64_t N) {
    // offsets of the last layer
    int in_off = 0, K = n.nin, poff = 0;
    for (int l = 0; l + 1 < n.n_layers; l++) {
        poff += K * n.width[l] + n.width[l];
        in_off += K;
        K = n.width[l];
    }
    int dcur = s.d0, dnxt = s.d1;
    {   // delta of the output unit
        const int out_off = in_off + K;
        s.col[dcur * kBlock] = wgt * gen_act_deriv(n.act[n.n_layers - 1], s.col[out_off * kBlock]);
    }
    for (int l = n.n_layers - 1; l >= 0; l--) {
        const int J = n.width[l];
        for (int k = 0; k < K; k++) s.col[(dnxt + k) * kBlock] = 0.0;
        for (int j = 0; j < J; j++) {
            const double dj = s.col[(dcur + j) * kBlock];
            if (acc != nullptr) {
                double* ab = acc + (int64_t)(poff + K * J + j) * N;
                *ab += dj;
            }
            for (int k = 0; k < K; k++) {
                const double hin = s.col[(in_off + k) * kBlock];
                if (acc != nullptr) {
                    double* aw = acc + (int64_t)(poff + k * J + j) * N;
                    *aw = fma(dj, hin, *aw);
                }
                s.col[(dnxt + k) * kBlock] = fma(s.w[poff + k * J + j], dj, s.col[(dnxt + k) * kBlock]);
            }
        }
        if (l > 0) {
            const int Kp = l > 1 ? n.width[l - 2] : n.nin;      // inputs of layer l - 1
            const int kind = n.act[l - 1];
            for (int k = 0; k < K; k++)
                s.col[(dnxt + k) * kBlock] *= gen_act_deriv(kind, s.col[(in_off + k) * kBlock]);
            poff -= Kp * K + K;
            in_off -= Kp;
            K = Kp;
        } else {
            for (int k = 0; k < K; k++) s.col[(s.dx + k) * kBlock] = s.col[(dnxt + k) * kBlock];
        }
        const int t = dcur; dcur = dnxt; dnxt = t;
    }
}

// ---------------------------------------------------------------------------------- model policies
// c-peptide cUDE: f(t, u) = A u + [k0 c0 + P(dG(t)) - P(0); 0] (+ du3 = P(dG(t)) - P(0) with three states), network
// inputs [dG(t), e^beta (, age)]  (src/c-peptide-models.jl:7-14,30-42,86-104)
template <int NSTATE>
struct GenCpep {
    static constexpr int NS = NSTATE;
    static constexpr bool IS_CPEP = true;
    using Args = CpepArgs;
    double a11, a12, a21, a22, f0, base, eb;
    int64_t i, N;
    const double* dG;
    cptr_t tp;
    int TG;
    __device__ __forceinline__ double init(const Args& a, const GenNet& n, const GenLds& s, int64_t i_, int64_t set, double (&y)[NS]) {
        i = i_; N = a.N; dG = a.dG; tp = as_const(a.tp); TG = a.TG;
        const double k0 = a.k0[i], k1 = a.k1[i], k2 = a.k2[i], c0 = a.c0[i];
        a11 = -(k0 + k2); a12 = k1; a21 = k2; a22 = -k1; f0 = k0 * c0;
        eb = exp(a.cond[set * a.set_stride_cond + i]);
        double chk = eb * 0.0;
        s.col[1 * kBlock] = eb;
        if (n.nin > 2) { s.col[2 * kBlock] = a.age[i]; chk = fma(a.age[i], 0.0, chk); }
        for (int m = 0; m < TG; m++) chk = fma(dG[(int64_t)m * N + i], 0.0, chk);
        y[0] = c0;
        y[1] = (k2 / k1) * c0;
        if (NS > 2) y[2] = 0.0;
        base = 0.0;
        return chk;
    }
    // glucose(t) - glucose(t_0), linear between the knots (DataInterpolations.LinearInterpolation)
    __device__ __forceinline__ double forcing_input(double t) const {
        int j = 0;
        double tlo = tp[0], thi = tp[1];
        for (int m = 1; m < TG - 1; m++) {
            const double tm = tp[m];
            if (tm <= t) { j = m; tlo = tm; thi = tp[m + 1]; }
        }
        const double glo = dG[(int64_t)j * N + i], ghi = dG[(int64_t)(j + 1) * N + i];
        return fma(t - tlo, (ghi - glo) / (thi - tlo), glo);
    }
    __device__ __forceinline__ void set_inputs(const GenLds& s, double t, const double (&)[NS]) const { s.col[0] = forcing_input(t); }
    __device__ __forceinline__ void rhs(double out, const double (&u)[NS], double (&du)[NS]) const {
        const double q = out - base;
        du[0] = fma(a11, u[0], fma(a12, u[1], f0 + q));
        du[1] = fma(a21, u[0], a22 * u[1]);
        if (NS > 2) du[2] = q;
    }
    __device__ __forceinline__ double residual2(const Args& a, int oi, const double (&o)[NS], bool active) const {
        if (a.traj != nullptr && active) {
            double* tr = a.traj + (int64_t)NS * (oi + (int64_t)a.T * i);
            for (int s = 0; s < NS; s++) tr[s] = o[s];
        }
        if (a.obs == nullptr) return 0.0;
        const double r = o[0] - a.obs[(int64_t)oi * N + i];
        return r * r;
    }
    __device__ __forceinline__ void residual_bar(const Args& a, int oi, const double (&o)[NS], double (&ob)[NS]) const {
        ob[0] = 2.0 * (o[0] - a.obs[(int64_t)oi * N + i]);
        for (int s = 1; s < NS; s++) ob[s] = 0.0;
    }
    // weight of d out / d (params, inputs) in kb^T f, and the part of J_f^T kb that does not pass through the network
    __device__ __forceinline__ double vjp_linear(const double (&kb)[NS], double (&ub)[NS]) const {
        ub[0] += fma(a11, kb[0], a21 * kb[1]);
        ub[1] += fma(a12, kb[0], a22 * kb[1]);
        return NS > 2 ? kb[0] + kb[2] : kb[0];
    }
    // input gradient (already times the weight) -> state adjoint / conditional parameter
    __device__ __forceinline__ void vjp_inputs(const GenLds& s, double (&)[NS], double& gcond) const {
        gcond = fma(s.col[(s.dx + 1) * kBlock], eb, gcond);
    }
};

// suppression cUDE: f(u) = [-0.4 u1, 0.4 u1 - NN(u, e^theta), NN(u, e^theta) - 0.3 u3]  (suppression_model.jl:88-95)
struct GenSupp {
    static constexpr int NS = 3;
    static constexpr bool IS_CPEP = false;
    using Args = SuppArgs;
    double et, base;
    int64_t i, N;
    __device__ __forceinline__ double init(const Args& a, const GenNet&, const GenLds& s, int64_t i_, int64_t set, double (&y)[NS]) {
        i = i_; N = a.N;
        et = exp(a.cond[set * a.set_stride_cond + i]);
        s.col[3 * kBlock] = et;
        double chk = et * 0.0;
        for (int q = 0; q < 3; q++) {
            y[q] = a.data[((int64_t)q * a.T + 0) * N + i];
            chk = fma(y[q], 0.0, chk);
        }
        base = 0.0;
        return chk;
    }
    __device__ __forceinline__ void set_inputs(const GenLds& s, double, const double (&u)[NS]) const {
        for (int q = 0; q < 3; q++) s.col[q * kBlock] = u[q];
    }
    __device__ __forceinline__ void rhs(double out, const double (&u)[NS], double (&du)[NS]) const {
        du[0] = -0.4 * u[0];
        du[1] = fma(0.4, u[0], -out);
        du[2] = fma(-0.3, u[2], out);
    }
    __device__ __forceinline__ double residual2(const Args& a, int oi, const double (&o)[NS], bool active) const {
        double s2 = 0.0;
        for (int q = 0; q < 3; q++) {
            if (a.traj != nullptr && active) a.traj[q + 3 * (oi + (int64_t)a.T * i)] = o[q];
            const double r = o[q] - a.data[((int64_t)q * a.T + oi) * N + i];
            s2 = fma(r * a.iscale2[q], r, s2);
        }
        return s2;
    }
    __device__ __forceinline__ void residual_bar(const Args& a, int oi, const double (&o)[NS], double (&ob)[NS]) const {
        for (int q = 0; q < 3; q++) ob[q] = 2.0 * a.iscale2[q] * (o[q] - a.data[((int64_t)q * a.T + oi) * N + i]);
    }
    __device__ __forceinline__ double vjp_linear(const double (&kb)[NS], double (&ub)[NS]) const {
        ub[0] += fma(-0.4, kb[0], 0.4 * kb[1]);
        ub[2] += -0.3 * kb[2];
        return kb[2] - kb[1];
    }
    __device__ __forceinline__ void vjp_inputs(const GenLds& s, double (&ub)[NS], double& gcond) const {
        for (int q = 0; q < 3; q++) ub[q] += s.col[(s.dx + q) * kBlock];
        gcond = fma(s.col[(s.dx + 3) * kBlock], et, gcond);
    }
};

// ---------------------------------------------------------------------------------- the integrator
template <class M, bool GRAD>
__global__ __launch_bounds__(kBlock) void generic_kernel(typename M::Args a, GenNet net) {
    constexpr int NS = M::NS;
    constexpr int TROWS = 2 + NS;                  // tape entry: t_n, dt_n, y_n
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int P = net.n_params();
    const int64_t N = a.N;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < N;
    const int64_t i = active ? gid : N - 1;
    const int64_t set = blockIdx.y;
    // ---- LDS: [P] weights | per-lane columns (inputs, layer outputs, two delta buffers, input gradient) | reduction rows
    double* s_w = smem;
    GenLds s;
    {
        const double* nn = a.nn + set * a.set_stride_nn;
        for (int q = lane; q < P; q += kBlock) s_w[q] = nn[q];
        const int units = net.n_units(), mw = net.max_width() > net.nin ? net.max_width() : net.nin;
        s.w = s_w;
        s.col = smem + ((P + 63) & ~63) + lane;
        s.d0 = net.nin + units;
        s.d1 = s.d0 + mw;
        s.dx = s.d1 + mw;
        __syncthreads();
    }
    cptr_t tout = as_const(a.out_times);
    cptr_t obs_w = as_const(a.obs_w);
    ciptr_t obs_step = as_const(a.obs_step);
    const int n_out = a.T;
    const bool fixed = a.S > 0;

    M m;
    double y[NS];
    double chk = m.init(a, net, s, i, set, y);
    for (int q = 0; q < P; q++) chk = fma(s_w[q], 0.0, chk);       // NaN iff a parameter (or an input of this subject) is non-finite
    auto f = [&](double t, const double (&u)[NS], double (&du)[NS]) {
        m.set_inputs(s, t, u);
        m.rhs(gen_forward(net, s), u, du);
    };
    if constexpr (M::IS_CPEP) {                     // NN([0; e^beta (; age)]): time-invariant, evaluated once
        s.col[0] = 0.0;
        m.base = gen_forward(net, s);
    }
    const int cap = GRAD ? (fixed ? a.S : a.tape_cap) : 0;
    double* const tape = GRAD ? a.tape + (set * (int64_t)cap * TROWS) * N + i : nullptr;
#define TAPE(n, r) tape[((int64_t)(n) * TROWS + (r)) * N]

    const double abstol = a.abstol, reltol = a.reltol;
    const double t0 = a.t_begin, t1 = a.t_end;
    const double t_stop = t1 - 1e-14 * fmax(1.0, fabs(t1));
    double t = t0, dt = fixed ? a.h : 0.0, sse = chk;
    StepController ctl;
    int nxt = 0, n_acc = 0, n_steps = 0;
    bool failed = false;
    double K[7][NS];
    if (!fixed) {       // outputs at (or before) the initial time; (fixed-step: the tables place t_0 inside step 0, theta = 0)
        while (nxt < n_out && tout[nxt] <= t0 + 1e-12) {
            sse += m.residual2(a, nxt, y, active);
            nxt++;
        }
    }
    bool done = !(t < t_stop);
    f(t0, y, K[0]);
    if (!fixed && !done) {
        // Hairer's initial-step heuristic (as cude_adaptive.hip)
        double sk[NS], v0[NS], v1[NS], Y[NS], du[NS], v2[NS];
        for (int q = 0; q < NS; q++) {
            sk[q] = fma(reltol, fabs(y[q]), abstol);
            v0[q] = y[q] / sk[q];
            v1[q] = K[0][q] / sk[q];
        }
        const double d0 = rms(v0, NS), d1 = rms(v1, NS);
        dt = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        for (int q = 0; q < NS; q++) Y[q] = fma(dt, K[0][q], y[q]);
        f(t0 + dt, Y, du);
        for (int q = 0; q < NS; q++) v2[q] = (du[q] - K[0][q]) / sk[q];
        const double d2 = rms(v2, NS) / dt;
        const double dm = fmax(d1, d2);
        const double dt1 = dm <= 1e-15 ? fmax(1e-6, dt * 1e-3) : pow(0.01 / dm, 0.2);
        dt = fmin(fmin(100.0 * dt, dt1), t1 - t0);
    }
    // ------------------------------------------------------------------ forward sweep
    while (!__all(done || failed)) {
        const bool live = !done && !failed;
        if (!fixed) dt = fmin(dt, t1 - t);
        if (!live) dt = 0.0;
        double Y[NS];
        for (int st = 1; st < 7; st++) {
            for (int q = 0; q < NS; q++) {
                double accv = 0.0;
                for (int j = 0; j < st; j++) accv = fma(TS_A[st][j], K[j][q], accv);
                Y[q] = fma(dt, accv, y[q]);
            }
            f(st < 6 ? fma(TS_C[st], dt, t) : t + dt, Y, K[st]);
        }
        bool accept = true;
        if (!fixed) {
            double ev[NS];
            for (int q = 0; q < NS; q++) {
                double e = 0.0;
                for (int j = 0; j < 7; j++) e = fma(TS_BT[j], K[j][q], e);
                ev[q] = dt * e / fma(reltol, fmax(fabs(y[q]), fabs(Y[q])), abstol);
            }
            const double est = rms(ev, NS);
            if (live && !(fabs(est) <= 1.79769313486231570815e308)) failed = true;
            accept = ctl.judge(est);
            if (live && !failed) {
                n_steps++;
                if (n_steps >= kAdaptiveMaxSteps) failed = true;
            }
        }
        const bool commit = live && !failed && accept;
        if (commit) {
            if (fixed) {
                while (nxt < n_out && obs_step[nxt] == n_acc) {
                    double o[NS];
                    for (int q = 0; q < NS; q++) {
                        double v = 0.0;
                        for (int j = 0; j < 7; j++) v = fma(obs_w[nxt * 7 + j], K[j][q], v);
                        o[q] = fma(dt, v, y[q]);
                    }
                    sse += m.residual2(a, nxt, o, active);
                    nxt++;
                }
            } else {
                while (nxt < n_out && tout[nxt] <= t + dt + 1e-12) {
                    const double th = fmin(1.0, (tout[nxt] - t) / dt);
                    const bool at_end = fabs(th - 1.0) < 1e-12;
                    double o[NS];
                    for (int q = 0; q < NS; q++) {
                        double v = 0.0;
                        for (int j = 0; j < 7; j++) v = fma(saveat_weight(j, th, at_end), K[j][q], v);
                        o[q] = fma(dt, v, y[q]);
                    }
                    sse += m.residual2(a, nxt, o, active);
                    nxt++;
                }
            }
            if (GRAD) {
                if (n_acc < cap) {
                    if (active) {
                        TAPE(n_acc, 0) = t;
                        TAPE(n_acc, 1) = dt;
                        for (int q = 0; q < NS; q++) TAPE(n_acc, 2 + q) = y[q];
                    }
                } else {
                    failed = true;                    // more accepted steps than the tape holds
                }
            }
        }
        if (live && !failed) {
            if (accept) {
                n_acc++;
                t = fixed ? fma((double)n_acc, a.h, t0) : t + dt;
                for (int q = 0; q < NS; q++) { y[q] = Y[q]; K[0][q] = K[6][q]; }
                if (!fixed) dt = ctl.after_accept(dt);
                if (fixed ? n_acc >= a.S : !(t < t_stop)) done = true;
            } else {
                dt = ctl.after_reject(dt);
            }
        }
    }
    if (failed || nxt < n_out) sse = __builtin_nan("");      // failed solve => non-finite SSE => loss +Inf (reference :61-64)
    const bool bad = !(fabs(sse) <= 1.79769313486231570815e308);
    if (active && a.sse != nullptr) a.sse[set * a.set_stride_cond + i] = sse;
    if constexpr (M::IS_CPEP) {
        if (NS > 2 && a.auc != nullptr && active && set == 0) a.auc[i] = y[NS - 1];
    }
    if (active && a.tape_n != nullptr && set == 0 && !fixed) a.tape_n[i] = n_acc;
    double* out = a.partials + ((int64_t)set * gridDim.x + blockIdx.x) * (P + 2);
    double* s_red = smem + ((P + 63) & ~63) + (int64_t)(s.dx + net.nin) * kBlock;
    if constexpr (!GRAD) {
        const double v2[2] = {active ? sse : 0.0, (active && bad) ? 1.0 : 0.0};
        block_reduce_store<2>(v2, s_red, out + P, lane);
    } else {
        // ------------------------------------------------------------------ reverse sweep over the tape
        double* const acc = active ? a.gen_acc + (set * (int64_t)P) * N + i : nullptr;     // entry q at acc[q * N]
        if (active)
            for (int q = 0; q < P; q++) acc[(int64_t)q * N] = 0.0;
        double lam[NS], gcond = 0.0, wsum = 0.0;
        for (int q = 0; q < NS; q++) lam[q] = 0.0;
        const double gs = a.inv_n;
        int hi = n_out;                            // observations [hi, n_out) are already accounted for
        int n_max = bad ? 0 : n_acc;               // (a failed subject contributes no gradient: its loss is +Inf anyway)
        const int n_own = n_max;
        for (int off = 32; off >= 1; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off, 64));
        for (int n = n_max - 1; n >= 0; n--) {
            const bool on = n < n_own;             // a lane with fewer steps idles (zero adjoints) until its own come up
            const int src = on ? n : 0;
            const double tn = active && n_own > 0 ? TAPE(src, 0) : t0, h = active && n_own > 0 ? TAPE(src, 1) : 0.0;
            double yn[NS];
            for (int q = 0; q < NS; q++) yn[q] = active && n_own > 0 ? TAPE(src, 2 + q) : y[q];
            // ---- the seven stages of the step again: inputs Y_j and derivatives k_j
            double Yj[7][NS], kb[7][NS], yb[NS];
            for (int q = 0; q < NS; q++) Yj[0][q] = yn[q];
            f(tn, Yj[0], K[0]);
            for (int st = 1; st < 7; st++) {
                for (int q = 0; q < NS; q++) {
                    double accv = 0.0;
                    for (int j = 0; j < st; j++) accv = fma(TS_A[st][j], K[j][q], accv);
                    Yj[st][q] = fma(h, accv, yn[q]);
                }
                f(st < 6 ? fma(TS_C[st], h, tn) : tn + h, Yj[st], K[st]);
            }
            // ---- seeds: y_{n+1} = y_n + h sum_j a(7,j) k_j carries lam; every observation inside the step its residual
            for (int j = 0; j < 7; j++)
                for (int q = 0; q < NS; q++) kb[j][q] = 0.0;
            for (int q = 0; q < NS; q++) yb[q] = on ? lam[q] : 0.0;
            for (int j = 0; j < 6; j++)
                for (int q = 0; q < NS; q++) kb[j][q] = fma(h * TS_A[6][j], yb[q], kb[j][q]);
            while (on && hi > 0 && (fixed ? obs_step[hi - 1] == n : tout[hi - 1] > tn + 1e-12)) {
                const int oi = hi - 1;
                double w[7], o[NS], ob[NS];
                if (fixed) {
                    for (int j = 0; j < 7; j++) w[j] = obs_w[oi * 7 + j];
                } else {
                    const double th = fmin(1.0, (tout[oi] - tn) / h);
                    const bool at_end = fabs(th - 1.0) < 1e-12;
                    for (int j = 0; j < 7; j++) w[j] = saveat_weight(j, th, at_end);
                }
                for (int q = 0; q < NS; q++) {
                    double v = 0.0;
                    for (int j = 0; j < 7; j++) v = fma(w[j], K[j][q], v);
                    o[q] = fma(h, v, yn[q]);
                }
                m.residual_bar(a, oi, o, ob);
                for (int q = 0; q < NS; q++) {
                    const double g = gs * ob[q];
                    yb[q] += g;
                    for (int j = 0; j < 7; j++) kb[j][q] = fma(h * w[j], g, kb[j][q]);
                }
                hi--;
            }
            // ---- stage VJPs in reverse order: k_j = f(t_j, Y_j), Y_j = y_n + h sum_{i<j} a(j,i) k_i
            for (int st = 6; st >= 0; st--) {
                double ub[NS];
                for (int q = 0; q < NS; q++) ub[q] = 0.0;
                const double wgt = m.vjp_linear(kb[st], ub);
                m.set_inputs(s, st == 0 ? tn : (st < 6 ? fma(TS_C[st], h, tn) : tn + h), Yj[st]);
                (void)gen_forward(net, s);
                gen_backward(net, s, wgt, on ? acc : nullptr, N);
                if (on) {
                    m.vjp_inputs(s, ub, gcond);
                    wsum += wgt;
                }
                for (int q = 0; q < NS; q++) {
                    yb[q] += ub[q];
                    for (int j = 0; j < st; j++) kb[j][q] = fma(h * TS_A[st][j], ub[q], kb[j][q]);
                }
            }
            if (on)
                for (int q = 0; q < NS; q++) lam[q] = yb[q];
        }
        if constexpr (M::IS_CPEP) {                 // the baseline term: - sum(weights) * d NN([0; e^beta]) / d (params, beta)
            s.col[0] = 0.0;
            (void)gen_forward(net, s);
            gen_backward(net, s, -wsum, acc, N);
            double ub[NS];
            for (int q = 0; q < NS; q++) ub[q] = 0.0;
            m.vjp_inputs(s, ub, gcond);
        }
        if (active) a.g_cond[set * a.set_stride_cond + i] = bad ? __builtin_nan("") : gcond;
        // ---- the workgroup's partial row: P gradient sums, sum of SSE, failures
        for (int q = 0; q < P; q++) {
            const double v = wave_sum(active ? acc[(int64_t)q * N] : 0.0);
            if (lane == 0) out[q] = v;
        }
        const double l0 = wave_sum(active ? sse : 0.0), l1 = wave_sum((active && bad) ? 1.0 : 0.0);
        if (lane == 0) { out[P] = l0; out[P + 1] = l1; }
    }
#undef TAPE
}

template <class M>
hipError_t launch_generic(const GenNet& net, bool grad, const typename M::Args& a, hipStream_t s) {
    const int64_t nblocks = (a.N + kBlock - 1) / kBlock;
    const unsigned n_sets = a.n_sets > 0 ? (unsigned)a.n_sets : 1u;
    const size_t lds = gen_lds_bytes(net);
    if (lds > kGenMaxLds) return hipErrorInvalidValue;
    if (grad && (a.gen_acc == nullptr || a.tape == nullptr)) return hipErrorInvalidValue;
    hipError_t e;
    if (grad) {
        if (lds > 65536 &&
            (e = hipFuncSetAttribute((const void*)generic_kernel<M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        hipLaunchKernelGGL((generic_kernel<M, true>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a, net);
    } else {
        if (lds > 65536 &&
            (e = hipFuncSetAttribute((const void*)generic_kernel<M, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        hipLaunchKernelGGL((generic_kernel<M, false>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a, net);
    }
    return hipGetLastError();
}

}  // namespace

size_t gen_lds_bytes(const GenNet& net) {
    const int mw = net.max_width() > net.nin ? net.max_width() : net.nin;
    const size_t per_lane = (size_t)net.nin + net.n_units() + 2 * (size_t)mw + net.nin;
    return sizeof(double) * ((((size_t)net.n_params() + 63) & ~(size_t)63) + (per_lane + kRedRows) * kBlock);
}

hipError_t launch_cpep_generic(const NetShape& net, int n_state, bool grad, const CpepArgs& a, hipStream_t s) {
    if (!net.generic() || (net.gen.nin != 2 && net.gen.nin != 3) || (net.gen.nin == 3 && a.age == nullptr))
        return hipErrorInvalidValue;
    if (n_state == 3) return launch_generic<GenCpep<3>>(net.gen, grad, a, s);
    if (n_state == 2) return launch_generic<GenCpep<2>>(net.gen, grad, a, s);
    return hipErrorInvalidValue;
}

hipError_t launch_supp_generic(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s) {
    if (!net.generic() || net.gen.nin != 4) return hipErrorInvalidValue;
    return launch_generic<GenSupp>(net.gen, grad, a, s);
}

}  // namespace cude
