// Adaptive Tsit5 ensemble kernels for gfx950, stages unrolled -- the suppression model
// (solve(ensemble, Tsit5(), EnsembleThreads(); saveat, trajectories = N), suppression/src/suppression_model.jl:113,123,
//  and its gradient under AutoForwardDiff, :155).
//
// Same integrator, same arithmetic in the same order as adaptive_kernel<SuppAd<W, D>> (cude_adaptive.hip), which walks
// one network body through every phase with the stage rows in LDS.  Here the six evaluations of a trial step and the six
// VJPs of a reversed step are unrolled: tableau entries are literals, stage derivatives and adjoints sit in registers.
// The gradient launch also keeps, per accepted step, the inputs of stages 2..7 (states 2 and 3: 96 B beside the 40 B of
// (t, dt, y); state 1's inputs follow from y_n by the same arithmetic as in the forward sweep) and, per observation, the
// residual's derivative: the reverse sweep then re-runs nothing -- six VJPs per step instead of seven evaluations and
// six VJPs -- and needs no LDS beyond the final reduction's.  Against the one-body kernel (tools/abl_adaptive_bits.py):
// losses, trajectories and accepted steps bit for bit, gradients to 3e-15 (the compiler fuses other multiply-add pairs).
// 1e5 subjects, 4x3x3x3x3x3x1, one box: forward 0.468 -> 0.388 ms; gradient 1.814 ms (one body) -> 1.107 ms (unrolled,
// stages re-run from y_n, stage rows in LDS) -> 0.819 ms (inputs kept on the tape).
#include "cude_adaptive.h"

namespace cude {

template <class M, bool GRAD>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2)))
void adaptive_unrolled_supp_kernel(SuppArgs a) {
    static_assert(M::NEED_Y && M::NS == 3, "the suppression model");
    constexpr int NS = 3;
    constexpr int P = M::P;
    constexpr int TROWS = kSuppTapeRows;           // tape entry: t_n, dt_n, y_n, inputs of stages 2..7 (states 2, 3)
    using Net = typename M::NetT;
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    if constexpr (Net::USES_TANH) tanh_tab_init(lane);
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < a.N;
    const int64_t slot = active ? gid : a.N - 1;
    const int64_t i = a.perm != nullptr ? (int64_t)a.perm[slot] : slot;
    const int64_t set = blockIdx.y;
    cptr_t tout = as_const(a.out_times);
    const int n_out = a.T;

    M m;
    double y[NS];
    const double chk = m.init(a, nullptr, lane, i, set, y);
    auto rhs = [&](const double (&Y)[NS], double (&du)[NS]) {
        const double uh = Net::eval(m.p, m.c, Y);
        du[0] = -0.4 * Y[0];
        du[1] = fma(0.4, Y[0], -uh);
        du[2] = fma(-0.3, Y[2], uh);
    };
    double* const tape = GRAD ? a.tape + (set * adaptive_tape_rows(NS, a.tape_cap, a.T)) * a.N + slot : nullptr;
#define TAPE(n, r) tape[((int64_t)(n) * TROWS + (r)) * a.N]
#define SEED(oi, s) tape[((int64_t)a.tape_cap * TROWS + (oi) * 2 + (s)) * a.N]   /* d residual2(oi) / d state 2 + s */
    int n_acc = 0;
    if (GRAD) {                                    // entry 0 always holds finite numbers (parked lanes read it)
        TAPE(0, 0) = a.t_begin;
        TAPE(0, 1) = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) TAPE(0, 2 + s) = y[s];
    }
    const double abstol = a.abstol, reltol = a.reltol;
    const double t0 = a.t_begin, t1 = a.t_end;
    const double t_stop = t1 - 1e-14 * fmax(1.0, fabs(t1));
    double t = t0, dt = 0.0, sse = chk;
    StepController ctl;
    int nxt = 0;
    bool failed = false;
    while (nxt < n_out && tout[nxt] <= t0 + 1e-12) {
        sse += m.residual2(a, nxt, i, y, active);
        nxt++;
    }
    bool done = !(t < t_stop);
    int n_steps = 0;
    double K[7][NS];
    // ---- k1 = f(y0) and the f1 probe of Hairer's initial-step heuristic: one network body
    {
        double sk[NS], d0 = 0.0, d1 = 0.0;
#pragma unroll 1
        for (int r = 1; r < 3; r++) {
            double Y[NS];
            if (r == 1) {
#pragma unroll
                for (int s = 0; s < NS; s++) Y[s] = y[s];
            } else {
                double v0[NS], v1[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    sk[s] = fma(reltol, fabs(y[s]), abstol);
                    v0[s] = y[s] / sk[s];
                    v1[s] = K[0][s] / sk[s];
                }
                d0 = rms(v0, NS);
                d1 = rms(v1, NS);
                dt = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
#pragma unroll
                for (int s = 0; s < NS; s++) Y[s] = fma(dt, K[0][s], y[s]);
            }
            double du[NS];
            rhs(Y, du);
            if (r == 1) {
#pragma unroll
                for (int s = 0; s < NS; s++) K[0][s] = du[s];
            } else {
                double v2[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) v2[s] = (du[s] - K[0][s]) / sk[s];
                const double d2 = rms(v2, NS) / dt;
                const double dm = fmax(d1, d2);
                const double dt1 = dm <= 1e-15 ? fmax(1e-6, dt * 1e-3) : pow(0.01 / dm, 0.2);
                dt = fmin(fmin(100.0 * dt, dt1), t1 - t0);
            }
        }
    }
#pragma unroll 1
    while (true) {
        dt = fmin(dt, t1 - t);
        double ynew[NS];
        double yin[6][2];                          // GRAD: inputs of stages 2..7 of this trial step, states 2 and 3
#pragma unroll
        for (int st = 1; st <= 6; st++) {
            double acc[NS], Y[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) acc[s] = 0.0;
#pragma unroll
            for (int j = 0; j < st; j++) {
#pragma unroll
                for (int s = 0; s < NS; s++) acc[s] = fma(TS_A[st][j], K[j][s], acc[s]);
            }
#pragma unroll
            for (int s = 0; s < NS; s++) Y[s] = fma(dt, acc[s], y[s]);
            rhs(Y, K[st]);
            if (GRAD) { yin[st - 1][0] = Y[1]; yin[st - 1][1] = Y[2]; }
            if (st == 6) {
#pragma unroll
                for (int s = 0; s < NS; s++) ynew[s] = Y[s];
            }
        }
        double ev[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            double e = 0.0;
#pragma unroll
            for (int j = 0; j < 7; j++) e = fma(TS_BT[j], K[j][s], e);
            ev[s] = dt * e / fma(reltol, fmax(fabs(y[s]), fabs(ynew[s])), abstol);
        }
        const double est = rms(ev, NS);
        const bool live = !done && !failed;
        if (live && !(fabs(est) <= 1.79769313486231570815e308)) failed = true;
        const bool accept = ctl.judge(est);
        if (live && !failed) {
            n_steps++;
            if (n_steps >= kAdaptiveMaxSteps) failed = true;
        }
        if (accept) {
            while (__any(live && !failed && nxt < n_out && tout[nxt < n_out ? nxt : n_out - 1] <= t + dt + 1e-12)) {
                const bool mine = live && !failed && nxt < n_out && tout[nxt < n_out ? nxt : n_out - 1] <= t + dt + 1e-12;
                if (mine) {
                    const double th = fmin(1.0, (tout[nxt] - t) / dt);
                    double o[NS];
#pragma unroll
                    for (int s = 0; s < NS; s++) o[s] = 0.0;
                    const bool at_end = fabs(th - 1.0) < 1e-12;
#pragma unroll
                    for (int j = 0; j < 7; j++) {
                        const double w = saveat_weight(j, th, at_end);
#pragma unroll
                        for (int s = 0; s < NS; s++) o[s] = fma(w, K[j][s], o[s]);
                    }
#pragma unroll
                    for (int s = 0; s < NS; s++) o[s] = fma(dt, o[s], y[s]);
                    sse += m.residual2(a, nxt, i, o, active);
                    if (GRAD) {
                        double ob[NS];
                        m.residual_bar(a, nxt, i, o, ob);
                        SEED(nxt, 0) = ob[1];
                        SEED(nxt, 1) = ob[2];
                    }
                    nxt++;
                }
            }
        }
        if (GRAD && live && !failed && accept) {
            if (n_acc < a.tape_cap) {
                TAPE(n_acc, 0) = t;
                TAPE(n_acc, 1) = dt;
#pragma unroll
                for (int s = 0; s < NS; s++) TAPE(n_acc, 2 + s) = y[s];
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    TAPE(n_acc, kSuppTapeHead + 2 * q) = yin[q][0];
                    TAPE(n_acc, kSuppTapeHead + 2 * q + 1) = yin[q][1];
                }
                n_acc++;
            } else {
                failed = true;                    // more accepted steps than the tape holds
            }
        }
        if (!GRAD && live && !failed && accept) n_acc++;
        if (live && !failed) {
            if (accept) {
                t = t + dt;
#pragma unroll
                for (int s = 0; s < NS; s++) { y[s] = ynew[s]; K[0][s] = K[6][s]; }
                dt = ctl.after_accept(dt);
                if (!(t < t_stop)) done = true;
            } else {
                dt = ctl.after_reject(dt);
            }
        }
        if (done || failed) dt = 0.0;
        if (__all(done || failed)) break;
    }
    if (failed || nxt < n_out) sse = __builtin_nan("");
    const bool bad = !(fabs(sse) <= 1.79769313486231570815e308);
    if (active && a.sse != nullptr) a.sse[set * a.set_stride_cond + i] = sse;
    double* out = a.partials + ((int64_t)set * gridDim.x + blockIdx.x) * (P + 2);
    if constexpr (!GRAD) {
        if (active && a.tape_n != nullptr && set == 0) a.tape_n[i] = n_acc;
        const double v2[2] = {active ? sse : 0.0, (active && bad) ? 1.0 : 0.0};
        block_reduce_store<2>(v2, smem, out + P, lane);
    } else {
        // ------------------------------------------------------------------ reverse sweep over the tape
        constexpr int A0 = M::A0;
        double acc[Net::NACC];
#pragma unroll
        for (int q = 0; q < Net::NACC; q++) acc[q] = 0.0;
        double lam[NS], kcar[NS], wsum = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) { kcar[s] = 0.0; lam[s] = 0.0; }
        const double gs = a.inv_n;
        int hi = n_out;                            // observations [hi, n_out) are already accounted for
        int n_max = n_acc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off, 64));
        // a lane with fewer accepted steps idles on its last entry with zero adjoints until its own steps come up
        auto entry = [&](int n) { return n < n_acc ? n : (n_acc > 0 ? n_acc - 1 : 0); };
        // (t_n, dt_n, y_1(t_n)) of the next iteration are requested one iteration ahead; the stage inputs of a step are
        // requested at its top, each is first used one VJP later than the one before
        double e_ahead[3];
#pragma unroll
        for (int r = 0; r < 3; r++) e_ahead[r] = n_max > 0 ? TAPE(entry(n_max - 1), r) : 0.0;
#pragma unroll 1
        for (int n = n_max - 1; n >= 0; n--) {
            const bool on = n < n_acc;
            const int64_t e = entry(n);
            const double tn = e_ahead[0], h = e_ahead[1], y1 = e_ahead[2];
            double yin[6][2];
#pragma unroll
            for (int q = 5; q >= 0; q--) {
                yin[q][0] = TAPE(e, kSuppTapeHead + 2 * q);
                yin[q][1] = TAPE(e, kSuppTapeHead + 2 * q + 1);
            }
            if (n > 0) {
#pragma unroll
                for (int r = 0; r < 3; r++) e_ahead[r] = TAPE(entry(n - 1), r);
            }
            // state 1 at the stages: du1 = -0.4 u1 re-integrated from y_1(t_n) -- the forward sweep's own operations
            double k1s[6], u1[7];
            u1[0] = y1;
#pragma unroll
            for (int sq = 1; sq <= 6; sq++) {
                k1s[sq - 1] = -0.4 * u1[sq - 1];
                double t1s = 0.0;
#pragma unroll
                for (int j = 0; j < sq; j++) t1s = fma(TS_A[sq][j], k1s[j], t1s);
                u1[sq] = fma(h, t1s, y1);
            }
            StageRows<NS, false> B(nullptr, lane);
#pragma unroll
            for (int j = 0; j < 7; j++) {
#pragma unroll
                for (int s = 0; s < NS; s++) B.set(j, s, 0.0);
            }
            // ---- the observations that were saved from this step: adjoint of o = y_n + h sum_j w_j(theta) k_j
            double yb[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) yb[s] = 0.0;
            while (__any(on && hi > 0 && tout[hi > 0 ? hi - 1 : 0] > tn + 1e-12)) {
                const bool mine = on && hi > 0 && tout[hi > 0 ? hi - 1 : 0] > tn + 1e-12;
                if (mine) {
                    const int oi = hi - 1;
                    const double th = fmin(1.0, (tout[oi] - tn) / h);
                    const bool at_end = fabs(th - 1.0) < 1e-12;
                    double ob[NS];
                    ob[0] = 0.0;
                    ob[1] = SEED(oi, 0);
                    ob[2] = SEED(oi, 1);
#pragma unroll
                    for (int s = A0; s < NS; s++) { ob[s] *= gs; yb[s] += ob[s]; ob[s] *= h; }
#pragma unroll
                    for (int j = 0; j < 7; j++) {
                        const double w = saveat_weight(j, th, at_end);
#pragma unroll
                        for (int s = A0; s < NS; s++) B.set(j, s, fma(w, ob[s], B.get(j, s)));
                    }
                    hi--;
                }
            }
            // ---- stage VJPs, last stage first.  k_7 = f(y_{n+1}) is k_1 of the step after this one (FSAL): the adjoint of
            // that k_1 (kcar) is applied together with k_7's at the shared linearisation point.
#pragma unroll
            for (int sq = 6; sq >= 1; sq--) {
                double kb[NS], ub[NS], uu[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    kb[s] = s >= A0 ? B.get(sq, s) + (sq == 6 ? kcar[s] : 0.0) : 0.0;
                    ub[s] = (s >= A0 && sq == 6) ? lam[s] : 0.0;    // Y_7 = y_{n+1}
                }
                uu[0] = u1[sq];
                uu[1] = yin[sq - 1][0];
                uu[2] = yin[sq - 1][1];
                m.vjp(0.0, uu, kb, ub, acc, wsum);
#pragma unroll
                for (int s = A0; s < NS; s++) yb[s] += ub[s];
#pragma unroll
                for (int j = 0; j < sq; j++) {                 // Y_sq = y_n + h sum_{j<sq} a(sq, j) k_j
                    const double aj = h * TS_A[sq][j];
#pragma unroll
                    for (int s = A0; s < NS; s++) B.set(j, s, fma(aj, ub[s], B.get(j, s)));
                }
            }
            // k_1's adjoint is applied with k_7 of the step before (below for step 0)
#pragma unroll
            for (int s = 0; s < NS; s++) kcar[s] = s >= A0 ? B.get(0, s) + 0.0 : 0.0;
#pragma unroll
            for (int s = A0; s < NS; s++) lam[s] = yb[s];
        }
        if (active && a.tape_n != nullptr && set == 0) a.tape_n[i] = n_acc;
        {                                          // k_1 of the first step: linearisation point y_0 (entry 0 of the tape)
            double y0[NS], ub0[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) { y0[s] = TAPE(0, 2 + s); ub0[s] = 0.0; }
            m.vjp(a.t_begin, y0, kcar, ub0, acc, wsum);
        }
        double cst[M::NCST];
        m.finish_grad(a, i, set, acc, wsum, 0.0, cst);
        __syncthreads();
        if (active) a.g_cond[set * a.set_stride_cond + i] = Net::grad_cond(m.p, acc, cst);
        block_reduce_expand<Net, M::NCST>(acc, cst, active ? 1.0 : 0.0, active ? sse : 0.0, (active && bad) ? 1.0 : 0.0,
                                          smem, out, lane);
    }
#undef TAPE
#undef SEED
}

template <class M>
static hipError_t launch_unrolled_supp(const SuppArgs& a, bool grad, hipStream_t s) {
    const int64_t nblocks = (a.N + kBlock - 1) / kBlock;
    const unsigned n_sets = a.n_sets > 0 ? (unsigned)a.n_sets : 1u;
    const size_t lds = sizeof(double) * (size_t)kRedRows * kBlock;      // the final reduction's scratch
    if (grad) {
        if (a.tape == nullptr || a.tape_cap < 1 || a.g_cond == nullptr) return hipErrorInvalidValue;
        hipLaunchKernelGGL((adaptive_unrolled_supp_kernel<M, true>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a);
    } else {
        hipLaunchKernelGGL((adaptive_unrolled_supp_kernel<M, false>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_supp_adaptive_unrolled(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s) {
    if (net.general() || net.nin != 4 || a.T < 1) return hipErrorNotSupported;
#define X(W, D) if (net.width == W && net.depth == D) return launch_unrolled_supp<SuppAd<W, D>>(a, grad, s);
    CUDE_SUPP_AD_UNROLLED(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace cude
