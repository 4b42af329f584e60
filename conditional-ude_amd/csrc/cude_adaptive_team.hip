// Adaptive Tsit5 for SMALL populations of the c-peptide models: a step's five network evaluations on five waves.
//
// Replaces the same reference lines as cude_adaptive_unrolled.hip (solve(model.problem, p = theta, saveat = timepoints),
// src/parameter-estimation.jl:59; src/saem.jl:52) -- for the reference's own population sizes (57 ... 117 subjects, its 25
// restarts side by side), which since round 5 run in this mode by default (cude/api.py default_steps).
//
// Why: with one lane per subject a population of 57 is ONE wave, and that wave walks ~22 trial steps x 5 network
// evaluations forward and ~20 accepted steps x 5 VJPs backward one after the other: 245 us per loss + gradient, 4.5 x the
// fixed-step time-split path (tools/default_mode_cost.py), on a chip whose other 1 023 SIMDs idle.  The adaptive solve
// cannot be split in time (the step sizes are not known in advance), but WITHIN a trial step the five stage times are known
// as soon as (t, dt) are, and the network's input is the time alone (J_f = A): the five evaluations are
// independent of each other and of the state.  So a workgroup is a team of kTeam = 5 waves over the same 64 subjects:
//   * every wave carries the whole integrator redundantly -- state, controller, stage sums, error estimate, `saveat`
//     outputs: ~100 network-free multiply-adds per trial step, the same instructions on the same values in every wave,
//     hence the same bits, hence the same accept / reject decisions without anybody telling anybody;
//   * wave w evaluates the network at stage time w + 1 only and publishes the value in LDS (double-buffered by step
//     parity: ONE workgroup barrier per trial step);
//   * the reverse sweep: the adjoint recursion is network-free and sequential, the five VJPs of a reversed step need only
//     the weights it leaves behind.  A SIXTH wave (the scribe; gradient launches only -- it carries the integrator forward
//     like the others, evaluating nothing) runs the recursion and publishes a step's five weights, h and t_n in LDS
//     (double-buffered, one barrier per step); wave w < 5 applies the VJP of stage w + 1 of the step published BEFORE to
//     its own accumulators while the scribe is already on the next step (round 5, first form: every wave ran the
//     recursion itself, recursion + VJP back to back: 3.3 us per reversed step; now the longer of the two).  At the end
//     the five accumulator sets are added up in wave order through LDS and wave 0 finishes as the one-wave kernel does.
// Forward values (trajectories, SSE, accepted steps) are bit-identical to adaptive_unrolled_kernel -- the same arithmetic
// in the same order; gradients differ by the association of the five partial sums (1e-15).
#include "cude_adaptive.h"

namespace cude {

constexpr int kTeam = 5;                   // waves per workgroup = distinct network evaluations of a Tsit5 trial step

// what the other waves do while wave 0 runs a reduction that contains `n` workgroup barriers
__device__ __forceinline__ void team_barriers(int n) {
    for (int k = 0; k < n; k++) __syncthreads();
}

// stage adjoints of the step being reversed: registers, or one set of LDS rows per wave for the networks whose gradient
// accumulators fill the register file (as cude_adaptive_unrolled.hip)
template <class M, bool GRAD>
constexpr bool team_adjoints_in_lds() { return GRAD && M::NetT::NACC > 40; }

template <class M, bool GRAD>
__global__ __launch_bounds__(kBlock*(kTeam + (GRAD ? 1 : 0))) void adaptive_team_kernel(typename M::Args a) {
    static_assert(!M::NEED_Y, "constant-Jacobian models only");
    constexpr int NS = M::NS;
    constexpr int P = M::P;
    constexpr bool B_LDS = team_adjoints_in_lds<M, GRAD>();
    using Net = typename M::NetT;
    // LDS: [kRedRows] reduction scratch | [2][kTeam] published network values | TG glucose rows, TG - 1 slope rows |
    // (GRAD) [NACC] accumulator rows for the sum over the waves, (wide networks) [7 NS] stage-adjoint rows of the scribe,
    // [2][7] published weights / h / t_n of a reversed step + 2 rows for the recursion's closing sums
    extern __shared__ double smem[];
    const int lane = threadIdx.x % kBlock;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kBlock);
    double* const s_prod = smem + kRedRows * kBlock;
    double* const s_model = s_prod + 2 * kTeam * kBlock;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < a.N;
    const int64_t slot = active ? gid : a.N - 1;                          // position in the launch (lane order) ...
    const int64_t i = a.perm != nullptr ? (int64_t)a.perm[slot] : slot;   // ... and the subject that sits there
    const int64_t set = blockIdx.y;
    const bool lead = wave == 0;                   // the wave that writes what must be written once
    cptr_t tout = as_const(a.out_times);
    const int n_out = a.T;

    M m;
    double y[NS];
    // (every wave stores the same glucose rows to the same LDS words: identical values, each wave reads its own stores)
    const double chk = m.init(a, s_model, lane, i, set, y);
    double* const s_S = s_model + a.TG * kBlock;
    for (int j = 0; j + 1 < a.TG; j++)
        s_S[j * kBlock + lane] = (m.s_G[(j + 1) * kBlock + lane] - m.s_G[j * kBlock + lane]) / (m.tp[j + 1] - m.tp[j]);
    const double tp0 = m.tp[0];
    double kn[kUnrolledKnots - 2];
#pragma unroll
    for (int q = 0; q < kUnrolledKnots - 2; q++) kn[q] = q + 1 < a.TG - 1 ? m.tp[q + 1] : __builtin_inf();
    auto forcing = [&](double t) {
        int j = 0;
        double tlo = tp0;
#pragma unroll
        for (int q = 0; q < kUnrolledKnots - 2; q++) {
            if (kn[q] <= t) { j = q + 1; tlo = kn[q]; }
        }
        return fma(t - tlo, s_S[j * kBlock + lane], m.s_G[j * kBlock + lane]);
    };
    double* const tape = GRAD ? a.tape + (set * adaptive_tape_rows(NS, a.tape_cap, a.T)) * a.N + slot : nullptr;
#define TAPE(n) tape[(int64_t)(n) * a.N]
#define OUTV(oi) tape[((int64_t)a.tape_cap + (oi)) * a.N]
    int n_acc = 0;
    if (GRAD && lead && active) TAPE(0) = 0.0;
    const double abstol = a.abstol, reltol = a.reltol;
    const double t0 = a.t_begin, t1 = a.t_end;
    const double t_stop = t1 - 1e-14 * fmax(1.0, fabs(t1));
    double t = t0, dt = 0.0, sse = chk;
    StepController ctl;
    int nxt = 0;
    bool failed = false;
    while (nxt < n_out && tout[nxt] <= t0 + 1e-12) {
        sse += m.residual2(a, nxt, i, y, active && lead);
        nxt++;
    }
    bool done = !(t < t_stop);
    int n_steps = 0;
    double K[7][NS];
    // ---- NN([0; e^beta]), k1 = f(t0, y0) and the f1 probe of Hairer's initial-step heuristic: every wave, redundantly
    // (three evaluations of ~110; the third depends on the second)
    {
        double sk[NS], d0 = 0.0, d1 = 0.0;
#pragma unroll 1
        for (int r = 0; r < 3; r++) {
            double Y[NS], x = 0.0;
            if (r == 1) {
                x = forcing(t0);
#pragma unroll
                for (int s = 0; s < NS; s++) Y[s] = y[s];
            } else if (r == 2) {
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
                x = forcing(t0 + dt);
#pragma unroll
                for (int s = 0; s < NS; s++) Y[s] = fma(dt, K[0][s], y[s]);
            }
            const double prod = m.production(x);
            if (r == 0) { m.base = prod; continue; }
            double du[NS];
            m.finish_rhs(prod, Y, du);
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
    const double c_mine = TS_C[wave < kTeam ? wave + 1 : kTeam];   // stage time of this wave's evaluation: t + c dt  (c_6 = 1: t + dt)
    int par = 0;
#pragma unroll 1
    while (true) {
        dt = fmin(dt, t1 - t);
        // ---- this wave's one network evaluation of the trial step, published for the team
        if (wave < kTeam) s_prod[(par * kTeam + wave) * kBlock + lane] = m.production(forcing(fma(c_mine, dt, t)));
        __syncthreads();
        double prod[kTeam];
#pragma unroll
        for (int q = 0; q < kTeam; q++) prod[q] = s_prod[(par * kTeam + q) * kBlock + lane];
        par ^= 1;
        double ynew[NS];
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
            m.finish_rhs(prod[st < 6 ? st - 1 : kTeam - 1], Y, K[st]);        // c_6 = c_7 = 1: stage 7 reuses stage 6's value
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
                    sse += m.residual2(a, nxt, i, o, active && lead);
                    if (GRAD && lead && active) OUTV(nxt) = o[0];
                    nxt++;
                }
            }
        }
        if (GRAD && live && !failed && accept) {
            if (n_acc < a.tape_cap) {
                if (lead && active) TAPE(n_acc) = dt;
                n_acc++;
            } else {
                failed = true;
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
        if (__all(done || failed)) break;          // (every wave holds the same state: they all leave in the same step)
    }
    if (failed || nxt < n_out) sse = __builtin_nan("");
    const bool bad = !(fabs(sse) <= 1.79769313486231570815e308);
    if (lead && active && a.sse != nullptr) a.sse[set * a.set_stride_cond + i] = sse;
    if (lead && active && a.tape_n != nullptr && set == 0) a.tape_n[i] = n_acc;
    double* out = a.partials + ((int64_t)set * gridDim.x + blockIdx.x) * (P + 2);
    if constexpr (!GRAD) {
        if (lead) {
            const double v2[2] = {active ? sse : 0.0, (active && bad) ? 1.0 : 0.0};
            block_reduce_store<2>(v2, smem, out + P, lane);
        } else {
            team_barriers(2);
        }
    } else {
        __syncthreads();                           // the lead wave's tape and saved outputs are in memory for the others
        double acc[Net::NACC];
#pragma unroll
        for (int q = 0; q < Net::NACC; q++) acc[q] = 0.0;
        double wsum = 0.0, carry = 0.0;
        const double gs = a.inv_n;
        int n_max = n_acc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off, 64));
        double* const s_B = s_model + (2 * a.TG + Net::NACC) * kBlock;
        double* const s_pub = s_B + (B_LDS ? 7 * NS : 0) * kBlock;
        constexpr int kPub = 7;                    // five weights, h, t_n
        if (wave == kTeam) {
            // ---- the scribe: the network-free stage-adjoint recursion, a step ahead of the VJPs
            double lam[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) lam[s] = 0.0;
            int hi = n_out;
            double t_next = t;
            double h_ahead = 0.0;
            if (n_max > 0) h_ahead = TAPE(n_max - 1 < n_acc ? n_max - 1 : (n_acc > 0 ? n_acc - 1 : 0));
            int pb = 0;
#pragma unroll 1
            for (int n = n_max - 1; n >= 0; n--) {
                const bool on = n < n_acc;
                const double h = h_ahead;
                if (n > 0) h_ahead = TAPE(n - 1 < n_acc ? n - 1 : (n_acc > 0 ? n_acc - 1 : 0));
                const double tn = t_next - h;
                if (on) t_next = tn;
                StageRows<NS, B_LDS> B(s_B, lane);
                double yb[NS];
#pragma unroll
                for (int j = 0; j < 7; j++) {
#pragma unroll
                    for (int s = 0; s < NS; s++) B.set(j, s, 0.0);
                }
#pragma unroll
                for (int s = 0; s < NS; s++) yb[s] = 0.0;
                while (__any(on && hi > 0 && tout[hi > 0 ? hi - 1 : 0] > tn + 1e-12)) {
                    const bool mine = on && hi > 0 && tout[hi > 0 ? hi - 1 : 0] > tn + 1e-12;
                    if (mine) {
                        const int oi = hi - 1;
                        const double th = fmin(1.0, (tout[oi] - tn) / h);
                        const bool at_end = fabs(th - 1.0) < 1e-12;
                        double o[NS], ob[NS];
#pragma unroll
                        for (int s = 0; s < NS; s++) o[s] = 0.0;
                        o[0] = OUTV(oi);
                        m.residual_bar(a, oi, i, o, ob);
#pragma unroll
                        for (int s = 0; s < NS; s++) { ob[s] *= gs; yb[s] += ob[s]; ob[s] *= h; }
#pragma unroll
                        for (int j = 0; j < 7; j++) {
                            const double w = saveat_weight(j, th, at_end);
#pragma unroll
                            for (int s = 0; s < NS; s++) B.set(j, s, fma(w, ob[s], B.get(j, s)));
                        }
                        hi--;
                    }
                }
                double wacc = carry, wst[kTeam];
#pragma unroll
                for (int sq = 6; sq >= 0; sq--) {
                    double kb[NS], ub[NS];
#pragma unroll
                    for (int s = 0; s < NS; s++) { kb[s] = B.get(sq, s); ub[s] = sq == 6 ? lam[s] : 0.0; }
                    m.vjp_linear(kb, ub);
                    if (sq == 6) {
                        wacc += kb[0];
                    } else if (sq == 0) {
                        carry = kb[0];
                    } else {
                        wst[sq - 1] = sq == 5 ? wacc + kb[0] : kb[0];      // the weight of stage sq's VJP (wave sq - 1)
                    }
                    wsum += kb[0];
#pragma unroll
                    for (int s = 0; s < NS; s++) yb[s] += ub[s];
#pragma unroll
                    for (int j = 0; j < sq; j++) {
                        const double aj = h * TS_A[sq][j];
#pragma unroll
                        for (int s = 0; s < NS; s++) B.set(j, s, fma(aj, ub[s], B.get(j, s)));
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; s++) lam[s] = yb[s];
#pragma unroll
                for (int q = 0; q < kTeam; q++) s_pub[(pb * kPub + q) * kBlock + lane] = wst[q];
                s_pub[(pb * kPub + 5) * kBlock + lane] = h;
                s_pub[(pb * kPub + 6) * kBlock + lane] = tn;
                pb ^= 1;
                __syncthreads();
            }
            s_pub[(2 * kPub) * kBlock + lane] = wsum;
            s_pub[(2 * kPub + 1) * kBlock + lane] = carry;
        } else {
            // ---- wave w: the VJP of stage w + 1 of the step the scribe published last
            int pb = 0;
#pragma unroll 1
            for (int n = n_max - 1; n >= 0; n--) {
                __syncthreads();
                const double my_w = s_pub[(pb * kPub + wave) * kBlock + lane];
                const double h = s_pub[(pb * kPub + 5) * kBlock + lane], tn = s_pub[(pb * kPub + 6) * kBlock + lane];
                pb ^= 1;
                double dx[1] = {0.0};
                const double xx[1] = {forcing(fma(c_mine, h, tn))};
                Net::template eval_grad<false, decltype(acc), kAdaptivePin>(m.p, m.c, xx, my_w, acc, dx);
            }
        }
        __syncthreads();                           // the recursion's closing sums are in LDS for the lead wave
        if (lead) {
            wsum = s_pub[(2 * kPub) * kBlock + lane];
            carry = s_pub[(2 * kPub + 1) * kBlock + lane];
        }
        // ---- closing evaluations (k1 of the first step at t0, the baseline term): the lead wave; then the five accumulator
        // sets are added up in wave order
        double cst[M::NCST];
        if (lead) {
            m.finish_grad(a, i, set, acc, wsum, carry, cst);
        } else {
            cst[0] = Net::cond_input(a.cond[set * a.set_stride_cond + i]);
            if (M::NCST > 1) cst[M::NCST - 1] = a.age[i];
        }
        double* const s_acc = s_model + (2 * a.TG) * kBlock;
        __syncthreads();
        for (int ww = 0; ww < kTeam; ww++) {
            if (wave == ww) {
#pragma unroll
                for (int q = 0; q < Net::NACC; q++)
                    s_acc[q * kBlock + lane] = ww == 0 ? acc[q] : s_acc[q * kBlock + lane] + acc[q];
            }
            __syncthreads();
        }
        if (lead) {
#pragma unroll
            for (int q = 0; q < Net::NACC; q++) acc[q] = s_acc[q * kBlock + lane];
            if (active) a.g_cond[set * a.set_stride_cond + i] = Net::grad_cond(m.p, acc, cst);
            block_reduce_expand<Net, M::NCST>(acc, cst, active ? 1.0 : 0.0, active ? sse : 0.0, (active && bad) ? 1.0 : 0.0,
                                              smem, out, lane);
        } else {
            team_barriers(2 * ((P + 2 + kRedRows - 1) / kRedRows));
        }
    }
#undef TAPE
#undef OUTV
}

template <class M>
static hipError_t launch_team(const typename M::Args& a, bool grad, hipStream_t s) {
    const int64_t nblocks = (a.N + kBlock - 1) / kBlock;
    const unsigned n_sets = a.n_sets > 0 ? (unsigned)a.n_sets : 1u;
    const size_t lds = sizeof(double) * (size_t)(kRedRows + 2 * kTeam + 2 * a.TG + (grad ? M::NetT::NACC + 2 * 7 + 2 : 0) +
                                                 (grad && team_adjoints_in_lds<M, true>() ? 7 * M::NS : 0)) * kBlock;
    if (grad) {
        if (a.tape == nullptr || a.tape_cap < 1 || a.g_cond == nullptr) return hipErrorInvalidValue;
        // (networks with more than 64 accumulators spill heavily at the two waves per SIMD a team of five needs: they keep
        // the one-wave kernel for the gradient)
        if constexpr (M::NetT::NACC > 64) return hipErrorNotSupported;
        hipLaunchKernelGGL((adaptive_team_kernel<M, true>), dim3((unsigned)nblocks, n_sets), dim3(kBlock * (kTeam + 1)), lds, s, a);
    } else {
        hipLaunchKernelGGL((adaptive_team_kernel<M, false>), dim3((unsigned)nblocks, n_sets), dim3(kBlock * kTeam), lds, s, a);
    }
    return hipGetLastError();
}

// Small launches only (the team multiplies the waves by five: worth it while the one-wave-per-64-subjects grid leaves most
// of the chip idle), the network shapes of the reference's scripts and their neighbours (shape group
// 0 of cude_adaptive.h) on grids of at most kUnrolledKnots times; hipErrorNotSupported = not this kernel's case.
hipError_t launch_cpep_adaptive_team(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
    if (net.general() || net.generic() || net.symbolic() || a.team < 0) return hipErrorNotSupported;
    if (a.TG < 2 || a.TG > kUnrolledKnots || a.T < 1) return hipErrorNotSupported;
    const int64_t waves1 = ((a.N + kBlock - 1) / kBlock) * (a.n_sets > 0 ? a.n_sets : 1);
    if (waves1 > kTeamMaxWaves) return hipErrorNotSupported;
    if (grad && a.obs == nullptr) return hipErrorInvalidValue;
#define X(NIN, W, D) \
    if (net.nin == NIN && net.width == W && net.depth == D) return launch_team<CpepAd<Mlp<NIN, W, D, 1>>>(a, grad, s);
    CUDE_CPEP_AD_SHAPES_0(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace cude
