// Adaptive Tsit5 ensemble kernels for gfx950 -- what the reference actually runs.
//
// Replaces (reference repo paths; the solver itself is OrdinaryDiffEq, third-party, restated here from its documented
// defaults; the reference's stored objectives pin this restatement, tests/test_gpu_known_answers.py):
//   solve(model.problem, p = theta, saveat = timepoints, save_idxs = 1)          src/parameter-estimation.jl:59
//   solve(prob, saveat = timepoints, save_idxs = 1)                              src/saem.jl:52
//   solve(ensemble, Tsit5(), EnsembleThreads(); saveat, trajectories = N)        suppression/src/suppression_model.jl:113,123
// i.e. Tsit5 with abstol 1e-6 / reltol 1e-3, the PI step controller (beta1 = 7/50, beta2 = 2/25, gamma = 0.9,
// qmin = 0.2, qmax = 10), Hairer's initial-step heuristic, `saveat` through the free 4th-order interpolant, failure
// (non-finite error estimate, more than 1e5 steps) => +Inf loss.  Selected by cude_config.n_steps = 0: loss,
// per-subject SSE, trajectories, dense output, profiles, screening, Metropolis E-step, and -- GRAD -- the gradient.
//
// Gradient of the adaptive solve = what the reference's AutoForwardDiff computes (src/parameter-estimation.jl:165,
// suppression_model.jl:155): under ForwardDiff only p = theta carries partials, tspan and dt stay Float64, so the
// derivative is that of the accepted step sequence as fixed arithmetic (controller and initial-step heuristic are not
// differentiated).  Here: the forward sweep writes every accepted step to a per-subject tape in HBM ([step][rows]
// [subject], coalesced) -- (t_n, dt_n, y_n) for the suppression model, whose reverse sweep re-runs the stage evaluations
// from y_n; dt_n ALONE (8 B per step) for the c-peptide models, whose Jacobian is constant: their reverse sweep needs
// the stage TIMES only and steps back from the final time, t_n = t_{n+1} - dt_n (within an ulp of the forward sweep's
// t_n: the forward sum is not exactly invertible; 1e-16 relative in a network input) -- and the reverse sweep walks the
// tape backwards and applies the stage VJPs in reverse order, including the `saveat` interpolation weights of the
// observations that fell into the step.  (OrdinaryDiffEq's error norm under duals also weighs the partials -- its
// documented behaviour -- so the reference's accepted steps during a gradient call can differ from those of a plain
// solve; the sequence here is the plain solve's.)  Lanes with more accepted steps than the tape holds fail (+Inf).
//
// One lane = one subject, each with its own (t, dt, controller state): the lanes of a wave walk the same sequence of
// phases (k1, the f1 probe of the initial step, then stages 2..7 of step after step) with ONE inlined network body;
// a lane that has reached t_end stops committing and the wave leaves when all of its lanes are done (or after the
// solver's own step limit, so every wave terminates).
#include "cude_adaptive.h"

namespace cude {

// ---------------------------------------------------------------------------------- the integrator
// LDS: s_K [7][NS] stage derivatives (one row of kBlock doubles each; >= kRedRows rows for the final reduction),
// GRAD: s_B [7][NS] their adjoints and (models whose Jacobian depends on the state) s_Y [7][NS] the stage inputs of the
// step being reversed; then the model's own rows.
template <class M>
constexpr int adaptive_rows(bool grad) {
    constexpr int KROWS = 7 * M::NS > kRedRows ? 7 * M::NS : kRedRows;
    return KROWS + (grad ? 7 * M::NS * (M::NEED_Y ? 2 : 1) : 0);
}

template <class M, bool IS_CPEP, bool GRAD>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(adaptive_waves<M, GRAD>())))
void adaptive_kernel(typename M::Args a) {
    constexpr int NS = M::NS;
    constexpr int P = M::P;
    constexpr int KROWS = 7 * NS > kRedRows ? 7 * NS : kRedRows;
    constexpr int TROWS = M::NEED_Y ? kSuppTapeRows : 1;   // tape entry: t_n, dt_n, y_n (+ rows this kernel leaves unused:
                                                           // cude_kernels.h) -- or dt_n alone (constant Jacobian)
    extern __shared__ double smem[];
    double* s_K = smem;
    double* s_B = smem + KROWS * kBlock;
    double* s_Y = s_B + 7 * NS * kBlock;
    const int lane = threadIdx.x;
    if constexpr (M::NetT::USES_TANH) tanh_tab_init(lane);
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < a.N;
    const int64_t slot = active ? gid : a.N - 1;                       // position in the launch (lane order) ...
    const int64_t i = a.perm != nullptr ? (int64_t)a.perm[slot] : slot;  // ... and the subject that sits there
    const int64_t set = blockIdx.y;
    cptr_t tout = as_const(a.out_times);
    const int n_out = a.T;
#define KROW(j, s) s_K[((j) * NS + (s)) * kBlock + lane]

    M m;
    double y[NS];
    const double chk = m.init(a, smem + adaptive_rows<M>(GRAD) * kBlock, lane, i, set, y);
    double* const tape = GRAD ? a.tape + (set * adaptive_tape_rows(NS, a.tape_cap, a.T)) * a.N + slot : nullptr;
#define TAPE(n, r) tape[((int64_t)(n) * TROWS + (r)) * a.N]
#define OUTV(oi) tape[((int64_t)a.tape_cap * TROWS + (oi)) * a.N]     /* saved output (state 1) behind the steps */
    int n_acc = 0;
    if (GRAD) {                                    // entry 0 always holds finite numbers (parked lanes read it)
        if constexpr (M::NEED_Y) {
            TAPE(0, 0) = a.t_begin;
            TAPE(0, 1) = 0.0;
#pragma unroll
            for (int s = 0; s < NS; s++) TAPE(0, 2 + s) = y[s];
        } else {
            TAPE(0, 0) = 0.0;
        }
    }
    const double abstol = a.abstol, reltol = a.reltol;
    const double t0 = a.t_begin, t1 = a.t_end;
    const double t_stop = t1 - 1e-14 * fmax(1.0, fabs(t1));

    double t = t0, dt = 0.0, sse = chk;
    StepController ctl;
    double sk[NS], d0 = 0.0, d1 = 0.0;
    int nxt = 0;
    bool failed = false;
    double prod_last = 0.0;
    // outputs at (or before) the initial time
    while (nxt < n_out && tout[nxt] <= t0 + 1e-12) {
        sse += m.residual2(a, nxt, i, y, active);
        nxt++;
    }
    bool done = !(t < t_stop);
    int n_steps = 0;
    // phase -2: k1 = f(t0, y0); -1: f1 probe of the initial-step heuristic; 1..6: stages 2..7 of the current step
    int st = -2;
    double Y[NS], ynew[NS];
    // alternating issue priority of co-resident waves (see cpep_kernel): every 2^prio_shift evaluations
    int prio_shift = 0;
    unsigned prio_par = 0, it = 0;
    if constexpr (GRAD && IS_CPEP) {
        prio_shift = a.prio_shift;
        prio_par = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)) & 1u;       // HW_ID.wave_id
    }
#pragma unroll 1
    while (true) {
        if (GRAD && IS_CPEP && prio_shift > 0) {
            if ((((it++) >> prio_shift) ^ prio_par) & 1u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        double te;
        if (st == -2) {
            te = t0;
#pragma unroll
            for (int s = 0; s < NS; s++) Y[s] = y[s];
        } else if (st == -1) {
            // Hairer's heuristic, first half: d0 = |y0|, d1 = |f0| in the scaled norm
            double v0[NS], v1[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                sk[s] = fma(reltol, fabs(y[s]), abstol);
                v0[s] = y[s] / sk[s];
                v1[s] = KROW(0, s) / sk[s];
            }
            d0 = rms(v0, NS);
            d1 = rms(v1, NS);
            dt = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
            te = t0 + dt;
#pragma unroll
            for (int s = 0; s < NS; s++) Y[s] = fma(dt, KROW(0, s), y[s]);
        } else {
            if (st == 1) dt = fmin(dt, t1 - t);
            double acc[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) acc[s] = 0.0;
#pragma unroll 1
            for (int j = 0; j < st; j++) {
                const double aj = TS_A[st][j];
#pragma unroll
                for (int s = 0; s < NS; s++) acc[s] = fma(aj, KROW(j, s), acc[s]);
            }
#pragma unroll
            for (int s = 0; s < NS; s++) Y[s] = fma(dt, acc[s], y[s]);
            te = st < 6 ? fma(TS_C[st], dt, t) : t + dt;
        }
        // ---- the one right-hand-side evaluation of the loop body
        double du[NS];
        if constexpr (IS_CPEP) {
            if (st == -2) m.base = m.production(0.0);      // NN([0; e^beta]): time-invariant, evaluated once
            // the forcing depends on time only and c_6 = c_7 = 1: stage 7 (st == 6, wave-uniform) reuses stage 6's value
            if (st != 6) prod_last = m.production(m.forcing_input(te));
            m.finish_rhs(prod_last, Y, du);
        } else {
            const double uh = M::Net::eval(m.p, m.c, Y);
            du[0] = -0.4 * Y[0];
            du[1] = fma(0.4, Y[0], -uh);
            du[2] = fma(-0.3, Y[2], uh);
        }
        if (st == -2) {
#pragma unroll
            for (int s = 0; s < NS; s++) KROW(0, s) = du[s];
            st = -1;
            continue;
        }
        if (st == -1) {
            double v2[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) v2[s] = (du[s] - KROW(0, s)) / sk[s];
            const double d2 = rms(v2, NS) / dt;
            const double dm = fmax(d1, d2);
            const double dt1 = dm <= 1e-15 ? fmax(1e-6, dt * 1e-3) : pow(0.01 / dm, 0.2);
            dt = fmin(fmin(100.0 * dt, dt1), t1 - t0);
            st = 1;
            continue;
        }
#pragma unroll
        for (int s = 0; s < NS; s++) KROW(st, s) = du[s];
        if (st < 6) { st++; continue; }
        // ---- end of a trial step: Y = y_{n+1}, KROW(6) = k7
#pragma unroll
        for (int s = 0; s < NS; s++) ynew[s] = Y[s];
        double ev[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            double e = 0.0;
#pragma unroll 1
            for (int j = 0; j < 7; j++) e = fma(TS_BT[j], KROW(j, s), e);
            ev[s] = dt * e / fma(reltol, fmax(fabs(y[s]), fabs(ynew[s])), abstol);
        }
        const double est = rms(ev, NS);
        const bool live = !done && !failed;
        if (live && !(fabs(est) <= 1.79769313486231570815e308)) failed = true;     // NaN / Inf: the solve fails
        const bool accept = ctl.judge(est);
        if (live && !failed) {
            n_steps++;
            if (n_steps >= kAdaptiveMaxSteps) failed = true;
        }
        if (accept) {
            // saveat outputs inside (t, t + dt] through the interpolant
            while (__any(live && !failed && nxt < n_out && tout[nxt < n_out ? nxt : n_out - 1] <= t + dt + 1e-12)) {
                const bool mine = live && !failed && nxt < n_out && tout[nxt < n_out ? nxt : n_out - 1] <= t + dt + 1e-12;
                if (mine) {
                    const double th = fmin(1.0, (tout[nxt] - t) / dt);
                    double o[NS];
#pragma unroll
                    for (int s = 0; s < NS; s++) o[s] = 0.0;
                    const bool at_end = fabs(th - 1.0) < 1e-12;
#pragma unroll 1
                    for (int j = 0; j < 7; j++) {
                        const double w = saveat_weight(j, th, at_end);
#pragma unroll
                        for (int s = 0; s < NS; s++) o[s] = fma(w, KROW(j, s), o[s]);
                    }
#pragma unroll
                    for (int s = 0; s < NS; s++) o[s] = fma(dt, o[s], y[s]);
                    sse += m.residual2(a, nxt, i, o, active);
                    if (GRAD && !M::NEED_Y) OUTV(nxt) = o[0];
                    nxt++;
                }
            }
        }
        if (GRAD && live && !failed && accept) {
            if (n_acc < a.tape_cap) {
                if constexpr (M::NEED_Y) {
                    TAPE(n_acc, 0) = t;
                    TAPE(n_acc, 1) = dt;
#pragma unroll
                    for (int s = 0; s < NS; s++) TAPE(n_acc, 2 + s) = y[s];
                } else {
                    TAPE(n_acc, 0) = dt;
                }
                n_acc++;
            } else {
                failed = true;                    // more accepted steps than the tape holds
            }
        }
        if (!GRAD && live && !failed && accept) n_acc++;      // (forward launches report the count too: cude_adaptive_regroup)
        if (live && !failed) {
            if (accept) {
                t = t + dt;
#pragma unroll
                for (int s = 0; s < NS; s++) { y[s] = ynew[s]; KROW(0, s) = KROW(6, s); }
                dt = ctl.after_accept(dt);
                if (!(t < t_stop)) done = true;
            } else {
                dt = ctl.after_reject(dt);
            }
        }
        if (done || failed) dt = 0.0;                 // parked lane: harmless arithmetic until the wave leaves
        if (__all(done || failed)) break;
        st = 1;
    }
    if (failed || nxt < n_out) sse = __builtin_nan("");      // failed solve => non-finite SSE => loss +Inf (reference :61-64)
    const bool bad = !(fabs(sse) <= 1.79769313486231570815e308);
    if (active && a.sse != nullptr) a.sse[set * a.set_stride_cond + i] = sse;
    double* out = a.partials + ((int64_t)set * gridDim.x + blockIdx.x) * (P + 2);
    if constexpr (!GRAD) {
        if (active && a.tape_n != nullptr && set == 0) a.tape_n[i] = n_acc;
        const double v2[2] = {active ? sse : 0.0, (active && bad) ? 1.0 : 0.0};
        block_reduce_store<2>(v2, smem, out + P, lane);
    } else {
        // ------------------------------------------------------------------ reverse sweep over the tape
        using Net = typename M::NetT;
        constexpr int A0 = M::A0;
#define BROW(j, s) s_B[((j) * NS + (s)) * kBlock + lane]
#define YROW(j, s) s_Y[((j) * NS + (s)) * kBlock + lane]
        double acc[Net::NACC];
#pragma unroll
        for (int q = 0; q < Net::NACC; q++) acc[q] = 0.0;
        double lam[NS], wsum = 0.0, carry = 0.0;
        double k1_next[NS], kcar[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) { k1_next[s] = 0.0; kcar[s] = 0.0; }
#pragma unroll
        for (int s = 0; s < NS; s++) lam[s] = 0.0;
        const double gs = a.inv_n;
        int hi = n_out;                            // observations [hi, n_out) are already accounted for
        int n_max = n_acc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off, 64));
        double t_next = t;                         // constant-Jacobian models: end of the step being reversed (t = final time)
        // the step size of the next iteration is requested one iteration ahead (a dependent HBM round trip otherwise)
        double h_ahead = 0.0;
        if constexpr (!M::NEED_Y) {
            if (n_max > 0) h_ahead = TAPE(n_max - 1 < n_acc ? n_max - 1 : (n_acc > 0 ? n_acc - 1 : 0), 0);
        }
#pragma unroll 1
        for (int n = n_max - 1; n >= 0; n--) {
            // a lane with fewer accepted steps idles on its last entry with zero adjoints until its own steps come up
            if (IS_CPEP && prio_shift > 0) {               // (5 VJPs per step: switch every 2^(prio_shift - 2) steps)
                if ((((unsigned)n >> (prio_shift > 2 ? prio_shift - 2 : 0)) ^ prio_par) & 1u) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(0);
            }
            const bool on = n < n_acc;
            const int src = on ? n : (n_acc > 0 ? n_acc - 1 : 0);
            double tn, h;
            if constexpr (M::NEED_Y) {
                tn = TAPE(src, 0);
                h = TAPE(src, 1);
            } else {
                h = h_ahead;
                if (n > 0) h_ahead = TAPE(n - 1 < n_acc ? n - 1 : (n_acc > 0 ? n_acc - 1 : 0), 0);
                tn = t_next - h;
                if (on) t_next = tn;
            }
            if constexpr (!M::NEED_Y) {
                // ---- linear kinetics + a forcing that depends on time only: J_f = A, nothing to re-run.  The outputs
                // were saved by the forward sweep; stages 6 and 7 of this step and stage 1 of the next share one time,
                // hence one network VJP: 5 per step (+ 1 at t_0), as in the fixed-step kernel.
#pragma unroll 1
                for (int j = 0; j < 7; j++) {
#pragma unroll
                    for (int s = 0; s < NS; s++) BROW(j, s) = 0.0;
                }
                double yb[NS];
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
#pragma unroll 1
                        for (int j = 0; j < 7; j++) {
                            const double w = saveat_weight(j, th, at_end);
#pragma unroll
                            for (int s = 0; s < NS; s++) BROW(j, s) = fma(w, ob[s], BROW(j, s));
                        }
                        hi--;
                    }
                }
                double wacc = carry;                            // weight of the evaluation at t_n + h
#pragma unroll 1
                for (int sq = 6; sq >= 0; sq--) {
                    double kb[NS], ub[NS];
#pragma unroll
                    for (int s = 0; s < NS; s++) { kb[s] = BROW(sq, s); ub[s] = sq == 6 ? lam[s] : 0.0; }
                    if (sq == 6) {
                        m.vjp_linear(kb, ub);
                        wacc += kb[0];
                        wsum += kb[0];
                    } else if (sq == 0) {
                        m.vjp_linear(kb, ub);
                        carry = kb[0];                          // evaluated with the previous step's stages 6 and 7
                        wsum += kb[0];
                    } else {
                        m.vjp_linear(kb, ub);
                        m.vjp_net(sq == 5 ? tn + h : fma(TS_C[sq], h, tn), sq == 5 ? wacc + kb[0] : kb[0], acc);
                        wsum += kb[0];
                    }
#pragma unroll
                    for (int s = 0; s < NS; s++) yb[s] += ub[s];
#pragma unroll 1
                    for (int j = 0; j < sq; j++) {
                        const double aj = h * TS_A[sq][j];
#pragma unroll
                        for (int s = 0; s < NS; s++) BROW(j, s) = fma(aj, ub[s], BROW(j, s));
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; s++) lam[s] = yb[s];
            } else {
#pragma unroll
            for (int s = 0; s < NS; s++) y[s] = TAPE(src, 2 + s);
            // ---- re-run the stages of the step: k_1 .. k_6 and Y_7 = y_{n+1}.  k_7 = f(y_{n+1}) is k_1 of the step after
            // this one (FSAL), which the sweep has just re-run: it is carried over (k1_next), and so is the adjoint of
            // that k_1 (kcar), which is applied together with k_7's at the shared linearisation point -- six network
            // evaluations and six VJPs per step instead of seven.  A lane's LAST step (and an idling lane) has no
            // later step: the seventh evaluation is made whenever some lane of the wave needs it.
            const bool last = !(n + 1 < n_acc);
            const bool need7 = __any(last);
#pragma unroll 1
            for (int sq = 0; sq <= 6; sq++) {
                double uu[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) uu[s] = 0.0;
#pragma unroll 1
                for (int j = 0; j < sq; j++) {
                    const double aj = TS_A[sq][j];
#pragma unroll
                    for (int s = 0; s < NS; s++) uu[s] = fma(aj, KROW(j, s), uu[s]);
                }
#pragma unroll
                for (int s = 0; s < NS; s++) uu[s] = sq == 0 ? y[s] : fma(h, uu[s], y[s]);
                if (M::NEED_Y) {
#pragma unroll
                    for (int s = 0; s < NS; s++) YROW(sq, s) = uu[s];
                }
                const double te = sq == 0 ? tn : (sq < 6 ? fma(TS_C[sq], h, tn) : tn + h);
                double dd[NS];
                if (sq < 6 || need7) {
                    if constexpr (IS_CPEP) {
                        m.finish_rhs(m.production(m.forcing_input(te)), uu, dd);
                    } else {
                        const double uh = M::Net::eval(m.p, m.c, uu);
                        dd[0] = -0.4 * uu[0];
                        dd[1] = fma(0.4, uu[0], -uh);
                        dd[2] = fma(-0.3, uu[2], uh);
                    }
                }
                if (sq == 6 && !last) {
#pragma unroll
                    for (int s = 0; s < NS; s++) dd[s] = k1_next[s];
                }
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    KROW(sq, s) = dd[s];
                    if (s >= A0) BROW(sq, s) = 0.0;
                }
            }
#pragma unroll
            for (int s = 0; s < NS; s++) k1_next[s] = KROW(0, s);
            // ---- the observations that were saved from this step: adjoint of o = y_n + h sum_j w_j(theta) k_j
            double yb[NS];
#pragma unroll
            for (int s = A0; s < NS; s++) yb[s] = 0.0;
            while (__any(on && hi > 0 && tout[hi > 0 ? hi - 1 : 0] > tn + 1e-12)) {
                const bool mine = on && hi > 0 && tout[hi > 0 ? hi - 1 : 0] > tn + 1e-12;
                if (mine) {
                    const int oi = hi - 1;
                    const double th = fmin(1.0, (tout[oi] - tn) / h);
                    const bool at_end = fabs(th - 1.0) < 1e-12;
                    double w[7], o[NS], ob[NS];
#pragma unroll
                    for (int s = 0; s < NS; s++) o[s] = 0.0;
#pragma unroll 1
                    for (int j = 0; j < 7; j++) {
                        w[j] = saveat_weight(j, th, at_end);
#pragma unroll
                        for (int s = 0; s < NS; s++) o[s] = fma(w[j], KROW(j, s), o[s]);
                    }
#pragma unroll
                    for (int s = 0; s < NS; s++) o[s] = fma(h, o[s], y[s]);
                    m.residual_bar(a, oi, i, o, ob);
#pragma unroll
                    for (int s = A0; s < NS; s++) { ob[s] *= gs; yb[s] += ob[s]; ob[s] *= h; }
#pragma unroll 1
                    for (int j = 0; j < 7; j++) {
#pragma unroll
                        for (int s = A0; s < NS; s++) BROW(j, s) = fma(w[j], ob[s], BROW(j, s));
                    }
                    hi--;
                }
            }
            // ---- stage VJPs, last stage first
#pragma unroll 1
            for (int sq = 6; sq >= 0; sq--) {
                double kb[NS], ub[NS], uu[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    kb[s] = s >= A0 ? BROW(sq, s) + (sq == 6 ? kcar[s] : 0.0) : 0.0;
                    ub[s] = (s >= A0 && sq == 6) ? lam[s] : 0.0;    // Y_7 = y_{n+1}
                    uu[s] = M::NEED_Y ? YROW(sq, s) : 0.0;
                }
                if (sq == 0) {                                 // applied with k_7 of the step before (finish_grad for step 0)
#pragma unroll
                    for (int s = A0; s < NS; s++) kcar[s] = kb[s];
                    break;
                }
                const double te = sq < 6 ? fma(TS_C[sq], h, tn) : tn + h;
                m.vjp(te, uu, kb, ub, acc, wsum);
#pragma unroll
                for (int s = A0; s < NS; s++) yb[s] += ub[s];
#pragma unroll 1
                for (int j = 0; j < sq; j++) {                 // Y_sq = y_n + h sum_{j<sq} a(sq, j) k_j
                    const double aj = h * TS_A[sq][j];
#pragma unroll
                    for (int s = A0; s < NS; s++) BROW(j, s) = fma(aj, ub[s], BROW(j, s));
                }
            }
#pragma unroll
            for (int s = A0; s < NS; s++) lam[s] = yb[s];
            }
        }
        if (active && a.tape_n != nullptr && set == 0) a.tape_n[i] = n_acc;
        if constexpr (M::NEED_Y) {                 // k_1 of the first step: linearisation point y_0 (entry 0 of the tape)
            double y0[NS], ub0[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) { y0[s] = TAPE(0, 2 + s); ub0[s] = 0.0; }
            m.vjp(a.t_begin, y0, kcar, ub0, acc, wsum);
        }
        double cst[M::NCST];
        m.finish_grad(a, i, set, acc, wsum, carry, cst);
        __syncthreads();                   // the reduction scratch aliases s_K
        if (active) a.g_cond[set * a.set_stride_cond + i] = Net::grad_cond(m.p, acc, cst);
        block_reduce_expand<Net, M::NCST>(acc, cst, active ? 1.0 : 0.0, active ? sse : 0.0, (active && bad) ? 1.0 : 0.0,
                                          smem, out, lane);
#undef BROW
#undef YROW
    }
#undef KROW
#undef TAPE
#undef OUTV
}

// ---------------------------------------------------------------------------------- dispatch
template <class M, bool IS_CPEP>
static hipError_t launch_adaptive(const typename M::Args& a, int extra_rows, bool grad, hipStream_t s) {
    const int64_t nblocks = (a.N + kBlock - 1) / kBlock;
    const size_t lds = sizeof(double) * (size_t)(adaptive_rows<M>(grad) + extra_rows) * kBlock;
    const unsigned n_sets = a.n_sets > 0 ? (unsigned)a.n_sets : 1u;
    if (grad && (a.tape == nullptr || a.tape_cap < 1 || a.g_cond == nullptr)) return hipErrorInvalidValue;
    if (grad) {
        hipLaunchKernelGGL((adaptive_kernel<M, IS_CPEP, true>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a);
    } else {
        hipLaunchKernelGGL((adaptive_kernel<M, IS_CPEP, false>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a);
    }
    return hipGetLastError();
}

#ifndef CUDE_AD_PART
#define CUDE_AD_PART 0
#endif
#if CUDE_AD_PART == 0
// the one-body kernel of the suppression shapes the unrolled kernel does not cover (cude_adaptive.h CUDE_SUPP_AD_UNROLLED
// lists those it does; A/B builds with -DCUDE_ADAPT_ONE_BODY compile them here as well)
#ifdef CUDE_ADAPT_ONE_BODY
#define CUDE_SUPP_AD_SHAPES(X) X(3, 5) X(3, 2) X(4, 2) X(6, 2) X(5, 2) X(3, 3) X(8, 2) X(3, 4) X(4, 3) X(4, 4) X(5, 3) X(6, 3) X(3, 1) X(4, 1) X(6, 1) X(8, 1)
#else
#define CUDE_SUPP_AD_SHAPES(X) X(4, 2) X(6, 2) X(5, 2) X(8, 2) X(4, 3) X(4, 4) X(5, 3) X(6, 3) X(3, 1) X(4, 1) X(6, 1) X(8, 1)
#endif

// (the shapes compiled with the other activation functions: as CUDE_CPEP_GENERAL_SHAPES / CUDE_SUPP_GENERAL_SHAPES)
template <int NIN, int W, int D>
static hipError_t launch_cpep_adaptive_general(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
#define Y(HA, OA) \
    if (net.hact == HA && net.oact == OA) return launch_adaptive<CpepAd<CpepNetG<NIN, W, D, HA, OA>>, true>(a, a.TG, grad, s);
    CUDE_GENERAL_ACTS(Y)
#undef Y
    return hipErrorInvalidValue;
}
template <int W, int D>
static hipError_t launch_supp_adaptive_general(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s) {
#define Y(HA, OA) if (net.hact == HA && net.oact == OA) return launch_adaptive<SuppAd<W, D, HA, OA>, false>(a, 0, grad, s);
    CUDE_GENERAL_ACTS(Y)
#undef Y
    return hipErrorInvalidValue;
}

hipError_t launch_cpep_adaptive(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
    if (a.TG < 2 || a.TG > kMaxObs || a.T < 1) return hipErrorInvalidValue;
    if (grad && a.obs == nullptr) return hipErrorInvalidValue;
#ifndef CUDE_ADAPT_ONE_BODY                        /* (A/B builds: tools/abl_adaptive_bits.py) */
    if (!net.general() && a.TG <= kUnrolledKnots) {
        hipError_t e = launch_cpep_adaptive_team(net, grad, a, s);      // small launches: five waves per 64 subjects
        if (e != hipErrorNotSupported) return e;
        e = launch_cpep_adaptive_unrolled(net, grad, a, s);
        if (e != hipErrorNotSupported) return e;
    }
#endif
    if (net.symbolic())
        return a.cond_raw ? launch_adaptive<CpepAd<MmProd<true>>, true>(a, a.TG, grad, s)
                          : launch_adaptive<CpepAd<MmProd<false>>, true>(a, a.TG, grad, s);
    if (net.general()) {
#define X(NIN, W, D) if (net.nin == NIN && net.width == W && net.depth == D) return launch_cpep_adaptive_general<NIN, W, D>(net, grad, a, s);
        X(2, 4, 2) X(2, 6, 2) X(3, 4, 2)
#undef X
        return hipErrorInvalidValue;
    }
#define X(NIN, W, D) \
    if (net.nin == NIN && net.width == W && net.depth == D) return launch_adaptive<CpepAd<Mlp<NIN, W, D, 1>>, true>(a, a.TG, grad, s);
    CUDE_CPEP_AD_SHAPES_0(X)
#undef X
    hipError_t e = launch_cpep_adaptive_part1(net, grad, a, s);
    if (e == hipErrorNotSupported) e = launch_cpep_adaptive_part2(net, grad, a, s);
    return e == hipErrorNotSupported ? hipErrorInvalidValue : e;
}

hipError_t launch_supp_adaptive(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s) {
    if (net.nin != 4 || a.T < 1) return hipErrorInvalidValue;
#ifndef CUDE_ADAPT_ONE_BODY
    if (!net.general()) {
        const hipError_t e = launch_supp_adaptive_unrolled(net, grad, a, s);
        if (e != hipErrorNotSupported) return e;
    }
#endif
    if (net.general()) {
#define X(W, D) if (net.width == W && net.depth == D) return launch_supp_adaptive_general<W, D>(net, grad, a, s);
        X(3, 5) X(3, 3)
#undef X
        return hipErrorInvalidValue;
    }
#define X(W, D) if (net.width == W && net.depth == D) return launch_adaptive<SuppAd<W, D>, false>(a, 0, grad, s);
    CUDE_SUPP_AD_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}
#elif CUDE_AD_PART == 1
hipError_t launch_cpep_adaptive_part1(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
#define X(NIN, W, D) \
    if (net.nin == NIN && net.width == W && net.depth == D) return launch_adaptive<CpepAd<Mlp<NIN, W, D, 1>>, true>(a, a.TG, grad, s);
    CUDE_CPEP_AD_SHAPES_1(X)
#undef X
    return hipErrorNotSupported;
}
#else
hipError_t launch_cpep_adaptive_part2(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
#define X(NIN, W, D) \
    if (net.nin == NIN && net.width == W && net.depth == D) return launch_adaptive<CpepAd<Mlp<NIN, W, D, 1>>, true>(a, a.TG, grad, s);
    CUDE_CPEP_AD_SHAPES_2(X)
#undef X
    return hipErrorNotSupported;
}
#endif

}  // namespace cude
