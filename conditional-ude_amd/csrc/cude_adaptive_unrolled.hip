// Adaptive Tsit5 ensemble kernels for gfx950, stages unrolled -- the c-peptide models on the reference's sampling grid
// (solve(model.problem, p = theta, saveat = timepoints), src/parameter-estimation.jl:59; src/saem.jl:52).
//
#include "cude_adaptive.h"

namespace cude {

// The kernel of cude_adaptive.hip walks ONE network body through every phase: small code, but each evaluation pays for
// it -- the stage sums run over LDS rows with tableau entries fetched by index (a scalar load and an LDS round trip per
// term, exposed at the one or two waves per SIMD a 1e5-subject launch has).  For the constant-Jacobian (c-peptide)
// models this variant unrolls the six stages of a trial step and the five VJPs of a reversed step: stage derivatives
// and their adjoints live in registers, tableau entries are literals, the glucose slope of each knot interval is formed
// once (same operands, same quotient as CpepAd::forcing_input) instead of once per evaluation, and the knot search has
// no loop.  Same arithmetic in the same order: results are bit-identical to that kernel
// (tests/test_gpu_adaptive_grad.py).  At 1e5 subjects, 2x4x4x1: forward 0.282 -> 0.195 ms, gradient 0.524 -> 0.369 ms.

// stage adjoints of the step being reversed: registers, or LDS rows for the networks whose gradient accumulators fill
// the register file
#ifndef CUDE_ADAPT_BLDS_NACC
#define CUDE_ADAPT_BLDS_NACC 40
#endif
template <class M, bool GRAD>
constexpr bool unrolled_adjoints_in_lds() { return GRAD && M::NetT::NACC > CUDE_ADAPT_BLDS_NACC; }
template <class M, bool GRAD>
constexpr int unrolled_fixed_rows() { return kRedRows + (unrolled_adjoints_in_lds<M, GRAD>() ? 7 * M::NS : 0); }

template <class M, bool GRAD>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(adaptive_waves<M, GRAD>())))
void adaptive_unrolled_kernel(typename M::Args a) {
    static_assert(!M::NEED_Y, "constant-Jacobian models only");
    constexpr int NS = M::NS;
    constexpr int P = M::P;
    constexpr bool B_LDS = unrolled_adjoints_in_lds<M, GRAD>();
    constexpr int FIXED = unrolled_fixed_rows<M, GRAD>();
    // [kRedRows] reduction scratch, (large networks) 7 NS rows of stage adjoints, then TG glucose rows and TG - 1 slope rows
    extern __shared__ double smem[];
    double* const s_B = smem + kRedRows * kBlock;
    const int lane = threadIdx.x;
    if constexpr (M::NetT::USES_TANH) tanh_tab_init(lane);
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < a.N;
    const int64_t slot = active ? gid : a.N - 1;
    const int64_t i = a.perm != nullptr ? (int64_t)a.perm[slot] : slot;
    const int64_t set = blockIdx.y;
    cptr_t tout = as_const(a.out_times);
    const int n_out = a.T;

    M m;
    double y[NS];
    const double chk = m.init(a, smem + FIXED * kBlock, lane, i, set, y);
    double* const s_S = smem + (FIXED + a.TG) * kBlock;
    for (int j = 0; j + 1 < a.TG; j++)
        s_S[j * kBlock + lane] = (m.s_G[(j + 1) * kBlock + lane] - m.s_G[j * kBlock + lane]) / (m.tp[j + 1] - m.tp[j]);
    // knot search: the (at most kUnrolledKnots - 2) interior knots are held in scalar registers and compared without a
    // loop, so the searches of a step's five stage times overlap instead of queueing behind one scalar load each
    // (longer sampling grids run the kernel of cude_adaptive.hip)
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
    if (GRAD) TAPE(0) = 0.0;
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
    // ---- NN([0; e^beta]), k1 = f(t0, y0) and the f1 probe of Hairer's initial-step heuristic: one network body
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
    int prio_shift = 0;
    unsigned prio_par = 0, it = 0;
    if constexpr (GRAD) {
        prio_shift = a.prio_shift;
        prio_par = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)) & 1u;       // HW_ID.wave_id
    }
#pragma unroll 1
    while (true) {
        if (GRAD && prio_shift > 0) {              // (six evaluations per trial step)
            if ((((it++) >> (prio_shift > 2 ? prio_shift - 2 : 0)) ^ prio_par) & 1u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        dt = fmin(dt, t1 - t);
        double ynew[NS];
        double prod_last = 0.0;
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
            // the forcing depends on time only and c_6 = c_7 = 1: stage 7 reuses stage 6's value
            if (st != 6) prod_last = m.production(forcing(st < 6 ? fma(TS_C[st], dt, t) : t + dt));
            m.finish_rhs(prod_last, Y, K[st]);
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
                    if (GRAD) OUTV(nxt) = o[0];
                    nxt++;
                }
            }
        }
        if (GRAD && live && !failed && accept) {
            if (n_acc < a.tape_cap) {
                TAPE(n_acc) = dt;
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
        using Net = typename M::NetT;
        double acc[Net::NACC];
#pragma unroll
        for (int q = 0; q < Net::NACC; q++) acc[q] = 0.0;
        double lam[NS], wsum = 0.0, carry = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) lam[s] = 0.0;
        const double gs = a.inv_n;
        int hi = n_out;
        int n_max = n_acc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off, 64));
        double t_next = t;
        double h_ahead = 0.0;
        if (n_max > 0) h_ahead = TAPE(n_max - 1 < n_acc ? n_max - 1 : (n_acc > 0 ? n_acc - 1 : 0));
#pragma unroll 1
        for (int n = n_max - 1; n >= 0; n--) {
            if (prio_shift > 0) {
                if ((((unsigned)n >> (prio_shift > 2 ? prio_shift - 2 : 0)) ^ prio_par) & 1u) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(0);
            }
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
            double wacc = carry;
#pragma unroll
            for (int sq = 6; sq >= 0; sq--) {
                __builtin_amdgcn_sched_barrier(0);             // one VJP body at a time: the accumulators fill the file
                double kb[NS], ub[NS];
#pragma unroll
                for (int s = 0; s < NS; s++) { kb[s] = B.get(sq, s); ub[s] = sq == 6 ? lam[s] : 0.0; }
                m.vjp_linear(kb, ub);
                if (sq == 6) {
                    wacc += kb[0];
                } else if (sq == 0) {
                    carry = kb[0];
                } else {
                    double dx[1] = {0.0};
                    const double xx[1] = {forcing(sq == 5 ? tn + h : fma(TS_C[sq], h, tn))};
                    Net::template eval_grad<false, decltype(acc), kAdaptivePin>(m.p, m.c, xx, sq == 5 ? wacc + kb[0] : kb[0], acc, dx);
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
        }
        if (active && a.tape_n != nullptr && set == 0) a.tape_n[i] = n_acc;
        double cst[M::NCST];
        m.finish_grad(a, i, set, acc, wsum, carry, cst);
        __syncthreads();
        if (active) a.g_cond[set * a.set_stride_cond + i] = Net::grad_cond(m.p, acc, cst);
        block_reduce_expand<Net, M::NCST>(acc, cst, active ? 1.0 : 0.0, active ? sse : 0.0, (active && bad) ? 1.0 : 0.0,
                                          smem, out, lane);
    }
#undef TAPE
#undef OUTV
}

template <class M>
static hipError_t launch_unrolled(const typename M::Args& a, bool grad, hipStream_t s) {
    const int64_t nblocks = (a.N + kBlock - 1) / kBlock;
    const unsigned n_sets = a.n_sets > 0 ? (unsigned)a.n_sets : 1u;
    const size_t lds = sizeof(double) * (size_t)((grad ? unrolled_fixed_rows<M, true>() : unrolled_fixed_rows<M, false>()) +
                                                  2 * a.TG) * kBlock;
    if (grad) {
        if (a.tape == nullptr || a.tape_cap < 1 || a.g_cond == nullptr) return hipErrorInvalidValue;
        hipLaunchKernelGGL((adaptive_unrolled_kernel<M, true>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a);
    } else {
        hipLaunchKernelGGL((adaptive_unrolled_kernel<M, false>), dim3((unsigned)nblocks, n_sets), dim3(kBlock), lds, s, a);
    }
    return hipGetLastError();
}

#ifndef CUDE_AD_PART
#define CUDE_AD_PART 0
#endif
#if CUDE_AD_PART == 0
hipError_t launch_cpep_adaptive_unrolled(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
    if (net.general()) return hipErrorNotSupported;
    if (a.TG < 2 || a.TG > kUnrolledKnots || a.T < 1) return hipErrorNotSupported;
    if (grad && a.obs == nullptr) return hipErrorInvalidValue;
    if (net.symbolic())
        return a.cond_raw ? launch_unrolled<CpepAd<MmProd<true>>>(a, grad, s) : launch_unrolled<CpepAd<MmProd<false>>>(a, grad, s);
#define X(NIN, W, D) \
    if (net.nin == NIN && net.width == W && net.depth == D) return launch_unrolled<CpepAd<Mlp<NIN, W, D, 1>>>(a, grad, s);
    CUDE_CPEP_AD_SHAPES_0(X)
#undef X
    const hipError_t e = launch_cpep_adaptive_unrolled_part1(net, grad, a, s);
    return e != hipErrorNotSupported ? e : launch_cpep_adaptive_unrolled_part2(net, grad, a, s);
}
#elif CUDE_AD_PART == 1
hipError_t launch_cpep_adaptive_unrolled_part1(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
#define X(NIN, W, D) \
    if (net.nin == NIN && net.width == W && net.depth == D) return launch_unrolled<CpepAd<Mlp<NIN, W, D, 1>>>(a, grad, s);
    CUDE_CPEP_AD_SHAPES_1(X)
#undef X
    return hipErrorNotSupported;
}
#else
hipError_t launch_cpep_adaptive_unrolled_part2(const NetShape& net, bool grad, const CpepArgs& a, hipStream_t s) {
#define X(NIN, W, D) \
    if (net.nin == NIN && net.width == W && net.depth == D) return launch_unrolled<CpepAd<Mlp<NIN, W, D, 1>>>(a, grad, s);
    CUDE_CPEP_AD_SHAPES_2(X)
#undef X
    return hipErrorNotSupported;
}
#endif

}  // namespace cude
