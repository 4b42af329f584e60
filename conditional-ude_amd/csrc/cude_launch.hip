// libcude_hip.so -- launch selection and the ensemble launches: which kernels a loss / gradient evaluation runs on
// (one lane per subject, time-split, mixed), the second-stage reductions behind them, and every entry point that is a
// batch of such launches (multi-start screening, restarts side by side, per-subject fits, profiles, Metropolis steps).
#include "cude_ctx.h"

namespace cude {
namespace api {

// SUPP gradient launches keep the network activations of the forward sweep (instead of recomputing them in the
// reverse sweep) when that buffer is small: the launch is then latency-bound and 2/3 of the reverse sweep's
// instructions are worth 8*(D*W+1) bytes per evaluation; at 1e5 subjects the 2.3 GB each way would cost more than the
// recomputation.  Option "supp_store" = 0 / 1 overrides the size rule (tests, A/B runs).
size_t supp_act_doubles(const cude_ctx* c) {
    return (size_t)(6 * c->cfg.n_steps + 1) * (size_t)(c->net.depth * c->net.width + 1) * (size_t)c->N;
}
bool supp_keep_activations(const cude_ctx* c, int64_t n_sets) {
    if (c->net.general() || c->net.generic()) return false;      // (other activation functions / shapes: no kept activations)
    if (c->opt.supp_store == 0 || c->opt.supp_store == 1) return c->opt.supp_store == 1;
    return (double)n_sets * (double)supp_act_doubles(c) * 8.0 <= 256e6;
}

// c-peptide gradient launches (one lane per subject, single parameter set): option "cpep_keep" = 2 keeps the upper layers'
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
// "cpep_keep" = 0 (recompute everything) | 1 (keep the output unit's logistic derivative) | 2 (keep the upper layers)
int cpep_keep_mode(const cude_ctx* c) {
    const int m = c->opt.cpep_keep ? c->opt.cpep_keep : CUDE_CPEP_KEEP_DEFAULT;
    return (m == 1 || m == 2) && cpep_act_doubles(c) > 0 ? m : 0;
}
bool cpep_keep_activations(const cude_ctx* c) { return cpep_keep_mode(c) != 0; }

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
    a.gen_acc = c->gen_acc.p;
    a.team = c->opt.adaptive_team ? 0 : -1;
    a.perm = (adaptive(c) && !c->slot_of.empty()) ? c->perm.p : nullptr;
    return a;
}

// option "supp_ckpt" = "steps": gradient launches of the suppression model keep only the step states (744 B per subject at
// S = 30) and re-run the stages in the reverse sweep, instead of keeping every stage input (4.3 KB per subject)
bool supp_steps_only(const cude_ctx* c) { return c->opt.supp_ckpt_steps != 0; }

cude::SuppArgs supp_args(const cude_ctx* c) {
    cude::SuppArgs a{};
    a.ckpt_steps_only = (supp_steps_only(c) && !c->net.general()) ? 1 : 0;
    a.N = c->N;
    a.data = c->data.p;
    a.obs_step = c->obs_step.p; a.obs_w = c->obs_w.p;
    a.rho = c->supp_rho.p; a.obs_rho = c->supp_obs_rho.p;
    a.T = c->T; a.S = c->cfg.n_steps; a.h = step_size(c); a.inv_n = 1.0 / c->n_global;
    for (int s = 0; s < 3; s++) a.iscale2[s] = 1.0 / (c->scale[s] * c->scale[s]);
    a.out_times = c->tp_dev.p;
    a.t_begin = c->tp.front(); a.t_end = c->tp.back();
    a.abstol = c->abstol; a.reltol = c->reltol;
    a.tape = c->tape.p; a.tape_cap = c->tape_cap; a.tape_n = c->tape_n.p;
    a.gen_acc = c->gen_acc.p;
    a.perm = (adaptive(c) && !c->slot_of.empty()) ? c->perm.p : nullptr;
    return a;
}

// Tape of the adaptive gradient: per accepted step and subject the step size (c-peptide models: 8 B) or (t, dt, y) and the
// inputs of stages 2..7 (the suppression model: 136 B), + T saved outputs / 2 T residual derivatives.  The reference's
// problems take 10-40 steps at its tolerances; the capacity is what ~4 GB hold, between 64 and 1024 steps
// (option "tape_steps" overrides).  A subject with more accepted steps fails its gradient evaluation (+Inf), not the
// process.  Allocated by the first gradient evaluation (forward-only users of the adaptive mode never pay for it), never
// under stream capture.
// The fallback kernel of a general network (cude_generic.hip) keeps (t_n, dt_n, y_n) of EVERY step -- in the fixed-step
// mode too -- and a lane's P gradient accumulators in HBM: both are allocated here as well.
int64_t tape_doubles_per_set(const cude_ctx* c) {        // (after ensure_tape)
    const int n_state = c->cfg.model == CUDE_MODEL_SUPP ? 3 : (c->net.generic() ? c->cfg.n_state : 2);
    if (c->net.generic()) return (int64_t)(adaptive(c) ? c->tape_cap : c->cfg.n_steps) * (2 + n_state) * c->N;
    return adaptive(c) ? cude::adaptive_tape_rows(n_state, c->tape_cap, c->T) * c->N : 0;
}

int32_t ensure_tape(cude_ctx* c) {
    if (c->net.generic()) {
        if (c->tape.p && c->gen_acc.p) return CUDE_OK;
        if (c->capturing) return fail(CUDE_ERR_STATE, "gradient scratch of the general network not allocated before stream capture");
        const int rows = 2 + (c->cfg.model == CUDE_MODEL_SUPP ? 3 : c->cfg.n_state);
        int64_t cap = (int64_t)(4e9 / (8.0 * rows * (double)c->N));
        cap = std::max<int64_t>(64, std::min<int64_t>(1024, cap));
        if (c->opt.tape_steps > 0) cap = c->opt.tape_steps;
        c->tape_cap = (int)cap;
        HIP_TRY(c->tape.resize((size_t)tape_doubles_per_set(c)));
        HIP_TRY(c->gen_acc.resize((size_t)c->P * c->N));
        return CUDE_OK;
    }
    if (!adaptive(c) || c->tape.p) return CUDE_OK;
    if (c->capturing) return fail(CUDE_ERR_STATE, "adaptive gradient tape not allocated before stream capture");
    const int64_t N = c->N;
    const int n_state = c->cfg.model == CUDE_MODEL_SUPP ? 3 : 2;
    int64_t cap = (int64_t)(4e9 / (8.0 * cude::adaptive_tape_rows(n_state) * (double)N));
    cap = std::max<int64_t>(64, std::min<int64_t>(1024, cap));
    if (c->opt.tape_steps > 0) cap = c->opt.tape_steps;
    c->tape_cap = (int)cap;
    HIP_TRY(c->tape.resize((size_t)cude::adaptive_tape_rows(n_state, (int)cap, c->T) * N));
    return CUDE_OK;
}

namespace {
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
    a2.adj_map = (c->opt.scan_map && !forward_only) ? c->adj_map.p : nullptr;
    a2.scan_bulk_blocks = c->opt.scan_bulk ? c->n_cu : 0;          // (one workgroup per compute unit: its LDS is the scan's)
    a2.base.blk0 = all_blocks ? 0 : c->blk0; a2.base.blk_count = 0;
    return a2;
}

}  // namespace

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
    c->n_cu = n_cu;
    c->half_slots = (int64_t)n_cu * 4;
    if (adaptive(c)) return CUDE_OK;
    c->slots_one = (int64_t)n_cu * std::max(1, cude::cpep_grad_waves_per_cu(c->net, c->cfg.n_state, c->T));
    c->half_slots = (int64_t)n_cu * 4;
    const int forced = c->opt.cpep_path;        // option "cpep_path": 0 = the cost model below decides
    if (forced == 1) return CUDE_OK;
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
    // Small populations (at most two workgroups per compute unit before splitting): the slot model above takes every
    // resident wave for full throughput, but a SIMD gives 1, 1.33, 1.36, 1.38 ... of a lone wave's rate to 1, 2, 3, 4+
    // waves (profiles/r02/ubench_fma_latency.txt) -- it chose 15 chunks for 1e4 and 2e4 subjects where 6 are 5 % and 11 %
    // faster, and 30 for 4 000 where 15 are 8 % faster (profiles/r05/sweep_chunks_small.txt).  Model fitted to that sweep:
    // a wave of 5S/L + 3 evaluations at 1.36 us each, slowed by the waves it shares its SIMD with, + 0.25 us of scan per chunk.
    if (c->nblocks <= 2 * (int64_t)n_cu) {
        const double simds = (double)n_cu * 4.0;
        int Ls = 0;
        double best_s = 0.0;
        for (int d = 2; d <= S; d++) {
            if (S % d) continue;
            const double w = std::ceil((double)c->nblocks * d / simds);
            const double thr = w <= 1.0 ? 1.0 : (w <= 2.0 ? 1.33 : (w <= 3.0 ? 1.36 : 1.38));
            const double cost = 1.36 * (5.0 * S / d + 3.0) * w / thr + 0.25 * d;
            if (Ls == 0 || cost < best_s) { best_s = cost; Ls = d; }
        }
        if (Ls >= 2) L = Ls;
    }
    // Mixed launch (more than one machine-fill of subjects): whole rounds of the one-lane kernel, the remainder --
    // which would otherwise be a second, mostly idle round -- time-split on its own.  Cost = the two parts one after
    // the other (they do overlap at the seam; not counted).
    int64_t blk0 = 0;
    const int64_t slots_one = (int64_t)n_cu * occ_one;
    // Only between one and two machine-fills: with two or more whole rounds the one-lane launch's own tail is amortised
    // and the mixed launch measured slower (300 000 subjects 1.461 against 1.366 ms, 1e6 4.415 against 4.183 ms).
    if (c->nblocks > slots_one && c->nblocks < 2 * slots_one && c->opt.mixed) {
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
    if (blk0 == 0 && occ_one == 8 && c->nblocks > half && c->nblocks < slots_one && c->opt.mixed) {
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
    if (blk0 == 0 && occ_one >= 12 && c->opt.mixed) {
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
    if (c->opt.debug_selector)
        fprintf(stderr, "[cude] chunk selector: nblocks=%lld CUs=%d waves/CU one-lane=%d reverse=%d -> L=%d, one-lane blocks %lld\n",
                (long long)c->nblocks, n_cu, occ_one, occ_rev, L, (long long)blk0);
    if (forced == 2) { L = c->opt.path_chunks; blk0 = 0; }
    if (forced == 3) {                                        // "3:<one-lane blocks>:<L>" (tests)
        blk0 = std::min<int64_t>(std::max<int64_t>(c->opt.path_blk0, 0), c->nblocks - 1);
        L = c->opt.path_chunks;
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
    if (blk0 > 0 && !c->stream2 && !c->opt.mixed_one_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    cude::CpepArgs a = cpep_args(c);
    cude::Cpep2Args a2 = chunk_args(c, a);
    HIP_TRY(cude::cpep2_prepare());
    HIP_TRY(cude::launch_cpep2_homog(a2, c->stream));
    HIP_TRY(c->adj_map.resize((size_t)cude::adj_map_rows(T) * N));          // the scan's adjoint recursion as a linear map
    HIP_TRY(cude::launch_cpep2_adjmap(a2, c->adj_map.p, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // cs (host vector) dies here
    // ---- the forward-only split.  Model (fitted to profiles/r03/forward_chunks.txt): a SIMD that holds w waves of e
    // evaluations each needs e * w / thr(w) evaluation times (thr = 1, 1.33, 1.36, 1.38 ... for 1, 2, 3, 4+ waves: the
    // issue rates of profiles/r02/ubench_fma_latency.txt), e = 5S/L + 2, and the scan adds ~0.6 evaluation times per chunk.
    c->chunks_f = 0;
    if (c->opt.fwd_split && forced != 2 && forced != 3) {
        const double simds = (double)n_cu * 4.0;
        int Lf = 0;
        double best_f = 0.0;
        for (int d = 2; d <= S; d++) {
            if (S % d) continue;
            const double w = std::ceil((double)c->nblocks * d / simds);
            const double thr = w <= 1.0 ? 1.0 : (w <= 2.0 ? 1.33 : (w <= 3.0 ? 1.36 : 1.38));
            // (0.6 per chunk dates from the one-wave scan; the eight-wave scan stitches a chunk in ~0.1 us and plain forward
            //  calls of <= 2 000 subjects would gain ~1 us from 30 chunks instead of 15 -- but the speculative Metropolis
            //  rounds launch 3 or 7 parameter sets on this split, and with 30 chunks their waves no longer have a SIMD each:
            //  1 250 subjects x 100 steps 1.25 -> 1.34 ms.  Kept.)
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
        if (c->opt.debug_selector)
            fprintf(stderr, "[cude] forward-only split: L_f=%d (gradient L=%d)\n", Lf, L);
    }
    return CUDE_OK;
}

// Alternating issue priority in the one-lane gradient kernel (CpepArgs::prio_shift): for a launch of `blocks` workgroups
// that is a single round with two waves on (some of) the SIMDs.  (Ablation builds: CUDE_PRIO_SHIFT=k overrides, 0 = never.)
int prio_shift_for(const cude_ctx* c, int64_t blocks) {
    if (c->opt.prio_shift >= 0) return c->opt.prio_shift;
    // two waves on (some of) the SIMDs and no third: also the kernels that could hold three (2-4-4-1 / 2 states at
    // 120 000 ... 131 072 subjects: 0.451 -> 0.426 ms); not the one-wave kernels (2-7-7-1: +2 %)
    return (c->slots_one >= 2 * c->half_slots && blocks > c->half_slots && blocks <= 2 * c->half_slots) ? 5 : 0;
}

// launches the ensemble kernel + second-stage reduction (+ all-reduce, + L2 term)
// cond_ov / sse_ov: evaluate at other conditional parameters / write the per-subject SSE elsewhere and stop
// after the ensemble kernels (used by the Metropolis E-step, which needs neither loss nor gradient).
int32_t run_ensemble(cude_ctx* c, bool grad, double* traj_dev, bool local_only, const double* cond_ov, double* sse_ov) {
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
            if (!grad && !sse_ov && !distributed(c) && c->cfg.lambda == 0.0 && c->pinned_pairs &&
                c->pinned_pairs_n >= c->nblocks && !c->capturing && c->opt.fused_final) {
                a2.final_host = c->pinned_pairs;
                fused_final = true;
                // the host watches the pairs arrive (finish_loss) instead of going through the runtime's completion wait
                // (forward call at 1e4 subjects 56.9 -> 51.6 us, at 57 subjects 41.3 -> 36.6 us): every slot starts as a
                // NaN no kernel produces.  Option "poll_pinned" = 0 (CUDE_NO_POLL_PINNED=1): plain hipStreamSynchronize.
                c->poll_pairs = c->opt.poll_pinned && c->poll_ok && watch;
                if (c->poll_pairs) {
                    volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(c->pinned_pairs);
                    for (int64_t q = 0; q < 2 * c->nblocks; q++) w[q] = kPairSentinel;
                }
            }
            a2.defer_chunk_sum = grad && c->blk0 == 0 && c->opt.fused_tail;      // (summed by the one tail launch below)
            HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, grad, a2, c->stream));
        } else {
            if (grad && !adaptive(c)) a.prio_shift = prio_shift_for(c, c->nblocks);
            if (grad && adaptive(c)) {      // adaptive gradient kernel: at most two waves per SIMD (measured -4.8 %)
                a.prio_shift = c->opt.prio_shift >= 0
                                   ? c->opt.prio_shift
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
    if (adaptive(c)) {
        c->have_counts = true;              // (adaptive launches leave every subject's accepted-step count behind)
        c->have_tape = grad;                // (a forward launch overwrites the counts cude_adaptive_steps pairs the tape with)
    }
    if (sse_ov) return CUDE_OK;
    const int P = c->P;
    const bool fold = c->fold_advance && grad && !local_only;
    // who sums over the ranks: the reduction kernels themselves through the peer-write exchange (no further launch; the
    // pair [sum loss, n_failed] they leave is already global), or an RCCL all-reduce behind them
    const bool xc = c->xchg.ready && !local_only;
    const bool rccl = !xc && c->comm != nullptr && !local_only;
    const cude::XchgArgs xargs = xc ? xchg_args(c) : cude::XchgArgs{};
    const cude::XchgArgs* xq = xc ? &xargs : nullptr;
    const cude::TailAdvance adv_args = tail_advance(c);
    const cude::TailAdvance* adv_red = (fold && !rccl && c->cfg.lambda == 0.0) ? &adv_args : nullptr;
    const cude::TailAdvance* adv_l2 = (fold && c->cfg.lambda != 0.0) ? &adv_args : nullptr;
    c->advance_done = adv_red != nullptr || adv_l2 != nullptr;
    // no L2 term, no RCCL call behind, not capturing: the reduction that finishes [sum loss, n_failed] writes the pair into
    // the page-locked result buffer too, and finish_loss watches it arrive instead of queueing a copy and waiting for the stream
    double* host_tail = nullptr;
    {
        // (not with the exchange: a wait that gave up is reported through the status word, read after the stream's end)
        if (c->opt.poll_pinned && c->poll_ok && watch && c->pinned && !rccl && !xc && c->cfg.lambda == 0.0 && !c->capturing &&
            !local_only && !fused_final) {
            host_tail = c->pinned + P;
            volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(host_tail);
            w[0] = kPairSentinel; w[1] = kPairSentinel;
        }
        c->tail_in_pinned = host_tail != nullptr;
    }
    if (grad && is_cpep(c) && c->chunks > 1 && c->blk0 > 0) {
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->blk0, P + 2, 0, P, c->g_nn.p, c->stream, 1, c->param_mask.p, P));
        HIP_TRY(cude::launch_reduce_cols(c->partials2.p, (c->nblocks - c->blk0) * c->chunks, P, 0, P, c->g_nn.p, c->stream, 1,
                                         c->param_mask.p, P, 0, /*accumulate=*/true, nullptr, nullptr, xq));
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, P, 2, c->g_nn.p, c->stream, 1, nullptr, 0, 0, false,
                                         adv_red, host_tail, xq));
    } else if (grad && is_cpep(c) && c->chunks > 1 && c->opt.fused_tail) {
        // one launch: the network gradient over the reverse chunks' rows, the loss / failure columns over the scan's rows
        // (+ state advance, page-locked pair, exchange) and the chunks' shares of the conditional gradient
        cude::ChunkedTailArgs ta{};
        ta.partials2 = c->partials2.p; ta.rows2 = c->nblocks * c->chunks;
        ta.partials = c->partials.p; ta.rows = c->nblocks;
        ta.P = P; ta.out = c->g_nn.p; ta.out_stride = P + 2;
        ta.mask = c->param_mask.p; ta.n_mask = P;
        if (adv_red) ta.adv = *adv_red;
        ta.host_tail = host_tail;
        if (xq) ta.xchg = *xq;
        ta.g_cond_part = c->g_cond_part.p; ta.L = c->chunks; ta.N = c->N; ta.g_cond = c->g_cond.p; ta.g_cond_set_stride = c->N;
        HIP_TRY(cude::launch_chunked_tail(ta, 1, c->stream));
    } else if (grad && is_cpep(c) && c->chunks > 1) {
        HIP_TRY(cude::launch_reduce_cols(c->partials2.p, c->nblocks * c->chunks, P, 0, P, c->g_nn.p, c->stream, 1,
                                         c->param_mask.p, P, 0, false, nullptr, nullptr, xq));
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, P, 2, c->g_nn.p, c->stream, 1, nullptr, 0, 0, false,
                                         adv_red, host_tail, xq));
    } else if (grad) {
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, 0, P + 2, c->g_nn.p, c->stream, 1,
                                         c->param_mask.p, P, 0, false, adv_red, host_tail, xq));
    } else if (fused_final) {
        c->loss_in_pinned = true;
    } else {
        HIP_TRY(cude::launch_reduce_cols(c->partials.p, c->nblocks, P + 2, P, 2, c->g_nn.p, c->stream, 1, nullptr, 0, 0, false,
                                         nullptr, host_tail, xq));
    }
    if (local_only) return CUDE_OK;   // the caller reduces across ranks and applies the L2 term
    if (rccl) {
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
    bool arrived = false, watch_timed_out = false;
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
            watch_timed_out = true;
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
        watch_timed_out = !arrived;
    }
    c->poll_pairs = false;
    if (!arrived) HIP_TRY(hipStreamSynchronize(c->stream));
    if (watch_timed_out && !distributed(c)) {
        // Either this launch takes longer than the watch's limit (then the few microseconds a watch saves do not matter to
        // it) or device stores to this host memory do not become visible while a kernel runs (a non-coherent mapping:
        // every later watch would burn its whole limit): in both cases this context goes back to plain stream waits.
        c->poll_ok = false;
        if (c->opt.debug_selector)
            fprintf(stderr, "[cude] a watched result did not arrive within its time limit: plain stream waits from now on\n");
    }
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
    if (c->xchg.ready) {        // a wait of the exchange that ran out of time (in ANY column, not only the loss's)
        if (arrived) HIP_TRY(hipStreamSynchronize(c->stream));
        const int32_t xrc = xchg_check(c);
        if (xrc) return xrc;
    }
    c->last_failed = (int64_t)std::llround(tmp[P + 1]);
    if (g_nn_host) std::memcpy(g_nn_host, tmp, P * sizeof(double));
    if (loss) *loss = (c->last_failed > 0 || !std::isfinite(tmp[P])) ? std::numeric_limits<double>::infinity()
                                                                     : tmp[P] / c->n_global;
    return CUDE_OK;
}

// (see cude_adaptive_regroup in include/cude.h)
// Entry points that launch a large adaptive population again and again (restarts trained side by side, per-subject
// fits) keep the launch ordered by accepted-step count as cude_adam_run does: once the counts of a first evaluation are
// there, then after every kRegroupEvals-th evaluation they make (the counts drift with the parameters).  Per-subject
// results do not depend on the order; the shared gradient's summation order does (rounding).
int32_t maybe_regroup(cude_ctx* c) {
    constexpr int64_t kRegroupEvals = 200;
    if (!adaptive(c) || c->N < 8192 || !c->opt.auto_regroup || !c->have_counts || c->capturing || c->net.generic()) return CUDE_OK;
    if (!c->slot_of.empty() && c->evals_since_regroup < kRegroupEvals) { c->evals_since_regroup++; return CUDE_OK; }
    return adaptive_regroup(c, nullptr, nullptr);
}

int32_t adaptive_regroup(cude_ctx* c, int32_t* spread_before, int32_t* spread_after) {
    if (c->net.generic()) return fail(CUDE_ERR_UNSUPPORTED, "the fallback kernel of a general network runs in the caller's order");
    if (!adaptive(c) || !c->have_counts) return fail(CUDE_ERR_STATE, "no adaptive evaluation on this context yet");
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
    c->evals_since_regroup = 0;
    return CUDE_OK;
}

// Bytes of scratch one multi-set launch may use (partial rows, stage inputs, tapes): sets per launch = this / the
// scratch of one set.  More sets per launch fill the chip better (25 sets of 1e5 subjects: 39 000 waves in one grid
// instead of 25 grids that each end in a part-filled round), and the per-set results do not depend on it.
double sets_scratch_budget(cude_ctx* c) {
    if (c->scratch_budget > 0.0) return c->scratch_budget;
    size_t free_b = 0, total_b = 0;
    c->scratch_budget = 512e6;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b >= ((size_t)64 << 30))
        c->scratch_budget = std::min(16e9, std::max(512e6, 0.25 * (double)free_b));
    else
        (void)hipGetLastError();
    return c->scratch_budget;
}

// Loss sums and gradients of n_sets parameter sets that are ALREADY on the device (set k: nn + k * stride_nn,
// cond + k * stride_cond), everything queued on the context's stream, no synchronisation, no host copy:
//   g_cond + k * stride_cond   <- dL/dcond of set k                               [N]
//   out + k * (P + 2)          <- [masked network gradient (P); sum of SSE; failed subjects], summed over the ranks
// (the L2 term and the loss value are the caller's: launch_finish_sets).  The set index is the grid's second dimension
// (third on the time-split path); sets are launched in groups as large as the scratch budget allows.
int32_t eval_sets_device(cude_ctx* c, int64_t n_sets, const double* nn, int64_t stride_nn, const double* cond,
                         int64_t stride_cond, double* g_cond, double* out) {
    int32_t rc = CUDE_OK;
    const int P = c->P, S = c->cfg.n_steps;
    const int64_t N = c->N, nb = c->nblocks;
    const bool supp = c->cfg.model == CUDE_MODEL_SUPP;
    // Small populations: K restarts of a few dozen subjects are K single-wave chains on the one-lane kernel (25 waves on
    // 1024 SIMDs, each paying the full single-wave latency).  When the population itself runs time-split (chunks > 1) and
    // the sets do not fill the chip either, the time-split kernels take the set index as a third grid dimension: K x L
    // short waves instead.  Same kernels as cude_loss_grad on this context, so a set's result is bit-identical to it.
    const int L = c->chunks;
    const bool split = !supp && L > 1 && c->blk0 == 0 && nb * (int64_t)std::min<int64_t>(n_sets, 64) <= 512 &&
                       c->opt.ms_split;
    if ((rc = ensure_tape(c))) return rc;                 // (fixes the capacity the per-set tapes share)
    if ((rc = maybe_regroup(c))) return rc;
    const bool gen = c->net.generic();
    const int64_t tape_rows = tape_doubles_per_set(c) / N;
    const double per_set = 8.0 * ((double)nb * (P + 2) + (double)tape_rows * N + (gen ? (double)P * N : 0.0) +
                                  (supp && !adaptive(c) ? (double)cude::supp_ckpt_rows(S, c->T) * N : 0.0) +
                                  (split ? (double)L * (3 + c->T) * N + 5.0 * S * N + (double)L * N + (double)L * nb * P : 0.0));
    // sets per launch: bounded by the grid's y / z dimension and the scratch budget
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_sets, split ? 16384 : 32768),
                                                                 (int64_t)(sets_scratch_budget(c) / per_set)));
    if (split) {
        HIP_TRY(c->ms_fsum.reserve((size_t)chunk * L * (3 + c->T) * N));
        HIP_TRY(c->ms_wts.reserve((size_t)chunk * 5 * S * N));
        HIP_TRY(c->ms_gcp.reserve((size_t)chunk * L * N));
        HIP_TRY(c->ms_p2.reserve((size_t)chunk * L * nb * P));
    }
    HIP_TRY(c->ms_part.reserve((size_t)chunk * nb * (P + 2)));
    if (supp && !adaptive(c) && !gen) HIP_TRY(c->ms_ckpt.reserve((size_t)chunk * cude::supp_ckpt_rows(S, c->T) * N));
    if (adaptive(c) || gen) HIP_TRY(c->ms_tape.reserve((size_t)chunk * tape_rows * N));
    if (gen) HIP_TRY(c->ms_gacc.reserve((size_t)chunk * P * N));
    for (int64_t k0 = 0; k0 < n_sets; k0 += chunk) {
        const int64_t kn = std::min<int64_t>(chunk, n_sets - k0);
        const double* nn_k = nn + k0 * stride_nn;
        const double* cond_k = cond + k0 * stride_cond;
        double* g_cond_k = g_cond + k0 * stride_cond;
        double* out_k = out + k0 * (P + 2);
        if (split) {
            cude::CpepArgs a = cpep_args(c);
            a.cond = cond_k; a.nn = nn_k;
            a.g_cond = g_cond_k; a.partials = c->ms_part.p;
            a.sse = nullptr; a.traj = nullptr; a.auc = nullptr;
            a.n_sets = (int32_t)kn; a.set_stride_nn = stride_nn; a.set_stride_cond = stride_cond;
            cude::Cpep2Args a2 = chunk_args(c, a);
            a2.fsum = c->ms_fsum.p; a2.wts = c->ms_wts.p; a2.g_cond_part = c->ms_gcp.p; a2.partials2 = c->ms_p2.p;
            a2.defer_chunk_sum = c->opt.fused_tail;
            HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, true, a2, c->stream));
            // network gradient: the reverse chunks' partial rows; loss / failure columns: the scan's
            if (c->opt.fused_tail) {
                cude::ChunkedTailArgs ta{};
                ta.partials2 = c->ms_p2.p; ta.rows2 = nb * L; ta.partials = c->ms_part.p; ta.rows = nb;
                ta.P = P; ta.out = out_k; ta.out_stride = P + 2; ta.mask = c->param_mask.p; ta.n_mask = P;
                ta.g_cond_part = c->ms_gcp.p; ta.L = L; ta.N = N; ta.g_cond = g_cond_k; ta.g_cond_set_stride = stride_cond;
                HIP_TRY(cude::launch_chunked_tail(ta, (int)kn, c->stream));
            } else {
                HIP_TRY(cude::launch_reduce_cols(c->ms_p2.p, nb * L, P, 0, P, out_k, c->stream, (int)kn, c->param_mask.p, P, P + 2));
                HIP_TRY(cude::launch_reduce_cols(c->ms_part.p, nb, P + 2, P, 2, out_k, c->stream, (int)kn));
            }
        } else if (!supp) {
            cude::CpepArgs a = cpep_args(c);
            a.cond = cond_k; a.nn = nn_k;
            a.g_cond = g_cond_k; a.partials = c->ms_part.p;
            a.n_sets = (int32_t)kn; a.set_stride_nn = stride_nn; a.set_stride_cond = stride_cond;
            if (adaptive(c) || gen) a.tape = c->ms_tape.p;     // (tape_n: the counts of set 0, for the re-ordering)
            if (gen) a.gen_acc = c->ms_gacc.p;
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, true, a, c->stream));     // one-lane kernel: the sets fill the chip
        } else {
            cude::SuppArgs a = supp_args(c);
            a.cond = cond_k; a.nn = nn_k;
            a.ckpt = c->ms_ckpt.p; a.g_cond = g_cond_k; a.partials = c->ms_part.p;
            if (adaptive(c) || gen) a.tape = c->ms_tape.p;
            if (gen) a.gen_acc = c->ms_gacc.p;
            if (!adaptive(c) && !a.ckpt_steps_only && supp_keep_activations(c, kn)) {
                HIP_TRY(c->ms_act.reserve((size_t)kn * supp_act_doubles(c)));
                a.act = c->ms_act.p;
            }
            a.n_sets = (int32_t)kn; a.set_stride_nn = stride_nn; a.set_stride_cond = stride_cond;
            HIP_TRY(cude::launch_supp(c->net, true, a, c->stream));
        }
        if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }   // (step counts of the first set; its tape is ms_tape)
        if (!split)
            HIP_TRY(cude::launch_reduce_cols(c->ms_part.p, nb, P + 2, 0, P + 2, out_k, c->stream, (int)kn, c->param_mask.p, P));
        if (distributed(c) && (rc = allreduce_dev(c, out_k, (size_t)kn * (P + 2)))) return rc;
    }
    return CUDE_OK;
}

}  // namespace api
}  // namespace cude

using namespace cude::api;

extern "C" {

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
        if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }
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
        if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }
        HIP_TRY(cude::launch_reduce_sets(d_part.p, (int)kn, nb, P + 2, P, d_out.p, c->stream));
        if (distributed(c) && (rc = allreduce_dev(c, d_out.p, (size_t)kn * 2))) return rc;
        HIP_TRY(hipMemcpyAsync(h_out.data(), d_out.p, kn * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->xchg.ready && (rc = xchg_check(c))) return rc;
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
        if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }
        HIP_TRY(cude::launch_reduce_sets(d_part.p, (int)kn, nb, P + 2, P, d_sums.p, c->stream));
        if (distributed(c) && (rc = allreduce_dev(c, d_sums.p, (size_t)kn * 2))) return rc;
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
        if (c->xchg.ready && (rc = xchg_check(c))) return rc;
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
    const int P = c->P;
    const int64_t N = c->N, K = n_sets;
    HIP_TRY(c->ms_nn.reserve((size_t)K * P));
    HIP_TRY(c->ms_cond.reserve((size_t)K * N));
    HIP_TRY(c->ms_gcond.reserve((size_t)K * N));
    HIP_TRY(c->ms_out.reserve((size_t)K * (P + 2)));
    HIP_TRY(c->ms_f.reserve((size_t)K));
    HIP_TRY(hipMemcpyAsync(c->ms_nn.p, nn_sets, (size_t)K * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ms_cond.p, cond_sets, (size_t)K * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = eval_sets_device(c, K, c->ms_nn.p, P, c->ms_cond.p, N, c->ms_gcond.p, c->ms_out.p))) return rc;
    // loss and L2 term per set, in the arithmetic of l2_term_kernel: a set's loss and gradient are bit-identical to
    // cude_loss_grad at the same parameters
    cude::FinishSetsArgs fa{};
    fa.P = P; fa.out = c->ms_out.p; fa.nn = c->ms_nn.p; fa.stride_nn = P; fa.lambda = c->cfg.lambda; fa.n_global = c->n_global;
    fa.mask = c->param_mask.p; fa.f = c->ms_f.p;
    HIP_TRY(cude::launch_finish_sets(fa, (int)K, c->stream));
    HIP_TRY(hipMemcpyAsync(losses, c->ms_f.p, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpy2DAsync(g_nn_sets, (size_t)P * sizeof(double), c->ms_out.p, (size_t)(P + 2) * sizeof(double),
                             (size_t)P * sizeof(double), (size_t)K, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(g_cond_sets, c->ms_gcond.p, (size_t)K * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return c->xchg.ready ? xchg_check(c) : CUDE_OK;
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
    // (the fallback kernel of a general network keeps (t_n, dt_n, y_n) for either model, as the suppression kernels do)
    const bool supp = c->cfg.model == CUDE_MODEL_SUPP || c->net.generic();
    const int rows = c->net.generic() ? 2 + (c->cfg.model == CUDE_MODEL_SUPP ? 3 : c->cfg.n_state)
                                      : cude::adaptive_tape_rows(supp ? 3 : 2);
    const int m = std::min(std::min(n, cap), c->tape_cap);
    const int64_t slot = c->slot_of.empty() ? subject : c->slot_of[(size_t)subject];      // the tape is in launch order
    if (m < 1) return CUDE_OK;
    // entry k, row r of the tape: one double every rows * N.  Suppression model: rows 0 and 1 are (t_n, dt_n); c-peptide
    // models keep dt_n alone and t_n is the forward sweep's own running sum t_{n+1} = t_n + dt_n from the initial time
    // (the same additions in the same order: the same bits)
    std::vector<double> dts;
    double* dt_dst = dt_out;
    if (!supp && !dt_dst) { dts.resize((size_t)m); dt_dst = dts.data(); }
    if (supp && t_out)
        HIP_TRY(hipMemcpy2DAsync(t_out, sizeof(double), c->tape.p + slot, (size_t)rows * c->N * sizeof(double), sizeof(double),
                                 (size_t)m, hipMemcpyDeviceToHost, c->stream));
    if (dt_dst)
        HIP_TRY(hipMemcpy2DAsync(dt_dst, sizeof(double), c->tape.p + (size_t)(supp ? 1 : 0) * c->N + slot,
                                 (size_t)rows * c->N * sizeof(double), sizeof(double), (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!supp && t_out) {
        double t = c->tp.front();
        for (int k = 0; k < m; k++) { t_out[k] = t; t += dt_dst[k]; }
    }
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
        if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }
        HIP_TRY(hipMemcpyAsync(sse_out + k0 * N, d_sse.p, kn * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
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
    // Several probes per forward launch (option "fit_spec"): every launch below is one wave's whole solve per 64 subjects,
    // and a small population leaves the chip empty -- the grid values ride as parameter sets of one launch, and because a
    // golden-section step has two outcomes, the probes of the next d steps are a heap of 2^d - 1 brackets evaluated together
    // (cude_common.hip fit_tree_*).  The same expressions on the same values: the same brackets and result as the
    // one-probe-per-launch form, 138 forward launches -> 1 + 48 / d + 1.
    int fdepth = c->opt.fit_spec;
    const bool fsplit = is_cpep(c) && !adaptive(c) && c->chunks > 1;
    if (fdepth < 0) {
        const int64_t waves1 = c->nblocks * (fsplit ? (c->chunks_f > 1 ? c->chunks_f : c->chunks) : 1);
        const int64_t room = fsplit ? 4096 : 1024;
        fdepth = 30 * waves1 <= room ? 4 : (14 * waves1 <= room ? 3 : (6 * waves1 <= room ? 2 : 1));
    }
    if (c->net.generic()) fdepth = 0;
    fdepth = std::min(fdepth, (int)cude::kFitSpecMaxDepth);
    if (fdepth >= 1) {
        const int P = c->P;
        const int64_t nb = c->nblocks;
        const int Lf = fsplit ? (c->chunks_f > 1 ? c->chunks_f : c->chunks) : 1;
        // sets per launch: the tree's, and for the grid scan as many as ~256 MB of per-set scratch allow
        const int tree_sets = 2 * ((1 << fdepth) - 1);
        const double per_set = 8.0 * ((double)N * (2 + (fsplit ? (double)Lf * (3 + c->T) : 0.0)) + (double)nb * (P + 2));
        const int grid_sets = (int)std::max<int64_t>(1, std::min<int64_t>(n_grid, (int64_t)(256e6 / per_set)));
        const int max_sets = std::max(tree_sets, grid_sets);
        DevBuf<double> d_cand, d_sse, d_part, d_vals;
        HIP_TRY(d_cand.resize((size_t)max_sets * N));
        HIP_TRY(d_sse.resize((size_t)max_sets * N));
        HIP_TRY(d_vals.resize((size_t)n_grid));
        if (fsplit) {
            HIP_TRY(c->ms_fsum.reserve((size_t)max_sets * Lf * (3 + c->T) * N));
            HIP_TRY(c->ms_part.reserve((size_t)max_sets * nb * (P + 2)));
        } else {
            HIP_TRY(d_part.resize((size_t)max_sets * nb * (P + 2)));
        }
        auto solve_sets = [&](int n_sets) -> int32_t {           // SSE of every subject at cand[set][subject]
            if (is_cpep(c)) {
                cude::CpepArgs a = cpep_args(c);
                a.cond = d_cand.p; a.nn = c->nn.p; a.sse = d_sse.p; a.traj = nullptr; a.auc = nullptr;
                a.g_cond = c->g_cond.p; a.partials = fsplit ? c->ms_part.p : d_part.p;
                a.n_sets = n_sets; a.set_stride_nn = 0; a.set_stride_cond = N;
                if (fsplit) {
                    cude::Cpep2Args a2 = chunk_args(c, a, /*all_blocks=*/true, /*forward_only=*/true);
                    a2.fsum = c->ms_fsum.p;
                    HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, false, a2, c->stream));
                } else {
                    HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, false, a, c->stream));
                }
            } else {
                cude::SuppArgs a = supp_args(c);
                a.cond = d_cand.p; a.nn = c->nn.p; a.sse = d_sse.p; a.traj = nullptr;
                a.g_cond = c->g_cond.p; a.partials = d_part.p;
                a.n_sets = n_sets; a.set_stride_nn = 0; a.set_stride_cond = N;
                HIP_TRY(cude::launch_supp(c->net, false, a, c->stream));
            }
            if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }
            return CUDE_OK;
        };
        std::vector<double> vals((size_t)n_grid);
        for (int k = 0; k < n_grid; k++) vals[k] = (k == n_grid - 1) ? upper : std::fma((double)k, f.step, lower);
        HIP_TRY(hipMemcpyAsync(d_vals.p, vals.data(), (size_t)n_grid * sizeof(double), hipMemcpyHostToDevice, c->stream));
        for (int k0 = 0; k0 < n_grid; k0 += grid_sets) {      // coarse scan of the box
            const int kn = std::min(grid_sets, n_grid - k0);
            HIP_TRY(cude::launch_fill_rows(N, kn, d_vals.p + k0, d_cand.p, c->stream));
            if ((rc = solve_sets(kn))) return rc;
            HIP_TRY(cude::launch_fit_grid_all(f, k0, kn, d_vals.p, d_sse.p, c->stream));
            if (k0 == 0 && (rc = maybe_regroup(c))) return rc; // (adaptive mode: the first launch has told the step counts)
        }
        HIP_TRY(hipStreamSynchronize(c->stream));             // vals (host vector) was read by the copy above
        HIP_TRY(cude::launch_fit(1, f, 0, 0.0, c->stream));
        for (int it = 0; it < n_iters;) {                     // golden section inside the bracket, d steps per launch
            const int d = std::min(fdepth, n_iters - it);
            HIP_TRY(cude::launch_fit_tree(f, d, 0, 0, d_cand.p, d_sse.p, c->stream));
            if ((rc = solve_sets(2 * ((1 << d) - 1)))) return rc;
            it += d;
            HIP_TRY(cude::launch_fit_tree(f, d, 1, it == n_iters ? 1 : 0, d_cand.p, d_sse.p, c->stream));
        }
        if ((rc = run_ensemble(c, false, nullptr, true, f.c, sse_c))) return rc;       // at the returned midpoint
        HIP_TRY(cude::launch_fit(3, f, 0, 0.0, c->stream));
        HIP_TRY(hipMemcpyAsync(cond_out, f.c, N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (objective_out) HIP_TRY(hipMemcpyAsync(objective_out, f.fc, N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (sse_out) HIP_TRY(hipMemcpyAsync(sse_out, sse_c, N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return CUDE_OK;
    }
    // everything below is queued on the stream; the only synchronisation is the copy-back at the end
    for (int k = 0; k < n_grid; k++) {                    // coarse scan of the box
        const double x = (k == n_grid - 1) ? upper : std::fma((double)k, f.step, lower);
        HIP_TRY(cude::launch_fill(N, x, f.c, c->stream));
        if (k == 1 && (rc = maybe_regroup(c))) return rc;     // (adaptive mode: the first probe has told the step counts)
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

}  // extern "C"

namespace {
// Steps per speculative round of cude_mh_chain (0 = one step per launch pair, the plain fused path).  Option "mh_spec":
// 0 off, 2 ... 4 forced; -1 (default) by size: a round of depth d evaluates 2^d - 1 candidates per subject in one launch,
// which is nearly free while the candidates' forward chunks run side by side on otherwise idle SIMDs and costs their
// full multiple once the chip is filled.  Rule = waves of the round's forward launch (candidates x workgroups x chunks of
// the forward split): depth 3 while they fit two waves per SIMD (the resident-weights variant of the forward kernel),
// depth 2 up to ~3 per SIMD.  profiles/r05/estep_speculative.txt (one MI355X, 2-4-4-1, 100 steps, us per step plain ->
// speculative): 625 subjects 23.8 -> 12.2 (depth 3), 1 250: 23.9 -> 13.7 (2; 14.5 with 3), 1 600: 23.7 -> 16.0 (2; 14.6 with 3),
// 2 500: 23.6 -> 16.5 (2), 5 000: 25.7 -> 23.5 (2), 1e4: 30.9 -> 29.6 (2), 12 000 and above: slower, off.  With the
// eight-wave scan (which helps the plain step more) 1e4 is a draw (30.2 -> 29.6 in that tool, 30.1 -> 31.2 in bench.py's
// 1e4 x 100): off from ~8 000 subjects.
int mh_spec_depth(const cude_ctx* c, int n_mc) {
    int d = c->opt.mh_spec;
    if (d < 0) {
        const int64_t waves1 = c->nblocks * (c->chunks_f > 1 ? c->chunks_f : c->chunks);
        d = 7 * waves1 <= 2048 ? 3 : (3 * waves1 <= 2400 ? 2 : 0);
    }
    d = std::min(d, cude::kMhSpecMaxDepth);
    if (d > n_mc) d = n_mc;
    return d >= 2 ? d : 0;
}

// The same speculation for the contexts whose forward solve is ONE launch over lanes = subjects (the adaptive solve -- the
// API mirrors' default --, the one-lane fixed-step kernel): a Metropolis step there is a wave's whole solve (84 us for
// 57 ... 1 000 subjects in the adaptive mode) on a chip that a small population leaves empty, so the 2^d - 1 candidate
// states run side by side as parameter sets of the launch (grid rows) and mh_spec_kernel resolves the d decisions.
// Depth while the candidates' workgroups still have a SIMD each (profiles/r05/estep_speculative.txt, adaptive mode, 100
// steps, ms per E-step plain -> depth 2 / 3 / 4: 57 subjects 10.4 -> 5.6 / 4.0 / 3.1, 1 000: 12.1 -> 6.3 / 4.5 / 3.5,
// 4 000: 12.4 -> - / 6.0 / 4.7, 1e4: 12.6 -> 8.7 / 8.3 / -).
int mh_spec_depth_one_launch(const cude_ctx* c, int n_mc) {
    int d = c->opt.mh_spec;
    if (d < 0) d = 15 * c->nblocks <= 1024 ? 4 : (7 * c->nblocks <= 1100 ? 3 : (3 * c->nblocks <= 1024 ? 2 : 0));
    d = std::min(d, cude::kMhSpecMaxDepth);
    if (d > n_mc) d = n_mc;
    return d >= 2 ? d : 0;
}
}  // namespace

extern "C" {

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
    const bool fused = m.carry_sse && is_cpep(c) && !adaptive(c) && c->chunks > 1 && c->opt.mh_fuse;
    // Speculative steps (MhSpecArgs, cude_kernels.h): d steps per dependent launch pair -- forward chunks of the 2^d - 1
    // candidate states as parameter sets of ONE launch, then a scan launch that holds a subject's candidates in one
    // workgroup and resolves the d decisions behind their SSEs -- instead of one pair per step.
    // Pays while the candidates still fit the chip beside each other (a shard of an E-step spread over 8 GPUs).
    const int spec = fused ? mh_spec_depth(c, n_mc) : 0;
    if (spec >= 2) {
        const int P = c->P, Lf = (c->chunks_f > 1 ? c->chunks_f : c->chunks), T = c->T;
        const int64_t nb = c->nblocks, max_sets = (1 << spec) - 1;
        DevBuf<double> d_cand, d_sse_sets;
        HIP_TRY(d_cand.resize((size_t)max_sets * N));
        HIP_TRY(d_sse_sets.resize((size_t)max_sets * N));
        HIP_TRY(c->ms_fsum.reserve((size_t)max_sets * Lf * (3 + T) * N));
        HIP_TRY(c->ms_part.reserve((size_t)max_sets * nb * (P + 2)));
        cude::MhSpecArgs sa{};
        sa.mh = m;
        sa.mh.key = cude::RngKey{c->rng_seed, c->rng_offset, 0};
        sa.cand = d_cand.p; sa.sse_sets = d_sse_sets.p; sa.proposal_std = proposal_std;
        sa.depth_resolve = 0; sa.depth_next = std::min(spec, n_mc);
        sa.step_resolve = sa.step_next = c->rng_step;
        sa.z_rows = device_rng ? nullptr : d_z.p;
        HIP_TRY(cude::launch_mh_spec(sa, c->stream));                 // the first round's candidates
        int round = 0;
        for (int k = 0; k < n_mc; round++) {
            const int d = std::min(spec, n_mc - k), sets = (1 << d) - 1;
            cude::CpepArgs a = cpep_args(c);
            a.cond = d_cand.p; a.nn = c->nn.p; a.sse = d_sse_sets.p; a.traj = nullptr; a.auc = nullptr;
            a.g_cond = c->g_cond.p; a.partials = c->ms_part.p;
            a.n_sets = sets; a.set_stride_nn = 0; a.set_stride_cond = N;       // one network, `sets` candidate states
            cude::Cpep2Args a2 = chunk_args(c, a, /*all_blocks=*/true, /*forward_only=*/true);
            a2.fsum = c->ms_fsum.p;
            sa.depth_resolve = d;
            sa.depth_next = std::min(spec, n_mc - k - d);
            sa.step_resolve = c->rng_step + k;
            sa.step_next = c->rng_step + k + d;
            sa.u_rows = device_rng ? nullptr : d_u.p + (size_t)k * N;
            sa.z_rows = device_rng ? nullptr : d_z.p + (size_t)(k + d) * N;
            // (caller's draws + samples: the states of steps k .. k+d-1 overwrite the normals of those steps, which the
            //  PREVIOUS resolver consumed; the next round's normals are rows k+d ...)
            sa.samples = samples ? d_z.p + (size_t)k * N : nullptr;
            a2.spec_slots = 1 << d;             // the scan launch resolves the round itself
            a2.spec = sa;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (c->timing && round % 4 == 0) {
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
            k += d;
        }
        if (samples)
            HIP_TRY(hipMemcpyAsync(samples, d_z.p, (size_t)n_mc * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (accepted) HIP_TRY(hipMemcpyAsync(accepted, d_acc.p, N * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (device_rng) c->rng_step += n_mc;
        return CUDE_OK;
    }
    const int spec1 = (!fused && m.carry_sse && is_cpep(c) && (adaptive(c) || c->chunks <= 1) && !c->net.generic())
                          ? mh_spec_depth_one_launch(c, n_mc) : 0;
    if (spec1 >= 2) {
        const int P = c->P;
        const int64_t nb = c->nblocks, max_sets = (1 << spec1) - 1;
        DevBuf<double> d_cand, d_sse_sets, d_part;
        HIP_TRY(d_cand.resize((size_t)max_sets * N));
        HIP_TRY(d_sse_sets.resize((size_t)max_sets * N));
        HIP_TRY(d_part.resize((size_t)max_sets * nb * (P + 2)));
        cude::MhSpecArgs sa{};
        sa.mh = m;
        sa.mh.key = cude::RngKey{c->rng_seed, c->rng_offset, 0};
        sa.cand = d_cand.p; sa.sse_sets = d_sse_sets.p; sa.proposal_std = proposal_std;
        sa.depth_resolve = 0; sa.depth_next = std::min(spec1, n_mc);
        sa.step_resolve = sa.step_next = c->rng_step;
        sa.z_rows = device_rng ? nullptr : d_z.p;
        HIP_TRY(cude::launch_mh_spec(sa, c->stream));                 // the first round's candidates
        for (int k = 0; k < n_mc;) {
            const int d = std::min(spec1, n_mc - k), sets = (1 << d) - 1;
            cude::CpepArgs a = cpep_args(c);
            a.cond = d_cand.p; a.nn = c->nn.p; a.sse = d_sse_sets.p; a.traj = nullptr; a.auc = nullptr;
            a.g_cond = c->g_cond.p; a.partials = d_part.p;
            a.n_sets = sets; a.set_stride_nn = 0; a.set_stride_cond = N;       // one network, `sets` candidate states
            HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, false, a, c->stream));
            sa.depth_resolve = d;
            sa.depth_next = std::min(spec1, n_mc - k - d);
            sa.step_resolve = c->rng_step + k;
            sa.step_next = c->rng_step + k + d;
            sa.u_rows = device_rng ? nullptr : d_u.p + (size_t)k * N;
            sa.z_rows = device_rng ? nullptr : d_z.p + (size_t)(k + d) * N;
            sa.samples = samples ? d_z.p + (size_t)k * N : nullptr;            // (as in the time-split rounds above)
            HIP_TRY(cude::launch_mh_spec(sa, c->stream));
            k += d;
        }
        if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }      // (step counts of the last round's set 0)
        if (samples)
            HIP_TRY(hipMemcpyAsync(samples, d_z.p, (size_t)n_mc * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (accepted) HIP_TRY(hipMemcpyAsync(accepted, d_acc.p, N * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (device_rng) c->rng_step += n_mc;
        return CUDE_OK;
    }
    // gamma < 1 (the stochastic-approximation phase): the next state is a blend whose likelihood is not known, so a step
    // needed two solves one after the other (the proposal, then the current state again).  The two possible next states
    // depend on (p, q, gamma) alone: they are solved in the SAME launch as the proposal, as three parameter sets, and the
    // decision picks state and SSE (mh_accept_blend_kernel) -- one solve launch per step, the same chain.
    if (!m.carry_sse && is_cpep(c) && !c->net.generic() && c->opt.mh_pair) {
        const int P = c->P, sets = 3;
        const int64_t nb = c->nblocks;
        const bool split = !adaptive(c) && c->chunks > 1;
        DevBuf<double> d_cand, d_sse3, d_part;
        HIP_TRY(d_cand.resize((size_t)sets * N));
        HIP_TRY(d_sse3.resize((size_t)sets * N));
        if (split) {
            const int Lf = (c->chunks_f > 1 ? c->chunks_f : c->chunks);
            HIP_TRY(c->ms_fsum.reserve((size_t)sets * Lf * (3 + c->T) * N));
            HIP_TRY(c->ms_part.reserve((size_t)sets * nb * (P + 2)));
        } else {
            HIP_TRY(d_part.resize((size_t)sets * nb * (P + 2)));
        }
        // ... and d such steps per launch by speculation (MhSpecArgs::blend: a node of the candidate heap is its state AND
        // its proposal, 2 (2^d - 1) parameter sets) while the sets' waves still have a SIMD each
        int spec3 = c->opt.mh_spec;
        {
            const int64_t waves1 = nb * (split ? (c->chunks_f > 1 ? c->chunks_f : c->chunks) : 1);
            // (profiles/r05/estep_speculative.txt, gamma = 0.25, 100 steps, ms per E-step two launches per step -> one -> depth
            //  2 / 3: adaptive 57 subjects 20.3 -> 11.0 -> 5.6 / 4.0, 4 000: 24.0 -> 12.8 -> 8.7 / 6.0, 1e4: 24.3 -> 17.3 -> 8.7 /
            //  11.5; fixed 30 steps (time-split) 57: 4.10 -> 2.39 -> 1.27 / 1.01, 1 000: 4.06 -> 2.39 -> 1.51 / 1.57, 4 000:
            //  4.28 -> 3.68 -> 3.02 / 3.61, 1e4: 5.94 -> 6.17 -> 5.10 / 6.62)
            if (spec3 < 0) {
                if (split) spec3 = 14 * waves1 <= 2048 ? 3 : (6 * waves1 <= 6000 ? 2 : 0);
                else spec3 = 14 * waves1 <= 1024 ? 3 : (6 * waves1 <= 1024 ? 2 : 0);
            }
            spec3 = std::min(std::min(spec3, (int)cude::kMhSpecMaxDepthBlend), (int)n_mc);
            if (spec3 < 2) spec3 = 0;
        }
        if (spec3 >= 2) {
            const int64_t max_sets = 2 * ((1 << spec3) - 1);
            HIP_TRY(d_cand.resize((size_t)max_sets * N));
            HIP_TRY(d_sse3.resize((size_t)max_sets * N));
            if (split) {
                const int Lf = (c->chunks_f > 1 ? c->chunks_f : c->chunks);
                HIP_TRY(c->ms_fsum.reserve((size_t)max_sets * Lf * (3 + c->T) * N));
                HIP_TRY(c->ms_part.reserve((size_t)max_sets * nb * (P + 2)));
            } else {
                HIP_TRY(d_part.resize((size_t)max_sets * nb * (P + 2)));
            }
            cude::MhSpecArgs sa{};
            sa.mh = m;
            sa.mh.key = cude::RngKey{c->rng_seed, c->rng_offset, 0};
            sa.blend = 1;
            sa.cand = d_cand.p; sa.sse_sets = d_sse3.p; sa.proposal_std = proposal_std;
            sa.depth_resolve = 0; sa.depth_next = std::min(spec3, n_mc);
            sa.step_resolve = sa.step_next = c->rng_step;
            sa.z_rows = device_rng ? nullptr : d_z.p;
            HIP_TRY(cude::launch_mh_spec(sa, c->stream));             // the first round's candidates
            for (int k = 0; k < n_mc;) {
                const int d = std::min(spec3, n_mc - k), nsets = 2 * ((1 << d) - 1);
                cude::CpepArgs a = cpep_args(c);
                a.cond = d_cand.p; a.nn = c->nn.p; a.sse = d_sse3.p; a.traj = nullptr; a.auc = nullptr;
                a.g_cond = c->g_cond.p; a.partials = split ? c->ms_part.p : d_part.p;
                a.n_sets = nsets; a.set_stride_nn = 0; a.set_stride_cond = N;
                if (split) {
                    cude::Cpep2Args a2 = chunk_args(c, a, /*all_blocks=*/true, /*forward_only=*/true);
                    a2.fsum = c->ms_fsum.p;
                    HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, false, a2, c->stream));
                } else {
                    HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, false, a, c->stream));
                }
                sa.depth_resolve = d;
                sa.depth_next = std::min(spec3, n_mc - k - d);
                sa.step_resolve = c->rng_step + k;
                sa.step_next = c->rng_step + k + d;
                sa.u_rows = device_rng ? nullptr : d_u.p + (size_t)k * N;
                sa.z_rows = device_rng ? nullptr : d_z.p + (size_t)(k + d) * N;
                sa.samples = samples ? d_z.p + (size_t)k * N : nullptr;
                HIP_TRY(cude::launch_mh_spec(sa, c->stream));
                k += d;
            }
            if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }
            if (samples)
                HIP_TRY(hipMemcpyAsync(samples, d_z.p, (size_t)n_mc * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            if (accepted) HIP_TRY(hipMemcpyAsync(accepted, d_acc.p, N * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (device_rng) c->rng_step += n_mc;
            return CUDE_OK;
        }
        if ((rc = run_ensemble(c, false, nullptr, true, c->cond.p, d_sc.p))) return rc;      // SSE of the starting state
        for (int k = 0; k < n_mc; k++) {
            m.key = cude::RngKey{c->rng_seed, c->rng_offset, c->rng_step + k};
            HIP_TRY(cude::launch_mh_blend_candidates(N, c->cond.p, device_rng ? nullptr : d_z.p + (size_t)k * N, m.key,
                                                     proposal_std, gamma, d_cand.p, c->stream));
            cude::CpepArgs a = cpep_args(c);
            a.cond = d_cand.p; a.nn = c->nn.p; a.sse = d_sse3.p; a.traj = nullptr; a.auc = nullptr;
            a.g_cond = c->g_cond.p; a.partials = split ? c->ms_part.p : d_part.p;
            a.n_sets = sets; a.set_stride_nn = 0; a.set_stride_cond = N;       // one network, three candidate states
            if (split) {
                cude::Cpep2Args a2 = chunk_args(c, a, /*all_blocks=*/true, /*forward_only=*/true);
                a2.fsum = c->ms_fsum.p;
                HIP_TRY(cude::launch_cpep2(c->net, c->cfg.n_state, false, a2, c->stream));
            } else {
                HIP_TRY(cude::launch_cpep(c->net, c->cfg.n_state, false, a, c->stream));
            }
            m.u = device_rng ? nullptr : d_u.p + (size_t)k * N;
            HIP_TRY(cude::launch_mh_accept_blend(m, d_cand.p, d_sse3.p, c->stream));
            if (samples)
                HIP_TRY(hipMemcpyAsync(d_z.p + (size_t)k * N, c->cond.p, N * sizeof(double), hipMemcpyDeviceToDevice,
                                       c->stream));
        }
        if (adaptive(c)) { c->have_counts = true; c->have_tape = false; }
        if (samples)
            HIP_TRY(hipMemcpyAsync(samples, d_z.p, (size_t)n_mc * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (accepted) HIP_TRY(hipMemcpyAsync(accepted, d_acc.p, N * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (device_rng) c->rng_step += n_mc;
        return CUDE_OK;
    }
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

}  // extern "C"
