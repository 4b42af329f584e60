// Counter-based random draws and the Metropolis accept step, shared by the stand-alone kernels (cude_common.hip) and
// the fused Metropolis step of the time-split path (cude_cpep2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cude_kernels.h"

namespace cude {

// Device-side draws (when the host supplies none): Philox4x32-10 (Salmon et al., SC'11), key = seed, counter =
// (global subject index [64 bits], Metropolis step, kind).  Block kind 0 of a (subject, step) gives the step's standard
// normal (Box-Muller on two 53-bit uniforms), block kind 1 its uniform.  Counter-based: a draw depends only on (seed,
// subject, step), not on the launch shape or on how the subjects are sharded over processes.
__host__ __device__ inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// 53 random bits -> (0, 1): ((x >> 11) + 0.5) * 2^-53
__host__ __device__ inline double u01(uint32_t hi, uint32_t lo) {
    const uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * 1.1102230246251565404e-16;
}
__device__ __forceinline__ void rng_block(const RngKey& key, int64_t i, uint32_t kind, uint32_t (&c)[4]) {
    const uint64_t g = (uint64_t)(key.subject_offset + i);
    c[0] = (uint32_t)g;
    c[1] = (uint32_t)(g >> 32);
    c[2] = (uint32_t)key.step;
    c[3] = ((uint32_t)((uint64_t)key.step >> 32) << 1) | kind;
    philox4x32_10(c, (uint32_t)key.seed, (uint32_t)(key.seed >> 32));
}
__device__ __forceinline__ double rng_normal(const RngKey& key, int64_t i) {
    uint32_t c[4];
    rng_block(key, i, 0u, c);
    const double u1 = u01(c[0], c[1]), u2 = u01(c[2], c[3]);
    return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}
__device__ __forceinline__ double rng_uniform(const RngKey& key, int64_t i) {
    uint32_t c[4];
    rng_block(key, i, 1u, c);
    return u01(c[0], c[1]);
}

// proposal of subject i: state + proposal_std * standard normal (host row z, or the device stream)
__device__ __forceinline__ double mh_proposal(const double* __restrict__ p, const double* __restrict__ z, const RngKey& key,
                                              double proposal_std, int64_t i) {
    return fma(z != nullptr ? z[i] : rng_normal(key, i), proposal_std, p[i]);
}

// The decision of one Metropolis-Hastings step (src/saem.jl:86-108): proposal q with SSE sn against state p with SSE sc,
// uniform draw u.  Its two per-state terms are written so that their bits do not depend on whether the compiler
// contracts a multiply-add (halving is exact; the likelihood's multiply-add is an explicit fma): the stand-alone kernel,
// the fused scan and the speculative resolver -- which forms the terms of all candidates before it walks them -- agree
// bit for bit.
__device__ __forceinline__ double mh_prior_term(const MhArgs& a, double x) {        // -logpdf(Normal(mu, sd), x) + const
    const double z = (x - a.prior_mean) / a.prior_sd;
    return 0.5 * (z * z);
}
__device__ __forceinline__ double mh_tempered_ll(const MhArgs& a, double sse) {     // -Inf on solver failure (:59-62)
    const bool ok = fabs(sse) <= 1.79769313486231570815e308;
    return (ok ? fma(-sse, a.inv_2s2, a.ll_const) : -__builtin_huge_val()) / a.temperature;
}
__device__ __forceinline__ bool mh_decide_terms(double logu, double prior_p, double prior_q, double llt_cur, double llt_new) {
    return logu < (prior_p - prior_q) + (llt_new - llt_cur);            // NaN compares false
}
__device__ __forceinline__ bool mh_decide(const MhArgs& a, double p, double q, double sc, double sn, double u) {
    return mh_decide_terms(log(u), mh_prior_term(a, p), mh_prior_term(a, q), mh_tempered_ll(a, sc), mh_tempered_ll(a, sn));
}

// the stochastic-approximation update of the state (src/saem.jl:177-186): (1 - gamma) p + gamma x, x = the proposal when it was
// accepted, else p itself (which does NOT give p back bit for bit when gamma < 1).  One explicit form: the kernels that
// evaluate the two possible next states before the decision (mh_blend_candidates_kernel) must form the same bits.
__device__ __forceinline__ double mh_blend(double gamma, double p, double x) { return fma(gamma, x, (1.0 - gamma) * p); }

// accept / reject of one Metropolis-Hastings step for subject i (src/saem.jl:86-108 and the stochastic-approximation
// update :177-186): q = proposal, sn = its SSE
__device__ __forceinline__ void mh_accept_one(const MhArgs& a, int64_t i, double q, double sn) {
    const double p = a.p[i];
    const double u = a.u != nullptr ? a.u[i] : rng_uniform(a.key, i);
    const bool acc = mh_decide(a, p, q, a.sse_cur[i], sn, u);
    a.p[i] = mh_blend(a.gamma, p, acc ? q : p);
    if (acc) {
        a.accepted[i] += 1;
        if (a.carry_sse) a.sse_cur[i] = sn;
    }
}

// Resolver of a speculative round for subject i (MhSpecArgs, cude_kernels.h): the decisions of mh_decide /
// mh_accept_one, with the path-independent parts of every node hoisted out of the walk -- same expressions on the same
// values, hence the same bits as d sequential steps.
__device__ __forceinline__ void mh_spec_resolve(const MhSpecArgs& a, int64_t i) {
    constexpr int D = kMhSpecMaxDepth, NODES = (1 << D) - 1;
    const int64_t N = a.mh.N;
    const MhArgs& m = a.mh;
    // draws of both rounds first: they depend on (seed, subject, step) only
    double logu[D], zn[D];
#pragma unroll
    for (int l = 0; l < D; l++) {
        RngKey key = m.key;
        if (l < a.depth_resolve) {
            key.step = a.step_resolve + l;
            logu[l] = log(a.u_rows != nullptr ? a.u_rows[(int64_t)l * N + i] : rng_uniform(key, i));
        }
        if (l < a.depth_next) {
            key.step = a.step_next + l;
            zn[l] = a.z_rows != nullptr ? a.z_rows[(int64_t)l * N + i] : rng_normal(key, i);
        }
    }
    double s = m.p[i];
    if (a.depth_resolve > 0) {
        const int nodes = (1 << a.depth_resolve) - 1;
        double q[NODES], sn[NODES], pri[NODES], llt[NODES];      // per node: proposal, its SSE, -0.5 zq^2 parts, ll / T
#pragma unroll
        for (int v = 0; v < NODES; v++) {
            if (v < nodes) {
                q[v] = a.cand[(int64_t)v * N + i];
                sn[v] = a.sse_sets[(int64_t)v * N + i];
            }
        }
        double sc = m.sse_cur[i];
        // the state's own terms, then every candidate's (what mh_decide forms per step)
        double pri_s = mh_prior_term(m, s), llt_s = mh_tempered_ll(m, sc);
#pragma unroll
        for (int v = 0; v < NODES; v++) {
            if (v < nodes) {
                pri[v] = mh_prior_term(m, q[v]);
                llt[v] = mh_tempered_ll(m, sn[v]);
            }
        }
        int v = 1, nacc = 0;
#pragma unroll
        for (int l = 0; l < D; l++) {
            if (l < a.depth_resolve) {
                // (select the node's values without dynamic register indexing: a chain of compares over the level)
                double qv = 0.0, snv = 0.0, priv = 0.0, lltv = 0.0;
#pragma unroll
                for (int w = (1 << l); w < (2 << l); w++)
                    if (w == v) { qv = q[w - 1]; snv = sn[w - 1]; priv = pri[w - 1]; lltv = llt[w - 1]; }
                const bool acc = mh_decide_terms(logu[l], pri_s, priv, llt_s, lltv);
                s = mh_blend(m.gamma, s, acc ? qv : s);                  // (gamma == 1: the statement of mh_accept_one)
                if (acc) { nacc++; sc = snv; pri_s = priv; llt_s = lltv; }
                v = 2 * v + (acc ? 1 : 0);
                if (a.samples != nullptr) a.samples[(int64_t)l * N + i] = s;
            }
        }
        m.p[i] = s;
        m.sse_cur[i] = sc;
        if (nacc) m.accepted[i] += nacc;
    }
    if (a.depth_next > 0) {
        double st[1 << D];                          // st[v]: state of node v
        st[1] = s;
#pragma unroll
        for (int l = 0; l < D; l++) {
            if (l < a.depth_next) {
#pragma unroll
                for (int v = 1 << l; v < (2 << l); v++) {
                    const double qn = fma(zn[l], a.proposal_std, st[v]);      // (= mh_proposal from that state)
                    a.cand[(int64_t)(v - 1) * N + i] = qn;
                    if (l + 1 < D) { st[2 * v] = st[v]; st[2 * v + 1] = qn; }
                }
            }
        }
    }
}

// The same for gamma < 1 (MhSpecArgs::blend): the state after a step is a BLEND -- mh_blend(s, q) if the proposal was
// accepted, mh_blend(s, s) if not -- whose likelihood is not carried by anything, so every node v of the heap contributes
// TWO parameter sets: its state s(v) (rows [v - 1]) and its proposal q(v) = s(v) + std z_l (rows [nodes + v - 1]);
// s(1) = the chain state, s(2v) = mh_blend(s(v), s(v)), s(2v + 1) = mh_blend(s(v), q(v)).  The walk reads both SSEs of the
// node it stands on.  Same expressions on the same values as mh_accept_one step by step: the same bits.
__device__ __forceinline__ void mh_spec_resolve_blend(const MhSpecArgs& a, int64_t i) {
    constexpr int D = kMhSpecMaxDepthBlend, NODES = (1 << D) - 1;
    const int64_t N = a.mh.N;
    const MhArgs& m = a.mh;
    double logu[D], zn[D];
#pragma unroll
    for (int l = 0; l < D; l++) {
        RngKey key = m.key;
        if (l < a.depth_resolve) {
            key.step = a.step_resolve + l;
            logu[l] = log(a.u_rows != nullptr ? a.u_rows[(int64_t)l * N + i] : rng_uniform(key, i));
        }
        if (l < a.depth_next) {
            key.step = a.step_next + l;
            zn[l] = a.z_rows != nullptr ? a.z_rows[(int64_t)l * N + i] : rng_normal(key, i);
        }
    }
    double s = m.p[i];
    if (a.depth_resolve > 0) {
        const int nodes = (1 << a.depth_resolve) - 1;
        double sv[NODES], q[NODES], pri_s[NODES], pri_q[NODES], llt_s[NODES], llt_q[NODES];
#pragma unroll
        for (int v = 0; v < NODES; v++) {
            if (v < nodes) {
                sv[v] = a.cand[(int64_t)v * N + i];
                q[v] = a.cand[(int64_t)(nodes + v) * N + i];
                pri_s[v] = mh_prior_term(m, sv[v]);
                pri_q[v] = mh_prior_term(m, q[v]);
                llt_s[v] = mh_tempered_ll(m, a.sse_sets[(int64_t)v * N + i]);
                llt_q[v] = mh_tempered_ll(m, a.sse_sets[(int64_t)(nodes + v) * N + i]);
            }
        }
        int v = 1, nacc = 0;
#pragma unroll
        for (int l = 0; l < D; l++) {
            if (l < a.depth_resolve) {
                double svv = 0.0, qv = 0.0, ps = 0.0, pq = 0.0, ls = 0.0, lq = 0.0;
#pragma unroll
                for (int w = (1 << l); w < (2 << l); w++)
                    if (w == v) { svv = sv[w - 1]; qv = q[w - 1]; ps = pri_s[w - 1]; pq = pri_q[w - 1]; ls = llt_s[w - 1]; lq = llt_q[w - 1]; }
                const bool acc = mh_decide_terms(logu[l], ps, pq, ls, lq);
                s = mh_blend(m.gamma, svv, acc ? qv : svv);
                if (acc) nacc++;
                v = 2 * v + (acc ? 1 : 0);
                if (a.samples != nullptr) a.samples[(int64_t)l * N + i] = s;
            }
        }
        m.p[i] = s;
        if (nacc) m.accepted[i] += nacc;
    }
    if (a.depth_next > 0) {
        const int nodes = (1 << a.depth_next) - 1;
        double st[1 << D];
        st[1] = s;
#pragma unroll
        for (int l = 0; l < D; l++) {
            if (l < a.depth_next) {
#pragma unroll
                for (int v = 1 << l; v < (2 << l); v++) {
                    const double qn = fma(zn[l], a.proposal_std, st[v]);      // (= mh_proposal from that state)
                    a.cand[(int64_t)(v - 1) * N + i] = st[v];
                    a.cand[(int64_t)(nodes + v - 1) * N + i] = qn;
                    if (l + 1 < D) { st[2 * v] = mh_blend(m.gamma, st[v], st[v]); st[2 * v + 1] = mh_blend(m.gamma, st[v], qn); }
                }
            }
        }
    }
}

}  // namespace cude
