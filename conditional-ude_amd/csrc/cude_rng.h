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

// accept / reject of one Metropolis-Hastings step for subject i (src/saem.jl:86-108 and the stochastic-approximation
// update :177-186): q = proposal, sn = its SSE
__device__ __forceinline__ void mh_accept_one(const MhArgs& a, int64_t i, double q, double sn) {
    const double p = a.p[i];
    // logpdf(Normal(mu, sd), x) differences: the normalisation cancels
    const double zq = (q - a.prior_mean) / a.prior_sd, zp = (p - a.prior_mean) / a.prior_sd;
    const double prior_ratio = -0.5 * zq * zq + 0.5 * zp * zp;
    const double inf = __builtin_huge_val();
    const double sc = a.sse_cur[i];
    const bool okn = fabs(sn) <= 1.79769313486231570815e308, okc = fabs(sc) <= 1.79769313486231570815e308;
    const double ll_new = okn ? a.ll_const - sn * a.inv_2s2 : -inf;     // -Inf on solver failure (:59-62)
    const double ll_cur = okc ? a.ll_const - sc * a.inv_2s2 : -inf;
    const double ratio = ll_new / a.temperature - ll_cur / a.temperature;
    const double u = a.u != nullptr ? a.u[i] : rng_uniform(a.key, i);
    const bool acc = log(u) < prior_ratio + ratio;                      // NaN compares false
    a.p[i] = (1.0 - a.gamma) * p + a.gamma * (acc ? q : p);
    if (acc) {
        a.accepted[i] += 1;
        if (a.carry_sse) a.sse_cur[i] = sn;
    }
}

}  // namespace cude
