// Branch-free fp64 activations for gfx950.
//
// The path is bound by fp64 VALU issue and ~70 % of the arithmetic is tanh/softplus
// (SURVEY.md 8(d)).  The ROCm device-library tanh/exp/log keep <1 ulp over the whole double
// range with double-double arithmetic (~250 VALU ops per tanh on gfx950); the network only
// needs ABSOLUTE accuracy ~1e-15 on outputs that are O(1).  Measured issue costs on MI355X
// (tools/ubench/valu_rates.hip): every fp64 VALU op (fma, mul, add, min, rndne, cvt, ldexp)
// takes one 1.71 ns SIMD slot per wave-instruction, v_rcp_f64 takes four.  Hence:
//   exp(2s) : Cody-Waite reduction to |s| <= ln2/4 + degree-10 near-minimax polynomial
//             (tools/fit_exp_poly.py, max rel err 3.8e-16) + v_ldexp_f64           (16 slots)
//   1/d     : v_rcp_f64 + one cubic Newton step r0*(1+e+e^2)                          (7 slots)
//   tanh    : addition theorem on a grid, tanh(a+b) = (T_a + tanh b)/(1 + T_a tanh b) with a = k/8 the nearest grid
//             point (T_a from a 161-entry table, one LDS read on the device), |b| <= 1/16 and tanh b replaced by its
//             [3/4] Pade approximant n/d (continued fraction cut at 7: error b^9/99225 <= 1.5e-16):
//             tanh|x| = (T d + n)/(d + T n) -- 13 slots for numerator and denominator (the exponential it replaces
//             took 18: round 3, -5 slots of 24 per tanh); the W quotients of a layer share ONE v_rcp_f64 through
//             prefix products (3(W-1) multiplies).  -DCUDE_TANH_EXP selects the previous form,
//             sign(x)*(1 - 2/(exp(2|x|)+1)).
//   softplus(x) = max(x,0) + log1p(exp(-|x|)); logistic derivative from the same exponential
// Reference semantics: softplus(x) = log(1+exp(x)) (src/neural-network.jl:13-15); the stable form
// differs from it by rounding only (and does not overflow for x > 709).
// Max abs error vs libm is asserted in tests/test_math_host.py.
#pragma once
#include <math.h>

#include "cude_tanh_table.h"
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CUDE_HD __host__ __device__ __forceinline__
#else
#define CUDE_HD inline
#endif

#ifndef CUDE_FMA_C_CONSTRAINT
#define CUDE_FMA_C_CONSTRAINT "v"
#endif

namespace cude {

// 1/d for d in [1, 1e290]
CUDE_HD double m_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rcp(d);       // ~2^-23 relative
#else
    double r0 = 1.0 / d;                              // host model of a low-precision seed:
    {                                                 // keep 23 mantissa bits of the exact quotient
        unsigned long long u;
        __builtin_memcpy(&u, &r0, 8);
        u &= ~((1ull << 29) - 1ull);
        __builtin_memcpy(&r0, &u, 8);
    }
#endif
    const double e = fma(-d, r0, 1.0);
    const double t = fma(e, e, e);                    // e + e^2 ; error after the step is e^3
    return fma(r0, t, r0);
}

// fma(a, b, c) with a constant addend.  Left to itself LLVM forms the two-address v_fmac for a Horner step whose
// addend it has hoisted into VGPRs and copies the constant into the accumulator first (a v_mov_b64 -- a VALU slot -- per
// step, and two VGPRs per constant); written out, the addend is an SGPR pair of a three-address v_fma_f64.
CUDE_HD double m_fma_c(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CUDE_NO_FMA_C)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), CUDE_FMA_C_CONSTRAINT(c));
    return r;
#else
    return fma(a, b, c);
#endif
}
// the same with the addend in an SGPR pair (no VGPRs held; the constant is rematerialised by the scalar unit)
CUDE_HD double m_fma_cs(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CUDE_NO_FMA_CS)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return fma(a, b, c);
#endif
}

template <bool CS>
CUDE_HD double m_fma_sel(double a, double b, double c) {
    if constexpr (CS) return m_fma_cs(a, b, c);
    else return fma(a, b, c);
}

// exp(2 x) for x in [-354, 354].  CS: Horner steps written as three-address FMAs with SGPR addends (m_fma_cs) -- for the
// kernels whose ONLY exponential is the output unit's (table tanh): there LLVM hoists the ten coefficients into VGPRs and
// copies one into the accumulator per step; where the tanh layers evaluate exponentials too the coefficients already
// live in SGPRs and the plain form schedules better (the asm statements are not interleaved with neighbouring chains).
template <bool CS>
CUDE_HD double m_exp2x_t(double x) {
#ifdef CUDE_MAGIC_ROUND
    // round-to-nearest through the 1.5*2^52 trick: the integer lands in the low mantissa bits of t, so the
    // fp64-rate v_rndne_f64 / v_cvt_i32_f64 / v_ldexp_f64 become one subtraction and one 32-bit shift-add
    const double t = fma(x, 2.88539008177792681472, 6755399441055744.0);
    const double n = t - 6755399441055744.0;
    long long tb;
    __builtin_memcpy(&tb, &t, 8);
    const int ni = (int)(unsigned)tb;
#else
    const double n = rint(x * 2.88539008177792681472);                  // 2*log2(e)
    const int ni = (int)n;
#endif
    double s = fma(n, -3.46573590184561908245e-01, x);                   // ln2/2 hi
    s = fma(n, -9.54107464635293850010e-11, s);                          // ln2/2 lo ; |s| <= ln2/4
    double p = 2.82893898152419560454e-04;
    p = m_fma_sel<CS>(p, s, 1.41517725676594432853e-03);
    p = m_fma_sel<CS>(p, s, 6.34918511283140765689e-03);
    p = m_fma_sel<CS>(p, s, 2.53966979461632859361e-02);
    p = m_fma_sel<CS>(p, s, 8.88888891679270320978e-02);
    p = m_fma_sel<CS>(p, s, 2.66666668341369039741e-01);
    p = m_fma_sel<CS>(p, s, 6.66666666665170271067e-01);
    p = m_fma_sel<CS>(p, s, 1.33333333332435244323e+00);
    p = m_fma_sel<CS>(p, s, 2.00000000000000222045e+00);
    p = m_fma_sel<CS>(p, s, 2.00000000000001332268e+00);
    p = fma(p, s, 1.0);
#ifdef CUDE_MAGIC_ROUND
    // p in [0.70, 1.42] and |n| <= 1022 for the argument range used (|x| <= 354): adding n to the exponent
    // field cannot leave the normal range
    long long pb;
    __builtin_memcpy(&pb, &p, 8);
    pb += (long long)ni << 52;
    __builtin_memcpy(&p, &pb, 8);
    return p;
#else
    return ldexp(p, ni);
#endif
}

CUDE_HD double m_exp2x(double x) { return m_exp2x_t<false>(x); }

CUDE_HD double m_exp(double y) { return m_exp2x(0.5 * y); }

// denominators exp(2|z|)+1 of tanh, clamped so that products of up to 8 of them stay finite
CUDE_HD double m_tanh_den(double z) { return m_exp2x(fmin(fabs(z), 20.0)) + 1.0; }

CUDE_HD double m_tanh(double x) {
    const double r = m_rcp(m_tanh_den(x));
    return copysign(fma(-2.0, r, 1.0), x);
}

// ---- tanh by table + addition theorem (see the header comment)
constexpr int kTanhEntries = CUDE_TANH_KMAX + 1;
// 1.5 * 2^(52 - log2 grid): adding it rounds |x| to the grid, leaving k = round(|x| * grid) in the low mantissa bits
constexpr double kTanhMagic = 6755399441055744.0 / CUDE_TANH_GRID;
#if !defined(__HIP_DEVICE_COMPILE__)
static const double kTanhTableHost[kTanhEntries] = {CUDE_TANH_TABLE_VALUES};
#endif
// Rational part of tanh b on the grid cell: tanh b ~ n / d with n = b (105 + 10 u) / 10, d = (105 + 45 u + u^2) / 10,
// u = b^2.  Written so that no instruction needs two non-inline constants (gfx950's VOP3 reads one SGPR pair per
// instruction: a second constant would be moved into VGPRs first, two v_mov_b32 each).  Returns the table index.
CUDE_HD int m_tanh_cell(double x, double& n, double& d) {
    const double xa = fmin(fabs(x), (double)CUDE_TANH_KMAX / CUDE_TANH_GRID);
    const double t = xa + kTanhMagic;                                 // low mantissa bits = k = round(|x| * grid)
    const double b = xa - (t - kTanhMagic);                           // exact, |b| <= 1/(2 grid)
    const double u = b * b;
    n = b * (u + 10.5);
    d = fma(u, u + 45.0, 105.0) * 0.1;                                // in [10.5, 10.52]
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2loint(t);
#else
    unsigned long long tb;
    __builtin_memcpy(&tb, &t, 8);
    return (int)(unsigned)tb;
#endif
}
// numerator and denominator of tanh|x|;  tab[k] = tanh(k / CUDE_TANH_GRID)
CUDE_HD void m_tanh_nd(double x, const double* tab, double& num, double& den) {
    double n, d;
    const double T = tab[m_tanh_cell(x, n, d)];
    num = fma(T, d, n);
    den = fma(T, n, d);                                               // in [10.5, 11.2]
}
CUDE_HD double m_tanh_tab(double x, const double* tab) {
    double num, den;
    m_tanh_nd(x, tab, num, den);
    return copysign(num * m_rcp(den), x);
}
// A whole layer with ONE reciprocal.  Source order = the order that keeps the fewest values alive: per unit, the table
// read is requested first and the rational part (which does not need it) is formed while it is in flight -- three
// doubles per unit (n, d, T) until all units are through, then two (numerator with the sign, denominator), the same
// footprint as the exponential form's (denominator, prefix product, sign).
template <int W>
CUDE_HD void m_tanh_vec_tab(const double (&z)[W], double (&t)[W], const double* tab) {
    static_assert(W <= 8, "batched reciprocal: product of denominators must stay below 1e308");
    double n[W], d[W], T[W];
    int zh[W];                                       // high words of z: all that the final copysign needs of it
#pragma unroll
    for (int j = 0; j < W; j++) {
        const int k = m_tanh_cell(z[j], n[j], d[j]);
        T[j] = tab[k];
#if defined(__HIP_DEVICE_COMPILE__)
        zh[j] = __double2hiint(z[j]);
#else
        unsigned long long zb;
        __builtin_memcpy(&zb, &z[j], 8);
        zh[j] = (int)(zb >> 32);
#endif
    }
    double pre[W];
#pragma unroll
    for (int j = 0; j < W; j++) {
        const double nj = n[j];
        n[j] = fma(T[j], d[j], nj);                  // numerator of tanh|z|
        d[j] = fma(T[j], nj, d[j]);                  // denominator
        pre[j] = j == 0 ? d[0] : pre[j - 1] * d[j];
    }
    double r = m_rcp(pre[W - 1]);
#pragma unroll
    for (int j = W - 1; j >= 0; j--) {
        double v;
        if (j > 0) {
            v = n[j] * (r * pre[j - 1]);
            r = r * d[j];
        } else {
            v = n[0] * r;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        t[j] = __hiloint2double((__double2hiint(v) & 0x7fffffff) | (zh[j] & (int)0x80000000), __double2loint(v));
#else
        unsigned long long vb;
        __builtin_memcpy(&vb, &v, 8);
        vb = (vb & 0x7fffffffffffffffull) | ((unsigned long long)(unsigned)(zh[j] & (int)0x80000000) << 32);
        __builtin_memcpy(&t[j], &vb, 8);
#endif
    }
}

// t[j] = tanh(z[j]) for a whole layer with ONE reciprocal (prefix-product trick).
// (A lock-step variant that interleaves the W Horner chains was measured 5 % slower: two waves per
// SIMD already cover the FMA latency and the extra live registers cost more than they buy.)
template <int W>
CUDE_HD void m_tanh_vec(const double (&z)[W], double (&t)[W]) {
    static_assert(W <= 8, "batched reciprocal: product of denominators must stay below 1e308");
    double d[W], pre[W];
#pragma unroll
    for (int j = 0; j < W; j++) d[j] = m_tanh_den(z[j]);
    pre[0] = d[0];
#pragma unroll
    for (int j = 1; j < W; j++) pre[j] = pre[j - 1] * d[j];
    double r = m_rcp(pre[W - 1]);
#pragma unroll
    for (int j = W - 1; j >= 1; j--) {
        const double inv = r * pre[j - 1];
        r = r * d[j];
        t[j] = copysign(fma(-2.0, inv, 1.0), z[j]);
    }
    t[0] = copysign(fma(-2.0, r, 1.0), z[0]);
}

// t[j] = tanh(z[j]) given E[j] = exp(2 z[j]) (any sign of z): 1 - 2/(E+1), one shared reciprocal.
// E is clamped to e^40 (tanh = 1 - 8.5e-18 there) so that the product of W <= 8 denominators stays finite.
template <int W>
CUDE_HD void m_tanh_from_exp(const double (&E)[W], double (&t)[W]) {
    static_assert(W <= 8, "batched reciprocal: product of denominators must stay below 1e308");
    double d[W], pre[W];
#pragma unroll
    for (int j = 0; j < W; j++) d[j] = fmin(E[j], 2.35385266837019985408e17) + 1.0;
    pre[0] = d[0];
#pragma unroll
    for (int j = 1; j < W; j++) pre[j] = pre[j - 1] * d[j];
    double r = m_rcp(pre[W - 1]);
#pragma unroll
    for (int j = W - 1; j >= 1; j--) {
        const double inv = r * pre[j - 1];
        r = r * d[j];
        t[j] = fma(-2.0, inv, 1.0);
    }
    t[0] = fma(-2.0, r, 1.0);
}

// softplus value and logistic derivative (LONE: this is the kernel's only exponential, see m_exp2x_t)
template <bool LONE>
CUDE_HD double m_softplus_t(double x, double* sig) {
    const double e = m_exp2x_t<LONE>(fmax(-0.5 * fabs(x), -350.0));     // exp(-|x|) in (0, 1]
    const double d = 1.0 + e;                                    // in (1, 2]
    // log(d) = 2 atanh(s), s = (v-1)/(v+1), v = d or d/2 so that v in [1/sqrt2, sqrt2]
    const bool big = d > 1.41421356237309504880;
    const double num = big ? e - 1.0 : e;
    const double den = big ? e + 3.0 : e + 2.0;
    const double rp = m_rcp(d * den);                            // one reciprocal for 1/d and 1/den
    const double inv_d = rp * den;
    const double s = num * (rp * d);
    *sig = x >= 0.0 ? inv_d : e * inv_d;
    const double z = s * s;                                      // <= 0.02944
#ifdef CUDE_LOG_TAYLOR
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
#else
    // atanh(s)/s in z = s^2: degree-7 interpolant at the Chebyshev nodes of [0, 0.02944] (max relative error 1.2e-18;
    // tools/fit_exp_poly.py), three steps shorter than the series cut at z^10
    double p = 7.40485518032763800900e-02;
    p = m_fma_c(p, z, 7.65626407418209531386e-02);
    p = m_fma_c(p, z, 9.09181584011466425999e-02);
    p = m_fma_c(p, z, 1.11110985283630239739e-01);
    p = m_fma_c(p, z, 1.42857143803208408439e-01);
    p = m_fma_c(p, z, 1.99999999996511690359e-01);
    p = m_fma_c(p, z, 3.33333333333338255322e-01);
    p = fma(p, z, 1.0);
#endif
    const double l = 2.0 * s * p;
    return fmax(x, 0.0) + (big ? l + 0.693147180559945309417 : l);
}

CUDE_HD double m_softplus(double x, double* sig) { return m_softplus_t<false>(x, sig); }

CUDE_HD double m_softplus_val(double x) {
    double sig;
    return m_softplus(x, &sig);
}

}  // namespace cude
