// Branch-free fp64 activations for gfx950.
//
// The path is bound by fp64 VALU issue and ~70 % of the arithmetic is tanh/softplus
// (SURVEY.md 8(d)).  The ROCm device-library tanh/exp/log keep <1 ulp over the whole double
// range with double-double arithmetic (~250 VALU ops per tanh on gfx950); the network only
// needs ABSOLUTE accuracy ~1e-16 on outputs that are O(1), so these versions use
//   exp : Cody-Waite reduction + degree-12 polynomial + v_ldexp_f64            (~18 ops)
//   1/d : v_rcp_f64 + two Newton steps (d is in [1, 2.4e17], no scaling needed)   (5 ops)
//   tanh(x)     = sign(x) * (1 - 2/(exp(2|x|)+1))
//   softplus(x) = max(x,0) + log1p(exp(-|x|)),  sigmoid from the same exponential
// Reference semantics: softplus(x) = log(1+exp(x)) (src/neural-network.jl:13-15); the stable form
// differs from it by rounding only (and does not overflow for x > 709).
// Max abs error vs libm (tests/test_math_host.py): tanh 2.3e-16, softplus 3e-16, sigmoid 2e-16.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CUDE_HD __host__ __device__ __forceinline__
#else
#define CUDE_HD inline
#endif

namespace cude {

CUDE_HD double m_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
#else
    double r = (double)(1.0f / (float)d);   // host model of a low-precision seed
#endif
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
#if !defined(__HIP_DEVICE_COMPILE__)
    e = fma(-d, r, 1.0);                    // float seed needs one more step than v_rcp_f64
    r = fma(r, e, r);
#endif
    return r;
}

// exp(y) for y in [-708, 708]
CUDE_HD double m_exp(double y) {
    const double n = rint(y * 1.4426950408889634074);             // log2(e)
    double r = fma(n, -6.93147180369123816490e-01, y);             // ln2 hi
    r = fma(n, -1.90821492927058770002e-10, r);                    // ln2 lo
    double p = 2.08767569878680989792e-09;                         // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);                     // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);                     // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);                     // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);                     // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);                     // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);                     // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);                     // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);                     // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);                     // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

CUDE_HD double m_tanh(double x) {
    const double y = fmin(fabs(x) * 2.0, 40.0);
    const double r = m_rcp(m_exp(y) + 1.0);
    return copysign(fma(-2.0, r, 1.0), x);
}

// log(u) for u in [1, 2]
CUDE_HD double m_log_1_2(double u) {
    const bool big = u > 1.41421356237309504880;
    const double v = big ? 0.5 * u : u;
    const double s = (v - 1.0) * m_rcp(v + 1.0);
    const double z = s * s;
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
    const double l = 2.0 * s * p;
    return big ? l + 0.693147180559945309417 : l;
}

// softplus value and logistic derivative
CUDE_HD double m_softplus(double x, double* sig) {
    const double e = m_exp(fmax(-fabs(x), -700.0));      // in (0, 1]
    const double d = 1.0 + e;
    const double r = m_rcp(d);
    *sig = x >= 0.0 ? r : e * r;
    return fmax(x, 0.0) + m_log_1_2(d);
}

CUDE_HD double m_softplus_val(double x) {
    const double e = m_exp(fmax(-fabs(x), -700.0));
    return fmax(x, 0.0) + m_log_1_2(1.0 + e);
}

}  // namespace cude
