"""Near-minimax polynomial for g(s) = exp(2 s) on |s| <= ln2/4 (Chebyshev interpolation in 60-digit
arithmetic, rounded to double), used by cude_math.h m_exp2x.  Prints coefficients and the achieved max
relative error of the double-precision Horner/FMA evaluation."""
import mpmath as mp
import numpy as np
mp.mp.dps = 60
a = mp.log(2) / 4

def cheb_fit(deg):
    n = deg + 1
    xs = [a * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    ys = [mp.exp(2 * x) for x in xs]
    # solve Vandermonde in high precision
    V = mp.matrix(n, n)
    for i, x in enumerate(xs):
        for j in range(n):
            V[i, j] = x ** j
    c = mp.lu_solve(V, mp.matrix(ys))
    return [c[j] for j in range(n)]

for deg in (9, 10, 11, 12):
    c = cheb_fit(deg)
    # constrain c0 = 1, c1 = 2 exactly? keep fitted but report
    cd = [float(v) for v in c]
    s = np.linspace(-float(a), float(a), 200001)
    # Horner in float64 (numpy has no fma; emulate with longdouble to approximate fma rounding)
    p = np.full_like(s, cd[-1], dtype=np.longdouble)
    for k in range(deg - 1, -1, -1):
        p = (p * s.astype(np.longdouble) + np.longdouble(cd[k])).astype(np.float64).astype(np.longdouble)
    ref = np.array([mp.exp(2 * mp.mpf(float(v))) for v in s[::200]], dtype=object)
    err = max(abs(mp.mpf(float(pv)) / r - 1) for pv, r in zip(p[::200], ref))
    print(deg, "max rel err", mp.nstr(err, 4))
    if deg in (10, 11):
        print("  coeffs:", ", ".join(f"{v:.20e}" for v in cd))


# ---- atanh(s)/s as a polynomial in z = s^2 on [0, ((sqrt2 - 1)/(sqrt2 + 1))^2] (m_softplus_t: log d = 2 atanh(s)):
# interpolant at the Chebyshev nodes of the interval; degree 7 reaches 1.2e-18 (the series needs z^10 for 6e-19)
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
zmax = smax ** 2


def atanh_over_s(z):
    if z == 0:
        return mp.mpf(1)
    s = mp.sqrt(z)
    return mp.atanh(s) / s


def cheb_fit_atanh(deg):
    n = deg + 1
    xs = [zmax / 2 * (1 + mp.cos(mp.pi * (2 * k + 1) / (2 * n))) for k in range(n)]
    V = mp.matrix(n, n)
    for i, x in enumerate(xs):
        for j in range(n):
            V[i, j] = x ** j
    c = mp.lu_solve(V, mp.matrix([atanh_over_s(x) for x in xs]))
    return [c[j] for j in range(n)]


print("atanh(s)/s in z = s^2 on [0, %.6f]:" % float(zmax))
for deg in (6, 7, 8):
    c = cheb_fit_atanh(deg)
    zs = [zmax * mp.mpf(k) / 2000 for k in range(2001)]
    err = max(abs(sum(c[j] * z ** j for j in range(deg + 1)) / atanh_over_s(z) - 1) for z in zs)
    print(deg, "max rel err", mp.nstr(err, 4))
    if deg == 7:
        print("  coeffs:", ", ".join(f"{float(v):.20e}" for v in c))
