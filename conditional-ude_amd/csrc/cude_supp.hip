// Suppression conditional-UDE ensemble kernels for gfx950 (3-state nonlinear ODE whose NN input
// is the state): fixed-step Tsit5 forward + discrete adjoint with step-state checkpoints.
//
// Replaces (reference repo paths):
//   ude_lsup!                      suppression/src/suppression_model.jl:88-95
//   get_prob_func / simul          suppression/src/suppression_model.jl:97-115 (EnsembleThreads)
//   suppression_loss (+ its ForwardDiffSensitivity gradient)   :117-130, :155
//
// One lane = one subject.  The forward sweep checkpoints y_n (3 doubles per step) to an HBM
// scratch laid out [step][state][subject] (coalesced); the reverse sweep reloads y_n,
// recomputes the stage states and applies the stage VJPs in reverse order.
#include "cude_device.h"
#include "cude_kernels.h"

namespace cude {

template <int W, int D>
struct SuppRhs {
    using Net = Mlp<4, W, D, 3>;
    __device__ static __forceinline__ void f(cptr_t p, const double (&c)[W], const double (&u)[3], double (&du)[3]) {
        const double uh = Net::eval(p, c, u);
        du[0] = -0.4 * u[0];
        du[1] = fma(0.4, u[0], -uh);
        du[2] = fma(-0.3, u[2], uh);
    }
    // ub += J_f(u)^T kb ; acc += (df/dparams)^T kb
    __device__ static __forceinline__ void vjp(cptr_t p, const double (&c)[W], const double (&u)[3],
                                               const double (&kb)[3], double (&ub)[3], double (&acc)[Net::NACC]) {
        const double wgt = kb[2] - kb[1];
        double dx[3] = {0.0, 0.0, 0.0};
        Net::template eval_grad<true>(p, c, u, wgt, acc, dx);
        ub[0] += fma(-0.4, kb[0], fma(0.4, kb[1], dx[0]));
        ub[1] += dx[1];
        ub[2] += fma(-0.3, kb[2], dx[2]);
    }
};

template <int W, int D, bool GRAD>
__global__ __launch_bounds__(kBlock) void supp_kernel(SuppArgs a) {
    using R = SuppRhs<W, D>;
    using Net = typename R::Net;
    constexpr int P = Net::P;
    extern __shared__ double s_res[];   // [T][3][kBlock]

    const int lane = threadIdx.x;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < a.N;
    const int64_t i = active ? gid : a.N - 1;
    const int64_t N = a.N;
    cptr_t p = as_const(a.nn);
    cptr_t obs_w = as_const(a.obs_w);
    ciptr_t obs_step = as_const(a.obs_step);
    const int S = a.S, T = a.T;
    const double h = a.h;

    double cst[1] = {exp(a.cond[i])};
    double c[W];
    Net::first_layer_offset(p, cst, c);

    double y[3];
#pragma unroll
    for (int s = 0; s < 3; s++) y[s] = a.data[((int64_t)s * T + 0) * N + i];
    const double y0[3] = {y[0], y[1], y[2]};

    double K[7][3];
    R::f(p, c, y, K[0]);
    double sse = fma(cst[0], 0.0, Net::param_check(p));   // NaN iff a parameter / theta is non-finite
    int oi = 0;
    for (int n = 0; n < S; n++) {
        if (GRAD) {
#pragma unroll
            for (int s = 0; s < 3; s++) a.ckpt[((int64_t)n * 3 + s) * N + i] = y[s];
        }
        double Y[3];
#pragma unroll
        for (int st = 1; st < 7; st++) {
#pragma unroll
            for (int s = 0; s < 3; s++) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < st; j++) acc = fma(Tab::a(st, j), K[j][s], acc);
                Y[s] = fma(h, acc, y[s]);
            }
            R::f(p, c, Y, K[st]);
        }
        while (oi < T && obs_step[oi] == n) {
#pragma unroll
            for (int s = 0; s < 3; s++) {
                double o = 0.0;
#pragma unroll
                for (int j = 0; j < 7; j++) o = fma(obs_w[oi * 7 + j], K[j][s], o);
                o = fma(h, o, y[s]);
                const double r = o - a.data[((int64_t)s * T + oi) * N + i];
                sse = fma(r * a.iscale2[s], r, sse);
                if (GRAD) s_res[(oi * 3 + s) * kBlock + lane] = r;
                if (a.traj != nullptr && active) a.traj[s + 3 * (oi + (int64_t)T * i)] = o;
            }
            oi++;
        }
#pragma unroll
        for (int s = 0; s < 3; s++) { y[s] = Y[s]; K[0][s] = K[6][s]; }
    }
    const bool failed = !(fabs(sse) <= 1.79769313486231570815e308);
    if (active && a.sse != nullptr) a.sse[i] = sse;
    double red_loss = active ? sse : 0.0;
    double red_fail = (active && failed) ? 1.0 : 0.0;
    double* out = a.partials + (int64_t)blockIdx.x * (P + 2);

    if (!GRAD) {
        red_loss = wave_sum(red_loss);
        red_fail = wave_sum(red_fail);
        if (lane == 0) { out[P] = red_loss; out[P + 1] = red_fail; }
        return;
    } else {
        double acc[Net::NACC];
#pragma unroll
        for (int q = 0; q < Net::NACC; q++) acc[q] = 0.0;
        double lam[3] = {0.0, 0.0, 0.0};   // adjoint of y_{n+1}
        double kap[3] = {0.0, 0.0, 0.0};   // adjoint of k_7 of step n from step n+1's use as k_1
        const double gs = 2.0 * a.inv_n;
        oi = T - 1;
        for (int n = S - 1; n >= 0; n--) {
            double yn[3];
#pragma unroll
            for (int s = 0; s < 3; s++) yn[s] = a.ckpt[((int64_t)n * 3 + s) * N + i];
            // recompute the stage states of this step
            double Ys[6][3];                 // Ys[st-1] = input of stage st+1 (st = 1..6; last = y_{n+1})
            {
                double Kf[6][3];
                R::f(p, c, yn, Kf[0]);
#pragma unroll
                for (int st = 1; st < 7; st++) {
#pragma unroll
                    for (int s = 0; s < 3; s++) {
                        double t = 0.0;
#pragma unroll
                        for (int j = 0; j < st; j++) t = fma(Tab::a(st, j), Kf[j][s], t);
                        Ys[st - 1][s] = fma(h, t, yn[s]);
                    }
                    if (st < 6) R::f(p, c, Ys[st - 1], Kf[st]);
                }
            }
            double kb[7][3];
#pragma unroll
            for (int j = 0; j < 6; j++) { kb[j][0] = 0.0; kb[j][1] = 0.0; kb[j][2] = 0.0; }
#pragma unroll
            for (int s = 0; s < 3; s++) kb[6][s] = kap[s];
            double yb[3] = {0.0, 0.0, 0.0};
            while (oi >= 0 && obs_step[oi] == n) {
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    const double g = gs * a.iscale2[s] * s_res[(oi * 3 + s) * kBlock + lane];
                    yb[s] += g;
                    const double hg = h * g;
#pragma unroll
                    for (int j = 0; j < 7; j++) kb[j][s] = fma(obs_w[oi * 7 + j], hg, kb[j][s]);
                }
                oi--;
            }
            // stage 7 at y_{n+1}
            R::vjp(p, c, Ys[5], kb[6], lam, acc);
#pragma unroll
            for (int s = 0; s < 3; s++) {
                yb[s] += lam[s];
                const double hl = h * lam[s];
#pragma unroll
                for (int j = 0; j < 6; j++) kb[j][s] = fma(Tab::a(6, j), hl, kb[j][s]);
            }
#pragma unroll
            for (int st = 5; st >= 1; st--) {
                double Yb[3] = {0.0, 0.0, 0.0};
                R::vjp(p, c, Ys[st - 1], kb[st], Yb, acc);
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    yb[s] += Yb[s];
                    const double hy = h * Yb[s];
#pragma unroll
                    for (int j = 0; j < st; j++) kb[j][s] = fma(Tab::a(st, j), hy, kb[j][s]);
                }
            }
#pragma unroll
            for (int s = 0; s < 3; s++) { lam[s] = yb[s]; kap[s] = kb[0][s]; }
        }
        // k_1 of step 0 = f(y_0): y_0 is data, but the network parameters and exp(theta) enter
        {
            double ub[3] = {0.0, 0.0, 0.0};
            R::vjp(p, c, y0, kap, ub, acc);
        }
        double g[P];
        double dcond;
        Net::expand(p, acc, cst, g, &dcond);
        if (active) a.g_cond[i] = dcond;
        const double keep = active ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < P; q++) {
            const double v = wave_sum(g[q] * keep);
            if (lane == 0) out[q] = v;
        }
        red_loss = wave_sum(red_loss);
        red_fail = wave_sum(red_fail);
        if (lane == 0) { out[P] = red_loss; out[P + 1] = red_fail; }
    }
}

template <int W, int D, bool GRAD>
static hipError_t launch_one(const SuppArgs& a, hipStream_t s) {
    const int64_t nblocks = (a.N + kBlock - 1) / kBlock;
    const size_t lds = GRAD ? sizeof(double) * (size_t)a.T * 3 * kBlock : 0;
    hipLaunchKernelGGL((supp_kernel<W, D, GRAD>), dim3((unsigned)nblocks), dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

#define CUDE_SUPP_SHAPES(X) X(3, 5) X(3, 2) X(4, 2) X(6, 2)

bool supp_shape_supported(const NetShape& net) {
    if (net.nin != 4) return false;
#define X(W, D) if (net.width == W && net.depth == D) return true;
    CUDE_SUPP_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_supp(const NetShape& net, bool grad, const SuppArgs& a, hipStream_t s) {
    if (net.nin != 4) return hipErrorInvalidValue;
#define X(W, D) if (net.width == W && net.depth == D) return grad ? launch_one<W, D, true>(a, s) : launch_one<W, D, false>(a, s);
    CUDE_SUPP_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cude
