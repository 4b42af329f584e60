/* CPU oracle (C, forward-mode dual numbers + OpenMP) for the conditional-UDE population
 * ensemble solve + gradient.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/libcude_oracle.so, loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product library
 * (conditional-ude_amd/csrc) never links or calls it.
 *
 * PARITY STATUS: the reference is Julia with no tests and cannot run in the build image.  Suppression path: pinned
 * by the reference's stored lambda = 1 objectives (known answers, tests/test_known_answers.py: this file reproduces
 * them to 1.26e-6, the reference solver's own tolerance); c-peptide path: pinned at FIGURE RESOLUTION (~1e-4) by
 * the trajectories and objectives plotted in the reference's vector figures (tests/test_figure_pins.py, through
 * oracle/cude_oracle.py, with which this file agrees to 1e-12), not at fp64 tolerance; soft-pinned by stored
 * training results.  See oracle/cude_oracle.py for the full statement.
 *
 * This file restates the reference's OWN differentiation method: ForwardDiff dual numbers
 * (AutoForwardDiff, /root/reference/src/parameter-estimation.jl:231,370;
 * suppression/src/suppression_model.jl:123,155) pushed through the as-written model:
 *   softplus log(1+exp(x))                src/neural-network.jl:13-15
 *   SimpleChains TurboDense MLP           src/neural-network.jl:42-58 ([vec_colmajor(W); b] per layer)
 *   van_cauter_parameters                 src/c-peptide-models.jl:30-42
 *   c_peptide_kinetics!                   src/c-peptide-models.jl:7-14
 *   conditional_production                src/c-peptide-models.jl:86-94 (baseline NN([0;e^b]) re-evaluated
 *                                         in every RHS call, exactly as written)
 *   loss single / population              src/parameter-estimation.jl:56-68,126-140
 *   ude_lsup!, suppression_loss           suppression/src/suppression_model.jl:88-95,117-130
 * with the per-subject sparsity the reference lacks (each subject carries P+1 partials:
 * the shared network parameters and its own conditional parameter; the reference carries
 * N+P, i.e. O(N^2) work).  OpenMP static scheduling over subjects stands in for
 * EnsembleThreads (suppression_model.jl:113,123).  Solver: fixed-step Tsit5 + dense output,
 * tableau from SURVEY.md Appendix A (OrdinaryDiffEq is not vendored in the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NPMAX 128
#define MAXT 32
#define MAXW 16

typedef struct { double v; double d[NPMAX]; } dual;

static const double TC[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
static const double TA[7][6] = {
    {0},
    {0.161},
    {-0.008480655492356989, 0.335480655492357},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
static const double TR[7][4] = {
    {1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216},
    {0.0, 0.13169999999999998, -0.2234, 0.1017},
    {0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253},
    {0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902},
    {0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928},
    {0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661},
    {0.0, 1.5, -4.0, 2.5}};

static void interp_weights(double th, double* w) {
    if (fabs(th - 1.0) < 1e-12) { for (int j = 0; j < 6; j++) w[j] = TA[6][j]; w[6] = 0.0; return; }
    for (int i = 0; i < 7; i++)
        w[i] = ((TR[i][3] * th + TR[i][2]) * th + TR[i][1]) * th * th + TR[i][0] * th;
}

static void locate_obs(const double* tp, int T, int S, int* step, double* theta) {
    double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    for (int i = 0; i < T; i++) {
        double x = (tp[i] - t0) / h;
        int n = (int)ceil(x - 1e-9) - 1;
        if (n < 0) n = 0;
        if (n > S - 1) n = S - 1;
        step[i] = n;
        theta[i] = (tp[i] - (t0 + n * h)) / h;
    }
}

/* ---------------------------------------------------------------- dual arithmetic */
static inline void d_copy(dual* r, const dual* x, int np) { r->v = x->v; for (int i = 0; i < np; i++) r->d[i] = x->d[i]; }
static inline void d_const(dual* r, double v, int np) { r->v = v; for (int i = 0; i < np; i++) r->d[i] = 0.0; }
static inline void d_axpy(dual* r, double a, const dual* x, int np) { /* r += a*x */
    r->v += a * x->v; for (int i = 0; i < np; i++) r->d[i] += a * x->d[i]; }
static inline void d_mulparam(dual* r, double w, int widx, const dual* x, int np) { /* r += W*x, W a seeded parameter */
    r->v += w * x->v; for (int i = 0; i < np; i++) r->d[i] += w * x->d[i];
    if (widx >= 0 && widx < np) r->d[widx] += x->v; }
static inline void d_tanh(dual* r, const dual* x, int np) {
    double t = tanh(x->v), g = 1.0 - t * t; r->v = t; for (int i = 0; i < np; i++) r->d[i] = g * x->d[i]; }
static inline void d_softplus(dual* r, const dual* x, int np) {
    double e = exp(x->v), g = e / (1.0 + e); r->v = log(1.0 + e);
    for (int i = 0; i < np; i++) r->d[i] = g * x->d[i]; }

/* MLP on duals.  nn parameters are seeded at partial index = parameter index (if want_grad). */
static void mlp_dual(dual* out, const dual* in, int nin, int width, int depth, const double* nn, int seed, int np) {
    dual h[MAXW], g[MAXW], z;
    int fan = nin, off = 0;
    for (int i = 0; i < nin; i++) d_copy(&h[i], &in[i], np);
    for (int l = 0; l < depth; l++) {
        for (int j = 0; j < width; j++) {
            int bi = off + fan * width + j;
            d_const(&z, nn[bi], np);
            if (seed && bi < np) z.d[bi] = 1.0;
            for (int i = 0; i < fan; i++) {
                int wi = off + j + width * i;
                d_mulparam(&z, nn[wi], seed ? wi : -1, &h[i], np);
            }
            d_tanh(&g[j], &z, np);
        }
        for (int j = 0; j < width; j++) d_copy(&h[j], &g[j], np);
        off += fan * width + width;
        fan = width;
    }
    int bi = off + fan;
    d_const(&z, nn[bi], np);
    if (seed && bi < np) z.d[bi] = 1.0;
    for (int i = 0; i < fan; i++) d_mulparam(&z, nn[off + i], seed ? off + i : -1, &h[i], np);
    d_softplus(out, &z, np);
}

static int n_params(int nin, int width, int depth) {
    int p = 0, fan = nin;
    for (int l = 0; l < depth; l++) { p += width * fan + width; fan = width; }
    return p + fan + 1;
}

typedef struct {
    int model;            /* 0 = CPEP, 1 = SUPP */
    int ns, nin, width, depth, np, seed, P;
    const double* nn;
    /* CPEP per-subject */
    double k0, k1, k2, c0, age; int covariate;
    const double* tp; const double* G; int T;
    dual eb;              /* exp(conditional) */
} rhs_ctx;

static double lin_interp(const double* tk, const double* u, int T, double t) {
    int j = 0;
    while (j + 1 < T && tk[j + 1] <= t) j++;
    if (j > T - 2) j = T - 2;
    double slope = (u[j + 1] - u[j]) / (tk[j + 1] - tk[j]);
    return u[j] + (t - tk[j]) * slope;
}

static void rhs_eval(const rhs_ctx* c, double t, const dual* u, dual* du) {
    int np = c->np;
    if (c->model == 0) {
        double dG = lin_interp(c->tp, c->G, c->T, t) - lin_interp(c->tp, c->G, c->T, c->tp[0]);
        dual in[3], a, b;
        d_const(&in[0], dG, np); d_copy(&in[1], &c->eb, np); d_const(&in[2], c->age, np);
        mlp_dual(&a, in, c->nin, c->width, c->depth, c->nn, c->seed, np);
        d_const(&in[0], 0.0, np);
        mlp_dual(&b, in, c->nin, c->width, c->depth, c->nn, c->seed, np);
        dual prod; d_copy(&prod, &a, np); d_axpy(&prod, -1.0, &b, np);
        d_const(&du[0], c->k0 * c->c0, np);
        d_axpy(&du[0], -(c->k0 + c->k2), &u[0], np);
        d_axpy(&du[0], c->k1, &u[1], np);
        d_axpy(&du[0], 1.0, &prod, np);
        d_const(&du[1], 0.0, np);
        d_axpy(&du[1], -c->k1, &u[1], np);
        d_axpy(&du[1], c->k2, &u[0], np);
        if (c->ns == 3) d_copy(&du[2], &prod, np);
    } else {
        dual in[4], uh;
        d_copy(&in[0], &u[0], np); d_copy(&in[1], &u[1], np); d_copy(&in[2], &u[2], np); d_copy(&in[3], &c->eb, np);
        mlp_dual(&uh, in, 4, c->width, c->depth, c->nn, c->seed, np);
        d_const(&du[0], 0.0, np); d_axpy(&du[0], -0.4, &u[0], np);
        d_const(&du[1], 0.0, np); d_axpy(&du[1], 0.4, &u[0], np); d_axpy(&du[1], -1.0, &uh, np);
        d_copy(&du[2], &uh, np); d_axpy(&du[2], -0.3, &u[2], np);
    }
}

/* Fixed-step Tsit5 with dense output at the observation times. out[T][ns]. */
static void solve_fixed(const rhs_ctx* c, const dual* u0, const double* tp, int T, int S,
                        const int* ostep, const double* otheta, dual* out) {
    int ns = c->ns, np = c->np;
    double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    dual y[3], ynew[3], Y[3], k[7][3];
    for (int s = 0; s < ns; s++) d_copy(&y[s], &u0[s], np);
    rhs_eval(c, t0, y, k[0]);
    for (int n = 0; n < S; n++) {
        double tn = t0 + n * h;
        for (int i = 1; i < 7; i++) {
            for (int s = 0; s < ns; s++) {
                dual acc; d_const(&acc, 0.0, np);
                for (int j = 0; j < i; j++) d_axpy(&acc, TA[i][j], &k[j][s], np);
                d_copy(&Y[s], &y[s], np); d_axpy(&Y[s], h, &acc, np);
            }
            if (i < 6) rhs_eval(c, tn + TC[i] * h, Y, k[i]);
            else { for (int s = 0; s < ns; s++) d_copy(&ynew[s], &Y[s], np); rhs_eval(c, t0 + (n + 1) * h, ynew, k[6]); }
        }
        for (int ti = 0; ti < T; ti++) if (ostep[ti] == n) {
            double w[7]; interp_weights(otheta[ti], w);
            for (int s = 0; s < ns; s++) {
                dual acc; d_const(&acc, 0.0, np);
                for (int j = 0; j < 7; j++) d_axpy(&acc, w[j], &k[j][s], np);
                d_copy(&out[ti * ns + s], &y[s], np); d_axpy(&out[ti * ns + s], h, &acc, np);
            }
        }
        for (int s = 0; s < ns; s++) { d_copy(&y[s], &ynew[s], np); d_copy(&k[0][s], &k[6][s], np); }
    }
}

static void van_cauter(double age, int t2dm, double* k0, double* k1, double* k2) {
    double sh = t2dm ? 4.52 : 4.95, fr = t2dm ? 0.78 : 0.76, lo = 0.14 * age + 29.2;
    *k1 = fr * (log(2.0) / lo) + (1 - fr) * (log(2.0) / sh);
    *k0 = (log(2.0) / sh) * (log(2.0) / lo) / *k1;
    *k2 = (log(2.0) / sh) + (log(2.0) / lo) - *k0 - *k1;
}

int cude_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Population loss (mean SSE on state 1) and, if want_grad, its gradient by forward duals.
 * glucose/cpeptide are N x T row-major.  traj (optional) is N x T x n_state.
 * Returns the number of subjects with a non-finite SSE (loss is +Inf then, as the reference). */
int cude_oracle_cpep(int N, int T, const double* tp, const double* glucose, const double* cpeptide,
                     const double* age, const uint8_t* t2dm, int covariate,
                     int nin, int width, int depth, const double* nn, const double* beta,
                     int n_steps, int n_state, int want_grad, int nthreads,
                     double* loss, double* sse, double* g_nn, double* g_beta, double* traj) {
    int P = n_params(nin, width, depth);
    int np = want_grad ? P + 1 : 0;
    if (np > NPMAX || T > MAXT || width > MAXW) return -1;
    int ostep[MAXT]; double oth[MAXT];
    locate_obs(tp, T, n_steps, ostep, oth);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    double* gacc = (double*)calloc((size_t)nthreads * (P + 2), sizeof(double));
    int nfail = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : nfail)
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double* ga = gacc + (size_t)tid * (P + 2);
        dual* out = (dual*)malloc(sizeof(dual) * T * 3);
#pragma omp for schedule(static)
        for (int i = 0; i < N; i++) {
            rhs_ctx c; memset(&c, 0, sizeof(c));
            c.model = 0; c.ns = n_state; c.nin = nin; c.width = width; c.depth = depth;
            c.np = np; c.seed = want_grad; c.P = P; c.nn = nn;
            van_cauter(age[i], t2dm[i], &c.k0, &c.k1, &c.k2);
            c.c0 = cpeptide[(size_t)i * T]; c.age = age[i]; c.covariate = covariate;
            c.tp = tp; c.G = glucose + (size_t)i * T; c.T = T;
            d_const(&c.eb, exp(beta[i]), np);
            if (want_grad) c.eb.d[P] = c.eb.v;          /* d e^b / d b */
            dual u0[3];
            d_const(&u0[0], c.c0, np); d_const(&u0[1], (c.k2 / c.k1) * c.c0, np); d_const(&u0[2], 0.0, np);
            solve_fixed(&c, u0, tp, T, n_steps, ostep, oth, out);
            double s = 0.0;
            double gd[NPMAX];
            for (int q = 0; q < np; q++) gd[q] = 0.0;
            for (int ti = 0; ti < T; ti++) {
                const dual* o = &out[ti * n_state];
                double r = o->v - cpeptide[(size_t)i * T + ti];
                s += r * r;
                for (int q = 0; q < np; q++) gd[q] += 2.0 * r * o->d[q];
                if (traj) for (int st = 0; st < n_state; st++) traj[((size_t)i * T + ti) * n_state + st] = out[ti * n_state + st].v;
            }
            if (sse) sse[i] = s;
            if (!isfinite(s)) nfail += 1;
            ga[P] += s;
            if (want_grad) {
                for (int q = 0; q < P; q++) ga[q] += gd[q];
                g_beta[i] = gd[P] / N;
            }
        }
        free(out);
    }
    double tot = 0.0;
    if (want_grad) for (int q = 0; q < P; q++) g_nn[q] = 0.0;
    for (int t = 0; t < nthreads; t++) {
        tot += gacc[(size_t)t * (P + 2) + P];
        if (want_grad) for (int q = 0; q < P; q++) g_nn[q] += gacc[(size_t)t * (P + 2) + q];
    }
    if (want_grad) for (int q = 0; q < P; q++) g_nn[q] /= N;
    free(gacc);
    *loss = nfail ? INFINITY : tot / N;
    return nfail;
}

/* suppression_loss.  data is 3 x T x N column-major (Julia layout): data[s + 3*(t + T*i)].
 * loss = sum(((sim - data)/scale)^2)/N + lambda*sum(nn^2); scale[s] = mean_i max_t data[s,t,i]. */
int cude_oracle_supp(int N, int T, const double* tp, const double* data,
                     int width, int depth, const double* nn, const double* theta, double lambda,
                     int n_steps, int want_grad, int nthreads,
                     double* loss, double* sse, double* g_nn, double* g_theta, double* traj) {
    int P = n_params(4, width, depth);
    int np = want_grad ? P + 1 : 0;
    if (np > NPMAX || T > MAXT || width > MAXW) return -1;
    int ostep[MAXT]; double oth[MAXT];
    locate_obs(tp, T, n_steps, ostep, oth);
    double scale[3] = {0, 0, 0};
    for (int i = 0; i < N; i++) for (int s = 0; s < 3; s++) {
        double m = -INFINITY;
        for (int t = 0; t < T; t++) { double v = data[s + 3 * (t + (size_t)T * i)]; if (v > m) m = v; }
        scale[s] += m;
    }
    for (int s = 0; s < 3; s++) scale[s] /= N;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    double* gacc = (double*)calloc((size_t)nthreads * (P + 2), sizeof(double));
    int nfail = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : nfail)
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double* ga = gacc + (size_t)tid * (P + 2);
        dual* out = (dual*)malloc(sizeof(dual) * T * 3);
#pragma omp for schedule(static)
        for (int i = 0; i < N; i++) {
            rhs_ctx c; memset(&c, 0, sizeof(c));
            c.model = 1; c.ns = 3; c.nin = 4; c.width = width; c.depth = depth;
            c.np = np; c.seed = want_grad; c.P = P; c.nn = nn;
            d_const(&c.eb, exp(theta[i]), np);
            if (want_grad) c.eb.d[P] = c.eb.v;
            dual u0[3];
            for (int s = 0; s < 3; s++) d_const(&u0[s], data[s + 3 * ((size_t)T * i)], np);
            solve_fixed(&c, u0, tp, T, n_steps, ostep, oth, out);
            double s2 = 0.0, gd[NPMAX];
            for (int q = 0; q < np; q++) gd[q] = 0.0;
            for (int ti = 0; ti < T; ti++) for (int s = 0; s < 3; s++) {
                const dual* o = &out[ti * 3 + s];
                double r = (o->v - data[s + 3 * (ti + (size_t)T * i)]) / scale[s];
                s2 += r * r;
                for (int q = 0; q < np; q++) gd[q] += 2.0 * r * o->d[q] / scale[s];
                if (traj) traj[s + 3 * (ti + (size_t)T * i)] = o->v;
            }
            if (sse) sse[i] = s2;
            if (!isfinite(s2)) nfail += 1;
            ga[P] += s2;
            if (want_grad) {
                for (int q = 0; q < P; q++) ga[q] += gd[q];
                g_theta[i] = gd[P] / N;
            }
        }
        free(out);
    }
    double tot = 0.0, reg = 0.0;
    if (want_grad) for (int q = 0; q < P; q++) g_nn[q] = 0.0;
    for (int t = 0; t < nthreads; t++) {
        tot += gacc[(size_t)t * (P + 2) + P];
        if (want_grad) for (int q = 0; q < P; q++) g_nn[q] += gacc[(size_t)t * (P + 2) + q];
    }
    for (int q = 0; q < P; q++) reg += nn[q] * nn[q];
    if (want_grad) for (int q = 0; q < P; q++) g_nn[q] = g_nn[q] / N + 2.0 * lambda * nn[q];
    free(gacc);
    *loss = nfail ? INFINITY : tot / N + lambda * reg;
    return nfail;
}

/* ---------------------------------------------------------------- adaptive mode (plain doubles)
 * cude_oracle.solve_adaptive + cpep_rhs_scalar restated operation for operation in C, for N independent subjects at
 * once: OrdinaryDiffEq's Tsit5 with its default controller (PI, beta1 = 7/50, beta2 = 2/25, gamma = 0.9, qmin = 0.2,
 * qmax = 10), Hairer's initial-step heuristic and `saveat` through the free interpolant.  It exists so that the tests
 * which compare with what the reference RAN (tests/test_figure_pins.py) can scan a parameter finely; the Python
 * function stays the restatement that the known answers pin, and tests/test_oracle.py holds the two together.
 * cond[i] = exp(beta_i), or k_i for the symbolic production (width == 0: nn[0] dG / (dG + k) for dG >= 0, else 0).
 * out is N x n_out (state 1 at out_times; integration runs from out_times[0] to out_times[n_out-1]); a failed
 * subject's row is NaN.  Returns the number of failed subjects. */
/* No multiply-add contraction in this section, so that it rounds like the Python statement of the same formulas. */
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")

static void interp_weights_nc(double th, double* w) {
    if (fabs(th - 1.0) < 1e-12) { for (int j = 0; j < 6; j++) w[j] = TA[6][j]; w[6] = 0.0; return; }
    for (int i = 0; i < 7; i++)
        w[i] = ((TR[i][3] * th + TR[i][2]) * th + TR[i][1]) * th * th + TR[i][0] * th;
}

static double lin_interp_nc(const double* tk, const double* u, int T, double t) {
    int j = 0;
    while (j + 1 < T && tk[j + 1] <= t) j++;
    if (j > T - 2) j = T - 2;
    double slope = (u[j + 1] - u[j]) / (tk[j + 1] - tk[j]);
    return u[j] + (t - tk[j]) * slope;
}

static const double TBT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                              -0.1447110071732629, 0.5823571654525552, -0.45808210592918697, 0.015151515151515152};

static double mlp_plain(const double* in, int nin, int width, int depth, const double* nn) {
    double h[MAXW], nx[MAXW];
    int fan = nin, off = 0;
    for (int i = 0; i < nin; i++) h[i] = in[i];
    for (int l = 0; l < depth; l++) {
        for (int j = 0; j < width; j++) {
            double z = nn[off + fan * width + j];
            for (int i = 0; i < fan; i++) z = z + nn[off + j + width * i] * h[i];
            nx[j] = tanh(z);
        }
        off += fan * width + width;
        fan = width;
        for (int j = 0; j < width; j++) h[j] = nx[j];
    }
    double z = nn[off + fan];
    for (int i = 0; i < fan; i++) z = z + nn[off + i] * h[i];
    return log(1.0 + exp(z));
}

typedef struct {
    int nin, width, depth, covariate, T;
    const double *nn, *tp, *G;
    double k0, k1, k2, c0, age, cond;
} plain_ctx;

static void rhs_plain(const plain_ctx* c, double t, const double* u, double* du) {
    double dG = lin_interp_nc(c->tp, c->G, c->T, t) - c->G[0], prod;
    if (c->width == 0) {
        prod = dG >= 0 ? (c->nn[0] * dG) / (dG + c->cond) : 0.0;
    } else {
        double in[3] = {dG, c->cond, c->age}, in0[3] = {0.0, c->cond, c->age};
        prod = mlp_plain(in, c->nin, c->width, c->depth, c->nn) - mlp_plain(in0, c->nin, c->width, c->depth, c->nn);
    }
    du[0] = -(c->k0 + c->k2) * u[0] + c->k1 * u[1] + c->k0 * c->c0 + prod;
    du[1] = -c->k1 * u[1] + c->k2 * u[0];
}

static double rms2(double a, double b) { return sqrt((a * a + b * b) / 2); }

static int solve_adaptive_plain(const plain_ctx* c, const double* u0, int n_out, const double* tout,
                                double abstol, double reltol, double* out) {
    const double beta1 = 7.0 / 50, beta2 = 2.0 / 25, gamma = 0.9, qmin = 0.2, qmax = 10.0;
    double t0 = tout[0], t1 = tout[n_out - 1];
    double y[2] = {u0[0], u0[1]}, ynew[2], Y[2], k[7][2], f1[2], y1[2];
    int nxt = 1;
    out[0] = y[0];
    rhs_plain(c, t0, y, k[0]);
    double sk0 = abstol + reltol * fabs(y[0]), sk1 = abstol + reltol * fabs(y[1]);
    double d0 = rms2(y[0] / sk0, y[1] / sk1), d1 = rms2(k[0][0] / sk0, k[0][1] / sk1);
    double dt = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    for (int s = 0; s < 2; s++) y1[s] = y[s] + dt * k[0][s];
    rhs_plain(c, t0 + dt, y1, f1);
    double d2 = rms2((f1[0] - k[0][0]) / sk0, (f1[1] - k[0][1]) / sk1) / dt;
    double dm = d1 > d2 ? d1 : d2;
    double dt1 = dm <= 1e-15 ? fmax(1e-6, dt * 1e-3) : pow(0.01 / dm, 1.0 / 5);
    dt = fmin(fmin(100 * dt, dt1), t1 - t0);
    double t = t0, qold = 1e-4;
    for (int it = 0; it < 100000; it++) {
        if (t >= t1 - 1e-14 * fmax(1.0, fabs(t1))) break;
        dt = fmin(dt, t1 - t);
        for (int i = 1; i < 7; i++) {
            for (int s = 0; s < 2; s++) {
                double acc = 0.0;
                for (int j = 0; j < i; j++) acc = acc + TA[i][j] * k[j][s];
                Y[s] = y[s] + dt * acc;
            }
            if (i < 6) rhs_plain(c, t + TC[i] * dt, Y, k[i]);
            else { ynew[0] = Y[0]; ynew[1] = Y[1]; rhs_plain(c, t + dt, ynew, k[6]); }
        }
        double e[2];
        for (int s = 0; s < 2; s++) {
            double acc = 0.0;
            for (int j = 0; j < 7; j++) acc = acc + TBT[j] * k[j][s];
            e[s] = dt * acc / (abstol + reltol * fmax(fabs(y[s]), fabs(ynew[s])));
        }
        double est = rms2(e[0], e[1]);
        if (!isfinite(est)) return 1;
        double q11 = est > 0 ? pow(est, beta1) : 1e-12;
        if (est <= 1.0) {
            while (nxt < n_out && tout[nxt] <= t + dt + 1e-12) {
                double th = fmin(1.0, (tout[nxt] - t) / dt), w[7], acc = 0.0;
                interp_weights_nc(th, w);
                for (int j = 0; j < 7; j++) acc = acc + w[j] * k[j][0];
                out[nxt++] = y[0] + dt * acc;
            }
            double q = q11 / pow(qold, beta2);
            q = fmax(1 / qmax, fmin(1 / qmin, q / gamma));
            t = t + dt; y[0] = ynew[0]; y[1] = ynew[1]; k[0][0] = k[6][0]; k[0][1] = k[6][1];
            qold = fmax(est, 1e-4);
            dt = dt / q;
        } else {
            dt = dt / fmin(1 / qmin, q11 / gamma);
        }
    }
    return nxt < n_out;
}

int cude_oracle_cpep_adaptive(int N, int T, const double* tp, const double* glucose, const double* cpeptide,
                              const double* age, const uint8_t* t2dm, int covariate,
                              int nin, int width, int depth, const double* nn, const double* cond,
                              int n_out, const double* out_times, double abstol, double reltol, int nthreads,
                              double* out) {
    if (T > MAXT || width > MAXW || n_out < 2) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    int nfail = 0;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads) reduction(+ : nfail)
    for (int i = 0; i < N; i++) {
        plain_ctx c;
        c.nin = nin; c.width = width; c.depth = depth; c.covariate = covariate; c.T = T;
        c.nn = nn; c.tp = tp; c.G = glucose + (size_t)i * T;
        van_cauter(age[i], t2dm[i], &c.k0, &c.k1, &c.k2);
        c.c0 = cpeptide[(size_t)i * T]; c.age = age[i]; c.cond = cond[i];
        double u0[2] = {c.c0, (c.k2 / c.k1) * c.c0};
        if (solve_adaptive_plain(&c, u0, n_out, out_times, abstol, reltol, out + (size_t)i * n_out)) {
            for (int j = 0; j < n_out; j++) out[(size_t)i * n_out + j] = NAN;
            nfail += 1;
        }
    }
    return nfail;
}
/* The same solver on the suppression model's three states (cude_oracle.solve_adaptive + supp_rhs, suppression/src/
 * suppression_model.jl:88-95,107-115): N independent subjects, u0 = data[:, 1, i], outputs at the T observation times.
 * data is Julia's column-major 3 x T x N; etheta[i] = exp(theta_i); out is N x T x 3 (a failed subject's rows are NaN).
 * It exists so that the reference's stored objectives of whole result directories (hundreds of networks:
 * tests/test_known_answers_runs.py) can be checked in seconds; tests/test_oracle.py holds it to the Python statement. */
typedef struct { int width, depth; const double* nn; double eth; } supp_ctx;

static void rhs_supp_plain(const supp_ctx* c, const double* u, double* du) {
    double in[4] = {u[0], u[1], u[2], c->eth};
    double uh = mlp_plain(in, 4, c->width, c->depth, c->nn);
    du[0] = -0.4 * u[0];
    du[1] = 0.4 * u[0] - uh;
    du[2] = uh - 0.3 * u[2];
}

static double rms3(const double* v) { return sqrt((v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) / 3); }

static int solve_adaptive_supp(const supp_ctx* c, const double* u0, int n_out, const double* tout,
                               double abstol, double reltol, double* out) {
    const double beta1 = 7.0 / 50, beta2 = 2.0 / 25, gamma = 0.9, qmin = 0.2, qmax = 10.0;
    double t0 = tout[0], t1 = tout[n_out - 1];
    double y[3], ynew[3], Y[3], k[7][3], f1[3], y1[3], sk[3], v[3];
    for (int s = 0; s < 3; s++) { y[s] = u0[s]; out[s] = y[s]; }
    int nxt = 1;
    rhs_supp_plain(c, y, k[0]);
    for (int s = 0; s < 3; s++) sk[s] = abstol + reltol * fabs(y[s]);
    for (int s = 0; s < 3; s++) v[s] = y[s] / sk[s];
    double d0 = rms3(v);
    for (int s = 0; s < 3; s++) v[s] = k[0][s] / sk[s];
    double d1 = rms3(v);
    double dt = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    for (int s = 0; s < 3; s++) y1[s] = y[s] + dt * k[0][s];
    rhs_supp_plain(c, y1, f1);
    for (int s = 0; s < 3; s++) v[s] = (f1[s] - k[0][s]) / sk[s];
    double d2 = rms3(v) / dt;
    double dm = d1 > d2 ? d1 : d2;
    double dt1 = dm <= 1e-15 ? fmax(1e-6, dt * 1e-3) : pow(0.01 / dm, 1.0 / 5);
    dt = fmin(fmin(100 * dt, dt1), t1 - t0);
    double t = t0, qold = 1e-4;
    for (int it = 0; it < 100000; it++) {
        if (t >= t1 - 1e-14 * fmax(1.0, fabs(t1))) break;
        dt = fmin(dt, t1 - t);
        for (int i = 1; i < 7; i++) {
            for (int s = 0; s < 3; s++) {
                double acc = 0.0;
                for (int j = 0; j < i; j++) acc = acc + TA[i][j] * k[j][s];
                Y[s] = y[s] + dt * acc;
            }
            if (i < 6) rhs_supp_plain(c, Y, k[i]);
            else { for (int s = 0; s < 3; s++) ynew[s] = Y[s]; rhs_supp_plain(c, ynew, k[6]); }
        }
        double e[3];
        for (int s = 0; s < 3; s++) {
            double acc = 0.0;
            for (int j = 0; j < 7; j++) acc = acc + TBT[j] * k[j][s];
            e[s] = dt * acc / (abstol + reltol * fmax(fabs(y[s]), fabs(ynew[s])));
        }
        double est = rms3(e);
        if (!isfinite(est)) return 1;
        double q11 = est > 0 ? pow(est, beta1) : 1e-12;
        if (est <= 1.0) {
            while (nxt < n_out && tout[nxt] <= t + dt + 1e-12) {
                double th = fmin(1.0, (tout[nxt] - t) / dt), w[7];
                interp_weights_nc(th, w);
                for (int s = 0; s < 3; s++) {
                    double acc = 0.0;
                    for (int j = 0; j < 7; j++) acc = acc + w[j] * k[j][s];
                    out[nxt * 3 + s] = y[s] + dt * acc;
                }
                nxt++;
            }
            double q = q11 / pow(qold, beta2);
            q = fmax(1 / qmax, fmin(1 / qmin, q / gamma));
            t = t + dt;
            for (int s = 0; s < 3; s++) { y[s] = ynew[s]; k[0][s] = k[6][s]; }
            qold = fmax(est, 1e-4);
            dt = dt / q;
        } else {
            dt = dt / fmin(1 / qmin, q11 / gamma);
        }
    }
    return nxt < n_out;
}

int cude_oracle_supp_adaptive(int N, int T, const double* tp, const double* data, int width, int depth, const double* nn,
                              const double* etheta, double abstol, double reltol, int nthreads, double* out) {
    if (T > MAXT || T < 2 || width > MAXW) return -1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    int nfail = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : nfail)
    for (int i = 0; i < N; i++) {
        supp_ctx c;
        c.width = width; c.depth = depth; c.nn = nn; c.eth = etheta[i];
        const double* u0 = data + (size_t)i * T * 3;            /* data[:, 1, i] */
        if (solve_adaptive_supp(&c, u0, T, tp, abstol, reltol, out + (size_t)i * T * 3)) {
            for (int j = 0; j < T * 3; j++) out[(size_t)i * T * 3 + j] = NAN;
            nfail += 1;
        }
    }
    return nfail;
}
#pragma GCC pop_options

