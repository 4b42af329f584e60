/* CPU oracle, third algorithm: per-subject REVERSE-MODE gradient (hand-written discrete adjoint) + OpenMP.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as cude_oracle.c: built into oracle/libcude_oracle.so, loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product library never links or calls it).
 * PARITY STATUS: as cude_oracle.c (this file is held to it at 1e-12 by tests/test_oracle.py).
 *
 * What it is for: SURVEY.md 8(d) / BASELINE.md 3 specify the CPU baseline as "the identical discretisation, per-subject
 * reverse-mode gradient, schedule(static) over subjects" -- the cheapest honest CPU way to get the gradient the
 * reference gets from ForwardDiff (src/parameter-estimation.jl:370, suppression_model.jl:155).  cude_oracle.c restates
 * the reference's own method (forward duals, P+1 partials per subject: ~P times the work); this file is the
 * efficient CPU formulation and is what bench.py reports as cpu_baseline.
 *
 * Same model statements as cude_oracle.c, written independently of it and of the HIP kernels:
 *   softplus / SimpleChains MLP            src/neural-network.jl:13-15,42-58
 *   van_cauter_parameters, kinetics        src/c-peptide-models.jl:7-14,30-42
 *   conditional_production (baseline NN([0;e^b]) evaluated in every RHS call, as written)   :86-104
 *   loss single / population               src/parameter-estimation.jl:56-68,126-140
 *   ude_lsup!, suppression_loss            suppression/src/suppression_model.jl:88-95,117-130
 * The adjoint is the generic one of an explicit Runge-Kutta step with FSAL and dense-output observations: the
 * forward sweep keeps the network activations of every stage, the reverse sweep applies J_f^T stage by stage (no use
 * of the c-peptide model's linearity, which the HIP kernels exploit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RMAXW 16
#define RMAXD 8
#define RMAXT 32
#define RMAXIN 4

static const double RC[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
static const double RA[7][6] = {
    {0},
    {0.161},
    {-0.008480655492356989, 0.335480655492357},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
static const double RR[7][4] = {
    {1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216},
    {0.0, 0.13169999999999998, -0.2234, 0.1017},
    {0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253},
    {0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902},
    {0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928},
    {0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661},
    {0.0, 1.5, -4.0, 2.5}};

static void r_weights(double th, double* w) {
    if (fabs(th - 1.0) < 1e-12) { for (int j = 0; j < 6; j++) w[j] = RA[6][j]; w[6] = 0.0; return; }
    for (int i = 0; i < 7; i++) w[i] = ((RR[i][3] * th + RR[i][2]) * th + RR[i][1]) * th * th + RR[i][0] * th;
}

static int r_nparams(int nin, int width, int depth) {
    int p = 0, fan = nin;
    for (int l = 0; l < depth; l++) { p += width * fan + width; fan = width; }
    return p + fan + 1;
}

typedef struct { int nin, width, depth; const double* nn; } net_t;

/* forward pass keeping the hidden activations; returns softplus(z_out); *sig = logistic(z_out) */
static double net_fwd(const net_t* n, const double* x, double h[RMAXD][RMAXW], double* sig) {
    const double* p = n->nn;
    int fan = n->nin;
    const double* in = x;
    for (int l = 0; l < n->depth; l++) {
        const double* W = p;
        const double* b = p + (size_t)n->width * fan;
        for (int j = 0; j < n->width; j++) {
            double z = b[j];
            for (int i = 0; i < fan; i++) z += W[j + n->width * i] * in[i];
            h[l][j] = tanh(z);
        }
        p += (size_t)n->width * fan + n->width;
        in = h[l];
        fan = n->width;
    }
    double z = p[fan];
    for (int i = 0; i < fan; i++) z += p[i] * in[i];
    const double e = exp(z);
    *sig = e / (1.0 + e);
    return log(1.0 + e);
}

/* g += wgt * d out / d params;  xb += wgt * d out / d x   (activations from net_fwd at the same x) */
static void net_bwd(const net_t* n, const double* x, double h[RMAXD][RMAXW], double sig, double wgt, double* g,
                    double* xb) {
    int off[RMAXD + 1], fans[RMAXD + 1];
    int o = 0, fan = n->nin;
    for (int l = 0; l < n->depth; l++) { off[l] = o; fans[l] = fan; o += n->width * fan + n->width; fan = n->width; }
    off[n->depth] = o;
    const double dz = wgt * sig;
    double dh[RMAXW], dn[RMAXW];
    const double* po = n->nn + o;
    g[o + fan] += dz;
    for (int i = 0; i < fan; i++) { g[o + i] += dz * h[n->depth - 1][i]; dh[i] = dz * po[i]; }
    for (int l = n->depth - 1; l >= 0; l--) {
        const int fi = fans[l];
        const double* in = l > 0 ? h[l - 1] : x;
        const double* W = n->nn + off[l];
        double* gW = g + off[l];
        double d[RMAXW];
        for (int j = 0; j < n->width; j++) { d[j] = dh[j] * (1.0 - h[l][j] * h[l][j]); gW[n->width * fi + j] += d[j]; }
        for (int i = 0; i < fi; i++) {
            double s = 0.0;
            for (int j = 0; j < n->width; j++) { gW[j + n->width * i] += d[j] * in[i]; s += W[j + n->width * i] * d[j]; }
            dn[i] = s;
        }
        for (int i = 0; i < fi; i++) dh[i] = dn[i];
    }
    for (int i = 0; i < n->nin; i++) xb[i] += dh[i];
}

typedef struct {
    int model, ns;                    /* 0 = c-peptide, 1 = suppression */
    net_t net;
    double k0, k1, k2, c0, age, eb;   /* eb = exp(conditional) */
    const double* tp; const double* G; int T;
} rsub_t;

static double r_interp(const double* tk, const double* u, int T, double t) {
    int j = 0;
    while (j + 1 < T && tk[j + 1] <= t) j++;
    if (j > T - 2) j = T - 2;
    return u[j] + (t - tk[j]) * ((u[j + 1] - u[j]) / (tk[j + 1] - tk[j]));
}

/* activations of the (up to two) network evaluations of one RHS call, kept by the forward sweep for the reverse one */
typedef struct { double x[2][RMAXIN]; double h[2][RMAXD][RMAXW]; double sig[2]; } ract_t;

static void r_rhs(const rsub_t* c, double t, const double* u, double* du, ract_t* a) {
    if (c->model == 0) {
        const double dG = r_interp(c->tp, c->G, c->T, t) - r_interp(c->tp, c->G, c->T, c->tp[0]);
        const double x0[RMAXIN] = {dG, c->eb, c->age, 0}, x1[RMAXIN] = {0.0, c->eb, c->age, 0};
        memcpy(a->x[0], x0, sizeof(x0));
        memcpy(a->x[1], x1, sizeof(x1));
        const double prod = net_fwd(&c->net, a->x[0], a->h[0], &a->sig[0]) - net_fwd(&c->net, a->x[1], a->h[1], &a->sig[1]);
        du[0] = -(c->k0 + c->k2) * u[0] + c->k1 * u[1] + c->k0 * c->c0 + prod;
        du[1] = -c->k1 * u[1] + c->k2 * u[0];
        if (c->ns == 3) du[2] = prod;
    } else {
        const double x0[RMAXIN] = {u[0], u[1], u[2], c->eb};
        memcpy(a->x[0], x0, sizeof(x0));
        const double uh = net_fwd(&c->net, a->x[0], a->h[0], &a->sig[0]);
        du[0] = -0.4 * u[0];
        du[1] = 0.4 * u[0] - uh;
        du[2] = uh - 0.3 * u[2];
    }
}

/* ub += J_u^T w;  g += (d f / d theta)^T w;  *ebb += (d f / d eb)^T w   at the point whose activations are in a */
static void r_rhs_vjp(const rsub_t* c, ract_t* a, const double* w, double* ub, double* g, double* ebb) {
    double xb[RMAXIN] = {0, 0, 0, 0};
    if (c->model == 0) {
        ub[0] += -(c->k0 + c->k2) * w[0] + c->k2 * w[1];
        ub[1] += c->k1 * w[0] - c->k1 * w[1];
        const double wp = w[0] + (c->ns == 3 ? w[2] : 0.0);
        net_bwd(&c->net, a->x[0], a->h[0], a->sig[0], wp, g, xb);
        net_bwd(&c->net, a->x[1], a->h[1], a->sig[1], -wp, g, xb);
        *ebb += xb[1];
    } else {
        net_bwd(&c->net, a->x[0], a->h[0], a->sig[0], w[2] - w[1], g, xb);
        ub[0] += -0.4 * w[0] + 0.4 * w[1] + xb[0];
        ub[1] += xb[1];
        ub[2] += -0.3 * w[2] + xb[2];
        *ebb += xb[3];
    }
}

/* One subject: forward sweep keeping the network activations of every stage, then the reverse sweep.
 * obs[T][ns] observed values, ow[ns] weights of the squared residuals (0 = state not observed).
 * Returns the weighted SSE; if g != NULL accumulates gscale * d SSE / d theta into g and returns d SSE / d cond
 * (times gscale) in *gcond. */
static double r_subject(const rsub_t* c, const double* u0, const double* tp, int T, int S, const int* ostep,
                        const double* oth, const double* obs, const double* ow, double gscale, double* g,
                        double* gcond, ract_t* acts, double* res) {
    const int ns = c->ns;
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    /* acts[0] = k_1 of step 0; acts[1 + 6 n + (i - 1)] = stage i (1..6) of step n; res[T][ns] residuals */
    double y[3], k[7][3], Y[3];
    for (int s = 0; s < ns; s++) y[s] = u0[s];
    r_rhs(c, t0, y, k[0], &acts[0]);
    double sse = 0.0;
    for (int n = 0; n < S; n++) {
        const double tn = t0 + n * h;
        for (int i = 1; i < 7; i++) {
            for (int s = 0; s < ns; s++) {
                double a = 0.0;
                for (int j = 0; j < i; j++) a += RA[i][j] * k[j][s];
                Y[s] = y[s] + h * a;
            }
            r_rhs(c, i < 6 ? tn + RC[i] * h : t0 + (n + 1) * h, Y, k[i], &acts[1 + 6 * n + (i - 1)]);
        }
        for (int ti = 0; ti < T; ti++) if (ostep[ti] == n) {
            double w[7];
            r_weights(oth[ti], w);
            for (int s = 0; s < ns; s++) {
                double a = 0.0;
                for (int j = 0; j < 7; j++) a += w[j] * k[j][s];
                const double r = (y[s] + h * a) - obs[ti * ns + s];
                res[ti * ns + s] = r;
                sse += ow[s] * r * r;
            }
        }
        for (int s = 0; s < ns; s++) { y[s] = Y[s]; k[0][s] = k[6][s]; }
    }
    if (!g) return sse;
    double lam[3] = {0, 0, 0}, kap[3] = {0, 0, 0}, ebb = 0.0;
    for (int n = S - 1; n >= 0; n--) {
        ract_t* an = acts + 1 + 6 * n;
        double kb[7][3], yb[3] = {0, 0, 0};
        memset(kb, 0, sizeof(kb));
        for (int s = 0; s < ns; s++) kb[6][s] = kap[s];
        for (int ti = T - 1; ti >= 0; ti--) if (ostep[ti] == n) {
            double w[7];
            r_weights(oth[ti], w);
            for (int s = 0; s < ns; s++) {
                const double ob = gscale * 2.0 * ow[s] * res[ti * ns + s];
                yb[s] += ob;
                for (int j = 0; j < 7; j++) kb[j][s] += h * w[j] * ob;
            }
        }
        /* k_7 = f(t_{n+1}, y_{n+1}) */
        r_rhs_vjp(c, &an[5], kb[6], lam, g, &ebb);
        /* y_{n+1} = y_n + h sum_j a_7j k_j */
        for (int s = 0; s < ns; s++) {
            yb[s] += lam[s];
            for (int j = 0; j < 6; j++) kb[j][s] += h * RA[6][j] * lam[s];
        }
        for (int i = 5; i >= 1; i--) {
            double Yb[3] = {0, 0, 0};
            r_rhs_vjp(c, &an[i - 1], kb[i], Yb, g, &ebb);
            for (int s = 0; s < ns; s++) {
                yb[s] += Yb[s];
                for (int j = 0; j < i; j++) kb[j][s] += h * RA[i][j] * Yb[s];
            }
        }
        if (n == 0) {                       /* k_1 = f(t_0, y_0) is evaluated, not inherited */
            r_rhs_vjp(c, &acts[0], kb[0], yb, g, &ebb);
            for (int s = 0; s < ns; s++) kap[s] = 0.0;
        } else {
            for (int s = 0; s < ns; s++) kap[s] = kb[0][s];
        }
        for (int s = 0; s < ns; s++) lam[s] = yb[s];
    }
    *gcond = ebb * c->eb;                   /* conditional enters as exp(conditional) */
    return sse;
}

static void r_locate(const double* tp, int T, int S, int* step, double* theta) {
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    for (int i = 0; i < T; i++) {
        int n = (int)ceil((tp[i] - t0) / h - 1e-9) - 1;
        if (n < 0) n = 0;
        if (n > S - 1) n = S - 1;
        step[i] = n;
        theta[i] = (tp[i] - (t0 + n * h)) / h;
    }
}

static void r_van_cauter(double age, int t2dm, double* k0, double* k1, double* k2) {
    const double sh = t2dm ? 4.52 : 4.95, fr = t2dm ? 0.78 : 0.76, lo = 0.14 * age + 29.2;
    *k1 = fr * (log(2.0) / lo) + (1 - fr) * (log(2.0) / sh);
    *k0 = (log(2.0) / sh) * (log(2.0) / lo) / *k1;
    *k2 = (log(2.0) / sh) + (log(2.0) / lo) - *k0 - *k1;
}

/* Population loss and gradient of the c-peptide cUDE, reverse mode.  Arguments as cude_oracle_cpep (no trajectory
 * output).  Returns the number of failed subjects, -1 for unsupported sizes. */
int cude_oracle_cpep_rev(int N, int T, const double* tp, const double* glucose, const double* cpeptide,
                         const double* age, const uint8_t* t2dm, int covariate, int nin, int width, int depth,
                         const double* nn, const double* beta, int n_steps, int n_state, int want_grad, int nthreads,
                         double* loss, double* sse, double* g_nn, double* g_beta) {
    const int P = r_nparams(nin, width, depth);
    if (T > RMAXT || width > RMAXW || depth > RMAXD || nin > 3 || n_state < 2 || n_state > 3) return -1;
    (void)covariate;
    int ostep[RMAXT]; double oth[RMAXT];
    r_locate(tp, T, n_steps, ostep, oth);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    /* one accumulator row per thread, padded to whole cache lines and line-aligned: adjacent rows of P + 2 = 69 doubles
     * shared a line at every boundary, and every subject's update of that line bounced it between two cores */
    const size_t grow = ((size_t)(P + 2) + 7) / 8 * 8;
    double* gacc = (double*)aligned_alloc(64, (size_t)nthreads * grow * sizeof(double));
    memset(gacc, 0, (size_t)nthreads * grow * sizeof(double));
    int nfail = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : nfail)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double* ga = gacc + (size_t)tid * grow;
        ract_t* acts = (ract_t*)malloc(sizeof(ract_t) * ((size_t)6 * n_steps + 1));
        double* work = (double*)malloc(sizeof(double) * (size_t)T * 3);
        double* obs = (double*)malloc(sizeof(double) * (size_t)T * 3);
#pragma omp for schedule(static)
        for (int i = 0; i < N; i++) {
            rsub_t c;
            memset(&c, 0, sizeof(c));
            c.model = 0; c.ns = n_state;
            c.net.nin = nin; c.net.width = width; c.net.depth = depth; c.net.nn = nn;
            r_van_cauter(age[i], t2dm[i], &c.k0, &c.k1, &c.k2);
            c.c0 = cpeptide[(size_t)i * T]; c.age = age[i]; c.eb = exp(beta[i]);
            c.tp = tp; c.G = glucose + (size_t)i * T; c.T = T;
            const double u0[3] = {c.c0, (c.k2 / c.k1) * c.c0, 0.0};
            const double ow[3] = {1.0, 0.0, 0.0};
            for (int t = 0; t < T; t++) { obs[t * n_state] = cpeptide[(size_t)i * T + t]; obs[t * n_state + 1] = 0.0; if (n_state == 3) obs[t * n_state + 2] = 0.0; }
            double gc = 0.0;
            const double s = r_subject(&c, u0, tp, T, n_steps, ostep, oth, obs, ow, 1.0 / N, want_grad ? ga : NULL, &gc, acts, work);
            if (sse) sse[i] = s;
            if (!isfinite(s)) nfail += 1;
            ga[P] += s;
            if (want_grad) g_beta[i] = gc;
        }
        free(acts);
        free(work);
        free(obs);
    }
    double tot = 0.0;
    if (want_grad) for (int q = 0; q < P; q++) g_nn[q] = 0.0;
    for (int t = 0; t < nthreads; t++) {
        tot += gacc[(size_t)t * grow + P];
        if (want_grad) for (int q = 0; q < P; q++) g_nn[q] += gacc[(size_t)t * grow + q];
    }
    free(gacc);
    *loss = nfail ? INFINITY : tot / N;
    return nfail;
}

/* suppression_loss and its gradient, reverse mode.  Arguments as cude_oracle_supp (no trajectory output). */
int cude_oracle_supp_rev(int N, int T, const double* tp, const double* data, int width, int depth, const double* nn,
                         const double* theta, double lambda, int n_steps, int want_grad, int nthreads, double* loss,
                         double* sse, double* g_nn, double* g_theta) {
    const int P = r_nparams(4, width, depth);
    if (T > RMAXT || width > RMAXW || depth > RMAXD) return -1;
    int ostep[RMAXT]; double oth[RMAXT];
    r_locate(tp, T, n_steps, ostep, oth);
    double scale[3] = {0, 0, 0};
    for (int i = 0; i < N; i++) for (int s = 0; s < 3; s++) {
        double m = -INFINITY;
        for (int t = 0; t < T; t++) { const double v = data[s + 3 * (t + (size_t)T * i)]; if (v > m) m = v; }
        scale[s] += m;
    }
    for (int s = 0; s < 3; s++) scale[s] /= N;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    /* one accumulator row per thread, padded to whole cache lines and line-aligned: adjacent rows of P + 2 = 69 doubles
     * shared a line at every boundary, and every subject's update of that line bounced it between two cores */
    const size_t grow = ((size_t)(P + 2) + 7) / 8 * 8;
    double* gacc = (double*)aligned_alloc(64, (size_t)nthreads * grow * sizeof(double));
    memset(gacc, 0, (size_t)nthreads * grow * sizeof(double));
    int nfail = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : nfail)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double* ga = gacc + (size_t)tid * grow;
        ract_t* acts = (ract_t*)malloc(sizeof(ract_t) * ((size_t)6 * n_steps + 1));
        double* work = (double*)malloc(sizeof(double) * (size_t)T * 3);
#pragma omp for schedule(static)
        for (int i = 0; i < N; i++) {
            rsub_t c;
            memset(&c, 0, sizeof(c));
            c.model = 1; c.ns = 3;
            c.net.nin = 4; c.net.width = width; c.net.depth = depth; c.net.nn = nn;
            c.eb = exp(theta[i]);
            const double* obs = data + (size_t)3 * T * i;          /* [T][3]: s fastest */
            const double u0[3] = {obs[0], obs[1], obs[2]};
            const double ow[3] = {1.0 / (scale[0] * scale[0]), 1.0 / (scale[1] * scale[1]), 1.0 / (scale[2] * scale[2])};
            double gc = 0.0;
            const double s = r_subject(&c, u0, tp, T, n_steps, ostep, oth, obs, ow, 1.0 / N, want_grad ? ga : NULL, &gc, acts, work);
            if (sse) sse[i] = s;
            if (!isfinite(s)) nfail += 1;
            ga[P] += s;
            if (want_grad) g_theta[i] = gc;
        }
        free(acts);
        free(work);
    }
    double tot = 0.0, reg = 0.0;
    if (want_grad) for (int q = 0; q < P; q++) g_nn[q] = 0.0;
    for (int t = 0; t < nthreads; t++) {
        tot += gacc[(size_t)t * grow + P];
        if (want_grad) for (int q = 0; q < P; q++) g_nn[q] += gacc[(size_t)t * grow + q];
    }
    for (int q = 0; q < P; q++) reg += nn[q] * nn[q];
    if (want_grad) for (int q = 0; q < P; q++) g_nn[q] += 2.0 * lambda * nn[q];
    free(gacc);
    *loss = nfail ? INFINITY : tot / N + lambda * reg;
    return nfail;
}
