// Chunked ("time-split") c-peptide cUDE loss+gradient path for gfx950.
//
// Same mathematics as cude_cpep.hip (fixed-step Tsit5 + discrete adjoint of the c-peptide cUDE,
// reference: src/c-peptide-models.jl:7-14,86-94; src/parameter-estimation.jl:56-68,126-140,370), but
// each subject's S steps are split into L contiguous chunks that run in DIFFERENT lanes (grid row =
// chunk), which (a) triples the number of waves -- at the benchmark size one lane per subject gives only
// 1.9 waves per SIMD -- and (b) lets the forward sweep (131 VGPRs) run at its own, higher occupancy
// instead of inheriting the reverse sweep's 229 VGPRs.
//
// Why this is possible: the network input [dG(t), e^beta] does not depend on the state, so the ODE is
// f = A u + [k0 c0 + q(t); 0] with constant A.  Hence
//   * the state at the end of a chunk is AFFINE in the state at its start: y_out = M y_in + v, where
//     M (and the response H_tau of every observation to y_in) depend only on the subject's kinetics and
//     are precomputed once per population (cpep2_homog_kernel), and v / the forced observation parts
//     come from running the chunk from a ZERO entry state with the network forcing (cpep2_fwd_kernel);
//   * a per-subject scan stitches the chunks, forms the residuals and -- because the adjoint recursion
//     needs no network evaluation and no forward state (J_f = A) -- runs the whole stage-adjoint algebra
//     once, storing the weight of every network evaluation (cpep2_scan_kernel);
//   * the reverse lane of chunk c evaluates the network reverse sweeps of its own stage times with those
//     weights and accumulates its share of the gradient (cpep2_rev_kernel).
#include <cstdlib>
#include <mutex>
#include <set>

#include "cude_device.h"
#include "cude_kernels.h"
#include "cude_rng.h"

namespace cude {

// ---------------------------------------------------------------------------------- shared pieces
struct Kin {
    double a11, a12, a21, a22, f0;
};

// One fixed Tsit5 step of  y' = A y + [g_i; 0]  given k_1 (FSAL); g[i] is the forcing of stage i+1.
// Returns y_{n+1} in (Y1,Y2) and all stage derivatives in K.
__device__ __forceinline__ void rk_step(const Kin& k, double h, double y1, double y2, const double (&g)[7],
                                        double (&K)[7][2], double& Y1, double& Y2) {
#pragma unroll
    for (int st = 1; st < 7; st++) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int j = 0; j < st; j++) {
            t1 = fma(Tab::a(st, j), K[j][0], t1);
            t2 = fma(Tab::a(st, j), K[j][1], t2);
        }
        Y1 = fma(h, t1, y1);
        Y2 = fma(h, t2, y2);
        K[st][0] = fma(k.a11, Y1, fma(k.a12, Y2, g[st]));
        K[st][1] = fma(k.a21, Y1, k.a22 * Y2);
    }
}

// ---------------------------------------------------------------------------------- homogeneous responses
// Per subject, per chunk: M = d y(chunk end) / d y(chunk start); per observation: H = d y1(tau) / d y(start of
// the chunk holding tau).  Depends only on (k0,k1,k2), the step grid and the tableau: computed once.
__global__ __launch_bounds__(kBlock) void cpep2_homog_kernel(Cpep2Args a) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.base.N) return;
    const int64_t N = a.base.N;
    cptr_t obs_w = as_const(a.base.obs_w);
    ciptr_t obs_step = as_const(a.base.obs_step);
    ciptr_t cs = as_const(a.chunk_start);
    const int T = a.base.T;
    const double h = a.base.h;
    const double k0 = a.base.k0[i], k1 = a.base.k1[i], k2 = a.base.k2[i];
    const Kin kin{-(k0 + k2), k1, k2, -k1, 0.0};
    const double g0[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < a.L; c++) {
        const int n0 = cs[c], n1 = cs[c + 1];
        for (int b = 0; b < 2; b++) {
            double y1 = b == 0 ? 1.0 : 0.0, y2 = b == 0 ? 0.0 : 1.0;
            double K[7][2];
            K[0][0] = fma(kin.a11, y1, kin.a12 * y2);
            K[0][1] = fma(kin.a21, y1, kin.a22 * y2);
            int oi = 0;
            while (oi < T && obs_step[oi] < n0) oi++;
            for (int n = n0; n < n1; n++) {
                double Y1, Y2;
                rk_step(kin, h, y1, y2, g0, K, Y1, Y2);
                while (oi < T && obs_step[oi] == n) {
                    double o1 = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; j++) o1 = fma(obs_w[oi * 7 + j], K[j][0], o1);
                    a.hom_obs[((int64_t)oi * 2 + b) * N + i] = fma(h, o1, y1);
                    oi++;
                }
                y1 = Y1; y2 = Y2;
                K[0][0] = K[6][0]; K[0][1] = K[6][1];
            }
            a.hom_M[((int64_t)c * 4 + 0 + b) * N + i] = y1;      // M[0][b]
            a.hom_M[((int64_t)c * 4 + 2 + b) * N + i] = y2;      // M[1][b]
        }
    }
}

// ---------------------------------------------------------------------------------- forward (forced, zero entry)
// VWR: the upper-layer weights live in VGPRs (Mlp::VW) -- no scalar loads in the evaluations.  Costs ~100 VGPRs (two
// waves per SIMD instead of four), so the launcher picks it only while the grid fits two waves per SIMD anyway: the
// latency-bound regime of small populations (forward launch at 1e4 subjects 46.4 -> 45.0 us, 2e4 64.2 -> 61.5 us).
// The five network evaluations of a step are independent of each other and of the state (their inputs are times), so they
// are issued together -- five instruction streams for the scheduler to interleave instead of one dependent chain per
// evaluation -- with their inputs from LDS-resident glucose knots instead of a scalar search and a conditional reload per
// evaluation (rounds 1-3 walked one network body through the evaluations one at a time: forward launch at 1e4 subjects
// 42.4 -> 37.7 us, at 1e5 0.229 -> 0.214 ms, SAEM E-step 3.87 -> 3.38 ms per 1e4 x 100 draws; same bits).
template <int NIN, int W, int D, int NS, bool VWR = false>
__global__ __launch_bounds__(kBlock) void cpep2_fwd_kernel(Cpep2Args a) {
    using Net = CpepNet<NIN, W, D>;
    constexpr int NC = NIN - 1;
    extern __shared__ double smem[];
    const CpepArgs& b = a.base;
    const int lane = threadIdx.x;
    const int64_t gid = ((int64_t)blockIdx.x + b.blk0) * kBlock + lane;      // (blk0: mixed launch, see CpepArgs)
    const bool active = gid < b.N;
    const int64_t i = active ? gid : b.N - 1;
    const int64_t N = b.N;
    const int c_idx = blockIdx.y;
    // side-by-side parameter sets (restarts trained together on a small population): grid z = set; every per-set array
    // is [set][...] with the strides of one set (single evaluations are set 0 of 1)
    const int64_t set = blockIdx.z;
#ifdef CUDE_SCAN_TIMING
    unsigned long long tk[8];
    tk[0] = __builtin_amdgcn_s_memrealtime();
#define TKF(k) tk[k] = __builtin_amdgcn_s_memrealtime()
#define TKF_PIN(x) x        // (the values a stamp stands behind have to exist by then)
#else
#define TKF(k)
#define TKF_PIN(x)
#endif
    cptr_t p = as_const(b.nn + set * b.set_stride_nn);
    cptr_t phi = as_const(b.phi);
    cptr_t obs_w = as_const(b.obs_w);
    ciptr_t seg = as_const(b.seg);
    ciptr_t obs_step = as_const(b.obs_step);
    ciptr_t cs = as_const(a.chunk_start);
    const int T = b.T;
    const double h = b.h;
    const int n0 = cs[c_idx], n1 = cs[c_idx + 1];

    const double k0 = b.k0[i], k1 = b.k1[i], k2 = b.k2[i], c0 = b.c0[i];
    const Kin kin{-(k0 + k2), k1, k2, -k1, k0 * c0};
    double cst[NC];
    cst[0] = exp(a.mh_fused ? mh_proposal(a.mh.p, a.mh_z, a.mh.key, a.mh_std, i) : b.cond[set * b.set_stride_cond + i]);
    double* const fsum = a.fsum + set * ((int64_t)a.L * (3 + b.T) * b.N);
    if (NC > 1) cst[1] = b.age[i];
    double c[W];
    Net::first_layer_offset(p, cst, c);
    double chk = fma(cst[0], 0.0, Net::param_check(p));   // NaN iff a parameter / beta is non-finite
    if (NC > 1) chk = fma(cst[1], 0.0, chk);
    typename Net::VW vw;
    if constexpr (VWR) Net::load_vw(p, vw);

    double y1 = 0.0, y2 = 0.0, y3 = 0.0;                  // forced response from a ZERO entry state
    double qprev = 0.0, base = 0.0;
    double K1a = 0.0, K1b = 0.0;
    // first observation of this chunk: the count of observation steps before n0 (obs_step ascends) by a wave vote over
    // one vector load -- the scalar search loop cost up to T dependent round trips on a cold cache
    int oi;
    {
        const int* os = b.obs_step;
        const bool before = lane < T && os[lane < T ? lane : 0] < n0;
        oi = (int)__popcll(__ballot(before));
    }
    int n = n0;
    TKF_PIN(asm volatile("" ::"v"(chk), "v"(c[0])));
    TKF(1);
    if constexpr (Net::USES_TANH) tanh_tab_init(lane, !Net::LDS_BIAS);    // (here: its global read travels with the subject's own loads)
    Net::bias_init(b.nn + set * b.set_stride_nn, lane);
    TKF(2);
    {
        double* const s_G = smem;                         // [T][kBlock] glucose increments at the knots
        for (int m = 0; m < T; m++) {
            const double gv = b.dG[(int64_t)m * N + i];
            s_G[m * kBlock + lane] = gv;
            chk = fma(gv, 0.0, chk);
        }
        TKF_PIN(asm volatile("" ::"v"(chk)));
        TKF(3);
        auto input_at = [&](int e) {
            const int sg = seg[e];
            const double lo = s_G[sg * kBlock + lane];
            return fma(phi[e], s_G[(sg + 1) * kBlock + lane] - lo, lo);
        };
        auto net = [&](double xv) {
            const double x[1] = {xv};
            if constexpr (VWR) return Net::eval_vw(p, vw, c, x, false, nullptr);
            else return Net::eval(p, c, x);
        };
        {   // baseline NN([0; e^beta]) and the forcing at the chunk's first stage-1 time (exactly 0 for the first chunk)
            const double x1 = n0 > 0 ? input_at(5 * n0 - 1) : 0.0;
            const double v0 = net(0.0), v1 = net(x1);
            base = v0;
            qprev = v1 - base;
            K1a = kin.f0 + qprev;
            K1b = 0.0;
        }
        TKF_PIN(asm volatile("" ::"v"(K1a)));
        TKF(4);
#pragma unroll 1
        for (; n < n1; n++) {
            double xs[5], q[7], g[7];
#pragma unroll
            for (int j = 0; j < 5; j++) xs[j] = input_at(5 * n + j);
            q[0] = qprev;
#pragma unroll
            for (int j = 0; j < 5; j++) q[j + 1] = net(xs[j]) - base;
            q[6] = q[5];
#pragma unroll
            for (int j = 0; j < 7; j++) g[j] = kin.f0 + q[j];
            double K[7][2];
            K[0][0] = K1a;
            K[0][1] = K1b;
            double Y1, Y2;
            rk_step(kin, h, y1, y2, g, K, Y1, Y2);
            double y3n = y3;
            if (NS == 3) {
                double t3 = 0.0;
#pragma unroll
                for (int j = 0; j < 6; j++) t3 = fma(Tab::a(6, j), q[j], t3);
                y3n = fma(h, t3, y3);
            }
            while (oi < T && obs_step[oi] == n) {
                double o1 = 0.0;
#pragma unroll
                for (int j = 0; j < 7; j++) o1 = fma(obs_w[oi * 7 + j], K[j][0], o1);
                if (active) fsum[((int64_t)c_idx * (3 + T) + 3 + oi) * N + i] = fma(h, o1, y1) + chk;
                oi++;
            }
            y1 = Y1; y2 = Y2; y3 = y3n;
            K1a = K[6][0]; K1b = K[6][1];
            qprev = q[6];
        }
    }
    TKF_PIN(asm volatile("" ::"v"(y1), "v"(y2)));
    TKF(5);
    if (active) {
        double* f = fsum + (int64_t)c_idx * (3 + T) * N + i;
        f[0] = y1 + chk;
        f[N] = y2;
        f[2 * N] = y3;
    }
#ifdef CUDE_SCAN_TIMING
    TKF(6);
    if (lane == 0 && blockIdx.x == 0 && c_idx == 1 && a.mh_fused == 0)
        printf("fwd chunk 1 of %d (%d steps) [10 ns]: subject loads + exp + first layer %llu, table fills %llu, glucose rows %llu, two evaluations %llu, steps %llu, stores %llu\n",
               a.L, n1 - n0, tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5]);
#endif
#undef TKF
#undef TKF_PIN
}

// ---------------------------------------------------------------------------------- scan + residuals
__device__ __forceinline__ void adj_step(const Kin& k, double h, double gscale, cptr_t obs_w, ciptr_t obs_step,
                                         const double* s_res, int lane, int n, int& oi, int& oi_step, double& lam1,
                                         double& lam2, double& kap1, double& kap2, double (&w)[5]);

// Stitches the chunks (affine maps), forms residuals / SSE, and -- for the gradient -- runs the network-free
// stage-adjoint recursion ONCE per subject, storing the 5 network weights of every step (wts[5S][N]) so that
// the reverse lanes of all chunks only have to read theirs.
// SPEC (speculative Metropolis round, Cpep2Args::spec_slots): a workgroup holds every candidate set of its 64 / slots
// subjects; behind the SSEs the slot-0 lanes resolve the round and write the next one's candidates (mh_spec_resolve).
// (P = the network's parameter count, a run-time argument: it only places the (loss, failures) pair in the partial-sum rows,
//  and one instance per network shape was a third of this file's compile time)
template <bool SPEC = false>
__global__ __launch_bounds__(kBlock) void cpep2_scan_kernel(Cpep2Args a, int P) {
    extern __shared__ double smem[];
    const CpepArgs& b = a.base;
    const int lane = threadIdx.x;
    const int spb = SPEC ? kBlock / a.spec_slots : kBlock;                   // subjects per workgroup
    const int64_t gid = SPEC ? (int64_t)blockIdx.x * spb + lane % spb
                             : ((int64_t)blockIdx.x + b.blk0) * kBlock + lane;      // (blk0: mixed launch, see CpepArgs)
    const int64_t N = b.N;
    ciptr_t obs_step = as_const(b.obs_step);
    ciptr_t cs = as_const(a.chunk_start);
    const int T = b.T;
    const int64_t set_raw = SPEC ? lane / spb : blockIdx.y;                  // parameter set (see cpep2_fwd_kernel)
    const bool active = gid < N && (!SPEC || set_raw < b.n_sets);
    const int64_t i = gid < N ? gid : N - 1;
    const int64_t set = SPEC && set_raw >= b.n_sets ? 0 : set_raw;           // (the idle slot re-reads set 0, writes nothing)
    const double* const fsum = a.fsum + set * ((int64_t)a.L * (3 + T) * N);
    double* const wts = a.wts != nullptr ? a.wts + set * ((int64_t)5 * b.S * N) : nullptr;
    double* s_res = smem + kRedRows * kBlock;   // [T][kBlock] residuals, then [T][4][kBlock] what every observation needs
    const double k0 = b.k0[i], k1 = b.k1[i], k2 = b.k2[i], c0 = b.c0[i];
    double y1 = c0, y2 = (k2 / k1) * c0, y3 = 0.0, sse = 0.0;
    int oi = 0;
    // The scan is a chain of L dependent affine maps, but what it READS does not depend on the chain.  Rounds 2-4 requested
    // chunk c + 1's summary while chunk c was applied: a look-ahead of ONE short iteration, so every chunk still waited
    // most of an HBM round trip (measured: the scan took 25 us for 57 subjects as for 1e4 -- L = 30 round trips).  Now
    // (round 5): everything an observation needs (its two homogeneous responses, its forced part, its datum) is put into
    // LDS rows up front with all loads in flight together, and the chunk summaries arrive kScanGroup chunks at a time,
    // one group ahead.  Same operations in the same order on the same values.
    double* s_obs = s_res + T * kBlock;         // [T][4][kBlock]
    {
        constexpr int kObsBatch = 4;                // observations whose loads are in flight together
        int c_of = 0;
        for (int o0 = 0; o0 < T; o0 += kObsBatch) {
            double v[kObsBatch][4];
#pragma unroll
            for (int k = 0; k < kObsBatch; k++) {
                const int o = o0 + k < T ? o0 + k : T - 1;
                const int st = obs_step[o];
                while (c_of + 1 < a.L && cs[c_of + 1] <= st) c_of++;      // the chunk whose steps hold observation o
                v[k][0] = a.hom_obs[((int64_t)o * 2) * N + i];
                v[k][1] = a.hom_obs[((int64_t)o * 2 + 1) * N + i];
                v[k][2] = fsum[((int64_t)c_of * (3 + T) + 3 + o) * N + i];
                v[k][3] = b.obs[(int64_t)o * N + i];
            }
#pragma unroll
            for (int k = 0; k < kObsBatch; k++) {
                if (o0 + k < T) {
#pragma unroll
                    for (int q = 0; q < 4; q++) s_obs[(4 * (o0 + k) + q) * kBlock + lane] = v[k][q];
                }
            }
        }
    }
    constexpr int kScanGroup = 6;
    double nx[kScanGroup][7];
    auto request = [&](int c0) {                 // summaries and transfer matrices of chunks c0 ... c0 + kScanGroup - 1
#pragma unroll
        for (int k = 0; k < kScanGroup; k++) {
            const int c = c0 + k < a.L ? c0 + k : a.L - 1;
            const double* f = fsum + (int64_t)c * (3 + T) * N + i;
            const double* M = a.hom_M + (int64_t)c * 4 * N + i;
            nx[k][0] = f[0]; nx[k][1] = f[N]; nx[k][2] = f[2 * N];
            nx[k][3] = M[0]; nx[k][4] = M[N]; nx[k][5] = M[2 * N]; nx[k][6] = M[3 * N];
        }
    };
    request(0);
    for (int c0 = 0; c0 < a.L; c0 += kScanGroup) {
        double cu[kScanGroup][7];
#pragma unroll
        for (int k = 0; k < kScanGroup; k++)
#pragma unroll
            for (int q = 0; q < 7; q++) cu[k][q] = nx[k][q];
        if (c0 + kScanGroup < a.L) request(c0 + kScanGroup);
#pragma unroll
        for (int k = 0; k < kScanGroup; k++) {
            const int c = c0 + k;
            if (c < a.L) {
                const int n1 = cs[c + 1];
                while (oi < T && obs_step[oi] < n1) {
                    const double hy = fma(s_obs[(4 * oi + 0) * kBlock + lane], y1, s_obs[(4 * oi + 1) * kBlock + lane] * y2);
                    const double r = (s_obs[(4 * oi + 2) * kBlock + lane] + hy) - s_obs[(4 * oi + 3) * kBlock + lane];
                    sse = fma(r, r, sse);
                    s_res[oi * kBlock + lane] = r;
                    oi++;
                }
                const double n1y = cu[k][0] + fma(cu[k][3], y1, cu[k][4] * y2);
                const double n2y = cu[k][1] + fma(cu[k][5], y1, cu[k][6] * y2);
                y1 = n1y; y2 = n2y;
                y3 += cu[k][2];
            }
        }
    }
    const bool failed = !(fabs(sse) <= 1.79769313486231570815e308);
    if (active) {
        if (b.sse != nullptr) b.sse[set * b.set_stride_cond + i] = sse;
        if (b.auc != nullptr && set == 0) b.auc[i] = y3;
        if (a.mh_fused) mh_accept_one(a.mh, i, mh_proposal(a.mh.p, a.mh_z, a.mh.key, a.mh_std, i), sse);
    }
    if (wts != nullptr) {
        const Kin kin{-(k0 + k2), k1, k2, -k1, k0 * c0};
        cptr_t obs_w = as_const(b.obs_w);
        double lam1 = 0.0, lam2 = 0.0, kap1 = 0.0, kap2 = 0.0, w[5];
        const double gscale = 2.0 * b.inv_n;
        oi = T - 1;
        int oi_step = obs_step[oi];
        if (a.adj_map != nullptr) {
            // the recursion as the subject's linear map (Cpep2Args::adj_map): 36 multiply-adds per step; the response of
            // the NEXT observation to be met is requested while the steps before it run
            const double* mp = a.adj_map + i;
            double Phi[16], Wm[20], R[9];
#pragma unroll
            for (int q = 0; q < 16; q++) Phi[q] = mp[(int64_t)q * N];
#pragma unroll
            for (int q = 0; q < 20; q++) Wm[q] = mp[(int64_t)(16 + q) * N];
#pragma unroll
            for (int q = 0; q < 9; q++) R[q] = mp[(int64_t)(kAdjMapRows + 9 * oi + q) * N];
#pragma unroll 1
            for (int n = b.S - 1; n >= 0; n--) {
                const double in[4] = {lam1, lam2, kap1, kap2};
                double o[4];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    o[r] = fma(Phi[4 * r + 3], in[3], fma(Phi[4 * r + 2], in[2], fma(Phi[4 * r + 1], in[1], Phi[4 * r] * in[0])));
#pragma unroll
                for (int j = 0; j < 5; j++)
                    w[j] = fma(Wm[4 * j + 3], in[3], fma(Wm[4 * j + 2], in[2], fma(Wm[4 * j + 1], in[1], Wm[4 * j] * in[0])));
                while (oi_step == n) {
                    const double g = gscale * s_res[oi * kBlock + lane];
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = fma(R[r], g, o[r]);
#pragma unroll
                    for (int j = 0; j < 5; j++) w[j] = fma(R[4 + j], g, w[j]);
                    oi--;
                    oi_step = oi >= 0 ? obs_step[oi] : -1;
                    const int on = oi >= 0 ? oi : 0;
#pragma unroll
                    for (int q = 0; q < 9; q++) R[q] = mp[(int64_t)(kAdjMapRows + 9 * on + q) * N];
                }
                lam1 = o[0]; lam2 = o[1]; kap1 = o[2]; kap2 = o[3];
                if (active) {
#pragma unroll
                    for (int j = 0; j < 5; j++) wts[(int64_t)(5 * n + j) * N + i] = w[j];
                }
            }
        } else {
#pragma unroll 1
            for (int n = b.S - 1; n >= 0; n--) {
                adj_step(kin, b.h, gscale, obs_w, obs_step, s_res, lane, n, oi, oi_step, lam1, lam2, kap1, kap2, w);
                if (active) {
#pragma unroll
                    for (int j = 0; j < 5; j++) wts[(int64_t)(5 * n + j) * N + i] = w[j];
                }
            }
        }
    }
    if constexpr (SPEC) {
        // the workgroup's own stores of the candidates' SSEs (global memory) are visible to it behind the barrier
        __syncthreads();
        if (lane < spb && gid < N) mh_spec_resolve(a.spec, gid);
        return;
    }
    const double v2[2] = {active ? sse : 0.0, (active && failed) ? 1.0 : 0.0};
    block_reduce_store<2>(v2, smem, b.partials + (set * gridDim.x + blockIdx.x + b.blk0) * (P + 2) + P, lane,
                          a.final_host != nullptr ? a.final_host + 2 * (int64_t)blockIdx.x : nullptr);
}

// ---------------------------------------------------------------------------------- scan of a small launch
// On a nearly empty chip the scan is a LONE wave, and a lone wave issues one instruction every ~5 cycles whatever the
// instruction is (measured with s_memrealtime stamps, tools/scan_timing.py, 57 subjects, 32 chunks, 2.4 GHz: requesting its
// ~200 rows 6.9 us although they arrive 0.2 us after the last request, stitching 4.6 us, the 32 steps of the adjoint
// recursion 7.4 us = 550 cycles per step of ~100 instructions, the final sums 1.0 us: 20.5 us, none of it memory latency --
// a first version that only moved every read to the top of the kernel ran exactly as long).  Hence, for launches of at
// most one scan workgroup per compute unit, EIGHT waves share a block of 64 subjects:
//   1. every row the scan reads is requested up front, each wave an eighth of them, by loads that write LDS directly
//      (global_load_lds_dword: no destination registers, hundreds in flight; a lane's double travels as two dwords into
//      rows 256 B apart);
//   2. wave 0 stitches the chunks out of LDS (chunk boundaries by arithmetic, the next observation's step in a register:
//      no scalar load per chunk) and runs the adjoint recursion's four STATE components alone through all steps (16 of
//      the step's 36 multiply-adds, no stores), leaving the state at the head of every wave's segment of steps in LDS;
//   3. every wave then runs the full recursion (weights + stores) over its own segment from that state.
// The state components do not depend on the weights, so every number is produced by the same operations in the same
// order as in cpep2_scan_kernel: same bits.  Needs chunks of equal length (then every chunk has chunk 0's transfer
// matrix: 4 rows instead of 4 L) and the recursion as a linear map (Cpep2Args::adj_map).
constexpr int kScanWaves = 8;
// orders a wave's LDS writes against the reads of its other lanes (a wave's LDS operations execute in order: no s_barrier)
__device__ __forceinline__ void stage_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double lds_get_plain(const double* row, int lane) { return row[lane]; }
__device__ __forceinline__ void lds_fetch(const double* g, double* row) {
    using lds_ptr = __attribute__((address_space(3))) void*;
    const unsigned* gp = reinterpret_cast<const unsigned*>(g);
    __builtin_amdgcn_global_load_lds(gp, (lds_ptr)row, 4, 0, 0);
    __builtin_amdgcn_global_load_lds(gp + 1, (lds_ptr)(row + kBlock / 2), 4, 0, 0);
}
__device__ __forceinline__ double lds_get(const double* row, int lane) {
    const unsigned* r = reinterpret_cast<const unsigned*>(row);
    return __hiloint2double((int)r[kBlock + lane], (int)r[lane]);
}
inline int scan_bulk_rows(int T, int L, bool grad) {
    return kRedRows + 5 * T + 8 + 3 * L + (grad ? kAdjMapRows + 9 * T + 4 * kScanWaves : 0);
}

template <bool SPEC = false>
__global__ __launch_bounds__(kScanWaves* kBlock) void cpep2_scan_bulk_kernel(Cpep2Args a, int P) {
    extern __shared__ double smem[];
    const CpepArgs& b = a.base;
    const int lane = threadIdx.x & (kBlock - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int spb = SPEC ? kBlock / a.spec_slots : kBlock;                   // subjects per workgroup
    const int64_t gid = SPEC ? (int64_t)blockIdx.x * spb + lane % spb : (int64_t)blockIdx.x * kBlock + lane;
    const int64_t N = b.N;
    ciptr_t obs_step = as_const(b.obs_step);
    const int T = b.T, L = a.L, S = b.S;
    const int len = S / L;                                                   // steps per chunk
    const int64_t set_raw = SPEC ? lane / spb : blockIdx.y;                  // parameter set (see cpep2_fwd_kernel)
    const bool active = gid < N && (!SPEC || set_raw < b.n_sets);
    const int64_t i = gid < N ? gid : N - 1;
    const int64_t set = SPEC && set_raw >= b.n_sets ? 0 : set_raw;           // (the idle slot re-reads set 0, writes nothing)
    const double* const fsum = a.fsum + set * ((int64_t)L * (3 + T) * N);
    double* const wts = a.wts != nullptr ? a.wts + set * ((int64_t)5 * S * N) : nullptr;
    double* const s_res = smem + kRedRows * kBlock;     // [T] residuals
    double* const s_obs = s_res + T * kBlock;           // [T][4] what every observation needs
    double* const s_kin = s_obs + 4 * T * kBlock;       // k0, k1, k2, c0
    double* const s_M = s_kin + 4 * kBlock;             // the chunks' transfer matrix
    double* const s_f = s_M + 4 * kBlock;               // [L][3] chunk summaries
    double* const s_map = s_f + 3 * L * kBlock;         // [36 + 9 T] the adjoint recursion's map (gradient)
    double* const s_bnd = s_map + (kAdjMapRows + 9 * T) * kBlock;      // [kScanWaves][4] state at the head of a segment
#ifdef CUDE_SCAN_TIMING
    unsigned long long tk[8];
    tk[0] = __builtin_amdgcn_s_memrealtime();
#define TK(k) tk[k] = __builtin_amdgcn_s_memrealtime()
#else
#define TK(k)
#endif
    {
        // every wave walks its own eighth of each group of rows (a wave that skipped the others' rows one by one still
        // paid for their address arithmetic: 4.8 us for its 26 rows)
        if (wv < 4) lds_fetch((wv == 0 ? b.k0 : wv == 1 ? b.k1 : wv == 2 ? b.k2 : b.c0) + i, s_kin + wv * kBlock);
        else lds_fetch(a.hom_M + (int64_t)(wv - 4) * N + i, s_M + (wv - 4) * kBlock);
        for (int r = wv; r < 4 * T; r += kScanWaves) {
            const int o = r >> 2, q = r & 3;
            const double* g = q < 2 ? a.hom_obs + ((int64_t)o * 2 + q) * N + i
                                    : (q == 2 ? fsum + ((int64_t)(obs_step[o] / len) * (3 + T) + 3 + o) * N + i      // (the chunk whose steps hold o)
                                              : b.obs + (int64_t)o * N + i);
            lds_fetch(g, s_obs + r * kBlock);
        }
        for (int r = wv; r < 3 * L; r += kScanWaves) {
            const int c = r / 3, q = r - 3 * c;
            lds_fetch(fsum + ((int64_t)c * (3 + T) + q) * N + i, s_f + r * kBlock);
        }
        if (wts != nullptr) {
            const int rows = kAdjMapRows + 9 * T;
            for (int q = wv; q < rows; q += kScanWaves) lds_fetch(a.adj_map + (int64_t)q * N + i, s_map + q * kBlock);
        }
    }
    TK(1);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    TK(2);
    double sse = 0.0;
    bool failed = false;
    if (wv == 0) {
        const double k1 = lds_get(s_kin + kBlock, lane), k2 = lds_get(s_kin + 2 * kBlock, lane), c0 = lds_get(s_kin + 3 * kBlock, lane);
        double y1 = c0, y2 = (k2 / k1) * c0, y3 = 0.0;
        int oi = 0;
        int next_obs = obs_step[0];                     // step of the next observation to be met (ascending)
        const double M0 = lds_get(s_M, lane), M1 = lds_get(s_M + kBlock, lane), M2 = lds_get(s_M + 2 * kBlock, lane),
                     M3 = lds_get(s_M + 3 * kBlock, lane);
        constexpr int kGroup = 4;                       // chunks whose summaries are read from LDS together
        for (int cg = 0; cg < L; cg += kGroup) {
            double f[kGroup][3];
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const int c = cg + k < L ? cg + k : L - 1;
#pragma unroll
                for (int q = 0; q < 3; q++) f[k][q] = lds_get(s_f + (3 * c + q) * kBlock, lane);
            }
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const int c = cg + k;
                if (c < L) {
                    const int n1 = (c + 1) * len;
                    while (oi < T && next_obs < n1) {
                        const double hy = fma(lds_get(s_obs + (4 * oi + 0) * kBlock, lane), y1, lds_get(s_obs + (4 * oi + 1) * kBlock, lane) * y2);
                        const double r = (lds_get(s_obs + (4 * oi + 2) * kBlock, lane) + hy) - lds_get(s_obs + (4 * oi + 3) * kBlock, lane);
                        sse = fma(r, r, sse);
                        s_res[oi * kBlock + lane] = r;
                        oi++;
                        next_obs = oi < T ? obs_step[oi] : 0x7fffffff;
                    }
                    const double n1y = f[k][0] + fma(M0, y1, M1 * y2);
                    const double n2y = f[k][1] + fma(M2, y1, M3 * y2);
                    y1 = n1y; y2 = n2y;
                    y3 += f[k][2];
                }
            }
        }
        failed = !(fabs(sse) <= 1.79769313486231570815e308);
        TK(3);
        if (wts != nullptr) {
            // the recursion's state alone, through all steps but the first segment's: what every wave starts from
            stage_lds_sync();
            const int seg = (S + kScanWaves - 1) / kScanWaves;
            double st[4] = {0.0, 0.0, 0.0, 0.0};
            const double gscale = 2.0 * b.inv_n;
            int ob = T - 1;
            int ob_step = obs_step[ob];
            double Phi[16], R[4];
#pragma unroll
            for (int q = 0; q < 16; q++) Phi[q] = lds_get(s_map + q * kBlock, lane);
#pragma unroll
            for (int q = 0; q < 4; q++) R[q] = lds_get(s_map + (kAdjMapRows + 9 * ob + q) * kBlock, lane);
#pragma unroll 1
            for (int n = S - 1; n >= seg - 1; n--) {
                if ((n + 1) % seg == 0 || n == S - 1) {
                    const int k = n / seg;              // the segment whose first (highest) step is n
#pragma unroll
                    for (int r = 0; r < 4; r++) s_bnd[(4 * k + r) * kBlock + lane] = st[r];
                    if (n == seg - 1) break;
                }
                double o[4];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    o[r] = fma(Phi[4 * r + 3], st[3], fma(Phi[4 * r + 2], st[2], fma(Phi[4 * r + 1], st[1], Phi[4 * r] * st[0])));
                while (ob_step == n) {
                    const double g = gscale * s_res[ob * kBlock + lane];
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = fma(R[r], g, o[r]);
                    ob--;
                    ob_step = ob >= 0 ? obs_step[ob] : -1;
                    const int on = ob >= 0 ? ob : 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) R[q] = lds_get(s_map + (kAdjMapRows + 9 * on + q) * kBlock, lane);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) st[r] = o[r];
            }
        }
        TK(4);
        if (active) {
            if (b.sse != nullptr) b.sse[set * b.set_stride_cond + i] = sse;
            if (b.auc != nullptr && set == 0) b.auc[i] = y3;
            if (a.mh_fused) mh_accept_one(a.mh, i, mh_proposal(a.mh.p, a.mh_z, a.mh.key, a.mh_std, i), sse);
        }
    }
    if (wts != nullptr) {
        __syncthreads();
        const int seg = (S + kScanWaves - 1) / kScanWaves;
        const int n_hi = (wv + 1) * seg - 1 < S - 1 ? (wv + 1) * seg - 1 : S - 1, n_lo = wv * seg;
        if (n_lo <= n_hi) {
            double lam1 = lds_get_plain(s_bnd + (4 * wv + 0) * kBlock, lane), lam2 = lds_get_plain(s_bnd + (4 * wv + 1) * kBlock, lane),
                   kap1 = lds_get_plain(s_bnd + (4 * wv + 2) * kBlock, lane), kap2 = lds_get_plain(s_bnd + (4 * wv + 3) * kBlock, lane), w[5];
            const double gscale = 2.0 * b.inv_n;
            int oi = T - 1;
            while (oi >= 0 && obs_step[oi] > n_hi) oi--;        // the observations of later steps belong to other waves
            int oi_step = oi >= 0 ? obs_step[oi] : -1;
            const int o_first = oi >= 0 ? oi : 0;
            double Phi[16], Wm[20], R[9];
#pragma unroll
            for (int q = 0; q < 16; q++) Phi[q] = lds_get(s_map + q * kBlock, lane);
#pragma unroll
            for (int q = 0; q < 20; q++) Wm[q] = lds_get(s_map + (16 + q) * kBlock, lane);
#pragma unroll
            for (int q = 0; q < 9; q++) R[q] = lds_get(s_map + (kAdjMapRows + 9 * o_first + q) * kBlock, lane);
#pragma unroll 1
            for (int n = n_hi; n >= n_lo; n--) {
                const double in[4] = {lam1, lam2, kap1, kap2};
                double o[4];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    o[r] = fma(Phi[4 * r + 3], in[3], fma(Phi[4 * r + 2], in[2], fma(Phi[4 * r + 1], in[1], Phi[4 * r] * in[0])));
#pragma unroll
                for (int j = 0; j < 5; j++)
                    w[j] = fma(Wm[4 * j + 3], in[3], fma(Wm[4 * j + 2], in[2], fma(Wm[4 * j + 1], in[1], Wm[4 * j] * in[0])));
                while (oi_step == n) {
                    const double g = gscale * s_res[oi * kBlock + lane];
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = fma(R[r], g, o[r]);
#pragma unroll
                    for (int j = 0; j < 5; j++) w[j] = fma(R[4 + j], g, w[j]);
                    oi--;
                    oi_step = oi >= 0 ? obs_step[oi] : -1;
                    const int on = oi >= 0 ? oi : 0;
#pragma unroll
                    for (int q = 0; q < 9; q++) R[q] = lds_get(s_map + (kAdjMapRows + 9 * on + q) * kBlock, lane);
                }
                lam1 = o[0]; lam2 = o[1]; kap1 = o[2]; kap2 = o[3];
                if (active) {
#pragma unroll
                    for (int j = 0; j < 5; j++) wts[(int64_t)(5 * n + j) * N + i] = w[j];
                }
            }
        }
    }
    TK(5);
    if (wv != 0) return;
    if constexpr (SPEC) {
        // the wave's own stores of the candidates' SSEs (global memory) are visible to it behind the fence
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane < spb && gid < N) mh_spec_resolve(a.spec, gid);
        return;
    }
    {   // (sum SSE, failures) over the wave: block_reduce_store<2> of cude_device.h for one wave of a larger workgroup
        double* const out = b.partials + (set * gridDim.x + blockIdx.x) * (P + 2) + P;
        double* const out2 = a.final_host != nullptr ? a.final_host + 2 * (int64_t)blockIdx.x : nullptr;
        stage_lds_sync();
        smem[lane] = active ? sse : 0.0;
        smem[kBlockLanes + lane] = (active && failed) ? 1.0 : 0.0;
        stage_lds_sync();
        if (lane < 2) {
            double acc = 0.0;
#pragma unroll 8
            for (int l = 0; l < kBlockLanes; l++) acc += smem[lane * kBlockLanes + ((l + lane) & (kBlockLanes - 1))];
            out[lane] = acc;
            if (out2 != nullptr) out2[lane] = acc;
        }
    }
#ifdef CUDE_SCAN_TIMING
    TK(6);
    __builtin_amdgcn_s_waitcnt(0);
    TK(7);
    if (lane == 0 && blockIdx.x == 0 && a.mh_fused == 0)
        printf("scan L=%d grad=%d [10 ns]: issue %llu, arrive %llu, stitch %llu, state pass %llu, own segment %llu, reduce %llu, drain %llu\n",
               L, wts != nullptr, tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5], tk[7] - tk[6]);
#endif
#undef TK
}

// ---------------------------------------------------------------------------------- reverse sweep
// Stage-adjoint algebra of one step (J_f = A): consumes the adjoint (lam, kap) of (y_{n+1}, k_7), the
// observation seeds of the step, and returns the adjoint of (y_n, k_1) plus the 5 network weights.
// oi_step = obs_step[oi] (-1 when oi < 0), kept in a register by the caller: the step index is compared against it, and
// the table is read again only when an observation has been consumed (a scalar round trip per STEP before)
__device__ __forceinline__ void adj_step(const Kin& k, double h, double gscale, cptr_t obs_w, ciptr_t obs_step,
                                         const double* s_res, int lane, int n, int& oi, int& oi_step, double& lam1,
                                         double& lam2, double& kap1, double& kap2, double (&w)[5]) {
    double kb[7][2];
#pragma unroll
    for (int j = 0; j < 6; j++) { kb[j][0] = 0.0; kb[j][1] = 0.0; }
    kb[6][0] = kap1;
    kb[6][1] = kap2;
    double yb1 = 0.0, yb2 = 0.0;
    while (oi_step == n) {
        const double g = gscale * s_res[oi * kBlock + lane];
        yb1 += g;
        const double hg = h * g;
#pragma unroll
        for (int j = 0; j < 7; j++) kb[j][0] = fma(obs_w[oi * 7 + j], hg, kb[j][0]);
        oi--;
        oi_step = oi >= 0 ? obs_step[oi] : -1;
    }
    lam1 = fma(k.a11, kb[6][0], fma(k.a21, kb[6][1], lam1));
    lam2 = fma(k.a12, kb[6][0], fma(k.a22, kb[6][1], lam2));
    w[4] = kb[6][0];
    yb1 += lam1;
    yb2 += lam2;
    {
        const double hl1 = h * lam1, hl2 = h * lam2;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            kb[j][0] = fma(Tab::a(6, j), hl1, kb[j][0]);
            kb[j][1] = fma(Tab::a(6, j), hl2, kb[j][1]);
        }
    }
#pragma unroll
    for (int st = 5; st >= 1; st--) {
        const double Yb1 = fma(k.a11, kb[st][0], k.a21 * kb[st][1]);
        const double Yb2 = fma(k.a12, kb[st][0], k.a22 * kb[st][1]);
        if (st == 5) w[4] += kb[5][0];
        else w[st - 1] = kb[st][0];
        yb1 += Yb1;
        yb2 += Yb2;
        const double h1 = h * Yb1, h2 = h * Yb2;
#pragma unroll
        for (int j = 0; j < st; j++) {
            kb[j][0] = fma(Tab::a(st, j), h1, kb[j][0]);
            kb[j][1] = fma(Tab::a(st, j), h2, kb[j][1]);
        }
    }
    lam1 = yb1;
    lam2 = yb2;
    kap1 = kb[0][0];
    kap2 = kb[0][1];
}

// The step's adjoint algebra on unit inputs: the coefficients of Cpep2Args::adj_map, per subject, once per population.
// (adj_step itself is run, so the map IS the algebra's linearisation point by point; sums over its entries round
// differently from the stage-by-stage form: ~1e-16 per step.)
__global__ __launch_bounds__(kBlock) void cpep2_adjmap_kernel(Cpep2Args a, double* __restrict__ map) {
    extern __shared__ double s_res[];           // [T][kBlock]: unit residual of one observation, zeros for the others
    const CpepArgs& b = a.base;
    const int lane = threadIdx.x;
    const int64_t gid = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gid < b.N;
    const int64_t i = active ? gid : b.N - 1, N = b.N;
    cptr_t obs_w = as_const(b.obs_w);
    ciptr_t obs_step = as_const(b.obs_step);
    const int T = b.T;
    const double k0 = b.k0[i], k1 = b.k1[i], k2 = b.k2[i];
    const Kin kin{-(k0 + k2), k1, k2, -k1, 0.0};
    for (int c = 0; c < 4; c++) {
        double lam1 = c == 0, lam2 = c == 1, kap1 = c == 2, kap2 = c == 3, w[5];
        int oi = -1, oi_step = -1;
        adj_step(kin, b.h, 1.0, obs_w, obs_step, s_res, lane, 0, oi, oi_step, lam1, lam2, kap1, kap2, w);
        if (active) {
            map[(int64_t)(0 + c) * N + i] = lam1;
            map[(int64_t)(4 + c) * N + i] = lam2;
            map[(int64_t)(8 + c) * N + i] = kap1;
            map[(int64_t)(12 + c) * N + i] = kap2;
            for (int j = 0; j < 5; j++) map[(int64_t)(16 + 4 * j + c) * N + i] = w[j];
        }
    }
    for (int o = 0; o < T; o++) {
        for (int q = 0; q < T; q++) s_res[q * kBlock + lane] = q == o ? 1.0 : 0.0;
        const int n = obs_step[o];
        int oi = o;
        while (oi + 1 < T && obs_step[oi + 1] == n) oi++;       // the step's last observation: the recursion meets it first
        int oi_step = n;
        double lam1 = 0.0, lam2 = 0.0, kap1 = 0.0, kap2 = 0.0, w[5];
        adj_step(kin, b.h, 1.0, obs_w, obs_step, s_res, lane, n, oi, oi_step, lam1, lam2, kap1, kap2, w);
        if (active) {
            double* r = map + (int64_t)(kAdjMapRows + 9 * o) * N + i;
            r[0] = lam1; r[N] = lam2; r[2 * N] = kap1; r[3 * N] = kap2;
            for (int j = 0; j < 5; j++) r[(int64_t)(4 + j) * N] = w[j];
        }
    }
}

template <int NIN, int W, int D>
__global__ __launch_bounds__(kBlock) void cpep2_rev_kernel(Cpep2Args a) {
    using Net = CpepNet<NIN, W, D>;
    constexpr int P = Net::P;
    constexpr int NC = NIN - 1;
    extern __shared__ double smem[];
    double* s_red = smem;                       // [kRedRows][kBlock] after the sweep ...
    double* s_tab = smem;                       // ... [5][W][kBlock] layer-1 factor table during it (Net::HAS_TAB)
    const CpepArgs& b = a.base;
    const int lane = threadIdx.x;
    const int64_t gid = ((int64_t)blockIdx.x + b.blk0) * kBlock + lane;      // (blk0: mixed launch, see CpepArgs)
    const bool active = gid < b.N;
    const int64_t i = active ? gid : b.N - 1;
    const int64_t N = b.N;
    const int c_idx = blockIdx.y;
    const int64_t set = blockIdx.z;             // parameter set (see cpep2_fwd_kernel)
    cptr_t p = as_const(b.nn + set * b.set_stride_nn);
    const double* const wts = a.wts + set * ((int64_t)5 * b.S * b.N);
    cptr_t phi = as_const(b.phi);
    ciptr_t seg = as_const(b.seg);
    ciptr_t cs = as_const(a.chunk_start);
    ciptr_t stepk = as_const(b.stepk);
    cptr_t stepd = as_const(b.stepd);
    const int n0 = cs[c_idx], n1 = cs[c_idx + 1];

    double cst[NC];
    cst[0] = exp(b.cond[set * b.set_stride_cond + i]);
    if (NC > 1) cst[1] = b.age[i];
    double c[W];
    Net::first_layer_offset(p, cst, c);
    double acc[Net::NACC];
#pragma unroll
    for (int q = 0; q < Net::NACC; q++) acc[q] = 0.0;
    double dxdummy[1] = {0.0};
    double wtot = 0.0;
    int cur_seg = -1;
    double g_lo = 0.0, g_d = 0.0;
    // layer-1 exponent table, exactly as the reverse sweep of cpep_kernel (cude_cpep.hip): anchor = exp(2 z_j) at the
    // END of the step, factors reach back from there; a run that continues from a later chunk is re-anchored here
#ifndef CUDE_NO_TAB2
    constexpr bool kTab = Net::HAS_TAB;
#else
    constexpr bool kTab = false;
#endif
    typename Net::Exps A, E1;
    int kind = 0, s = 4, n = n1 - 1;
    bool run_ok = false, have_anchor = false;
    // own stage times in reverse order; e = 5 n0 - 1 stands for the baseline with weight -sum(own w)
    if constexpr (Net::USES_TANH) tanh_tab_init(lane, !Net::LDS_BIAS);
    Net::bias_init(b.nn + set * b.set_stride_nn, lane);
#pragma unroll 1
    for (int e = 5 * n1 - 1; e >= 5 * n0 - 1; e--) {
        const bool own = e >= 5 * n0;
        double xv = 0.0, wv;
        bool tab = false;
        if (own) {
            if constexpr (kTab) {
                if (s == 4) {                    // first evaluation (in reverse order) of step n
                    kind = stepk[3 * n + 1];     // 0 = straddles a knot, 1 = a run starts here (in reverse), 2 = continues
                    if (kind == 0) have_anchor = false;
                    if (kind == 2 && !have_anchor) kind = 1;      // the run started in a later chunk: anchor it here
                    if (kind == 2 && !run_ok) kind = 0;
                    if (kind == 1) {
                        const int s0 = stepk[3 * n + 2];
                        const double lo = b.dG[(int64_t)s0 * N + i];
                        const double d = b.dG[(int64_t)(s0 + 1) * N + i] - lo;
                        run_ok = !__any(!Net::tab_safe(p, c, lo, d, stepd[3 * n + 2]));
                        have_anchor = true;
                        if (run_ok) {
                            const double cr[5] = {Tab::c(1) - 1.0, Tab::c(2) - 1.0, Tab::c(3) - 1.0, Tab::c(4) - 1.0, -1.0};
                            Net::tab_build(p, d * stepd[3 * n + 2], cr, s_tab, lane);
                            Net::tab_anchor(p, c, fma(stepd[3 * n + 1], d, lo), A);
                        } else {
                            kind = 0;
                        }
                    } else if (kind == 2) {
#pragma unroll
                        for (int j = 0; j < W; j++) A.v[j] *= s_tab[(4 * W + j) * kBlock + lane];
                    }
                }
            }
            const int sg = seg[e];
            if (sg != cur_seg) {
                cur_seg = sg;
                g_lo = b.dG[(int64_t)sg * N + i];
                g_d = b.dG[(int64_t)(sg + 1) * N + i] - g_lo;
            }
            xv = fma(phi[e], g_d, g_lo);
            wv = wts[(int64_t)e * N + i];             // needed at the end of the evaluation only: the load stays in flight
            if constexpr (kTab) {
                tab = kind != 0;
                if (tab) {
                    const int sr = s < 4 ? s : 0;           // reads grouped and unconditional, as in cpep_kernel
                    double f[W];
#pragma unroll
                    for (int j = 0; j < W; j++) f[j] = s_tab[(sr * W + j) * kBlock + lane];
#pragma unroll
                    for (int j = 0; j < W; j++) asm volatile("" : "+v"(f[j]));
#pragma unroll
                    for (int j = 0; j < W; j++) E1.v[j] = s < 4 ? A.v[j] * f[j] : A.v[j];   // stage 5 sits at the anchor time
                }
            }
            if (s == 0) { s = 4; n--; } else s--;
        } else {
            wv = -wtot;
        }
        const double x[1] = {xv};
        Net::template eval_grad<false>(p, c, x, wv, acc, dxdummy, tab, &E1);
        if (own) wtot += wv;
    }
    if (active) a.g_cond_part[(set * a.L + c_idx) * N + i] = Net::grad_cond(p, acc, cst);
    double* out = a.partials2 + ((set * a.L + c_idx) * gridDim.x + blockIdx.x) * P;
    __syncthreads();                            // the table rows become the reduction buffer
    {
        // P columns only (the loss / failure columns belong to the scan kernel): expand 16 rows at a time into LDS
        const double keep = active ? 1.0 : 0.0;
#pragma unroll
        for (int c0 = 0; c0 < P; c0 += kRedRows) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kRedRows; r++)
                if (c0 + r < P) s_red[r * kBlockLanes + lane] = Net::grad_elem(c0 + r, acc, cst) * keep;
            __syncthreads();
            if (lane < kRedRows && c0 + lane < P) {
                double v = 0.0;
#pragma unroll 8
                for (int l = 0; l < kBlockLanes; l++) v += s_red[lane * kBlockLanes + ((l + lane) & (kBlockLanes - 1))];
                out[c0 + lane] = v;
            }
        }
    }
}

// g_cond[i] = sum_c part[c][i]
__global__ void cpep2_sum_chunks_kernel(const double* __restrict__ part, int L, int64_t N, double* __restrict__ out,
                                        int64_t out_set_stride, int64_t i0) {
    const int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    part += (int64_t)blockIdx.y * L * N;        // parameter set
    double s = 0.0;
    for (int c = 0; c < L; c++) s += part[(int64_t)c * N + i];
    out[(int64_t)blockIdx.y * out_set_stride + i] = s;
}

// ---------------------------------------------------------------------------------- dispatch
constexpr size_t kScanBulkMaxLds = 156 * 1024;      // (+ nothing static in the scan: 160 KB per compute unit)
// dynamic LDS above 64 KB has to be allowed per kernel, once
static hipError_t allow_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::set<const void*> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(kernel)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScanBulkMaxLds);
    if (e == hipSuccess) done.insert(kernel);
    return e;
}

template <int NIN, int W, int D>
static hipError_t run_shape(int n_state, bool grad, const Cpep2Args& a, hipStream_t s) {
    using Net = CpepNet<NIN, W, D>;
    const int64_t nblocks = (a.base.N + kBlock - 1) / kBlock - a.base.blk0;             // blocks [blk0, end)
    const unsigned n_sets = a.base.n_sets > 0 ? (unsigned)a.base.n_sets : 1u;
    if (nblocks < 1 || (a.base.blk0 > 0 && n_sets > 1)) return hipErrorInvalidValue;
    const dim3 grid2((unsigned)nblocks, (unsigned)a.L, n_sets);
#ifdef CUDE_ABLATION
    static const bool no_vw = getenv("CUDE_NO_VW2") != nullptr;
#else
    constexpr bool no_vw = false;
#endif
    // the weights move into VGPRs (VWR) while the grid fits two waves per SIMD at once
    const bool vwr = Net::HAS_VW_SMALL && !no_vw && nblocks * a.L * n_sets <= 2 * 1024;
    const size_t lds_b = sizeof(double) * (size_t)a.base.T * kBlock;
    if (vwr) {
        if constexpr (Net::HAS_VW_SMALL) {
            if (n_state == 3) hipLaunchKernelGGL((cpep2_fwd_kernel<NIN, W, D, 3, true>), grid2, dim3(kBlock), lds_b, s, a);
            else hipLaunchKernelGGL((cpep2_fwd_kernel<NIN, W, D, 2, true>), grid2, dim3(kBlock), lds_b, s, a);
        }
    } else if (n_state == 3) {
        hipLaunchKernelGGL((cpep2_fwd_kernel<NIN, W, D, 3>), grid2, dim3(kBlock), lds_b, s, a);
    } else {
        hipLaunchKernelGGL((cpep2_fwd_kernel<NIN, W, D, 2>), grid2, dim3(kBlock), lds_b, s, a);
    }
    Cpep2Args as = a;
    if (!grad) as.wts = nullptr;
    // launches of at most one scan workgroup per compute unit: every row through LDS up front (cpep2_scan_bulk_kernel)
    const size_t lds_bulk = sizeof(double) * (size_t)scan_bulk_rows(a.base.T, a.L, as.wts != nullptr) * kBlock;
    const int64_t scan_blocks = a.spec_slots > 0 ? (a.base.N + kBlock / a.spec_slots - 1) / (kBlock / a.spec_slots) : nblocks * n_sets;
    const bool bulk = a.scan_bulk_blocks > 0 && scan_blocks <= a.scan_bulk_blocks && a.base.blk0 == 0 && a.base.S % a.L == 0 &&
                      (as.wts == nullptr || a.adj_map != nullptr) && lds_bulk <= kScanBulkMaxLds;
    if (a.spec_slots > 0) {
        if (grad || a.spec_slots > kBlock || (int)n_sets >= a.spec_slots || (a.spec_slots & (a.spec_slots - 1)))
            return hipErrorInvalidValue;
        const int spb = kBlock / a.spec_slots;
        if (bulk) {
            hipError_t e = allow_lds((const void*)cpep2_scan_bulk_kernel<true>, lds_bulk);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((cpep2_scan_bulk_kernel<true>), dim3((unsigned)((a.base.N + spb - 1) / spb)),
                               dim3(kScanWaves * kBlock), lds_bulk, s, as, (int)Net::P);
            return hipGetLastError();
        }
        hipLaunchKernelGGL((cpep2_scan_kernel<true>), dim3((unsigned)((a.base.N + spb - 1) / spb)), dim3(kBlock),
                           sizeof(double) * (size_t)(kRedRows + 5 * a.base.T) * kBlock, s, as, (int)Net::P);
        return hipGetLastError();
    }
    if (bulk) {
        hipError_t e = allow_lds((const void*)cpep2_scan_bulk_kernel<false>, lds_bulk);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((cpep2_scan_bulk_kernel<false>), dim3((unsigned)nblocks, n_sets), dim3(kScanWaves * kBlock), lds_bulk, s, as,
                           (int)Net::P);
    } else {
        hipLaunchKernelGGL((cpep2_scan_kernel<false>), dim3((unsigned)nblocks, n_sets), dim3(kBlock),
                           sizeof(double) * (size_t)(kRedRows + 5 * a.base.T) * kBlock, s, as, (int)Net::P);
    }
    if (!grad) return hipGetLastError();
    constexpr int TABROWS = Net::HAS_TAB ? 5 * W : 0;
    const size_t lds_r = sizeof(double) * (size_t)(TABROWS > kRedRows ? TABROWS : kRedRows) * kBlock;
    hipLaunchKernelGGL((cpep2_rev_kernel<NIN, W, D>), grid2, dim3(kBlock), lds_r, s, a);
    if (a.defer_chunk_sum && a.base.blk0 == 0) return hipGetLastError();
    const int bs = 256;
    const int64_t i0 = a.base.blk0 * kBlock;
    hipLaunchKernelGGL(cpep2_sum_chunks_kernel, dim3((unsigned)((a.base.N - i0 + bs - 1) / bs), n_sets), dim3(bs), 0, s,
                       a.g_cond_part, a.L, a.base.N, a.base.g_cond, a.base.set_stride_cond, i0);
    return hipGetLastError();
}

#define CUDE_CPEP2_SHAPES(X) X(2, 4, 2) X(2, 6, 2) X(3, 4, 2) X(2, 8, 2) X(2, 4, 3) X(2, 3, 2) X(2, 5, 2) X(2, 7, 2) X(3, 6, 2) X(2, 4, 1) X(2, 6, 1) X(2, 6, 3) X(2, 8, 1) X(2, 8, 3) X(3, 8, 2) X(2, 3, 1) X(2, 5, 1) X(2, 7, 1) X(2, 3, 3) X(2, 5, 3) X(2, 7, 3) X(3, 4, 1) X(3, 6, 1) X(3, 4, 3)

bool cpep2_shape_supported(const NetShape& net, int n_state) {
    if (net.general() || net.generic()) return false;         // other activation functions / shapes: the one-lane kernels only
    if (n_state != 2 && n_state != 3) return false;
#define X(NIN, W, D) if (net.nin == NIN && net.width == W && net.depth == D) return true;
    CUDE_CPEP2_SHAPES(X)
#undef X
    return false;
}

// resident waves per CU of the reverse kernel (register / LDS limited): the slot count of the chunk-count selector
template <int NIN, int W, int D>
static int rev_occupancy() {
    int n = 0;
    constexpr int TABROWS = CpepNet<NIN, W, D>::HAS_TAB ? 5 * W : 0;
    const size_t lds_r = sizeof(double) * (size_t)(TABROWS > kRedRows ? TABROWS : kRedRows) * kBlock;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cpep2_rev_kernel<NIN, W, D>, kBlock, lds_r) != hipSuccess) return 0;
    return n;
}
int cpep2_rev_waves_per_cu(const NetShape& net) {
#define X(NIN, W, D) if (net.nin == NIN && net.width == W && net.depth == D) return rev_occupancy<NIN, W, D>();
    CUDE_CPEP2_SHAPES(X)
#undef X
    return 0;
}

// once per context, outside any stream capture: the eight-wave scan may ask for more than 64 KB of LDS
hipError_t cpep2_prepare() {
    hipError_t e = allow_lds((const void*)cpep2_scan_bulk_kernel<false>, kScanBulkMaxLds);
    if (e != hipSuccess) return e;
    return allow_lds((const void*)cpep2_scan_bulk_kernel<true>, kScanBulkMaxLds);
}

hipError_t launch_cpep2_homog(const Cpep2Args& a, hipStream_t s) {
    const int64_t nblocks = (a.base.N + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(cpep2_homog_kernel, dim3((unsigned)nblocks), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_cpep2_adjmap(const Cpep2Args& a, double* adj_map, hipStream_t s) {
    const int64_t nblocks = (a.base.N + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(cpep2_adjmap_kernel, dim3((unsigned)nblocks), dim3(kBlock), sizeof(double) * (size_t)a.base.T * kBlock, s, a,
                       adj_map);
    return hipGetLastError();
}

hipError_t launch_cpep2(const NetShape& net, int n_state, bool grad, const Cpep2Args& a, hipStream_t s) {
#define X(NIN, W, D) if (net.nin == NIN && net.width == W && net.depth == D) return run_shape<NIN, W, D>(n_state, grad, a, s);
    CUDE_CPEP2_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cude
