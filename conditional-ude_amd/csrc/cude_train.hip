// libcude_hip.so -- restarts trained side by side with their optimiser state resident on the device: the loop
//   for p in initials[selected]: _optimize(optfunc, p, ADAM(lr), n_adam, n_lbfgs)
// of `train` (src/parameter-estimation.jl:340-386, `_optimize` :170-183) and of `fit_suppression_model`
// (suppression/src/suppression_model.jl:132-177) for all restarts at once.  Per iteration the host queues a handful of
// launches and moves O(restarts) bytes: the restarts' parameters, their gradients, the Adam moments and the L-BFGS
// vectors (iterate, gradient, direction, ten (s, y) pairs) never leave the GPU.
//   Adam stage     eval_sets_device -> finish_sets_kernel (L2 term, loss, liveness, trace) -> adam_sets_kernel; no
//                  synchronisation at all between iterations.
//   L-BFGS stage   per round: eval_sets_device at the restarts' trial points -> finish_sets_kernel -> lbfgs_feed_kernel
//                  (one workgroup per restart: the whole of cude::Lbfgs::feed -- line-search decision, history update,
//                  two-loop recursion, next direction, next trial point x + alpha d) -> the restarts' phases come back
//                  (a few hundred bytes) so that finished restarts drop out of the next evaluation.
// The element-wise arithmetic is that of cude_optim.h (compiled without contraction); inner products are summed by a
// fixed tree per workgroup instead of left to right, which is the only difference from the host statement.
// A sharded population keeps the L-BFGS vectors on the host (cude::Lbfgs with its reducer: the inner products'
// conditional parts are summed over the ranks); its Adam stage is the device-resident one.
#include "cude_ctx.h"

namespace cude {

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double pick_max(double a, double b) { return a < b ? b : a; }     // std::max: a NaN second operand is ignored
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = pick_max(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace

__global__ __launch_bounds__(64) void finish_sets_kernel(FinishSetsArgs a) {
    const int k = blockIdx.x, lane = threadIdx.x, P = a.P;
    double* r = a.out + (int64_t)k * (P + 2);
    const double* w = a.nn + (int64_t)k * a.stride_nn;
    double* g = a.g_dst != nullptr ? a.g_dst + (int64_t)k * a.g_stride : r;
    double sum = r[P];
    const double nf = r[P + 1];
    if (a.lambda != 0.0) {
        // as l2_term_kernel: 64 strided partial sums, xor butterfly, one fma each for the loss sum and the gradient entries
        double ss = 0.0;
        for (int q = lane; q < P; q += 64) {
            ss = fma(w[q], w[q], ss);
            g[q] = fma(2.0 * a.lambda * (a.mask != nullptr ? a.mask[q] : 1.0), w[q], r[q]);
        }
        ss = wsum(ss);
        sum = fma(a.lambda * a.n_global, ss, sum);
    } else if (a.g_dst != nullptr) {
        for (int q = lane; q < P; q += 64) g[q] = r[q];
    }
    if (lane != 0) return;
    const bool bad = nf > 0.0 || !(fabs(sum) <= 1.79769313486231570815e308);
    const double loss = bad ? __builtin_inf() : sum / a.n_global;
    a.f[k] = loss;
    if (a.alive != nullptr) {
        if (bad) a.alive[k] = 0;
        if (a.trace != nullptr && a.alive[k] != 0) a.trace[(int64_t)k * a.trace_len + a.trace_pos] = loss;
    }
}

hipError_t launch_finish_sets(const FinishSetsArgs& a, int n_sets, hipStream_t s) {
    hipLaunchKernelGGL(finish_sets_kernel, dim3((unsigned)n_sets), dim3(64), 0, s, a);
    return hipGetLastError();
}

__global__ void adam_sets_kernel(AdamSetsArgs a) {
#pragma clang fp contract(off)
    const int k = blockIdx.y;
    if (a.alive[k] == 0) return;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double *x, *m, *v;
    double g;
    if (idx < a.N) {
        const int64_t o = (int64_t)k * a.N + idx;
        x = a.cond + o; m = a.m_cond + o; v = a.v_cond + o; g = a.g_cond[o];
    } else if (idx < a.N + a.P) {
        const int64_t q = idx - a.N, o = (int64_t)k * a.P + q;
        x = a.nn + o; m = a.m_nn + o; v = a.v_nn + o; g = a.out[(int64_t)k * (a.P + 2) + q];
    } else {
        return;
    }
    const double mm = a.b1 * *m + (1.0 - a.b1) * g;
    const double vv = a.b2 * *v + (1.0 - a.b2) * g * g;
    *m = mm;
    *v = vv;
    *x -= a.lr * (mm / a.c1) / (sqrt(vv / a.c2) + a.eps);
}

hipError_t launch_adam_sets(const AdamSetsArgs& a, int n_sets, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(adam_sets_kernel, dim3((unsigned)((a.N + a.P + bs - 1) / bs), (unsigned)n_sets), dim3(bs), 0, s, a);
    return hipGetLastError();
}

// trial[slot] = x + alpha d of the slot's restart (its iterate itself before the first evaluation)
__global__ void lbfgs_trial_kernel(LbfgsArgs a) {
#pragma clang fp contract(off)
    const int slot = blockIdx.y, r = a.act[slot];
    const LbfgsState& st = a.state[r];
    const double* X = a.X + (int64_t)r * a.n;
    const double* D = a.D + (int64_t)r * a.n;
    double* t = a.trial + (int64_t)slot * a.n;
    const bool step = st.phase == kLbfgsFinite || st.phase == kLbfgsArmijo;
    const double alpha = st.a2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x)
        t[i] = step ? X[i] + alpha * D[i] : X[i];
}

hipError_t launch_lbfgs_trial(const LbfgsArgs& a, int n_active, hipStream_t s) {
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (a.n + 1023) / 1024));
    hipLaunchKernelGGL(lbfgs_trial_kernel, dim3(gx, (unsigned)n_active), dim3(256), 0, s, a);
    return hipGetLastError();
}

// cude::Lbfgs::feed for the restart of workgroup `slot`: every thread carries the restart's scalar state (the same
// values in every thread: block reductions hand every thread the same sum), thread t owns the vector elements
// t, t + TPB, ... in every pass, so no pass reads what another thread wrote.
template <int TPB>
struct BlockRed {
    double* sh;     // [3][TPB / 64]
    __device__ __forceinline__ void sum3(double& a, double& b, double& c) const {
        constexpr int W = TPB / 64;
        a = wsum(a); b = wsum(b); c = wsum(c);
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[W + (threadIdx.x >> 6)] = b; sh[2 * W + (threadIdx.x >> 6)] = c; }
        __syncthreads();
        double ta = 0.0, tb = 0.0, tc = 0.0;
        for (int w = 0; w < W; w++) { ta += sh[w]; tb += sh[W + w]; tc += sh[2 * W + w]; }
        __syncthreads();
        a = ta; b = tb; c = tc;
    }
    __device__ __forceinline__ double sum(double a) const {
        double b = 0.0, c = 0.0;
        sum3(a, b, c);
        return a;
    }
    // two maxima and a sum
    __device__ __forceinline__ void max2sum(double& a, double& b, double& c) const {
        constexpr int W = TPB / 64;
        a = wmax(a); b = wmax(b); c = wsum(c);
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[W + (threadIdx.x >> 6)] = b; sh[2 * W + (threadIdx.x >> 6)] = c; }
        __syncthreads();
        double ta = 0.0, tb = 0.0, tc = 0.0;
        for (int w = 0; w < W; w++) { ta = pick_max(ta, sh[w]); tb = pick_max(tb, sh[W + w]); tc += sh[2 * W + w]; }
        __syncthreads();
        a = ta; b = tb; c = tc;
    }
};

template <int TPB>
__global__ __launch_bounds__(TPB) void lbfgs_feed_kernel(LbfgsArgs a) {
#pragma clang fp contract(off)
    __shared__ double sh[3 * (TPB / 64)];
    extern __shared__ double lds[];     // a.lds_doubles of them (launch_lbfgs_feed)
    const BlockRed<TPB> red{sh};
    constexpr int m = kLbfgsM, kMaxLs = 1000, kMaxFinite = 52;
    constexpr double kC1 = 1e-4, kRhoHi = 0.5, kRhoLo = 0.1;
    const int slot = blockIdx.x, r = a.act[slot], tid = threadIdx.x;
    LbfgsState st = a.state[r];
    if (st.phase == kLbfgsDone) return;
    __syncthreads();            // every wave has the state before thread 0 can store the new one (a rejected trial step
                                // reaches the end of this kernel without another barrier)
    const int64_t n = a.n;
    double* X = a.X + (int64_t)r * n;
    double* G = a.G + (int64_t)r * n;
    double* D = a.D + (int64_t)r * n;
    double* Sr = a.S + (int64_t)r * m * n;
    double* Yr = a.Y + (int64_t)r * m * n;
    const double* tr = a.trial + (int64_t)slot * n;
    const double* gn = a.g_trial + (int64_t)slot * n;
    const double f = a.f_trial[slot];
    bool start = false;         // top of the main loop: stop, or the next direction
    if (st.phase == kLbfgsFirst) {
        st.f = f;
        st.calls = 1;
        double gmax = 0.0, u0 = 0.0, u1 = 0.0;
        for (int64_t i = tid; i < n; i += TPB) {
            const double g = gn[i];
            G[i] = g;
            gmax = pick_max(gmax, fabs(g));
        }
        red.max2sum(gmax, u0, u1);
        st.converged = (fabs(st.f) <= 1.79769313486231570815e308 && gmax <= st.g_tol) ? 1 : 0;
        start = true;
    } else {
        st.n_eval++;
        bool armijo = true;
        if (st.phase == kLbfgsFinite) {             // first trial of a line search: halve until finite
            if (!(fabs(f) <= 1.79769313486231570815e308) && st.ls_it < kMaxFinite) {
                st.a1 = st.a2;
                st.a2 *= 0.5;
                st.ls_it++;
                armijo = false;
            } else {
                st.phi1 = st.f0;
                st.ls_it = 0;
                st.phase = kLbfgsArmijo;
            }
        }
        if (armijo && f > st.f0 + kC1 * st.a2 * st.dphi0) {       // (NaN compares false: accepted, the main loop then stops)
            st.ls_it++;
            if (st.ls_it > kMaxLs) {
                // LineSearchException: Optim takes the last trial step and stops (cude_optim.h, armijo)
                st.calls += st.n_eval;
                for (int64_t i = tid; i < n; i += TPB) { X[i] = tr[i]; G[i] = gn[i]; }
                st.f = f;
                st.it++;
                st.line_search_failed = 1;
                st.phase = kLbfgsDone;
            } else {
                const double f0 = st.f0, d0 = st.dphi0, a1 = st.a1, a2 = st.a2;
                double a_tmp;
                if (st.ls_it == 1) {
                    a_tmp = -(d0 * a2 * a2) / (2.0 * (f - f0 - d0 * a2));
                } else {
                    const double div = 1.0 / (a1 * a1 * a2 * a2 * (a2 - a1));
                    const double A = (a1 * a1 * (f - f0 - d0 * a2) - a2 * a2 * (st.phi1 - f0 - d0 * a1)) * div;
                    const double B = (-a1 * a1 * a1 * (f - f0 - d0 * a2) + a2 * a2 * a2 * (st.phi1 - f0 - d0 * a1)) * div;
                    if (fabs(A) <= 2.220446049250313e-16) {
                        a_tmp = d0 / (2.0 * B);
                    } else {
                        const double disc0 = B * B - 3.0 * A * d0;
                        const double disc = disc0 < 0.0 ? 0.0 : disc0;          // std::max(disc0, 0.0)
                        a_tmp = (-B + sqrt(disc)) / (3.0 * A);
                    }
                }
                st.a1 = a2;
                const double hi = a2 * kRhoHi, lo = a2 * kRhoLo;
                double a_new = (a_tmp != a_tmp) ? hi : (hi < a_tmp ? hi : a_tmp);   // NaNMath.min
                a_new = a_new < lo ? lo : a_new;                                     // std::max(a_new, lo)
                st.a2 = a_new;
                st.phi1 = f;
            }
        } else if (armijo) {
            // accepted: x <- x + alpha d, (s, y) into the ring, convergence tests
            st.calls += st.n_eval;
            const int hslot = (st.pseudo - 1) % m;
            double* Ss = Sr + (int64_t)hslot * n;
            double* Ys = Yr + (int64_t)hslot * n;
            const double alpha = st.a2;
            double moved = 0.0, gmax = 0.0, sy = 0.0;
            for (int64_t i = tid; i < n; i += TPB) {
                const double s = alpha * D[i], g1 = gn[i];
                const double y = g1 - G[i];
                const double x0 = X[i], xn = x0 + s;
                moved = pick_max(moved, fabs(xn - x0));
                gmax = pick_max(gmax, fabs(g1));
                sy += s * y;
                X[i] = xn; G[i] = g1; Ss[i] = s; Ys[i] = y;
            }
            red.max2sum(moved, gmax, sy);
            const double f_prev = st.f;
            st.f = f;
            if (a.trace != nullptr && st.accepted < st.maxiters && tid == 0)      // where Optim's callback fires
                a.trace[(int64_t)a.owner[r] * a.trace_len + a.trace_off + st.accepted] = f;
            st.it++;
            st.accepted++;
            st.f_flat = (fabs(f_prev - st.f) == 0.0) ? st.f_flat + 1 : 0;                 // successive_f_tol = 1
            st.converged = (moved == 0.0 || gmax <= st.g_tol || st.f_flat > 1) ? 1 : 0;
            if (!st.converged) {                        // update_h!
                const double rho = 1.0 / sy;
                if (fabs(rho) > 1.79769313486231570815e308) st.pseudo = 0;        // isinf
                else st.rho[hslot] = rho;
            }
            start = true;
        }
    }
    if (start) {
        if (!(st.it < st.maxiters) || st.converged || !(fabs(st.f) <= 1.79769313486231570815e308)) {
            st.phase = kLbfgsDone;
        } else {
            st.pseudo++;
            const int upper = st.pseudo - 1, lower = st.pseudo - m > 1 ? st.pseudo - m : 1;
            double dphi0 = 0.0;
            if (upper >= lower) {
                // two-loop recursion over the pairs lower..upper on q; every pass also forms the inner product the next
                // step needs (same products, same element order as a separate pass would take).  A small problem (the
                // reference's 57 + 37 unknowns) is a chain of ~25 dependent passes over a few hundred bytes each: there
                // the pairs, g and q are staged in LDS by one batch of independent loads, so that the chain pays the
                // memory latency once instead of once per pass (26 -> 8 us per round at n = 94).
                double alpha[m];
                const bool staged = a.lds_doubles >= (int64_t)(2 * m + 2) * n;
                double* q = staged ? lds + (int64_t)(2 * m) * n : D;
                const double* gq = staged ? lds + (int64_t)(2 * m + 1) * n : G;
                const double* Sq = staged ? lds : Sr;
                const double* Yq = staged ? lds + (int64_t)m * n : Yr;
                if (staged) {
                    for (int k = lower; k <= upper; k++) {
                        const int64_t o = (int64_t)((k - 1) % m) * n;
                        for (int64_t i = tid; i < n; i += TPB) {
                            lds[o + i] = Sr[o + i];
                            lds[(int64_t)m * n + o + i] = Yr[o + i];
                        }
                    }
                    for (int64_t i = tid; i < n; i += TPB) lds[(int64_t)(2 * m + 1) * n + i] = G[i];
                }
                const double* su = Sq + (int64_t)((upper - 1) % m) * n;
                const double* yu = Yq + (int64_t)((upper - 1) % m) * n;
                double acc = 0.0, sy = 0.0, yy = 0.0;
                for (int64_t i = tid; i < n; i += TPB) {
                    const double g = gq[i], s = su[i], y = yu[i];
                    q[i] = g;
                    acc += s * g;
                    sy += s * y;
                    yy += y * y;
                }
                red.sum3(acc, sy, yy);
                double dotv = acc;
                for (int k = upper; k >= lower; k--) {
                    const double al = st.rho[(k - 1) % m] * dotv;
                    alpha[k - lower] = al;
                    const double* yk = Yq + (int64_t)((k - 1) % m) * n;
                    if (k > lower) {
                        const double* sn = Sq + (int64_t)((k - 2) % m) * n;
                        acc = 0.0;
                        for (int64_t i = tid; i < n; i += TPB) {
                            const double v = q[i] - al * yk[i];
                            q[i] = v;
                            acc += sn[i] * v;
                        }
                        dotv = red.sum(acc);
                    } else {
                        for (int64_t i = tid; i < n; i += TPB) q[i] = q[i] - al * yk[i];
                    }
                }
                const double sc = sy / yy;                  // scaleinvH0
                const double* yl = Yq + (int64_t)((lower - 1) % m) * n;
                acc = 0.0;
                for (int64_t i = tid; i < n; i += TPB) {
                    const double v = q[i] * sc;
                    q[i] = v;
                    acc += yl[i] * v;
                }
                dotv = red.sum(acc);
                for (int k = lower; k <= upper; k++) {
                    const double b = st.rho[(k - 1) % m] * dotv;
                    const double coef = alpha[k - lower] - b;
                    const double* sk = Sq + (int64_t)((k - 1) % m) * n;
                    acc = 0.0;
                    if (k < upper) {
                        const double* yn = Yq + (int64_t)(k % m) * n;
                        for (int64_t i = tid; i < n; i += TPB) {
                            const double v = q[i] + coef * sk[i];
                            q[i] = v;
                            acc += yn[i] * v;
                        }
                        dotv = red.sum(acc);
                    } else {
                        for (int64_t i = tid; i < n; i += TPB) {
                            const double d = -(q[i] + coef * sk[i]);
                            D[i] = d;
                            acc += gq[i] * d;
                        }
                        dphi0 = red.sum(acc);
                    }
                }
            } else {
                double acc = 0.0;
                for (int64_t i = tid; i < n; i += TPB) {
                    const double g = G[i], d = -g;
                    D[i] = d;
                    acc += g * d;
                }
                dphi0 = red.sum(acc);
            }
            st.f0 = st.f;
            if (!(dphi0 < 0.0)) {                       // not a descent direction: restart from steepest descent
                st.pseudo = 1;
                double acc = 0.0;
                for (int64_t i = tid; i < n; i += TPB) {
                    const double g = G[i], d = -g;
                    D[i] = d;
                    acc += g * d;
                }
                dphi0 = red.sum(acc);
            }
            st.dphi0 = dphi0;
            if (!(dphi0 < 0.0)) {                       // zero (or non-finite) gradient
                st.phase = kLbfgsDone;
            } else {
                st.a1 = st.a2 = 1.0;                    // InitialStatic(alpha = 1)
                st.phi1 = st.f0;
                st.n_eval = 0;
                st.ls_it = 0;
                st.phase = kLbfgsFinite;
            }
        }
    }
    if (tid == 0) a.state[r] = st;
    // the point the next round evaluates (as lbfgs_trial_kernel, which runs only when the slots are re-assigned)
    if (st.phase == kLbfgsFinite || st.phase == kLbfgsArmijo) {
        double* t = a.trial + (int64_t)slot * n;
        const double alpha = st.a2;
        for (int64_t i = tid; i < n; i += TPB) t[i] = X[i] + alpha * D[i];
    }
}

hipError_t launch_lbfgs_feed(const LbfgsArgs& a_in, int n_active, hipStream_t s) {
    LbfgsArgs a = a_in;
    const int64_t stage = (int64_t)(2 * kLbfgsM + 2) * a.n;         // pairs, g and q in LDS when 60 KB hold them
    a.lds_doubles = stage * 8 <= 60 * 1024 ? stage : 0;
    const size_t lds = (size_t)a.lds_doubles * sizeof(double);
    if (a.n >= 16384) hipLaunchKernelGGL((lbfgs_feed_kernel<1024>), dim3((unsigned)n_active), dim3(1024), 0, s, a);
    else if (a.n >= 2048) hipLaunchKernelGGL((lbfgs_feed_kernel<256>), dim3((unsigned)n_active), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((lbfgs_feed_kernel<64>), dim3((unsigned)n_active), dim3(64), lds, s, a);
    return hipGetLastError();
}

}  // namespace cude

using namespace cude::api;

namespace {

// L-BFGS stage on host vectors (a sharded population; option "train_host"): one resumable cude::Lbfgs per restart,
// advanced in lock step, every round one multi-set evaluation through the host interface
int32_t lbfgs_stage_host(cude_ctx* c, int K, const std::vector<char>& alive, std::vector<double>& nn, std::vector<double>& cond,
                         int32_t adam_iters, int32_t lbfgs_iters, double* objective_out, double* loss_trace) {
    int32_t rc = CUDE_OK;
    const int P = c->P;
    const int64_t N = c->N, n = P + N, trace_len = (int64_t)adam_iters + lbfgs_iters;
    // L-BFGS takes inner products over [neural; conditional]: on a sharded population the conditional part of every
    // inner product / max-norm is reduced over the ranks (a few doubles per iteration, cude::Lbfgs reducer), so every
    // rank follows the same iterates
    std::vector<cude::Lbfgs> opt;
    std::vector<int> owner;                               // restart index of each machine
    std::vector<double> x0(n);
    for (int k = 0; k < K; k++) {
        if (!alive[k]) continue;
        std::copy(nn.begin() + (size_t)k * P, nn.begin() + (size_t)(k + 1) * P, x0.begin());
        std::copy(cond.begin() + (size_t)k * N, cond.begin() + (size_t)(k + 1) * N, x0.begin() + P);
        if (distributed(c)) opt.emplace_back(x0.data(), (int)n, lbfgs_iters, 10, 1e-8, P, lbfgs_comm_reduce, c);
        else opt.emplace_back(x0.data(), (int)n, lbfgs_iters);
        owner.push_back(k);
    }
    std::vector<double> b_nn, b_cond, b_f, b_gnn, b_gcond, gfull(n);
    std::vector<int> active;
    while (true) {
        active.clear();
        for (size_t q = 0; q < opt.size(); q++)
            if (!opt[q].done()) active.push_back((int)q);
        if (active.empty()) break;
        const int A = (int)active.size();
        b_nn.resize((size_t)A * P); b_cond.resize((size_t)A * N); b_f.resize(A);
        b_gnn.resize((size_t)A * P); b_gcond.resize((size_t)A * N);
        for (int a = 0; a < A; a++) {
            const double* x = opt[active[a]].pending();
            std::copy(x, x + P, b_nn.begin() + (size_t)a * P);
            std::copy(x + P, x + n, b_cond.begin() + (size_t)a * N);
        }
        if ((rc = cude_multistart_loss_grad(c, A, b_nn.data(), b_cond.data(), b_f.data(), b_gnn.data(), b_gcond.data()))) return rc;
        for (int a = 0; a < A; a++) {
            std::copy(b_gnn.begin() + (size_t)a * P, b_gnn.begin() + (size_t)(a + 1) * P, gfull.begin());
            std::copy(b_gcond.begin() + (size_t)a * N, b_gcond.begin() + (size_t)(a + 1) * N, gfull.begin() + P);
            cude::Lbfgs& o = opt[active[a]];
            const int before = o.accepted_steps();
            o.feed(b_f[a], gfull.data());
            if (loss_trace && o.accepted_steps() > before && before < lbfgs_iters)     // where Optim's callback fires
                loss_trace[(int64_t)owner[active[a]] * trace_len + adam_iters + before] = o.current_f();
            if (o.comm_failed()) return CUDE_ERR_COMM;     // message already set by the reducer
        }
    }
    for (size_t q = 0; q < opt.size(); q++) {
        const int k = owner[q];
        const std::vector<double>& x = opt[q].x();
        std::copy(x.begin(), x.begin() + P, nn.begin() + (size_t)k * P);
        std::copy(x.begin() + P, x.end(), cond.begin() + (size_t)k * N);
        objective_out[k] = opt[q].result().f;
    }
    return CUDE_OK;
}

// Adam stage on the host (option "train_host" = 2: the statement the device kernels are compared with)
int32_t adam_stage_host(cude_ctx* c, int K, std::vector<char>& alive, std::vector<double>& nn, std::vector<double>& cond,
                        int32_t adam_iters, double learning_rate, int64_t trace_len, double* loss_trace) {
    const int P = c->P;
    const int64_t N = c->N;
    std::vector<double> f(K), g_nn((size_t)K * P), g_cond((size_t)K * N);
    std::vector<double> m_nn((size_t)K * P, 0.0), v_nn((size_t)K * P, 0.0), m_c((size_t)K * N, 0.0), v_c((size_t)K * N, 0.0);
    for (int t = 1; t <= adam_iters; t++) {
        int32_t rc = cude_multistart_loss_grad(c, K, nn.data(), cond.data(), f.data(), g_nn.data(), g_cond.data());
        if (rc) return rc;
        for (int k = 0; k < K; k++) {
            if (!std::isfinite(f[k])) alive[k] = 0;
            if (!alive[k]) continue;
            if (loss_trace) loss_trace[(int64_t)k * trace_len + (t - 1)] = f[k];
            cude::adam_update(nn.data() + (size_t)k * P, g_nn.data() + (size_t)k * P, m_nn.data() + (size_t)k * P,
                              v_nn.data() + (size_t)k * P, P, t, learning_rate);
            cude::adam_update(cond.data() + (size_t)k * N, g_cond.data() + (size_t)k * N, m_c.data() + (size_t)k * N,
                              v_c.data() + (size_t)k * N, N, t, learning_rate);
        }
    }
    return CUDE_OK;
}

int32_t ensure_pinned(cude_ctx* c, size_t bytes) {
    if (bytes <= c->tr_pinned_bytes) return CUDE_OK;
    if (c->tr_pinned) (void)hipHostFree(c->tr_pinned);
    c->tr_pinned = nullptr;
    c->tr_pinned_bytes = 0;
    HIP_TRY(hipHostMalloc(&c->tr_pinned, bytes, hipHostMallocDefault));
    c->tr_pinned_bytes = bytes;
    return CUDE_OK;
}

}  // namespace

extern "C" {

int32_t cude_train_restarts(cude_ctx* c, int32_t n_sets, const double* nn_sets, const double* cond_sets,
                            int32_t adam_iters, double learning_rate, int32_t lbfgs_iters, double* nn_out,
                            double* cond_out, double* objective_out, double* loss_trace) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (n_sets < 1 || !nn_sets || !cond_sets || !nn_out || !cond_out || !objective_out || adam_iters < 0 ||
        lbfgs_iters < 0 || !(learning_rate > 0))
        return fail(CUDE_ERR_ARG, "bad argument");
    const int K = n_sets, P = c->P;
    const int64_t N = c->N, n = N + P;
    const int64_t trace_len = (int64_t)adam_iters + lbfgs_iters;
    const bool adam_on_host = c->opt.train_host >= 2;
    const bool lbfgs_on_host = c->opt.train_host >= 1 || distributed(c);
    for (int k = 0; k < K; k++) objective_out[k] = std::numeric_limits<double>::infinity();
    if (loss_trace)
        for (int64_t q = 0; q < (int64_t)K * trace_len; q++) loss_trace[q] = std::numeric_limits<double>::quiet_NaN();
    std::vector<char> alive(K, 1);
    if (adam_on_host) {
        std::vector<double> nn(nn_sets, nn_sets + (size_t)K * P), cond(cond_sets, cond_sets + (size_t)K * N);
        if ((rc = adam_stage_host(c, K, alive, nn, cond, adam_iters, learning_rate, trace_len, loss_trace))) return rc;
        if ((rc = lbfgs_stage_host(c, K, alive, nn, cond, adam_iters, lbfgs_iters, objective_out, loss_trace))) return rc;
        std::copy(nn.begin(), nn.end(), nn_out);
        std::copy(cond.begin(), cond.end(), cond_out);
        return CUDE_OK;
    }
    // ---- the restarts' parameters go up once
    HIP_TRY(c->ms_nn.reserve((size_t)K * P));
    HIP_TRY(c->ms_cond.reserve((size_t)K * N));
    HIP_TRY(c->ms_gcond.reserve((size_t)K * N));
    HIP_TRY(c->ms_out.reserve((size_t)K * (P + 2)));
    HIP_TRY(c->ms_f.reserve((size_t)K));
    HIP_TRY(c->tr_alive.reserve((size_t)K));
    HIP_TRY(hipMemcpyAsync(c->ms_nn.p, nn_sets, (size_t)K * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ms_cond.p, cond_sets, (size_t)K * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const bool dev_trace = loss_trace != nullptr && trace_len > 0;
    if (dev_trace) {
        HIP_TRY(c->tr_trace.reserve((size_t)K * trace_len));
        HIP_TRY(hipMemsetAsync(c->tr_trace.p, 0xff, (size_t)K * trace_len * sizeof(double), c->stream));    // all-ones = a NaN
    }
    {
        std::vector<int32_t> ones(K, 1);
        HIP_TRY(hipMemcpyAsync(c->tr_alive.p, ones.data(), (size_t)K * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));       // `ones` dies here (the caller's arrays may be pageable, too)
    }
    // ---- Adam, all restarts per launch; a restart whose loss becomes non-finite is dropped.  Nothing comes back.
    if (adam_iters > 0) {
        HIP_TRY(c->tr_m_nn.reserve((size_t)K * P)); HIP_TRY(c->tr_v_nn.reserve((size_t)K * P));
        HIP_TRY(c->tr_m_cond.reserve((size_t)K * N)); HIP_TRY(c->tr_v_cond.reserve((size_t)K * N));
        HIP_TRY(hipMemsetAsync(c->tr_m_nn.p, 0, (size_t)K * P * sizeof(double), c->stream));
        HIP_TRY(hipMemsetAsync(c->tr_v_nn.p, 0, (size_t)K * P * sizeof(double), c->stream));
        HIP_TRY(hipMemsetAsync(c->tr_m_cond.p, 0, (size_t)K * N * sizeof(double), c->stream));
        HIP_TRY(hipMemsetAsync(c->tr_v_cond.p, 0, (size_t)K * N * sizeof(double), c->stream));
        for (int t = 1; t <= adam_iters; t++) {
            if ((rc = eval_sets_device(c, K, c->ms_nn.p, P, c->ms_cond.p, N, c->ms_gcond.p, c->ms_out.p))) return rc;
            cude::FinishSetsArgs fa{};
            fa.P = P; fa.out = c->ms_out.p; fa.nn = c->ms_nn.p; fa.stride_nn = P; fa.lambda = c->cfg.lambda;
            fa.n_global = c->n_global; fa.mask = c->param_mask.p; fa.f = c->ms_f.p; fa.alive = c->tr_alive.p;
            fa.trace = dev_trace ? c->tr_trace.p : nullptr; fa.trace_len = trace_len; fa.trace_pos = t - 1;
            HIP_TRY(cude::launch_finish_sets(fa, K, c->stream));
            cude::AdamSetsArgs aa{};
            aa.N = N; aa.P = P; aa.cond = c->ms_cond.p; aa.nn = c->ms_nn.p;
            aa.m_cond = c->tr_m_cond.p; aa.v_cond = c->tr_v_cond.p; aa.m_nn = c->tr_m_nn.p; aa.v_nn = c->tr_v_nn.p;
            aa.g_cond = c->ms_gcond.p; aa.out = c->ms_out.p; aa.alive = c->tr_alive.p;
            aa.lr = learning_rate; aa.b1 = 0.9; aa.b2 = 0.999; aa.eps = 1e-8;
            aa.c1 = 1.0 - std::pow(aa.b1, t); aa.c2 = 1.0 - std::pow(aa.b2, t);
            HIP_TRY(cude::launch_adam_sets(aa, K, c->stream));
        }
    }
    if ((rc = ensure_pinned(c, std::max((size_t)K * sizeof(cude::LbfgsState), (size_t)K * sizeof(int32_t))))) return rc;
    HIP_TRY(hipMemcpyAsync(c->tr_pinned, c->tr_alive.p, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->xchg.ready && (rc = xchg_check(c))) return rc;
    for (int k = 0; k < K; k++) alive[k] = static_cast<const int32_t*>(c->tr_pinned)[k] != 0;
    if (lbfgs_on_host) {
        std::vector<double> nn((size_t)K * P), cond((size_t)K * N);
        HIP_TRY(hipMemcpyAsync(nn.data(), c->ms_nn.p, nn.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(cond.data(), c->ms_cond.p, cond.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (dev_trace)
            HIP_TRY(hipMemcpyAsync(loss_trace, c->tr_trace.p, (size_t)K * trace_len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if ((rc = lbfgs_stage_host(c, K, alive, nn, cond, adam_iters, lbfgs_iters, objective_out, loss_trace))) return rc;
        std::copy(nn.begin(), nn.end(), nn_out);
        std::copy(cond.begin(), cond.end(), cond_out);
        return CUDE_OK;
    }
    // ---- L-BFGS on the device: restart r of the R live ones keeps [conditional; network] vectors of n doubles
    std::vector<int32_t> owner;
    for (int k = 0; k < K; k++)
        if (alive[k]) owner.push_back(k);
    const int R = (int)owner.size();
    if (R > 0) {
        constexpr int m = cude::kLbfgsM;
        HIP_TRY(c->tr_x.reserve((size_t)R * n)); HIP_TRY(c->tr_g.reserve((size_t)R * n)); HIP_TRY(c->tr_d.reserve((size_t)R * n));
        HIP_TRY(c->tr_trial.reserve((size_t)R * n)); HIP_TRY(c->tr_gtrial.reserve((size_t)R * n));
        if (lbfgs_iters > 0) { HIP_TRY(c->tr_s.reserve((size_t)R * m * n)); HIP_TRY(c->tr_y.reserve((size_t)R * m * n)); }
        HIP_TRY(c->tr_state.reserve((size_t)R * sizeof(cude::LbfgsState)));
        HIP_TRY(c->tr_act.reserve((size_t)2 * K));
        std::vector<cude::LbfgsState> st((size_t)R);
        for (int r = 0; r < R; r++) {
            cude::LbfgsState s0{};
            s0.phase = cude::kLbfgsFirst; s0.maxiters = lbfgs_iters; s0.g_tol = 1e-8;
            s0.f = std::numeric_limits<double>::quiet_NaN();
            st[(size_t)r] = s0;
        }
        // iterate r = [conditional; network] of restart owner[r]: runs of consecutive live restarts move in one strided copy
        auto pack = [&](bool into_vectors) -> int32_t {
            for (int r = 0; r < R;) {
                int len = 1;
                while (r + len < R && owner[(size_t)(r + len)] == owner[(size_t)r] + len) len++;
                const int k = owner[(size_t)r];
                double* xv = c->tr_x.p + (size_t)r * n;
                double* cs = c->ms_cond.p + (size_t)k * N;
                double* ns = c->ms_nn.p + (size_t)k * P;
                if (into_vectors) {
                    HIP_TRY(hipMemcpy2DAsync(xv, (size_t)n * 8, cs, (size_t)N * 8, (size_t)N * 8, (size_t)len, hipMemcpyDeviceToDevice, c->stream));
                    HIP_TRY(hipMemcpy2DAsync(xv + N, (size_t)n * 8, ns, (size_t)P * 8, (size_t)P * 8, (size_t)len, hipMemcpyDeviceToDevice, c->stream));
                } else {
                    HIP_TRY(hipMemcpy2DAsync(cs, (size_t)N * 8, xv, (size_t)n * 8, (size_t)N * 8, (size_t)len, hipMemcpyDeviceToDevice, c->stream));
                    HIP_TRY(hipMemcpy2DAsync(ns, (size_t)P * 8, xv + N, (size_t)n * 8, (size_t)P * 8, (size_t)len, hipMemcpyDeviceToDevice, c->stream));
                }
                r += len;
            }
            return CUDE_OK;
        };
        if ((rc = pack(true))) return rc;
        HIP_TRY(hipMemcpyAsync(c->tr_state.p, st.data(), (size_t)R * sizeof(cude::LbfgsState), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->tr_act.p + K, owner.data(), (size_t)R * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));       // st / owner are read by the copies above
        cude::LbfgsArgs la{};
        la.n = n; la.state = reinterpret_cast<cude::LbfgsState*>(c->tr_state.p); la.act = c->tr_act.p;
        la.X = c->tr_x.p; la.G = c->tr_g.p; la.D = c->tr_d.p; la.S = c->tr_s.p; la.Y = c->tr_y.p;
        la.trial = c->tr_trial.p; la.g_trial = c->tr_gtrial.p; la.f_trial = c->ms_f.p;
        la.trace = dev_trace ? c->tr_trace.p : nullptr; la.owner = c->tr_act.p + K;
        la.trace_len = trace_len; la.trace_off = adam_iters;
        std::vector<int32_t> act((size_t)R);
        for (int r = 0; r < R; r++) act[(size_t)r] = r;
        // finished restarts leave the evaluation at the next look at the phases: after every round when an evaluation is
        // worth more than the host's turn-around, after every fourth when the launches are latency-bound anyway
        const int look_every = (int64_t)R * c->nblocks > 2048 ? 1 : 4;
        const cude::LbfgsState* host_st = static_cast<const cude::LbfgsState*>(c->tr_pinned);
        bool act_changed = true;
        for (int64_t round = 0; !act.empty(); round++) {
            const int A = (int)act.size();
            if (act_changed) {
                HIP_TRY(hipMemcpyAsync(c->tr_act.p, act.data(), (size_t)A * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));       // (pageable source: `act` is rebuilt below)
                act_changed = false;
                HIP_TRY(cude::launch_lbfgs_trial(la, A, c->stream));     // (otherwise the feed kernel has left the trial points)
            }
            if ((rc = eval_sets_device(c, A, c->tr_trial.p + N, n, c->tr_trial.p, n, c->tr_gtrial.p, c->ms_out.p))) return rc;
            cude::FinishSetsArgs fa{};
            fa.P = P; fa.out = c->ms_out.p; fa.nn = c->tr_trial.p + N; fa.stride_nn = n; fa.lambda = c->cfg.lambda;
            fa.n_global = c->n_global; fa.mask = c->param_mask.p; fa.f = c->ms_f.p;
            fa.g_dst = c->tr_gtrial.p + N; fa.g_stride = n;
            HIP_TRY(cude::launch_finish_sets(fa, A, c->stream));
            HIP_TRY(cude::launch_lbfgs_feed(la, A, c->stream));
            if ((round + 1) % look_every != 0) continue;
            HIP_TRY(hipMemcpyAsync(c->tr_pinned, c->tr_state.p, (size_t)R * sizeof(cude::LbfgsState), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const size_t before = act.size();
            act.clear();
            for (int r = 0; r < R; r++)
                if (host_st[r].phase != cude::kLbfgsDone) act.push_back(r);
            act_changed = act.size() != before;
        }
        for (int r = 0; r < R; r++) objective_out[owner[(size_t)r]] = host_st[r].f;
        if ((rc = pack(false))) return rc;      // the iterates back into the [K][P] / [K][N] sets
    }
    HIP_TRY(hipMemcpyAsync(nn_out, c->ms_nn.p, (size_t)K * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(cond_out, c->ms_cond.p, (size_t)K * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (dev_trace)
        HIP_TRY(hipMemcpyAsync(loss_trace, c->tr_trace.p, (size_t)K * trace_len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

}  // extern "C"
