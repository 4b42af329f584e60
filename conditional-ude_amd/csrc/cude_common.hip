// Small support kernels: population preparation (van Cauter kinetics, glucose increments),
// deterministic second-stage reduction of the per-workgroup partials, L2 term, and Adam.
//
// Replaces (reference repo paths):
//   van_cauter_parameters     src/c-peptide-models.jl:30-42 (dup src/saem.jl:7-21)
//   Optimisers.Adam update    used at src/parameter-estimation.jl:176, suppression_model.jl:164, saem.jl:128
//   lambda*sum(abs2, neural)  suppression/src/suppression_model.jl:128
#include <algorithm>

#include "cude_device.h"
#include "cude_kernels.h"
#include "cude_rng.h"
#include "cude_xchg.h"

namespace cude {

// glucose_tn / cpep_tn are already [T][N] (subject fastest).
__global__ void prepare_cpep_kernel(int64_t N, int T, const double* __restrict__ glucose_tn,
                                    const double* __restrict__ cpep_tn, const double* __restrict__ age,
                                    const uint8_t* __restrict__ t2dm, double* __restrict__ k0, double* __restrict__ k1,
                                    double* __restrict__ k2, double* __restrict__ c0, double* __restrict__ dG) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const bool d = t2dm[i] != 0;
    const double ln2 = 0.693147180559945309417232121458;
    const double sh = d ? 4.52 : 4.95, fr = d ? 0.78 : 0.76, lo = 0.14 * age[i] + 29.2;
    const double kk1 = fr * (ln2 / lo) + (1.0 - fr) * (ln2 / sh);
    const double kk0 = (ln2 / sh) * (ln2 / lo) / kk1;
    const double kk2 = (ln2 / sh) + (ln2 / lo) - kk0 - kk1;
    k0[i] = kk0;
    k1[i] = kk1;
    k2[i] = kk2;
    c0[i] = cpep_tn[i];
    const double g0 = glucose_tn[i];
    for (int t = 0; t < T; t++) dG[(int64_t)t * N + i] = glucose_tn[(int64_t)t * N + i] - g0;
}

hipError_t launch_prepare_cpep(int64_t N, int T, const double* glucose_tn, const double* cpep_tn, const double* age,
                               const uint8_t* t2dm, double* k0, double* k1, double* k2, double* c0, double* dG,
                               hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(prepare_cpep_kernel, dim3((unsigned)((N + bs - 1) / bs)), dim3(bs), 0, s, N, T, glucose_tn,
                       cpep_tn, age, t2dm, k0, k1, k2, c0, dG);
    return hipGetLastError();
}

// After an iteration's [sum loss, n_failed] is final: advance the running powers / step counter (unless the step is
// skipped because a subject failed) and append the pair to the loss trace.  One thread.
__device__ __forceinline__ void adam_advance(const TailAdvance& a, double loss_sum, double n_failed) {
    if (!(n_failed > 0.0)) {
        a.state[0] *= a.b1;
        a.state[1] *= a.b2;
        a.state[2] += 1.0;
    }
    const int64_t pos = (int64_t)a.state[3];
    if (pos < a.cap) {
        a.trace[2 * pos] = loss_sum;
        a.trace[2 * pos + 1] = n_failed;
    }
    a.state[3] = (double)(pos + 1);
}

// One workgroup per column; fixed-shape tree => bitwise reproducible for a given nblocks.
// mask (optional, n_mask entries): frozen shared parameters (cude_set_param_mask) -- column q < n_mask is scaled by mask[q]
// adv.state != nullptr: the workgroup of the last column (the failure count) sums the column before it (the loss) as
// well -- same tree, so the same bits as that column's own workgroup -- and advances the optimiser state: the update
// kernel behind this launch then needs no third launch for it (round 2: 3 launches per step tail, 4.7 us the last).
// xchg.seq != nullptr: the column's sum is combined with the other ranks' through the peer-write exchange (cude_xchg.h)
// before it is stored, so that `out` holds the sum over ALL ranks' subjects -- no collective launch behind this one,
// and the tail workgroup advances the optimiser state with the GLOBAL pair.  With adv the tail workgroup exchanges the
// loss column as well and the loss column's own workgroup stores nothing.
// (the body of reduce_partials_kernel for column q; shared with chunked_tail_kernel so that both form the same bits)
__device__ __forceinline__ void reduce_column(const double* __restrict__ partials, int64_t nblocks, int stride, int q,
                                              double* __restrict__ out, const double* __restrict__ mask, int n_mask,
                                              int out_stride, int accumulate, const TailAdvance& adv, double* host_tail,
                                              const XchgArgs& xchg, double* s, double* s2) {
    const bool tail = adv.state != nullptr && q == stride - 1 && blockIdx.y == 0;      // wave-uniform
    const bool xc = xchg.seq != nullptr && blockIdx.y == 0;
    if (xc && adv.state != nullptr && q == stride - 2) return;     // the tail workgroup exchanges and stores this column
    partials += (int64_t)blockIdx.y * nblocks * stride;      // multi-start: one row of the grid per parameter set
    out += (int64_t)blockIdx.y * out_stride;
    double v = 0.0, v2 = 0.0;
    if (tail) {
        for (int64_t b = threadIdx.x; b < nblocks; b += 256) {
            v += partials[b * stride + q];
            v2 += partials[b * stride + q - 1];
        }
    } else {
        for (int64_t b = threadIdx.x; b < nblocks; b += 256) v += partials[b * stride + q];
    }
    s[threadIdx.x] = v;
    s2[threadIdx.x] = v2;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s[threadIdx.x] += s[threadIdx.x + off];
            if (tail) s2[threadIdx.x] += s2[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double v0 = (mask != nullptr && q < n_mask) ? s[0] * mask[q] : s[0];
        double vq = accumulate ? out[q] + v0 : v0;               // accumulate: a second group of rows of the same launch
        double loss = s2[0];
        if (xc && tail) {               // loss sum (column q - 1) and failure count (q): both written, then both waited for
            xchg_combine2(xchg, q - 1, loss, vq);
            out[q - 1] = loss;
            if (host_tail != nullptr) host_tail[0] = loss;
        } else if (xc) {
            vq = xchg_combine(xchg, q, vq);
        }
        out[q] = vq;
        // [sum loss, n_failed] straight into page-locked host memory as well, where the host watches for them
        if (host_tail != nullptr && q >= stride - 2 && blockIdx.y == 0) host_tail[q - (stride - 2)] = vq;
        if (tail) adam_advance(adv, loss, vq);
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partials, int64_t nblocks,
                                                              int stride, int col0, double* __restrict__ out,
                                                              const double* __restrict__ mask, int n_mask, int out_stride,
                                                              int accumulate, TailAdvance adv, double* host_tail,
                                                              XchgArgs xchg) {
    __shared__ double s[256];
    __shared__ double s2[256];
    reduce_column(partials, nblocks, stride, col0 + blockIdx.x, out, mask, n_mask, out_stride, accumulate, adv, host_tail, xchg,
                  s, s2);
}

// Tail of a time-split gradient evaluation in ONE launch (round 5; three before: at the reference's population sizes every
// launch of the dependent chain costs ~5 us whatever it does): workgroups [0, P) reduce the network-gradient columns over
// the reverse chunks' rows (partials2, stride P), workgroups P and P + 1 the loss / failure columns over the scan's rows
// (partials, stride P + 2; tail work as in reduce_partials_kernel), the workgroups behind them add up the chunks' shares
// of d loss / d conditional (what cpep2_sum_chunks_kernel does).  Each output is formed by the code that formed it before.
__global__ __launch_bounds__(256) void chunked_tail_kernel(ChunkedTailArgs a) {
    __shared__ double s[256];
    __shared__ double s2[256];
    const int q = blockIdx.x;
    if (q < a.P) {
        reduce_column(a.partials2, a.rows2, a.P, q, a.out, a.mask, a.n_mask, a.out_stride, 0, TailAdvance{}, nullptr, a.xchg, s, s2);
    } else if (q < a.P + 2) {
        reduce_column(a.partials, a.rows, a.P + 2, q, a.out, nullptr, 0, a.out_stride, 0, a.adv, a.host_tail, a.xchg, s, s2);
    } else {
        const int64_t i = ((int64_t)(q - a.P - 2)) * 256 + threadIdx.x;
        if (i >= a.N) return;
        const double* part = a.g_cond_part + (int64_t)blockIdx.y * a.L * a.N;
        double v = 0.0;
        for (int c = 0; c < a.L; c++) v += part[(int64_t)c * a.N + i];
        a.g_cond[(int64_t)blockIdx.y * a.g_cond_set_stride + i] = v;
    }
}

hipError_t launch_chunked_tail(const ChunkedTailArgs& a, int n_sets, hipStream_t s) {
    if (a.P < 1 || a.out_stride < a.P + 2 || n_sets < 1) return hipErrorInvalidValue;
    if (a.xchg.seq != nullptr && (n_sets != 1 || a.P + 2 > a.xchg.cols)) return hipErrorInvalidValue;
    const unsigned gblocks = a.g_cond_part != nullptr ? (unsigned)((a.N + 255) / 256) : 0u;
    hipLaunchKernelGGL(chunked_tail_kernel, dim3((unsigned)a.P + 2u + gblocks, (unsigned)n_sets), dim3(256), 0, s, a);
    return hipGetLastError();
}

// reduces columns [col0, col0+ncol) of partials[nblocks][stride] into out[col0..]
hipError_t launch_reduce_cols(const double* partials, int64_t nblocks, int stride, int col0, int ncol, double* out,
                              hipStream_t s, int n_sets, const double* mask, int n_mask, int out_stride, bool accumulate,
                              const TailAdvance* adv, double* host_tail, const XchgArgs* xchg) {
    TailAdvance a{};
    if (adv != nullptr) {
        if (accumulate || col0 + ncol != stride || ncol < 2) return hipErrorInvalidValue;
        a = *adv;
    }
    XchgArgs x{};
    if (xchg != nullptr && xchg->seq != nullptr) {
        if (n_sets != 1 || stride > xchg->cols) return hipErrorInvalidValue;
        x = *xchg;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ncol, n_sets), dim3(256), 0, s, partials, nblocks, stride, col0,
                       out, mask, n_mask, out_stride > 0 ? out_stride : stride, accumulate ? 1 : 0, a,
                       col0 + ncol == stride ? host_tail : nullptr, x);
    return hipGetLastError();
}

// buf[k] <- combination over the ranks, for k in [0, count): one workgroup, lane t exchanges column t of a slab of
// `cols` elements at a time (the columns' sequence counters advance together on every rank: all ranks make the same calls)
__global__ __launch_bounds__(256) void xchg_allreduce_kernel(XchgArgs x, double* __restrict__ buf, int64_t count, int op) {
    for (int64_t k0 = 0; k0 < count; k0 += x.cols)
        for (int t = threadIdx.x; t < x.cols && k0 + t < count; t += 256) buf[k0 + t] = xchg_combine(x, t, buf[k0 + t], op);
}

hipError_t launch_xchg_allreduce(const XchgArgs& x, double* buf, int64_t count, int op, hipStream_t s) {
    if (x.seq == nullptr || x.cols < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(1), dim3(256), 0, s, x, buf, count, op);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void reduce_sets_kernel(const double* __restrict__ partials, int64_t nblocks, int stride,
                                                         int col0, double* __restrict__ out) {
    const int64_t k = blockIdx.x;
    const double* base = partials + k * nblocks * stride + col0;
    double a = 0.0, b = 0.0;
    for (int64_t r = threadIdx.x; r < nblocks; r += 64) {
        a += base[r * stride];
        b += base[r * stride + 1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
        out[2 * k] = a;
        out[2 * k + 1] = b;
    }
}

hipError_t launch_reduce_sets(const double* partials, int n_sets, int64_t nblocks, int stride, int col0, double* out,
                              hipStream_t s) {
    hipLaunchKernelGGL(reduce_sets_kernel, dim3(n_sets), dim3(64), 0, s, partials, nblocks, stride, col0, out);
    return hipGetLastError();
}

// ---- screening (first phase of `train`, src/parameter-estimation.jl:359-372): candidate losses and a running
// top-k on the device.
// loss[k] = sum_k / n_global + lambda * |nn_k|^2, +Inf for a set with a failed subject (the reference's convention)
__global__ void set_losses_kernel(int n_sets, const double* __restrict__ sums, const double* __restrict__ nn, int P,
                                  double lambda, double n_global, double* __restrict__ loss) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_sets) return;
    double reg = 0.0;
    if (lambda != 0.0)
        for (int q = 0; q < P; q++) reg = fma(nn[(int64_t)k * P + q], nn[(int64_t)k * P + q], reg);
    const double sum = sums[2 * k], nf = sums[2 * k + 1];
    const bool bad = nf > 0.0 || !(fabs(sum) <= 1.79769313486231570815e308);
    loss[k] = bad ? __builtin_inf() : fma(lambda, reg, sum / n_global);
}

hipError_t launch_set_losses(int n_sets, const double* sums, const double* nn, int P, double lambda, double n_global,
                             double* loss, hipStream_t s) {
    hipLaunchKernelGGL(set_losses_kernel, dim3((n_sets + 255) / 256), dim3(256), 0, s, n_sets, sums, nn, P, lambda,
                       n_global, loss);
    return hipGetLastError();
}

// `partialsortperm(losses, 1:n_keep)` as a running merge: the n_keep smallest of {current best (n_have)} U {this chunk
// (n_new)} by (loss, global index) -- n_keep passes of a block-wide arg-min over the union (n_keep is a few dozen).
// Union entry u < n_have is best slot u, otherwise chunk row u - n_have with global index first + (u - n_have).
// work[n_have + n_new] is scratch; sel_src[r] receives the union entry chosen for rank r.
__global__ __launch_bounds__(1024) void topk_merge_kernel(TopkArgs a) {
    __shared__ double s_v[1024];
    __shared__ long long s_i[1024];
    __shared__ int s_u[1024];
    const int tid = threadIdx.x, M = a.n_have + a.n_new;
    for (int u = tid; u < M; u += 1024) a.work[u] = u < a.n_have ? a.best_loss[u] : a.chunk_loss[u - a.n_have];
    __syncthreads();
    const int n_out = M < a.n_keep ? M : a.n_keep;
    for (int r = 0; r < n_out; r++) {
        double bv = __builtin_inf();
        long long bi = 0x7fffffffffffffffLL;
        int bu = -1;
        for (int u = tid; u < M; u += 1024) {
            const double v = a.work[u];
            if (v != v) continue;                                   // taken
            const long long gi = u < a.n_have ? a.best_idx[u] : a.first + (u - a.n_have);
            if (v < bv || (v == bv && gi < bi)) { bv = v; bi = gi; bu = u; }
        }
        s_v[tid] = bv; s_i[tid] = bi; s_u[tid] = bu;
        __syncthreads();
        for (int off = 512; off >= 1; off >>= 1) {
            if (tid < off) {
                const bool take = s_u[tid + off] >= 0 &&
                                  (s_u[tid] < 0 || s_v[tid + off] < s_v[tid] ||
                                   (s_v[tid + off] == s_v[tid] && s_i[tid + off] < s_i[tid]));
                if (take) { s_v[tid] = s_v[tid + off]; s_i[tid] = s_i[tid + off]; s_u[tid] = s_u[tid + off]; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            a.new_loss[r] = s_v[0];
            a.new_idx[r] = s_i[0];
            a.sel_src[r] = s_u[0];
            a.work[s_u[0]] = __builtin_nan("");
        }
        __syncthreads();
    }
}

// new_nn[r] / new_cond[r] <- the parameter set chosen for rank r (from the previous best buffers or from the chunk)
__global__ void topk_gather_kernel(TopkArgs a, int n_out, int P, int64_t N, const double* __restrict__ old_nn,
                                   const double* __restrict__ old_cond, const double* __restrict__ chunk_nn,
                                   const double* __restrict__ chunk_cond, double* __restrict__ new_nn,
                                   double* __restrict__ new_cond) {
    const int r = blockIdx.y;
    if (r >= n_out) return;
    const int u = a.sel_src[r];
    const double* snn = u < a.n_have ? old_nn + (int64_t)u * P : chunk_nn + (int64_t)(u - a.n_have) * P;
    const double* scd = u < a.n_have ? old_cond + (int64_t)u * N : chunk_cond + (int64_t)(u - a.n_have) * N;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < P + N; q += (int64_t)gridDim.x * blockDim.x) {
        if (q < P) new_nn[(int64_t)r * P + q] = snn[q];
        else new_cond[(int64_t)r * N + (q - P)] = scd[q - P];
    }
}

hipError_t launch_topk_merge(const TopkArgs& a, int P, int64_t N, const double* old_nn, const double* old_cond,
                             const double* chunk_nn, const double* chunk_cond, double* new_nn, double* new_cond,
                             hipStream_t s) {
    hipLaunchKernelGGL(topk_merge_kernel, dim3(1), dim3(1024), 0, s, a);
    const int M = a.n_have + a.n_new, n_out = M < a.n_keep ? M : a.n_keep;
    const unsigned gx = (unsigned)std::min<int64_t>(64, (P + N + 255) / 256);
    hipLaunchKernelGGL(topk_gather_kernel, dim3(gx, n_out), dim3(256), 0, s, a, n_out, P, N, old_nn, old_cond, chunk_nn,
                       chunk_cond, new_nn, new_cond);
    return hipGetLastError();
}

__global__ void l2_term_kernel(const double* __restrict__ nn, int P, double lambda, double n_global,
                               double* __restrict__ out, const double* __restrict__ mask, TailAdvance adv) {
    // single wave
    const int lane = threadIdx.x;
    double ss = 0.0;
    for (int q = lane; q < P; q += 64) {
        const double w = nn[q];
        ss = fma(w, w, ss);
        out[q] = fma(2.0 * lambda * (mask != nullptr ? mask[q] : 1.0), w, out[q]);
    }
    ss = wave_sum(ss);
    if (lane == 0) {
        const double loss = fma(lambda * n_global, ss, out[P]);
        out[P] = loss;
        if (adv.state != nullptr) adam_advance(adv, loss, out[P + 1]);
    }
}

hipError_t launch_l2_term(const double* nn, int P, double lambda, double n_global, double* out, hipStream_t s,
                          const double* mask, const TailAdvance* adv) {
    hipLaunchKernelGGL(l2_term_kernel, dim3(1), dim3(64), 0, s, nn, P, lambda, n_global, out, mask,
                       adv != nullptr ? *adv : TailAdvance{});
    return hipGetLastError();
}

// ---- SAEM E-step (src/saem.jl:86-108): proposal and accept/reject of one Metropolis-Hastings step for
// every subject.  The two likelihood evaluations in between are ordinary forward launches.
__global__ void mh_propose_kernel(int64_t N, const double* __restrict__ p, const double* __restrict__ z, RngKey key,
                                  double proposal_std, double* __restrict__ prop) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) prop[i] = mh_proposal(p, z, key, proposal_std, i);
}

__global__ void rng_draws_kernel(int64_t N, RngKey key, double* __restrict__ normals, double* __restrict__ uniforms) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (normals != nullptr) normals[i] = rng_normal(key, i);
    if (uniforms != nullptr) uniforms[i] = rng_uniform(key, i);
}

__global__ void mh_accept_kernel(MhArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    mh_accept_one(a, i, a.prop[i], a.sse_new[i]);
}

// gamma < 1: the next state is a BLEND of the current state with the proposal (accepted) or with itself (rejected) -- in
// either case a point whose likelihood is not known yet, which is why every step used to solve twice (the proposal, then
// the current state again).  Both blends depend on nothing but (p, q, gamma), so they are formed BEFORE the decision and
// solved in the SAME launch as the proposal (three parameter sets: on a chip that a small population leaves mostly empty
// three sets cost what one costs); the decision then picks state and carried SSE.  cand / sse: [3][N] = proposal, blend
// if accepted, blend if rejected.  The same expressions on the same values as the two-launch form: the same chain.
__global__ void mh_blend_candidates_kernel(int64_t N, const double* __restrict__ p, const double* __restrict__ z, RngKey key,
                                           double proposal_std, double gamma, double* __restrict__ cand) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double pi = p[i], q = mh_proposal(p, z, key, proposal_std, i);
    cand[i] = q;
    cand[N + i] = mh_blend(gamma, pi, q);
    cand[2 * N + i] = mh_blend(gamma, pi, pi);
}
__global__ void mh_accept_blend_kernel(MhArgs a, const double* __restrict__ cand, const double* __restrict__ sse) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const int64_t N = a.N;
    const double p = a.p[i];
    const double u = a.u != nullptr ? a.u[i] : rng_uniform(a.key, i);
    const bool acc = mh_decide(a, p, cand[i], a.sse_cur[i], sse[i], u);
    a.p[i] = acc ? cand[N + i] : cand[2 * N + i];
    a.sse_cur[i] = acc ? sse[N + i] : sse[2 * N + i];
    if (acc) a.accepted[i] += 1;
}

// One lane per subject: walks the candidate heap of the round just evaluated (depth_resolve levels), leaves the chain
// where the sequential steps would have left it, and writes the next round's candidates (MhSpecArgs, cude_kernels.h).
// Everything that does not depend on the path -- the draws, log(u), every candidate's prior term and tempered
// log-likelihood -- is formed first, for all nodes side by side (mh_spec_resolve, cude_rng.h); the walk itself is d
// compare-and-select steps.
__global__ void mh_spec_kernel(MhSpecArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.mh.N) return;
    if (a.blend) mh_spec_resolve_blend(a, i);
    else mh_spec_resolve(a, i);
}

hipError_t launch_mh_spec(const MhSpecArgs& a, hipStream_t s) {
    const int dmax = a.blend ? kMhSpecMaxDepthBlend : kMhSpecMaxDepth;
    if (a.depth_resolve < 0 || a.depth_resolve > dmax || a.depth_next < 0 || a.depth_next > dmax) return hipErrorInvalidValue;
    const int bs = 64;
    hipLaunchKernelGGL(mh_spec_kernel, dim3((unsigned)((a.mh.N + bs - 1) / bs)), dim3(bs), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_mh_propose(int64_t N, const double* p, const double* z, RngKey key, double proposal_std, double* prop,
                             hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(mh_propose_kernel, dim3((unsigned)((N + bs - 1) / bs)), dim3(bs), 0, s, N, p, z, key, proposal_std,
                       prop);
    return hipGetLastError();
}

hipError_t launch_rng_draws(int64_t N, RngKey key, double* normals, double* uniforms, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(rng_draws_kernel, dim3((unsigned)((N + bs - 1) / bs)), dim3(bs), 0, s, N, key, normals, uniforms);
    return hipGetLastError();
}

hipError_t launch_mh_accept(const MhArgs& a, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(mh_accept_kernel, dim3((unsigned)((a.N + bs - 1) / bs)), dim3(bs), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_mh_blend_candidates(int64_t N, const double* p, const double* z, RngKey key, double proposal_std, double gamma,
                                      double* cand, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(mh_blend_candidates_kernel, dim3((unsigned)((N + bs - 1) / bs)), dim3(bs), 0, s, N, p, z, key,
                       proposal_std, gamma, cand);
    return hipGetLastError();
}

hipError_t launch_mh_accept_blend(const MhArgs& a, const double* cand, const double* sse, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(mh_accept_blend_kernel, dim3((unsigned)((a.N + bs - 1) / bs)), dim3(bs), 0, s, a, cand, sse);
    return hipGetLastError();
}

// ---- per-subject 1-D fits with the shared parameters frozen (cude_fit_conditional): every subject minimises
// f_i(x) = SSE_i(x) + w (x - mu)^2 over a box; all subjects advance together, one forward launch per probe.
// Phase kernels between the forward launches keep the whole search on the device.
__device__ __forceinline__ double fit_objective(double sse, double x, double w, double mu) {
    const double f = fma(w * (x - mu), x - mu, sse);
    return fabs(f) <= 1.79769313486231570815e308 ? f : __builtin_huge_val();     // failed solve = +Inf
}

// coarse scan: probe k of the grid was evaluated at x for everybody; keep the best index per subject
__global__ void fit_grid_kernel(FitArgs a, int k, double x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const double f = fit_objective(a.sse_c[i], x, a.w, a.mu);
    if (k == 0 || f < a.fc[i]) {       // strict <: first minimum wins, as numpy.argmin
        a.fc[i] = f;
        a.best[i] = k;
    }
}

// bracket [grid[k-1], grid[k+1]] around the best grid point and the two golden-section probes inside it
__global__ void fit_bracket_kernel(FitArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    int k = (int)a.best[i];
    k = k < 1 ? 1 : (k > a.n_grid - 2 ? a.n_grid - 2 : k);
    const double lo = fma((double)(k - 1), a.step, a.lower), hi = fma((double)(k + 1), a.step, a.lower);
    a.a[i] = lo;
    a.b[i] = hi;
    a.c[i] = hi - a.gr * (hi - lo);
    a.d[i] = lo + a.gr * (hi - lo);
}

// one golden-section step from the SSEs at c and d; final = 1 writes the midpoint to c instead of new probes
__global__ void fit_golden_kernel(FitArgs a, int final) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    double lo = a.a[i], hi = a.b[i];
    const double c = a.c[i], d = a.d[i];
    const double fc = fit_objective(a.sse_c[i], c, a.w, a.mu), fd = fit_objective(a.sse_d[i], d, a.w, a.mu);
    if (fc < fd) hi = d; else lo = c;
    a.a[i] = lo;
    a.b[i] = hi;
    if (final) {
        a.c[i] = 0.5 * (lo + hi);
    } else {
        a.c[i] = hi - a.gr * (hi - lo);
        a.d[i] = lo + a.gr * (hi - lo);
    }
}

// ---- the same search with several probes per forward launch (cude_fit_conditional, option "fit_spec"): on a chip that a
// small population leaves mostly empty a launch of k parameter sets costs what a launch of one costs.
// Grid scan: all n_grid values as the sets of one launch; the scan's update in the scan's order.
__global__ void fit_grid_all_kernel(FitArgs a, int k0, int kn, const double* __restrict__ values, const double* __restrict__ sse_sets) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    double fbest = a.fc[i], kbest = a.best[i];
    for (int k = 0; k < kn; k++) {
        const double f = fit_objective(sse_sets[(int64_t)k * a.N + i], values[k0 + k], a.w, a.mu);
        if (k0 + k == 0 || f < fbest) { fbest = f; kbest = (double)(k0 + k); }
    }
    a.fc[i] = fbest;
    a.best[i] = kbest;
}
// Golden section: which half survives a step is a binary outcome, so the probes of the next `depth` steps form a heap --
// node v holds the bracket it would have and its two probes (rows 2 (v - 1) and 2 (v - 1) + 1 of cand), node 2v = the
// bracket after "left" (fc < fd), 2v + 1 after "right" -- all evaluated in one launch; the walk then takes `depth` steps.
// The same expressions on the same values as fit_golden_kernel step by step: the same brackets, the same result.
__global__ void fit_tree_candidates_kernel(FitArgs a, int depth, double* __restrict__ cand) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    constexpr int D = kFitSpecMaxDepth;
    double lo[1 << D], hi[1 << D];
    lo[1] = a.a[i];
    hi[1] = a.b[i];
#pragma unroll
    for (int l = 0; l < D; l++) {
        if (l < depth) {
#pragma unroll
            for (int v = 1 << l; v < (2 << l); v++) {
                const double c = hi[v] - a.gr * (hi[v] - lo[v]), d = lo[v] + a.gr * (hi[v] - lo[v]);
                cand[(int64_t)(2 * (v - 1)) * a.N + i] = c;
                cand[(int64_t)(2 * (v - 1) + 1) * a.N + i] = d;
                if (l + 1 < D) { lo[2 * v] = lo[v]; hi[2 * v] = d; lo[2 * v + 1] = c; hi[2 * v + 1] = hi[v]; }
            }
        }
    }
}
__global__ void fit_tree_resolve_kernel(FitArgs a, int depth, int final, const double* __restrict__ cand,
                                        const double* __restrict__ sse_sets) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    double lo = a.a[i], hi = a.b[i];
    int v = 1;
    for (int l = 0; l < depth; l++) {
        const int64_t rc = (int64_t)(2 * (v - 1)) * a.N + i, rd = rc + a.N;
        const double c = cand[rc], d = cand[rd];
        const double fc = fit_objective(sse_sets[rc], c, a.w, a.mu), fd = fit_objective(sse_sets[rd], d, a.w, a.mu);
        if (fc < fd) { hi = d; v = 2 * v; } else { lo = c; v = 2 * v + 1; }
    }
    a.a[i] = lo;
    a.b[i] = hi;
    if (final) a.c[i] = 0.5 * (lo + hi);
}

// objective at the returned point
__global__ void fit_finish_kernel(FitArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.N) a.fc[i] = fit_objective(a.sse_c[i], a.c[i], a.w, a.mu);
}

__global__ void fill_kernel(int64_t N, double v, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = v;
}

hipError_t launch_fit(int phase, const FitArgs& a, int k, double x, hipStream_t s) {
    const int bs = 256;
    const dim3 grid((unsigned)((a.N + bs - 1) / bs));
    switch (phase) {
        case 0: hipLaunchKernelGGL(fit_grid_kernel, grid, dim3(bs), 0, s, a, k, x); break;
        case 1: hipLaunchKernelGGL(fit_bracket_kernel, grid, dim3(bs), 0, s, a); break;
        case 2: hipLaunchKernelGGL(fit_golden_kernel, grid, dim3(bs), 0, s, a, k); break;
        case 3: hipLaunchKernelGGL(fit_finish_kernel, grid, dim3(bs), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fit_grid_all(const FitArgs& a, int k0, int kn, const double* values, const double* sse_sets, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(fit_grid_all_kernel, dim3((unsigned)((a.N + bs - 1) / bs)), dim3(bs), 0, s, a, k0, kn, values, sse_sets);
    return hipGetLastError();
}
hipError_t launch_fit_tree(const FitArgs& a, int depth, int resolve, int final, double* cand, const double* sse_sets, hipStream_t s) {
    if (depth < 1 || depth > kFitSpecMaxDepth) return hipErrorInvalidValue;
    const int bs = 256;
    const dim3 grid((unsigned)((a.N + bs - 1) / bs));
    if (resolve) hipLaunchKernelGGL(fit_tree_resolve_kernel, grid, dim3(bs), 0, s, a, depth, final, cand, sse_sets);
    else hipLaunchKernelGGL(fit_tree_candidates_kernel, grid, dim3(bs), 0, s, a, depth, cand);
    return hipGetLastError();
}

// out[k][i] = values[k]: the conditional-parameter sets of a likelihood-profile scan (every subject at the same value)
__global__ void fill_rows_kernel(int64_t N, const double* __restrict__ values, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[(int64_t)blockIdx.y * N + i] = values[blockIdx.y];
}

hipError_t launch_fill_rows(int64_t N, int n_rows, const double* values, double* out, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((N + bs - 1) / bs), (unsigned)n_rows), dim3(bs), 0, s, N, values,
                       out);
    return hipGetLastError();
}

hipError_t launch_fill(int64_t N, double v, double* out, hipStream_t s) {
    const int bs = 256;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((N + bs - 1) / bs)), dim3(bs), 0, s, N, v, out);
    return hipGetLastError();
}

// Adam exactly as Optimisers.jl: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// x -= lr * (m / (1-b1^t)) / (sqrt(v / (1-b2^t)) + eps).  Skipped when any subject failed
// (g_nn[P+1] > 0): the reference's optimiser would see an Inf objective there.
__global__ void adam_kernel(AdamArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.g_nn[a.P + 1] > 0.0) return;
    // bias corrections of this step from the device-resident running powers b^t (advanced before this launch: by the
    // reduction / L2 kernel that finished the iteration's loss, or by adam_advance_kernel), so that a captured hipGraph
    // of the iteration needs no per-iteration kernel arguments
    a.c1 = 1.0 - a.state[0];
    a.c2 = 1.0 - a.state[1];
    double *x, *m, *v;
    double g;
    if (idx < a.N) {
        x = a.cond + idx; m = a.m_cond + idx; v = a.v_cond + idx; g = a.g_cond[idx];
    } else if (idx < a.N + a.P) {
        const int64_t q = idx - a.N;
        x = a.nn + q; m = a.m_nn + q; v = a.v_nn + q; g = a.g_nn[q];
    } else {
        return;
    }
    const double mm = fma(a.b1, *m, (1.0 - a.b1) * g);
    const double vv = fma(a.b2, *v, (1.0 - a.b2) * g * g);
    *m = mm;
    *v = vv;
    *x -= a.lr * (mm / a.c1) / (sqrt(vv / a.c2) + a.eps);
}

// The state advance on its own (bring-your-own-collective and RCCL flows without an L2 term: the pair is final only
// after the all-reduce).  (Folding it into adam_kernel instead -- the workgroup that finishes last advances the state,
// found by an arrival counter -- was measured: ~500 device-scope atomics on one address cost more than the launch they
// save, step tail 20 -> 23 us with a relaxed atomic, 33 us with the fence a hand-over of data would need.)
__global__ void adam_advance_kernel(TailAdvance adv, const double* g_tail) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    adam_advance(adv, g_tail[0], g_tail[1]);
}

hipError_t launch_adam_advance(const TailAdvance& adv, const double* g_tail, hipStream_t s) {
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, s, adv, g_tail);
    return hipGetLastError();
}

hipError_t launch_adam(const AdamArgs& a, hipStream_t s) {
    const int bs = 256;
    const int64_t n = a.N + a.P;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, a);
    return hipGetLastError();
}

}  // namespace cude
