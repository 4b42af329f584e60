// Device-side building blocks shared by the gfx950 kernels: Tsit5 tableau, fp64 activations,
// and the tiny tanh/softplus MLP (forward and weighted reverse sweep) with the shared weights
// read through the scalar unit.
//
// Reference behaviour being replaced (paths relative to the reference repo):
//   softplus            src/neural-network.jl:13-15
//   SimpleChains MLP    src/neural-network.jl:42-58  -- params per layer [vec_colmajor(W); b]
//   Tsit5               OrdinaryDiffEq (third-party, not vendored); tableau: Tsitouras 2011
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cude_kernels.h"
#include "cude_math.h"

namespace cude {

// ------------------------------------------------------------------------------------ Tsit5
// wave-uniform table for rolled stage loops (scalar loads with a run-time stage index)
__device__ __constant__ const double TS_A[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {0.161, 0, 0, 0, 0, 0},
    {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};

// compile-time copy for fully unrolled code (immediates / SGPR literals instead of loads)
struct Tab {
    static constexpr double a(int i, int j) {
        constexpr double A[7][6] = {
            {0, 0, 0, 0, 0, 0},
            {0.161, 0, 0, 0, 0, 0},
            {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
            {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
            {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
            {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
            {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
             2.324710524099774}};
        return A[i][j];
    }
    static constexpr double c(int i) {
        constexpr double C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
        return C[i];
    }
};

// ------------------------------------------------------------------------------------ math
// Scalar (wave-uniform) read of a network parameter: the constant address space makes the
// compiler emit s_load_* for uniform addresses, so weights live in SGPRs and feed v_fma_f64 as
// the scalar operand -- no VGPRs, no LDS bandwidth for the 37-67 shared doubles.
typedef const __attribute__((address_space(4))) double* cptr_t;
typedef const __attribute__((address_space(4))) int32_t* ciptr_t;
__device__ __forceinline__ cptr_t as_const(const double* p) {
    return (cptr_t)(uintptr_t)p;
}
__device__ __forceinline__ ciptr_t as_const(const int32_t* p) {
    return (ciptr_t)(uintptr_t)p;
}

// Makes the compiler forget what it knows about *p so that the (cheap, asynchronous) scalar loads
// of the weights are re-issued per network evaluation instead of being hoisted out of the time
// loop, where 55+ doubles exceed the 102-SGPR budget and spill to VGPR lanes.
__device__ __forceinline__ cptr_t launder(cptr_t p) {
    asm volatile("" : "+s"(p));
    return p;
}

// tanh of a layer (cude_math.h): two forms.
//  * table + addition theorem (round 3): 5 issue slots fewer per tanh than the exponential form, but ~35 more live
//    VGPRs per layer (numerator, denominator and the table read of every unit are in flight together).  The table of
//    tanh(k/8) sits in LDS -- per-lane index, 161 doubles = 3 rows of every workgroup's static allocation; a kernel that
//    evaluates such a network calls tanh_tab_init() first.
//  * exponential form: sign(x) (1 - 2/(exp(2|x|) + 1)).
// Chosen per kernel family through Mlp's TT parameter (CpepNet / SuppNet below), the same in every kernel a family's
// results are compared across bit for bit (one-lane and time-split c-peptide kernels; the suppression kernel's modes).
// Measured on MI355X (tools/abl_run.sh, same box): suppression 4-3x5-1 gradient launch at 1e5 subjects 1.012 -> 0.958 ms;
// 2-4-4-1 forward call at 125 000 subjects 0.245 -> 0.223 ms, its gradient launch unchanged (0.430 / 0.433 ms: three
// waves per SIMD become two); but the width-6 gradient kernels sit at the 256-register line of two waves per SIMD and
// pay for the extra live values with scratch traffic (2-6-6-1 0.564 -> 0.603 ms), and the adaptive kernels -- latency
// chains that want resident waves more than short instruction streams -- would drop from three waves per SIMD to two.
// Hence: fixed-step c-peptide kernels (two inputs) up to width CUDE_TANH_TAB_MAXW, the fixed-step suppression kernel
// always, the adaptive kernels and the covariate model never.
#ifndef CUDE_TANH_TAB_MAXW
#define CUDE_TANH_TAB_MAXW 4
#endif
#ifdef CUDE_TANH_EXP
constexpr int kTanhTabMaxW = 0;
#else
constexpr int kTanhTabMaxW = CUDE_TANH_TAB_MAXW;
#endif
__device__ const double TANH_TAB_G[kTanhEntries] = {CUDE_TANH_TABLE_VALUES};
constexpr int kTanhRows = (kTanhEntries + 63) / 64;
__shared__ double s_tanh_tab[kTanhRows * 64];
// (sync = false: the caller fills another LDS table right behind and synchronises once -- bias_lds_init; every global
// read of a kernel's prologue that is issued behind a wait costs a cold-cache round trip of its own, ~1 us)
__device__ __forceinline__ void tanh_tab_init(int lane, bool sync = true) {
#pragma unroll
    for (int r = 0; r < kTanhRows; r++) {
        const int k = lane + 64 * r;
        s_tanh_tab[k] = TANH_TAB_G[k < kTanhEntries ? k : kTanhEntries - 1];
    }
    if (sync) __syncthreads();
}
// Hidden-layer biases in LDS (networks with LB set: the suppression kernel).  A bias seeds the accumulator of its unit
// before the layer's FMAs; coming from the scalar unit it costs a v_mov_b64 -- a VALU slot -- per unit (fma(w, h, b) with w
// AND b in SGPRs would need two constant-bus reads, gfx950 allows one), 15 per evaluation of a 4-3x5-1 network; read from LDS
// (all lanes the same address: a broadcast) it lands in the VGPR pair without touching the VALU.
__shared__ double s_bias[64];
template <int W, int D, int L1, int LH>
__device__ __forceinline__ void bias_lds_init(const double* p, int lane) {
    static_assert((D - 1) * W <= 64, "bias row");
    if (lane < (D - 1) * W) s_bias[lane] = p[L1 + (lane / W) * LH + W * W + lane % W];      // hidden layer 1 + lane / W
    __syncthreads();
}

template <int W, bool TT>
__device__ __forceinline__ void act_tanh_vec(const double (&z)[W], double (&t)[W]) {
    if constexpr (TT) m_tanh_vec_tab<W>(z, t, s_tanh_tab);
    else m_tanh_vec<W>(z, t);
}

// ---- the other activation functions `chain(widths, activations; output_activation)` can be built with
// (src/neural-network.jl:42-58: any hidden activation, any output activation; the reference's scripts use tanh and
// softplus, which the specialised paths of Mlp are written for).  Codes = cude_kernels.h kActHidden* / kActOut*.
// hidden layer of W units; the derivative is always a function of the OUTPUT h (what the reverse sweep has at hand)
template <int W, int HA, bool TT>
__device__ __forceinline__ void act_hidden_vec(const double (&z)[W], double (&h)[W]) {
    if constexpr (HA == kActHiddenTanh) {
        act_tanh_vec<W, TT>(z, h);
    } else if constexpr (HA == kActHiddenRelu) {
#pragma unroll
        for (int j = 0; j < W; j++) h[j] = fmax(z[j], 0.0);
    } else if constexpr (HA == kActHiddenSigmoid) {
        // logistic function 1 / (1 + exp(-z)), all W units through ONE reciprocal (prefix products, as the tanh layer):
        // e = exp(-|z|) in (0, 1], sigma(|z|) = 1 / (1 + e), sigma(-|z|) = e / (1 + e)
        static_assert(W <= 8, "batched reciprocal");
        double e[W], d[W], pre[W];
#pragma unroll
        for (int j = 0; j < W; j++) {
            e[j] = m_exp2x(fmax(-0.5 * fabs(z[j]), -350.0));
            d[j] = 1.0 + e[j];
            pre[j] = j == 0 ? d[0] : pre[j - 1] * d[j];
        }
        double r = m_rcp(pre[W - 1]);
#pragma unroll
        for (int j = W - 1; j >= 0; j--) {
            const double inv = j > 0 ? r * pre[j - 1] : r;
            if (j > 0) r = r * d[j];
            h[j] = z[j] >= 0.0 ? inv : e[j] * inv;
        }
    } else {
#pragma unroll
        for (int j = 0; j < W; j++) h[j] = z[j];
    }
}
template <int HA>
__device__ __forceinline__ double act_hidden_deriv(double h) {
    if constexpr (HA == kActHiddenTanh) return fma(-h, h, 1.0);
    else if constexpr (HA == kActHiddenRelu) return h > 0.0 ? 1.0 : 0.0;       // (0 at the kink, as ForwardDiff's max)
    else if constexpr (HA == kActHiddenSigmoid) return h * (1.0 - h);
    else return 1.0;
}
// output unit: value and derivative
template <int OA, bool LONE>
__device__ __forceinline__ double act_out(double x, double* sig) {
    if constexpr (OA == kActOutSoftplus) {
        return m_softplus_t<LONE>(x, sig);
    } else {
        *sig = 1.0;
        return x;
    }
}

// softplus(x) = log(1+exp(x)) (reference form, evaluated stably); *sig receives the logistic derivative.
// (LONE: no other exponential in the kernel -- networks with the table tanh; see m_exp2x_t)
template <bool LONE>
__device__ __forceinline__ double act_softplus(double x, double* sig) { return m_softplus_t<LONE>(x, sig); }
template <bool LONE>
__device__ __forceinline__ double act_softplus_val(double x) {
    double sig;
    return m_softplus_t<LONE>(x, &sig);
}

// ------------------------------------------------------------------------------------ MLP
// Network NIN -> W (tanh) x D -> 1 (softplus).  The first NV inputs vary per evaluation; the
// remaining NIN-NV inputs are constant per subject (exp(conditional) [, age]) and are folded
// into a per-subject first-layer offset  c_j = b1_j + sum_{i>=NV} W1[j,i]*cst_i.
//
// Weight traffic: the shared parameters are wave-uniform, so they are read with scalar loads and
// used as SGPR operands.  A 2-6-6-1 network has 55 doubles per evaluation (110 SGPRs) which does
// not fit next to the activation constants in the 102-SGPR budget; left alone, the scheduler
// hoists every s_load to the top of the evaluation and spills ~60 SGPRs to VGPR lanes
// (v_writelane/v_readlane = 25 % extra VALU instructions in the reverse sweep).  Therefore the
// weights are streamed one SimpleChains column (W contiguous doubles) at a time and
// sched_barriers keep each column's loads next to its W independent FMAs.
template <int W>
struct SCol {
    double v[W];
};
template <int W>
__device__ __forceinline__ SCol<W> ld_col(cptr_t p, int off) {
    p = launder(p);   // opaque base per column: keeps the IR from merging/hoisting the column loads
    SCol<W> c;
#pragma unroll
    for (int j = 0; j < W; j++) c.v[j] = p[off + j];
    return c;
}
// A use of the first loaded double: scalar loads return out of order, so the only wait there is is lgkmcnt(0), and it
// is placed before the first use.  Touching a group BEFORE the next group's loads are issued keeps that wait from
// covering the new loads as well.
template <int W>
__device__ __forceinline__ void touch(const SCol<W>& c) {
    asm volatile("" ::"s"(c.v[0]));
}
#ifndef CUDE_NO_FENCE
#define CUDE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define CUDE_FENCE()
#endif

#ifndef CUDE_COLGROUP
#define CUDE_COLGROUP 3       // three columns per scalar load + wait: -1 % (125 000 subjects) ... -1.7 % (1e6) on 2-6-6-1
#endif

// Pins an accumulator right behind its update.  An accumulator feeds nothing inside the time loop, so LLVM sinks the
// whole outer-product update of every layer into the loop latch and keeps the operands (d, h of every layer) alive until
// there: ~100 VGPRs of the suppression gradient kernel (275 -> one wave per SIMD).  An empty volatile asm that reads and
// writes the register holds the FMA where it was written.
template <class A>
struct AccPin {
    __device__ static __forceinline__ void pin(A& acc, int q) { acc.pin(q); }
};
template <int N>
struct AccPin<double[N]> {
    __device__ static __forceinline__ void pin(double (&a)[N], int q) {
#ifndef CUDE_NO_ACC_PIN
        asm volatile("" : "+v"(a[q]));
#endif
    }
};

// A wave-uniform bias seeding its unit's accumulator: one v_mov_b64 (left alone, LLVM copies the SGPR pair with two
// v_mov_b32 -- two VALU slots per unit and evaluation).
__device__ __forceinline__ double seed_from_sgpr(double b) {
#ifndef CUDE_NO_SEED_MOV64
    double z;
    asm("v_mov_b64 %0, %1" : "=v"(z) : "s"(b));
    return z;
#else
    return b;
#endif
}

// HA / OA: hidden and output activation (kActHidden* / kActOut*); anything but tanh / softplus runs on the general
// evaluation path only (no exponent table, no register-resident weights, no pipelined weight stream, biases from the
// scalar unit): correct for every combination, tuned for none
template <int NIN, int W, int D, int NV, bool TT_ = false, bool LB = false, int HA = kActHiddenTanh, int OA = kActOutSoftplus>
struct Mlp {
    static constexpr bool GENERAL = (HA != kActHiddenTanh || OA != kActOutSoftplus);
    static constexpr bool TT = TT_ && HA == kActHiddenTanh;         // tanh by table
    static constexpr bool LDS_BIAS = LB && D >= 2 && !GENERAL;
    // fills s_bias for this workgroup's network (call once, before the first evaluation, where LDS_BIAS)
    __device__ static __forceinline__ void bias_init(const double* p, int lane) {
        if constexpr (LDS_BIAS) bias_lds_init<W, D, W * NIN + W, W * W + W>(p, lane);
    }
    // weight columns fetched per scalar load (a column = W doubles); W must be a multiple of it
    static constexpr int CG = (W % CUDE_COLGROUP == 0) ? CUDE_COLGROUP : 1;
    static constexpr bool USES_TANH = TT;           // kernels then call tanh_tab_init() before the first evaluation
    static constexpr int NC = NIN - NV;
    static constexpr int NCST = W;                  // per-subject constants kept in registers (first-layer offsets)
    // the conditional parameter enters the network as exp(beta) (src/c-peptide-models.jl:90)
    __device__ static __forceinline__ double cond_input(double raw) { return exp(raw); }
    static constexpr int L1 = W * NIN + W;          // first layer params
    static constexpr int LH = W * W + W;            // each further hidden layer
    static constexpr int OUT = L1 + (D - 1) * LH;   // output layer offset
    static constexpr int P = OUT + W + 1;
    // accumulator layout for the reverse sweep
    static constexpr int G_W1V = 0;                 // [NV][W]  d/dW1[j,i], i<NV
    static constexpr int G_C = G_W1V + NV * W;      // [W]      d/dc_j
    static constexpr int G_H = G_C + W;             // (D-1) x (W*W + W)
    static constexpr int G_OUT = G_H + (D - 1) * LH;  // [W] + 1
    static constexpr int NACC = G_OUT + W + 1;

    __device__ static __forceinline__ void first_layer_offset(cptr_t p, const double (&cst)[NC > 0 ? NC : 1],
                                                              double (&c)[W]) {
#pragma unroll
        for (int j = 0; j < W; j++) {
            double z = p[W * NIN + j];
#pragma unroll
            for (int i = 0; i < NC; i++) z = fma(p[j + W * (NV + i)], cst[i], z);
            c[j] = z;
        }
    }

    // 0 when every parameter is finite, NaN otherwise.  The clamped activations swallow
    // NaN/Inf, so non-finite inputs are tracked explicitly to honour the reference's failure
    // convention (non-finite solve => loss = Inf).
    // (A vector form -- one parameter per lane, all loads in flight together, and a wave vote instead of this rolled
    // loop of s_load + wait -- was measured: -1 % on the latency-bound time-split forward launch, but its vote changes
    // the scalar register allocation of the one-lane gradient kernel, 13 -> 25 v_readlane per reverse evaluation,
    // +2 % there.  Kept behind CUDE_PARAM_CHECK_VECTOR.)
    __device__ static __forceinline__ double param_check(cptr_t p) {
#ifdef CUDE_PARAM_CHECK_VECTOR
        const int lane = (int)threadIdx.x & 63;
        bool bad = false;
#pragma unroll
        for (int q0 = 0; q0 < P; q0 += 64) {
            const int q = q0 + lane;
            const double v = p[q < P ? q : P - 1];
            bad |= !(fabs(v) <= 1.79769313486231570815e308);          // NaN or Inf
        }
        return __any(bad) ? __builtin_nan("") : 0.0;
#else
        double chk = 0.0;
        constexpr int P8 = P / 8 * 8;
#pragma unroll 1
        for (int q0 = 0; q0 < P8; q0 += 8) {      // 8 doubles = one s_load_dwordx16 per trip
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                c0 = fma(p[q0 + q], 0.0, c0);
                c1 = fma(p[q0 + q + 1], 0.0, c1);
            }
            chk += c0 + c1;
        }
#pragma unroll
        for (int q = P8; q < P; q++) chk = fma(p[q], 0.0, chk);
        return chk;
#endif
    }

    // ---- layer-1 exponent table (NV == 1: the only varying input is the forcing x(t)).
    // Along a linear piece of x(t) the layer-1 exponentials exp(2 z_j), z_j = W1[j,0] x + c_j, form a geometric
    // sequence in t, so inside a run of steps that stay within one piece they follow from an anchor value by ONE
    // multiply with a tabulated factor instead of a 20-instruction exponential: E_j(e) = A_j * T[s][j].
    // Measured on MI355X (tools/abl_bench.py): 2-6-6-1 gradient kernel -8.7 % at 125 000 subjects, -6 % at 1e6;
    // forward kernel -9 % / -1.5 %.  For width 4 the 36 extra VGPRs drop the kernel from 3 to 2 waves per SIMD and
    // the gain is lost (+2.7 % at 1e6); for width 8 (351 VGPRs, one wave per SIMD, 25 KB of LDS per wave) the
    // exposed LDS latency makes it slower (+6 % / +18 %).  So the table is compiled in for widths 6 and 7 only.
    static constexpr bool HAS_TAB = (NV == 1 && W >= 6 && W <= 7) && !GENERAL;
    struct Exps {
        double v[W];
    };
    // T[s][j] = exp(2 W1[j,0] * dx * coef[s]) -> s_tab[(s*W + j)*64 + lane]   (dx = change of x over one step)
    __device__ static __forceinline__ void tab_build(cptr_t p, double dx, const double (&coef)[5], double* s_tab,
                                                     int lane) {
        const SCol<W> col = ld_col<W>(p, 0);
#pragma unroll 1
        for (int s = 0; s < 5; s++) {
            const double cs = dx * coef[s];
#pragma unroll
            for (int j = 0; j < W; j++)
                s_tab[(s * W + j) * 64 + lane] = m_exp2x(fmin(fmax(col.v[j] * cs, -300.0), 300.0));
        }
    }
    // A_j = exp(2 (W1[j,0] x + c_j))
    __device__ static __forceinline__ void tab_anchor(cptr_t p, const double (&c)[W], double x, Exps& A) {
        const SCol<W> col = ld_col<W>(p, 0);
#pragma unroll
        for (int j = 0; j < W; j++) A.v[j] = m_exp2x(fmin(fmax(fma(col.v[j], x, c[j]), -340.0), 340.0));
    }
    // The recurrence is exact (every anchor, factor and product representable, nothing clamped) iff the
    // pre-activations stay within +-300 over the whole linear piece [x0, x0+dx] and move by at most 300 per step.
    // Returns false for a lane where that does not hold (absurdly large first-layer weights); the kernel then runs
    // the whole wave's run with direct exponentials, whose clamp at |z| = 20 is exact for any magnitude.
    __device__ static __forceinline__ bool tab_safe(cptr_t p, const double (&c)[W], double x0, double dx, double hl) {
        const SCol<W> col = ld_col<W>(p, 0);
        double worst = 0.0;
#pragma unroll
        for (int j = 0; j < W; j++) {
            const double zs = fma(col.v[j], x0, c[j]);
            worst = fmax(worst, fmax(fabs(zs), fabs(fma(col.v[j], dx, zs))));
            worst = fmax(worst, fabs(col.v[j] * dx * hl));
        }
        return worst <= 300.0;                          // false for NaN as well
    }

    // hidden activations of all layers: h[l][j]; returns the output pre-activation.
    // use_tab: the layer-1 activations come from the exponentials E1 instead of x.
    __device__ static __forceinline__ double forward(cptr_t p, const double (&c)[W], const double (&x)[NV],
                                                     double (&h)[D][W], bool use_tab = false,
                                                     const Exps* E1 = nullptr) {
        double z[W];
        if (use_tab) {
            m_tanh_from_exp<W>(E1->v, h[0]);
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const SCol<W> col = ld_col<W>(p, W * i);
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = fma(col.v[j], x[i], i == 0 ? c[j] : z[j]);
            }
            act_hidden_vec<W, HA, TT>(z, h[0]);
        }
#pragma unroll
        for (int l = 1; l < D; l++) {
            const int o = L1 + (l - 1) * LH;
            CUDE_FENCE();
            if constexpr (LDS_BIAS) {
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = s_bias[(l - 1) * W + j];
            } else {
                const SCol<W> b = ld_col<W>(p, o + W * W);
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = seed_from_sgpr(b.v[j]);
            }
#pragma unroll
            for (int i0 = 0; i0 < W; i0 += CG) {               // CG weight columns per scalar load + wait
                CUDE_FENCE();
                constexpr int kG = CG;
                const SCol<W * kG> col = ld_col<W * kG>(p, o + W * i0);
#pragma unroll
                for (int g = 0; g < kG; g++)
#pragma unroll
                    for (int j = 0; j < W; j++)
                        if (i0 + g < W) z[j] = fma(col.v[g * W + j], h[l - 1][i0 + g], z[j]);
            }
            CUDE_FENCE();
            act_hidden_vec<W, HA, TT>(z, h[l]);
        }
        CUDE_FENCE();
        const SCol<W> wo = ld_col<W>(p, OUT);
        double z0 = seed_from_sgpr(p[OUT + W]), z1 = 0.0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (i & 1) z1 = fma(wo.v[i], h[D - 1][i], z1);
            else z0 = fma(wo.v[i], h[D - 1][i], z0);
        }
        return z0 + z1;
    }

    // ---- layers 2..D and the output layer with the weights RESIDENT IN VGPRS (forward sweep of the gradient kernel:
    // its registers are sized by the reverse sweep's accumulators and sit idle during the forward sweep, so every lane
    // can afford its own copy of the (D-1)(W*W+W)+W+1 upper-layer weights -- no scalar loads, no s_waitcnt, no SGPR
    // pressure in the forward evaluations)
    // Measured on MI355X (tools/abl_bench.py, 2-6-6-1): gradient launch 0.641 -> 0.614 ms at 125 000 subjects, 4.56 ->
    // 4.30 ms at 1e6 (bit-identical results).  Enabled where it does not cost a resident wave: the 2-4-4-1 kernels sit
    // at 151 VGPRs = 3 waves per SIMD and would drop to 2.
    static constexpr int NVW = (D - 1) * LH + W + 1;
    static constexpr bool HAS_VW = (NV == 1 && NVW <= 49 && (W >= 6 || D >= 3)) && !GENERAL;
    // ... and in the launches that leave registers idle anyway (time-split forward chunks of a small population: at most
    // two waves per SIMD), also for the narrow networks
    static constexpr bool HAS_VW_SMALL = (NV == 1 && NVW <= 49) && !GENERAL;
    struct VW {
        double w[NVW];
    };
    __device__ static __forceinline__ void load_vw(cptr_t p, VW& v) {
#pragma unroll
        for (int q = 0; q < NVW; q++) {
            double t = p[L1 + q];
            asm volatile("" : "+v"(t));              // pin the copy in a VGPR pair (no re-materialising scalar load)
            v.w[q] = t;
        }
    }
    __device__ static __forceinline__ double eval_vw(cptr_t p, const VW& v, const double (&c)[W], const double (&x)[NV],
                                                     bool use_tab, const Exps* E1) {
        double h[W], z[W];
        if (use_tab) {
            m_tanh_from_exp<W>(E1->v, h);
        } else {
            p = launder(p);
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const SCol<W> col = ld_col<W>(p, W * i);
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = fma(col.v[j], x[i], i == 0 ? c[j] : z[j]);
            }
            act_tanh_vec<W, TT>(z, h);
        }
#pragma unroll
        for (int l = 1; l < D; l++) {
            const int o = (l - 1) * LH;
#pragma unroll
            for (int j = 0; j < W; j++) z[j] = v.w[o + W * W + j];
#pragma unroll
            for (int i = 0; i < W; i++)
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = fma(v.w[o + W * i + j], h[i], z[j]);
            act_tanh_vec<W, TT>(z, h);
        }
        constexpr int oo = (D - 1) * LH;
        double z0 = v.w[oo + W], z1 = 0.0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (i & 1) z1 = fma(v.w[oo + i], h[i], z1);
            else z0 = fma(v.w[oo + i], h[i], z0);
        }
        return act_softplus_val<TT>(z0 + z1);
    }

    // ---- kept activations (c-peptide gradient kernel, CpepArgs::act): the forward sweep hands out the tanh outputs of
    // the hidden layers 2..D and the output unit's logistic derivative, the reverse sweep forms layer 1 again (one
    // multiply per unit on the exponent-table path) and runs `backward` on them
    static constexpr int DEPTH = D, WIDTH = W;
    static constexpr int NKEEP = (D - 1) * W + 1;
    __device__ static __forceinline__ double eval_vw_keep(cptr_t p, const VW& v, const double (&c)[W],
                                                          const double (&x)[NV], bool use_tab, const Exps* E1,
                                                          double (&keep)[NKEEP]) {
        double h[W], z[W];
        if (use_tab) {
            m_tanh_from_exp<W>(E1->v, h);
        } else {
            p = launder(p);
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const SCol<W> col = ld_col<W>(p, W * i);
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = fma(col.v[j], x[i], i == 0 ? c[j] : z[j]);
            }
            act_tanh_vec<W, TT>(z, h);
        }
#pragma unroll
        for (int l = 1; l < D; l++) {
            const int o = (l - 1) * LH;
#pragma unroll
            for (int j = 0; j < W; j++) z[j] = v.w[o + W * W + j];
#pragma unroll
            for (int i = 0; i < W; i++)
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = fma(v.w[o + W * i + j], h[i], z[j]);
            act_tanh_vec<W, TT>(z, h);
#pragma unroll
            for (int j = 0; j < W; j++) keep[(l - 1) * W + j] = h[j];
        }
        constexpr int oo = (D - 1) * LH;
        double z0 = v.w[oo + W], z1 = 0.0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (i & 1) z1 = fma(v.w[oo + i], h[i], z1);
            else z0 = fma(v.w[oo + i], h[i], z0);
        }
        return act_softplus<TT>(z0 + z1, &keep[NKEEP - 1]);
    }
    // value and the output unit's logistic derivative (the only kept value of the KEEP = 1 gradient kernel)
    __device__ static __forceinline__ double eval_vw_sig(cptr_t p, const VW& v, const double (&c)[W],
                                                         const double (&x)[NV], bool use_tab, const Exps* E1,
                                                         double* sig) {
        double keep[NKEEP];
        const double y = eval_vw_keep(p, v, c, x, use_tab, E1, keep);
        *sig = keep[NKEEP - 1];
        return y;
    }
    // layer 1 alone
    __device__ static __forceinline__ void layer1(cptr_t p, const double (&c)[W], const double (&x)[NV], double (&h0)[W],
                                                  bool use_tab, const Exps* E1) {
        if (use_tab) {
            m_tanh_from_exp<W>(E1->v, h0);
        } else {
            p = launder(p);
            double z[W];
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const SCol<W> col = ld_col<W>(p, W * i);
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = fma(col.v[j], x[i], i == 0 ? c[j] : z[j]);
            }
            act_tanh_vec<W, TT>(z, h0);
        }
    }

    // value only
    __device__ static __forceinline__ double eval(cptr_t p, const double (&c)[W], const double (&x)[NV],
                                                  bool use_tab = false, const Exps* E1 = nullptr) {
        p = launder(p);
        double h[D][W];
        // (the pipelined weight stream of eval_grad_pf does not pay here: measured +1 ... +7 % on the forward-only
        // kernels, which run at 3-4 waves per SIMD and lose a wave or spill SGPRs to the extra groups in flight)
        double sig;
        return act_out<OA, TT>(forward(p, c, x, h, use_tab, E1), &sig);
    }

    // ---- value + weighted reverse sweep with a SOFTWARE-PIPELINED weight stream.
    // In `forward` / `backward` every column group is loaded right where it is used: the wave then sits out the whole
    // scalar-cache latency ~7 times per evaluation (s_load ... s_waitcnt lgkmcnt(0) back to back), which is most of
    // what a lone wave loses against the VALU issue rate; with two waves per SIMD the second one only partly fills
    // those holes.  Here the loads of the NEXT group (or the next layer's bias + first group, or the output weights)
    // are issued before the FMAs of the current one, and the first group of a layer before the previous layer's
    // activation functions, so each wait finds its data landed.  At most two groups are in flight.
    // Measured on MI355X (tools/abl_bench.py, A/B in one process, bit-identical results), gradient launch at 125 000
    // subjects: 2-6-6-1 0.6086 -> 0.5926 ms (-2.6 %; -1.4 % at 1e6, -1.2 % on the time-split path at 1e5), 2-4-4-1
    // -2.4 %, 3-4-4-1 -1.9 %, 2-8-8-1 -0.9 %, 2-4-4-4-1 -4.8 %; with the wait of a group placed before the next group's
    // request (`touch`) 0.5826 ms (-3.8 %; 2-8-8-1 -4 %); suppression 4-3x5-1 at 1e5 subjects 1.612 -> 1.564 ms.
    // Groups of two columns: with three in flight twice (36 + 36 SGPRs) the allocator spills to VGPR lanes and the gain
    // is lost (0.6088 ms).
#ifdef CUDE_PF_GROUP
    static constexpr int CGP = (W % CUDE_PF_GROUP == 0) ? CUDE_PF_GROUP : 1;
#else
    static constexpr int CGP = (W == 3) ? 3 : (W % 2 == 0) ? 2 : 1;  // columns per group of the pipelined stream
#endif
    static constexpr int NG = W / CGP;                // column groups per hidden layer
    // hidden-layer accumulators pinned behind their update (AccPin): where the network's input is the ODE state, the
    // evaluation is followed by stage-adjoint algebra the updates would otherwise be sunk behind
#ifdef CUDE_PIN_LAYERS
    static constexpr bool kPinLayers = CUDE_PIN_LAYERS;
#else
    static constexpr bool kPinLayers = (NV > 1);
#endif
    static constexpr bool HAS_PF = (D >= 2) && !GENERAL;
    // forward half: hidden activations h, output pre-activation returned; wo and (KEEP) the last hidden layer's last
    // column group stay loaded for the backward half
    // SKIP_OUT: the output unit is not evaluated (its logistic derivative is supplied: eval_grad_pf with SIG_IN)
    template <bool KEEP, bool SKIP_OUT = false>
    __device__ static __forceinline__ double forward_pf(cptr_t p, const double (&c)[W], const double (&x)[NV],
                                                        double (&h)[D][W], bool use_tab, const Exps* E1, SCol<W>& wo,
                                                        SCol<W * CGP>& glast) {
        double z[W];
        SCol<W> bias[D];                 // bias[l]: hidden layer l (l >= 1)
        SCol<W * CGP> grp[D][NG];        // grp[l][g]: column group g of hidden layer l (l >= 1)
        double bo = 0.0;
        if constexpr (!LDS_BIAS) bias[1] = ld_col<W>(p, L1 + W * W);
        grp[1][0] = ld_col<W * CGP>(p, L1);
        CUDE_FENCE();
        if (use_tab) {
            m_tanh_from_exp<W>(E1->v, h[0]);
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const SCol<W> col = ld_col<W>(p, W * i);
#pragma unroll
                for (int j = 0; j < W; j++) z[j] = fma(col.v[j], x[i], i == 0 ? c[j] : z[j]);
            }
            act_tanh_vec<W, TT>(z, h[0]);
        }
#pragma unroll
        for (int l = 1; l < D; l++) {
            const int o = L1 + (l - 1) * LH;
#pragma unroll
            for (int j = 0; j < W; j++) z[j] = LDS_BIAS ? s_bias[(l - 1) * W + j] : seed_from_sgpr(bias[l].v[j]);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                CUDE_FENCE();
                touch(grp[l][g]);        // the wait for THIS group goes here, before the next one is requested
                if (g + 1 < NG) {
                    grp[l][g + 1] = ld_col<W * CGP>(p, o + W * CGP * (g + 1));
                } else if (l + 1 < D) {
                    if constexpr (!LDS_BIAS) bias[l + 1] = ld_col<W>(p, o + LH + W * W);
                    grp[l + 1][0] = ld_col<W * CGP>(p, o + LH);
                } else {
                    wo = ld_col<W>(p, OUT);
                    bo = launder(p)[OUT + W];
                }
                CUDE_FENCE();
#pragma unroll
                for (int k = 0; k < CGP; k++)
#pragma unroll
                    for (int j = 0; j < W; j++) z[j] = fma(grp[l][g].v[k * W + j], h[l - 1][g * CGP + k], z[j]);
            }
            CUDE_FENCE();
            act_tanh_vec<W, TT>(z, h[l]);
        }
        CUDE_FENCE();
        // the last hidden layer's last column group is needed again right after the output unit
        if (KEEP) glast = ld_col<W * CGP>(p, L1 + (D - 2) * LH + W * CGP * (NG - 1));
        CUDE_FENCE();
        if (SKIP_OUT) return 0.0;
        double z0 = seed_from_sgpr(bo), z1 = 0.0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (i & 1) z1 = fma(wo.v[i], h[D - 1][i], z1);
            else z0 = fma(wo.v[i], h[D - 1][i], z0);
        }
        return z0 + z1;
    }

    // SIG_IN: the output unit's logistic derivative comes from the caller (kept by the forward sweep, CpepArgs::act);
    // neither the output pre-activation nor the softplus is evaluated (52 of ~375 instructions of a 2-6-6-1 reverse
    // evaluation) and the return value is 0
    template <bool WANT_DX, class A, bool SIG_IN = false, bool PIN = kPinLayers, int DX0 = 0>
    __device__ static __forceinline__ double eval_grad_pf(cptr_t p, const double (&c)[W], const double (&x)[NV],
                                                          double wgt, A& acc, double (&dx)[NV],
                                                          bool use_tab, const Exps* E1, double sig_in = 0.0) {
        p = launder(p);
        double h[D][W];
        SCol<W * CGP> grp[D][NG];        // grp[l][g]: column group g of hidden layer l (l >= 1), backward order
        SCol<W> wo;
        const double zo = forward_pf<true, SIG_IN>(p, c, x, h, use_tab, E1, wo, grp[D - 1][NG - 1]);
        double sig = sig_in;
        double y = 0.0;
        if (!SIG_IN) y = act_softplus<TT>(zo, &sig);
        // ------------------------------------------------ backward (column groups from the last to the first)
        const double dz = wgt * sig;
        acc[G_OUT + W] += dz;
        double dh[W];
#pragma unroll
        for (int i = 0; i < W; i++) {
            acc[G_OUT + i] = fma(dz, h[D - 1][i], acc[G_OUT + i]);
            dh[i] = dz * wo.v[i];
        }
#pragma unroll
        for (int l = D - 1; l >= 1; l--) {
            const int o = L1 + (l - 1) * LH;
            const int go = G_H + (l - 1) * LH;
            double d[W];
#pragma unroll
            for (int j = 0; j < W; j++) {
                d[j] = dh[j] * fma(-h[l][j], h[l][j], 1.0);
                acc[go + W * W + j] += d[j];
            }
#pragma unroll
            for (int g = NG - 1; g >= 0; g--) {
                CUDE_FENCE();
                touch(grp[l][g]);
                if (g > 0) grp[l][g - 1] = ld_col<W * CGP>(p, o + W * CGP * (g - 1));
                else if (l > 1) grp[l - 1][NG - 1] = ld_col<W * CGP>(p, o - LH + W * CGP * (NG - 1));
                CUDE_FENCE();
#pragma unroll
                for (int k = 0; k < CGP; k++) {
                    const int i = g * CGP + k;
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        acc[go + j + W * i] = fma(d[j], h[l - 1][i], acc[go + j + W * i]);
                        if (j & 1) s1 = fma(grp[l][g].v[k * W + j], d[j], s1);
                        else s0 = fma(grp[l][g].v[k * W + j], d[j], s0);
                    }
                    dh[i] = s0 + s1;
                }
                if (PIN) {
#pragma unroll
                    for (int k = 0; k < CGP; k++)
#pragma unroll
                        for (int j = 0; j < W; j++) AccPin<A>::pin(acc, go + j + W * (g * CGP + k));
                }
            }
            if (PIN) {
#pragma unroll
                for (int j = 0; j < W; j++) AccPin<A>::pin(acc, go + W * W + j);
            }
            CUDE_FENCE();
        }
#pragma unroll
        for (int j = 0; j < W; j++) {
            const double d = dh[j] * fma(-h[0][j], h[0][j], 1.0);
            dh[j] = d;
            acc[G_C + j] += d;
#pragma unroll
            for (int i = 0; i < NV; i++) acc[G_W1V + i * W + j] = fma(d, x[i], acc[G_W1V + i * W + j]);
        }
        if (WANT_DX) {
#pragma unroll
            for (int i = DX0; i < NV; i++) {
                CUDE_FENCE();
                const SCol<W> col = ld_col<W>(p, W * i);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int j = 0; j < W; j++) {
                    if (j & 1) s1 = fma(col.v[j], dh[j], s1);
                    else s0 = fma(col.v[j], dh[j], s0);
                }
                dx[i] += s0 + s1;
            }
        }
        return y;
    }

    // value + weighted reverse sweep:  acc += wgt * d(out)/d(params);  if WANT_DX,
    // dx[i] += wgt * d(out)/dx_i.  Returns the network output.
    // PIN: hidden-layer accumulators pinned behind their update (default: the network's own rule, kPinLayers; the adaptive
    // kernels ask for it -- their evaluation sits inside a conditional of the replay loop and, unpinned, every
    // accumulator is copied to another register at the end of it: 32 v_mov_b64 per evaluation of a 2-4-4-1 network)
    // DX0: first input whose derivative is wanted (the suppression model's input 1 has no adjoint: cude_supp.hip)
    template <bool WANT_DX, class A, bool PIN = kPinLayers, int DX0 = 0>
    __device__ static __forceinline__ double eval_grad(cptr_t p, const double (&c)[W], const double (&x)[NV],
                                                       double wgt, A& acc, double (&dx)[NV],
                                                       bool use_tab = false, const Exps* E1 = nullptr) {
#ifndef CUDE_NO_PREFETCH
        if constexpr (HAS_PF) return eval_grad_pf<WANT_DX, A, false, PIN, DX0>(p, c, x, wgt, acc, dx, use_tab, E1);
#endif
        p = launder(p);
        double h[D][W];
        const double zo = forward(p, c, x, h, use_tab, E1);
        double sig;
        const double y = act_out<OA, TT>(zo, &sig);
        backward<WANT_DX, A, PIN, DX0>(p, x, h, sig, wgt, acc, dx);
        return y;
    }

    // value with the activations kept for a later `backward`: h = tanh outputs of every layer, *sig = logistic
    // derivative of the output unit
    __device__ static __forceinline__ double eval_keep(cptr_t p, const double (&c)[W], const double (&x)[NV],
                                                       double (&h)[D][W], double* sig) {
        p = launder(p);
        return act_out<OA, TT>(forward(p, c, x, h), sig);
    }

    // weighted reverse sweep from kept activations:  acc += wgt * d(out)/d(params);  if WANT_DX,
    // dx[i] += wgt * d(out)/dx_i
    template <bool WANT_DX, class A, bool PIN = kPinLayers, int DX0 = 0>
    __device__ static __forceinline__ void backward(cptr_t p, const double (&x)[NV], const double (&h)[D][W], double sig,
                                                    double wgt, A& acc, double (&dx)[NV]) {
        const double dz = wgt * sig;
        acc[G_OUT + W] += dz;
        double dh[W];
        CUDE_FENCE();
        {
            const SCol<W> wo = ld_col<W>(p, OUT);
#pragma unroll
            for (int i = 0; i < W; i++) {
                acc[G_OUT + i] = fma(dz, h[D - 1][i], acc[G_OUT + i]);
                dh[i] = dz * wo.v[i];
            }
        }
#pragma unroll
        for (int l = D - 1; l >= 1; l--) {
            const int o = L1 + (l - 1) * LH;
            const int go = G_H + (l - 1) * LH;
            double d[W];
#pragma unroll
            for (int j = 0; j < W; j++) {
                d[j] = dh[j] * act_hidden_deriv<HA>(h[l][j]);
                acc[go + W * W + j] += d[j];
            }
#pragma unroll
            for (int i0 = 0; i0 < W; i0 += CG) {
                CUDE_FENCE();
                constexpr int kG = CG;
                const SCol<W * kG> col = ld_col<W * kG>(p, o + W * i0);
#pragma unroll
                for (int g = 0; g < kG; g++) {
                    const int i = i0 + g;
                    if (i < W) {
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int j = 0; j < W; j++) {
                            acc[go + j + W * i] = fma(d[j], h[l - 1][i], acc[go + j + W * i]);
                            if (j & 1) s1 = fma(col.v[g * W + j], d[j], s1);
                            else s0 = fma(col.v[g * W + j], d[j], s0);
                        }
                        dh[i] = s0 + s1;
                        if (PIN) {
#pragma unroll
                            for (int j = 0; j < W; j++) AccPin<A>::pin(acc, go + j + W * i);
                        }
                    }
                }
            }
            if (PIN) {
#pragma unroll
                for (int j = 0; j < W; j++) AccPin<A>::pin(acc, go + W * W + j);
            }
            CUDE_FENCE();
        }
#pragma unroll
        for (int j = 0; j < W; j++) {
            const double d = dh[j] * act_hidden_deriv<HA>(h[0][j]);
            dh[j] = d;
            acc[G_C + j] += d;
#pragma unroll
            for (int i = 0; i < NV; i++) acc[G_W1V + i * W + j] = fma(d, x[i], acc[G_W1V + i * W + j]);
        }
        if (WANT_DX) {
#pragma unroll
            for (int i = DX0; i < NV; i++) {
                CUDE_FENCE();
                const SCol<W> col = ld_col<W>(p, W * i);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int j = 0; j < W; j++) {
                    if (j & 1) s1 = fma(col.v[j], dh[j], s1);
                    else s0 = fma(col.v[j], dh[j], s0);
                }
                dx[i] += s0 + s1;
            }
        }
    }

    // Entry q of the gradient in the SimpleChains parameter order, read off the accumulators (q is a compile-time
    // constant wherever this is called from an unrolled loop, so it folds to one register or one multiply).
    template <class A>
    __device__ static __forceinline__ double grad_elem(int q, const A& acc,
                                                       const double (&cst)[NC > 0 ? NC : 1]) {
        if (q < W * NIN) {
            const int j = q % W, i = q / W;
            return i < NV ? acc[G_W1V + i * W + j] : acc[G_C + j] * cst[i - NV < NC ? i - NV : 0];
        }
        if (q < L1) return acc[G_C + (q - W * NIN)];
        return acc[G_H + (q - L1)];
    }
    // d/d(conditional) through cst[0] = exp(conditional):  cst0 * sum_j dc_j * W1[j,NV]
    template <class A>
    __device__ static __forceinline__ double grad_cond(cptr_t p, const A& acc,
                                                       const double (&cst)[NC > 0 ? NC : 1]) {
        double s = 0.0;
        if (NC > 0) {
#pragma unroll
            for (int j = 0; j < W; j++) s = fma(acc[G_C + j], p[j + W * NV], s);
            s *= cst[0];
        }
        return s;
    }

    // Expand the accumulators into the SimpleChains parameter order; also d/d(conditional) through
    // cst[0] = exp(conditional):  dcond = cst0 * sum_j dc_j * W1[j,NV].
    __device__ static __forceinline__ void expand(cptr_t p, const double (&acc)[NACC],
                                                  const double (&cst)[NC > 0 ? NC : 1], double (&g)[P], double* dcond) {
#pragma unroll
        for (int j = 0; j < W; j++) {
#pragma unroll
            for (int i = 0; i < NV; i++) g[j + W * i] = acc[G_W1V + i * W + j];
#pragma unroll
            for (int i = 0; i < NC; i++) g[j + W * (NV + i)] = acc[G_C + j] * cst[i];
            g[W * NIN + j] = acc[G_C + j];
        }
#pragma unroll
        for (int q = 0; q < (D - 1) * LH + W + 1; q++) g[L1 + q] = acc[G_H + q];
        double s = 0.0;
        if (NC > 0) {
#pragma unroll
            for (int j = 0; j < W; j++) s = fma(acc[G_C + j], p[j + W * NV], s);
            s *= cst[0];
        }
        *dcond = s;
    }
};

// the networks of the fixed-step kernel families with their tanh form (see act_tanh_vec)
// (not the covariate model, NIN = 3: its raw-age input (20 ... 79) saturates first-layer units, and where tanh is within
// 1e-8 of +-1 the quotient form's last-place errors show in 1 - h^2 -- gradients off by up to 2e-8 of their largest
// entry in a randomised sweep, tools/fuzz_parity.py; the exponential form 1 - 2/(E + 1) rounds like libm there)
#ifndef CUDE_CPEP_LDS_BIAS
#define CUDE_CPEP_LDS_BIAS 1
#endif
#ifndef CUDE_LDS_BIAS_MAXW
#define CUDE_LDS_BIAS_MAXW CUDE_TANH_TAB_MAXW
#endif
template <int NIN, int W, int D>
using CpepNet = Mlp<NIN, W, D, 1, (W <= kTanhTabMaxW && NIN == 2),
                    ((W <= kTanhTabMaxW || W <= CUDE_LDS_BIAS_MAXW) && NIN == 2 && CUDE_CPEP_LDS_BIAS != 0)>;
#ifdef CUDE_TANH_EXP
template <int W, int D>
using SuppNet = Mlp<4, W, D, 3, false>;
#else
#ifndef CUDE_SUPP_LDS_BIAS
#define CUDE_SUPP_LDS_BIAS 1
#endif
template <int W, int D>
using SuppNet = Mlp<4, W, D, 3, true, (CUDE_SUPP_LDS_BIAS != 0)>;
#endif

// Networks with other activation functions (Mlp::GENERAL): compiled for the shapes the reference's experiments use and
// the combinations below -- hidden tanh | relu | sigmoid, output softplus | identity, (tanh, softplus) being the
// specialised default.  Y(hidden, output)
#define CUDE_GENERAL_ACTS(Y) Y(0, 1) Y(1, 0) Y(1, 1) Y(2, 0) Y(2, 1)
template <int NIN, int W, int D, int HA, int OA>
using CpepNetG = Mlp<NIN, W, D, 1, false, false, HA, OA>;
template <int W, int D, int HA, int OA>
using SuppNetG = Mlp<4, W, D, 3, false, false, HA, OA>;
inline bool general_acts_compiled(int hact, int oact) {
#define Y(HA, OA) if (hact == HA && oact == OA) return true;
    CUDE_GENERAL_ACTS(Y)
#undef Y
    return false;
}

// ------------------------------------------------------------------------------------ analytic production
// Drop-in for Mlp<2, W, D, 1> in the c-peptide kernel: the production term found by symbolic regression,
//   production(dG, k) = dG >= 0 ? p0*dG/(dG + k) : 0,   p0 = 1.78 in the reference
// (c-peptide/03-symreg.jl:37-40, evaluated through analytic_production src/c-peptide-models.jl:68-75; the same
// term in src/saem-symreg.jl:23-29).  One shared parameter p0 and the per-subject k, which is the conditional
// parameter itself (RAW, 03-symreg.jl:99-106) or its exponential (saem-symreg.jl:57-59 km_pop*exp(eta)).
template <bool RAW>
struct MmProd {
    static constexpr bool USES_TANH = false;
    static constexpr bool LDS_BIAS = false;
    __device__ static __forceinline__ void bias_init(const double*, int) {}
    static constexpr int NC = 1, NCST = 1, P = 1;
    static constexpr int NACC = 2;                  // [d/dp0, d/dk]
    static constexpr bool HAS_TAB = false;          // one division per evaluation: nothing to tabulate
    static constexpr bool HAS_VW = false;
    static constexpr bool HAS_VW_SMALL = false;
    static constexpr int DEPTH = 0, WIDTH = 1, NKEEP = 1;
    struct VW {};
    struct Exps {
        double v[1];
    };
    __device__ static __forceinline__ double cond_input(double raw) { return RAW ? raw : exp(raw); }
    __device__ static __forceinline__ void first_layer_offset(cptr_t, const double (&cst)[1], double (&c)[1]) {
        c[0] = cst[0];
    }
    __device__ static __forceinline__ double param_check(cptr_t p) { return fma(p[0], 0.0, 0.0); }
    __device__ static __forceinline__ double eval(cptr_t p, const double (&c)[1], const double (&x)[1], bool = false,
                                                  const Exps* = nullptr) {
        const double v = (p[0] * x[0]) / (x[0] + c[0]);
        return x[0] >= 0.0 ? v : 0.0;
    }
    template <bool WANT_DX, class A, bool PIN = false>
    __device__ static __forceinline__ double eval_grad(cptr_t p, const double (&c)[1], const double (&x)[1],
                                                       double wgt, A& acc, double (&dx)[1], bool = false,
                                                       const Exps* = nullptr) {
        const double r = 1.0 / (x[0] + c[0]);
        const double f = x[0] * r;                  // d/dp0
        const bool pos = x[0] >= 0.0;
        const double p0 = p[0];
        acc[0] += pos ? wgt * f : 0.0;
        acc[1] += pos ? -(wgt * p0) * f * r : 0.0;  // d/dk
        if (WANT_DX) dx[0] += pos ? (wgt * p0) * c[0] * r * r : 0.0;
        return pos ? p0 * f : 0.0;
    }
    __device__ static __forceinline__ void expand(cptr_t, const double (&acc)[NACC], const double (&cst)[1],
                                                  double (&g)[P], double* dcond) {
        g[0] = acc[0];
        *dcond = RAW ? acc[1] : acc[1] * cst[0];
    }
    template <class A>
    __device__ static __forceinline__ double grad_elem(int, const A& acc, const double (&)[1]) { return acc[0]; }
    template <class A>
    __device__ static __forceinline__ double grad_cond(cptr_t, const A& acc, const double (&cst)[1]) {
        return RAW ? acc[1] : acc[1] * cst[0];
    }
};

// ------------------------------------------------------------------------------------ accumulator containers
// The gradient functions above take any container with operator[] for the NACC partial sums: a plain register array,
// or this one, which keeps the hidden layers' accumulators in registers and the first- and output-layer ones -- each
// touched once per evaluation, at the two ends of the backward sweep -- in LDS rows (lds = row base + lane).  For a
// kernel that already sits at one wave per SIMD with registers to spare nowhere (suppression 4-3x5-1: 64 accumulators
// = 128 VGPRs), every accumulator that leaves the register file is one the compiler no longer shuttles through AGPRs.
template <class Net, int NHL = 0>        // NHL: how many of the LAST hidden layers' accumulators also live in LDS
struct SplitAcc {
    static constexpr int HI = Net::G_OUT - NHL * Net::LH;            // accumulators [G_H, HI) stay in registers
    static constexpr int NREG = HI - Net::G_H;
    static constexpr int NLDS = Net::NACC - NREG;                    // first layer, last NHL hidden layers, output layer
    double r[NREG > 0 ? NREG : 1];
    double* lds;
    __device__ __forceinline__ double& operator[](int q) {
        return q < Net::G_H ? lds[q * 64] : (q >= HI ? lds[(q - NREG) * 64] : r[q - Net::G_H]);
    }
    __device__ __forceinline__ const double& operator[](int q) const {
        return q < Net::G_H ? lds[q * 64] : (q >= HI ? lds[(q - NREG) * 64] : r[q - Net::G_H]);
    }
    __device__ __forceinline__ void pin(int q) {
#ifndef CUDE_NO_ACC_PIN
        if (q >= Net::G_H && q < HI) asm volatile("" : "+v"(r[q - Net::G_H]));
#endif
    }
};

// ------------------------------------------------------------------------------------ reductions
constexpr int kBlockLanes = 64;
constexpr int kRedRows = 16;   // rows of the LDS transpose used by the wave reduction

// Sum v[0..NV) over the 64 lanes of the (single-wave) workgroup and store the sums to out[0..NV).
// Chunked LDS transpose: rolled code (small i-cache footprint), fixed order (deterministic).
template <int NV>
__device__ __forceinline__ void block_reduce_store(const double (&v)[NV], double* s_red, double* out, int lane,
                                                   double* out2 = nullptr /* second destination of the sums, or null */) {
#pragma unroll
    for (int c0 = 0; c0 < NV; c0 += kRedRows) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRedRows; r++)
            if (c0 + r < NV) s_red[r * kBlockLanes + lane] = v[c0 + r];
        __syncthreads();
        if (lane < kRedRows && c0 + lane < NV) {
            double acc = 0.0;
#pragma unroll 8
            for (int l = 0; l < kBlockLanes; l++) acc += s_red[lane * kBlockLanes + ((l + lane) & (kBlockLanes - 1))];
            out[c0 + lane] = acc;
            if (out2 != nullptr) out2[c0 + lane] = acc;
        }
    }
}

// The same reduction for a gradient that is still in accumulator form: rows are expanded kRedRows at a time straight
// into the LDS transpose, so the P-vector never exists in registers next to the accumulators (the epilogue used to be
// the register peak of the gradient kernels: 336 VGPRs = one wave per SIMD for the suppression model).
// out[0..P) = sum over lanes of keep * grad_elem(q); out[P], out[P+1] = sum of extra0, extra1.
template <class Net, int NCST, class A>
__device__ __forceinline__ void block_reduce_expand(const A& acc, const double (&cst)[NCST], double keep,
                                                    double extra0, double extra1, double* s_red, double* out, int lane) {
    constexpr int NV = Net::P + 2;
#pragma unroll
    for (int c0 = 0; c0 < NV; c0 += kRedRows) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRedRows; r++) {
            const int q = c0 + r;
            if (q < Net::P) s_red[r * kBlockLanes + lane] = Net::grad_elem(q, acc, cst) * keep;
            else if (q == Net::P) s_red[r * kBlockLanes + lane] = extra0;
            else if (q == Net::P + 1) s_red[r * kBlockLanes + lane] = extra1;
        }
        __syncthreads();
        if (lane < kRedRows && c0 + lane < NV) {
            double a = 0.0;
#pragma unroll 8
            for (int l = 0; l < kBlockLanes; l++) a += s_red[lane * kBlockLanes + ((l + lane) & (kBlockLanes - 1))];
            out[c0 + lane] = a;
        }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace cude
