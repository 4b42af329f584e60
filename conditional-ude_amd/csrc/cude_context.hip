// libcude_hip.so -- host side of the C ABI declared in include/cude.h: contexts, populations, parameters.
// Owns the device-resident population (subject-major SoA), the solver tables and the HIP stream; the launches are in
// cude_launch.hip, the optimisers in cude_optimise.hip, the multi-GPU exchange in cude_comm.hip.
#include <algorithm>

#include "cude_ctx.h"

namespace {

thread_local std::string g_err;

// ---------------------------------------------------------------------------------- solver tables
const double TA7[6] = {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
                       2.324710524099774};
// rows 2..6 of the tableau (row 7 = TA7): a(st, j), j < st
const double TA[7][6] = {{0, 0, 0, 0, 0, 0},
                         {0.161, 0, 0, 0, 0, 0},
                         {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
                         {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
                         {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
                         {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
                         {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
                          2.324710524099774}};
const double TC[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
const double TR[7][4] = {{1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216},
                         {0.0, 0.13169999999999998, -0.2234, 0.1017},
                         {0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253},
                         {0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902},
                         {0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928},
                         {0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661},
                         {0.0, 1.5, -4.0, 2.5}};

using cude::api::interp_weights;

// observation tau lies in step n with t_n < tau <= t_{n+1}  (tau = t_0 -> step 0, theta 0)
void locate_obs(const std::vector<double>& tp, int S, std::vector<int32_t>& step, std::vector<double>& w) {
    const int T = (int)tp.size();
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    step.resize(T);
    w.resize((size_t)T * 7);
    for (int i = 0; i < T; i++) {
        const double x = (tp[i] - t0) / h;
        int n = (int)std::ceil(x - 1e-9) - 1;
        if (n < 0) n = 0;
        if (n > S - 1) n = S - 1;
        step[i] = n;
        interp_weights((tp[i] - (t0 + n * h)) / h, &w[(size_t)i * 7]);
    }
}

// glucose segment + fraction for the 5 distinct stage times of every step (c2..c5 and 1)
void glucose_tables(const std::vector<double>& tp, int S, std::vector<int32_t>& seg, std::vector<double>& phi) {
    const int T = (int)tp.size();
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S;
    seg.resize((size_t)S * 5);
    phi.resize((size_t)S * 5);
    for (int n = 0; n < S; n++) {
        for (int s = 0; s < 5; s++) {
            const double t = (s < 4) ? (t0 + n * h) + TC[s + 1] * h : t0 + (n + 1) * h;
            int j = 0;
            while (j + 1 < T && tp[j + 1] <= t) j++;
            if (j > T - 2) j = T - 2;
            seg[(size_t)n * 5 + s] = j;
            phi[(size_t)n * 5 + s] = (t - tp[j]) / (tp[j + 1] - tp[j]);
        }
    }
}

// Suppression model, state 1: du1 = -0.4 u1 (suppression/src/suppression_model.jl:91).  Fixed-step Tsit5 on a scalar
// linear equation multiplies by constants of z = -0.4 h alone: stage input Y_st = u1_n rho_st with rho_0 = 1,
// rho_st = 1 + z sum_{j<st} a(st, j) rho_j, and u1_{n+1} = u1_n rho_6.  rho[e] = u1 at evaluation e = 6n + st relative
// to u1(t_0); obs_rho[oi] = the dense output at observation oi (step[oi], weights w[oi][0..6]) relative to u1(t_0).
void linear_state_tables(int S, double h, const std::vector<int32_t>& step, const std::vector<double>& w,
                         std::vector<double>& rho, std::vector<double>& obs_rho) {
    const double z = -0.4 * h;
    double r[7];
    r[0] = 1.0;
    for (int st = 1; st <= 6; st++) {
        double acc = 0.0;
        for (int j = 0; j < st; j++) acc = std::fma(TA[st][j], r[j], acc);
        r[st] = std::fma(z, acc, 1.0);
    }
    rho.assign((size_t)6 * S + 1, 1.0);
    std::vector<double> pw((size_t)S + 1, 1.0);             // R^n
    for (int n = 0; n < S; n++) {
        for (int st = 1; st <= 6; st++) rho[(size_t)6 * n + st] = pw[n] * r[st];
        pw[n + 1] = rho[(size_t)6 * n + 6];
    }
    const size_t T = step.size();
    obs_rho.assign(T, 1.0);
    for (size_t oi = 0; oi < T; oi++) {
        double acc = 0.0;
        for (int j = 0; j < 7; j++) acc = std::fma(w[oi * 7 + j], r[j], acc);
        obs_rho[oi] = pw[step[oi]] * std::fma(z, acc, 1.0);
    }
}

// Per-step tables of the layer-1 exponent recurrence (CpepArgs::stepk / stepd): a step is "inside" glucose piece j
// when [t_n, t_n+h] lies within [tp[j], tp[j+1]]; consecutive inside steps of one piece form a run.
void step_tables(const std::vector<double>& tp, int S, std::vector<int32_t>& k, std::vector<double>& d) {
    const int T = (int)tp.size();
    const double t0 = tp[0], h = (tp[T - 1] - tp[0]) / S, tol = 1e-9 * h;
    k.assign((size_t)S * 3, 0);
    d.assign((size_t)S * 3, 0.0);
    std::vector<int> piece(S, -1);
    for (int n = 0; n < S; n++) {
        const double ta = t0 + n * h, tb = t0 + (n + 1) * h;
        int j = 0;
        while (j + 1 < T - 1 && tp[j + 1] <= ta + tol) j++;
        const double len = tp[j + 1] - tp[j];
        if (ta >= tp[j] - tol && tb <= tp[j + 1] + tol) piece[n] = j;
        k[(size_t)n * 3 + 2] = j;
        d[(size_t)n * 3 + 0] = (ta - tp[j]) / len;
        d[(size_t)n * 3 + 1] = (tb - tp[j]) / len;
        d[(size_t)n * 3 + 2] = h / len;
    }
    // a run is re-anchored with fresh exponentials every kReanchor steps so that the rounding of the anchor
    // recurrence (one multiply per step) stays below 256 ulp however many steps a piece holds
    constexpr int kReanchor = 256;
    for (int n = 0, pos = 0; n < S; n++) {
        if (piece[n] < 0) { pos = 0; continue; }
        const bool cont = n > 0 && piece[n - 1] == piece[n] && pos + 1 < kReanchor;
        k[(size_t)n * 3 + 0] = cont ? 2 : 1;
        pos = cont ? pos + 1 : 0;
    }
    for (int n = S - 1, pos = 0; n >= 0; n--) {
        if (piece[n] < 0) { pos = 0; continue; }
        const bool cont = n + 1 < S && piece[n + 1] == piece[n] && pos + 1 < kReanchor;
        k[(size_t)n * 3 + 1] = cont ? 2 : 1;
        pos = cont ? pos + 1 : 0;
    }
}

}  // namespace

namespace cude {
namespace api {

int32_t fail(int32_t code, const std::string& msg) {
    g_err = msg;
    return code;
}

void interp_weights(double th, double* w) {
    if (std::fabs(th - 1.0) < 1e-12) {
        for (int j = 0; j < 6; j++) w[j] = TA7[j];
        w[6] = 0.0;
        return;
    }
    for (int i = 0; i < 7; i++) w[i] = ((TR[i][3] * th + TR[i][2]) * th + TR[i][1]) * th * th + TR[i][0] * th;
}


}  // namespace api
}  // namespace cude

// Page-locked host memory the device writes results into while the host watches the words change (finish_loss):
// coherent (fine-grained) and mapped, asked for explicitly -- the default flags leave coherence to the HIP_HOST_COHERENT
// environment, and non-coherent host memory would show the stores only at the end of the stream.
constexpr unsigned kWatchedHostFlags = hipHostMallocCoherent | hipHostMallocMapped;

namespace cude {
namespace api {

namespace {
struct OptionEntry {
    const char* name;       // cude_set_option
    const char* env;        // read once at cude_create
    int Options::*field;    // nullptr: parsed by hand (cpep_path, supp_ckpt)
    bool env_negates;       // CUDE_NO_X: presence of the variable = 0
    bool ablation;          // environment variable honoured in -DCUDE_ABLATION builds only
};
const OptionEntry kOptions[] = {
    {"cpep_path", "CUDE_CPEP_PATH", nullptr, false, false},
    {"cpep_keep", "CUDE_CPEP_KEEP", &Options::cpep_keep, false, false},
    {"supp_store", "CUDE_SUPP_STORE", &Options::supp_store, false, false},
    {"supp_ckpt", "CUDE_SUPP_CKPT", nullptr, false, false},
    {"tape_steps", "CUDE_TAPE_STEPS", &Options::tape_steps, false, false},
    {"exp_table", "CUDE_NO_EXPTAB", &Options::exp_table, true, false},
    {"ms_split", "CUDE_NO_MS_SPLIT", &Options::ms_split, true, false},
    {"train_host", "CUDE_TRAIN_HOST", &Options::train_host, false, false},
    {"mh_spec", "CUDE_MH_SPEC", &Options::mh_spec, false, false},
    {"fit_spec", "CUDE_FIT_SPEC", &Options::fit_spec, false, false},
    {"adaptive_team", "CUDE_NO_ADAPTIVE_TEAM", &Options::adaptive_team, true, false},
    {"auto_regroup", "CUDE_NO_AUTO_REGROUP", &Options::auto_regroup, true, false},
    {"poll_pinned", "CUDE_NO_POLL_PINNED", &Options::poll_pinned, true, false},
    {"force_fallback", "CUDE_FORCE_FALLBACK", &Options::force_fallback, false, false},
    {"debug_selector", "CUDE_DEBUG_SELECTOR", &Options::debug_selector, false, false},
    {"xchg_allow_plain", "CUDE_ALLOW_PLAIN_MAILBOX", &Options::xchg_allow_plain, false, false},
    {"xchg_fail_kinds", "CUDE_XCHG_FAIL_KINDS", &Options::xchg_fail_kinds, false, false},
    {"mixed", "CUDE_NO_MIXED", &Options::mixed, true, true},
    {"mixed_one_stream", "CUDE_MIXED_ONE_STREAM", &Options::mixed_one_stream, false, true},
    {"fwd_split", "CUDE_NO_FWD_SPLIT", &Options::fwd_split, true, true},
    {"fused_final", "CUDE_NO_FUSED_FINAL", &Options::fused_final, true, true},
    {"fused_tail", "CUDE_NO_FUSED_TAIL", &Options::fused_tail, true, true},
    {"scan_map", "CUDE_NO_SCAN_MAP", &Options::scan_map, true, true},
    {"scan_bulk", "CUDE_NO_SCAN_BULK", &Options::scan_bulk, true, true},
    {"mh_fuse", "CUDE_NO_MH_FUSE", &Options::mh_fuse, true, true},
    {"mh_pair", "CUDE_NO_MH_PAIR", &Options::mh_pair, true, true},
    {"graph", "CUDE_NO_GRAPH", &Options::graph, true, true},
    {"graph_unroll", "CUDE_GRAPH_UNROLL", &Options::graph_unroll, false, true},
    {"prio_shift", "CUDE_PRIO_SHIFT", &Options::prio_shift, false, true},
};
}  // namespace

// `chain(widths, activation_functions; input_dims, output_activation)` (src/neural-network.jl:42-58) as the fallback
// kernel's network: n_hidden widths, n_hidden + 1 activation codes (CUDE_ACT_*; the last one is the output layer's)
int32_t make_general_network(int32_t model, int32_t nin, int32_t n_hidden, const int32_t* widths, const int32_t* acts,
                             cude::GenNet* out) {
    if (n_hidden < 1 || !widths || !acts) return fail(CUDE_ERR_ARG, "a network needs at least one hidden layer");
    if (n_hidden + 1 > cude::kGenMaxLayers)
        return fail(CUDE_ERR_UNSUPPORTED, "more than " + std::to_string(cude::kGenMaxLayers - 1) + " hidden layers");
    const int want_in = model == CUDE_MODEL_SUPP ? 4 : nin;
    if (model == CUDE_MODEL_SUPP ? nin != 4 : (nin != 2 && nin != 3))
        return fail(CUDE_ERR_ARG, "network inputs: 2 (glucose, conditional) or 3 (+ age) for the c-peptide model, 4 for the suppression model");
    cude::GenNet g;
    g.nin = want_in;
    g.n_layers = n_hidden + 1;
    for (int l = 0; l <= n_hidden; l++) {
        g.width[l] = l < n_hidden ? widths[l] : 1;
        g.act[l] = acts[l];
        if (g.width[l] < 1 || g.width[l] > 4096) return fail(CUDE_ERR_ARG, "layer widths must lie in 1 ... 4096");
        if (acts[l] < cude::kGenActTanh || acts[l] > cude::kGenActIdentity)
            return fail(CUDE_ERR_ARG, "activation codes: 0 tanh, 1 relu, 2 sigmoid, 3 softplus, 4 identity");
    }
    if (cude::gen_lds_bytes(g) > cude::kGenMaxLds)
        return fail(CUDE_ERR_UNSUPPORTED, "this network needs " + std::to_string(cude::gen_lds_bytes(g) / 1024) +
                                              " KB of LDS per workgroup (weights + one activation column per lane); 160 KB fit");
    *out = g;
    return CUDE_OK;
}

// P-sized buffers of a context (cude_create; again when cude_set_network changes the network)
int32_t alloc_network_buffers(cude_ctx* c) {
    const int P = c->P;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr;
    if (hipHostMalloc((void**)&c->pinned, (size_t)(P + 2) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
        c->pinned = nullptr;
    (void)hipGetLastError();
    if (c->nn.resize(P) || c->g_nn.resize(P + 2) || c->m_nn.resize(P) || c->v_nn.resize(P)) return fail(CUDE_ERR_HIP, "hipMalloc failed");
    (void)hipMemsetAsync(c->g_nn.p, 0, (P + 2) * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->m_nn.p, 0, P * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->v_nn.p, 0, P * sizeof(double), c->stream);
    return CUDE_OK;
}

int32_t apply_option(cude_ctx* c, const char* name, const char* value) {
    if (!name || !value) return fail(CUDE_ERR_ARG, "null option name / value");
    // activation functions of `chain(widths, activations; output_activation)` (src/neural-network.jl:42-58): part of the
    // network's shape, fixed before the population is uploaded
    const bool hidden = std::strcmp(name, "hidden_activation") == 0;
    if (hidden || std::strcmp(name, "output_activation") == 0) {
        if (c->have_pop) return fail(CUDE_ERR_STATE, "set the activation functions before the population");
        if (c->cfg.model == CUDE_MODEL_CPEP_SYM) return fail(CUDE_ERR_ARG, "the symbolic model has no network");
        cude::NetShape net = c->net;
        if (hidden) {
            if (std::strcmp(value, "tanh") == 0) net.hact = cude::kActHiddenTanh;
            else if (std::strcmp(value, "relu") == 0) net.hact = cude::kActHiddenRelu;
            else if (std::strcmp(value, "sigmoid") == 0) net.hact = cude::kActHiddenSigmoid;
            else return fail(CUDE_ERR_UNSUPPORTED, std::string("hidden activation not compiled in: ") + value +
                                                       " (tanh, relu, sigmoid)");
        } else {
            if (std::strcmp(value, "softplus") == 0) net.oact = cude::kActOutSoftplus;
            else if (std::strcmp(value, "identity") == 0) net.oact = cude::kActOutIdentity;
            else return fail(CUDE_ERR_UNSUPPORTED, std::string("output activation not compiled in: ") + value +
                                                       " (softplus, identity)");
        }
        const bool ok = c->cfg.model == CUDE_MODEL_SUPP ? cude::supp_shape_supported(net)
                                                        : cude::cpep_shape_supported(net, c->cfg.n_state);
        if (!ok) {          // no tuned kernel for this shape with these functions: the fallback kernel
            const int Dh = net.generic() ? net.gen.n_layers - 1 : net.depth;
            std::vector<int32_t> w((size_t)Dh), act((size_t)Dh + 1);
            const int hcode = net.hact == cude::kActHiddenRelu ? cude::kGenActRelu
                              : (net.hact == cude::kActHiddenSigmoid ? cude::kGenActSigmoid : cude::kGenActTanh);
            for (int l = 0; l < Dh; l++) {
                w[(size_t)l] = net.generic() ? net.gen.width[l] : net.width;
                act[(size_t)l] = hidden || !net.generic() ? hcode : net.gen.act[l];
            }
            act[(size_t)Dh] = !hidden || !net.generic() ? (net.oact == cude::kActOutIdentity ? cude::kGenActIdentity : cude::kGenActSoftplus)
                                                        : net.gen.act[Dh];
            int32_t rc = make_general_network(c->cfg.model, c->cfg.nn_in, Dh, w.data(), act.data(), &net.gen);
            if (rc) return rc;
        }
        c->net = net;
        drop_graph(c);
        return CUDE_OK;
    }
    for (const OptionEntry& o : kOptions) {
        if (std::strcmp(o.name, name) != 0) continue;
        Options& opt = c->opt;
        if (o.field) {
            char* end = nullptr;
            const long v = std::strtol(value, &end, 10);
            if (end == value) return fail(CUDE_ERR_ARG, std::string("option ") + name + ": not an integer: " + value);
            opt.*(o.field) = (int)v;
        } else if (std::strcmp(name, "supp_ckpt") == 0) {
            opt.supp_ckpt_steps = std::strcmp(value, "steps") == 0 ? 1 : 0;
        } else {        // cpep_path: "" | "0" | "1" | "2:L" | "3:B:L"
            opt.cpep_path = 0; opt.path_chunks = 0; opt.path_blk0 = 0;
            if (value[0] == '1') opt.cpep_path = 1;
            else if (value[0] == '2' && value[1] == ':') { opt.cpep_path = 2; opt.path_chunks = atoi(value + 2); }
            else if (value[0] == '3' && value[1] == ':') {
                opt.cpep_path = 3;
                opt.path_blk0 = atoll(value + 2);
                const char* q = std::strchr(value + 2, ':');
                opt.path_chunks = q ? atoi(q + 1) : 2;
            } else if (value[0] != '\0' && value[0] != '0') {
                return fail(CUDE_ERR_ARG, std::string("option cpep_path: expected \"1\", \"2:L\" or \"3:B:L\", got ") + value);
            }
        }
        drop_graph(c);          // (captured iterations were built under the old options)
        return CUDE_OK;
    }
    return fail(CUDE_ERR_ARG, std::string("unknown option: ") + name);
}

// the one place the library reads its own environment variables (cude_comm.hip checks the runtime's IPC mode besides)
static void options_from_environment(cude_ctx* c) {
    for (const OptionEntry& o : kOptions) {
#ifndef CUDE_ABLATION
        if (o.ablation) continue;
#endif
        const char* v = getenv(o.env);
        if (v) (void)apply_option(c, o.name, o.env_negates ? "0" : v);
    }
}

}  // namespace api
}  // namespace cude

using namespace cude::api;

namespace {

int32_t alloc_common(cude_ctx* c) {
    const int64_t N = c->N;
    drop_graph(c);
    c->nblocks = (N + cude::kBlock - 1) / cude::kBlock;
    HIP_TRY(c->cond.resize(N));
    HIP_TRY(c->g_cond.resize(N));
    HIP_TRY(c->sse.resize(N));
    HIP_TRY(c->partials.resize((size_t)c->nblocks * (c->P + 2)));
    if (c->pinned_pairs_n < c->nblocks && c->nblocks <= 8192) {      // (bigger populations are not launch-bound)
        if (c->pinned_pairs) (void)hipHostFree(c->pinned_pairs);
        c->pinned_pairs = nullptr;
        c->pinned_pairs_n = 0;
        if (hipHostMalloc((void**)&c->pinned_pairs, (size_t)c->nblocks * 2 * sizeof(double), kWatchedHostFlags) ==
            hipSuccess)
            c->pinned_pairs_n = c->nblocks;
        else
            c->pinned_pairs = nullptr;
    }
    HIP_TRY(c->perm.resize(0));          // a new population starts in its own order
    c->slot_of.clear();
    HIP_TRY(c->tape.resize(0));          // adaptive gradient tape: allocated by the first gradient evaluation
    HIP_TRY(c->gen_acc.resize(0));
    c->tape_cap = 0;
    HIP_TRY(c->tape_n.resize(adaptive(c) ? (size_t)N : 0));      // accepted-step counts: written by every adaptive launch
    c->have_tape = false;
    c->have_counts = false;
    c->evals_since_regroup = 0;
    c->run_iters = 0;
    c->regroup_done_at = -1;
    HIP_TRY(c->m_cond.resize(N));
    HIP_TRY(c->v_cond.resize(N));
    HIP_TRY(hipMemsetAsync(c->cond.p, 0, N * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->m_cond.p, 0, N * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->v_cond.p, 0, N * sizeof(double), c->stream));
    c->have_cond = false;
    c->adam_t = 0;
    return CUDE_OK;
}

int32_t upload_tables(cude_ctx* c, bool glucose) {
    HIP_TRY(c->tp_dev.resize(c->tp.size()));
    HIP_TRY(hipMemcpyAsync(c->tp_dev.p, c->tp.data(), c->tp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (adaptive(c)) {                       // no step grid: the kernels locate knots and outputs themselves
        HIP_TRY(hipStreamSynchronize(c->stream));
        return CUDE_OK;
    }
    std::vector<int32_t> step;
    std::vector<double> w;
    locate_obs(c->tp, c->cfg.n_steps, step, w);
    HIP_TRY(c->obs_step.resize(step.size()));
    HIP_TRY(c->obs_w.resize(w.size()));
    HIP_TRY(hipMemcpyAsync(c->obs_step.p, step.data(), step.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->obs_w.p, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (!glucose) {                          // suppression model: the closed-form Tsit5 factors of its first state
        std::vector<double> rho, obs_rho;
        linear_state_tables(c->cfg.n_steps, step_size(c), step, w, rho, obs_rho);
        HIP_TRY(c->supp_rho.resize(rho.size()));
        HIP_TRY(c->supp_obs_rho.resize(obs_rho.size()));
        HIP_TRY(hipMemcpyAsync(c->supp_rho.p, rho.data(), rho.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->supp_obs_rho.p, obs_rho.data(), obs_rho.size() * sizeof(double), hipMemcpyHostToDevice,
                               c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));          // rho / obs_rho die at the end of this block
    }
    if (glucose) {
        std::vector<int32_t> seg;
        std::vector<double> phi;
        glucose_tables(c->tp, c->cfg.n_steps, seg, phi);
        HIP_TRY(c->seg.resize(seg.size()));
        HIP_TRY(c->phi.resize(phi.size()));
        HIP_TRY(hipMemcpyAsync(c->seg.p, seg.data(), seg.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->phi.p, phi.data(), phi.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        std::vector<int32_t> sk;
        std::vector<double> sd;
        step_tables(c->tp, c->cfg.n_steps, sk, sd);
        if (!c->opt.exp_table)                             // option "exp_table" = 0: direct exponentials everywhere
            for (size_t q = 0; q < sk.size(); q += 3) sk[q] = sk[q + 1] = 0;
        HIP_TRY(c->stepk.resize(sk.size()));
        HIP_TRY(c->stepd.resize(sd.size()));
        HIP_TRY(hipMemcpyAsync(c->stepk.p, sk.data(), sk.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->stepd.p, sd.data(), sd.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));          // sk / sd die at the end of this block
    }
    HIP_TRY(hipStreamSynchronize(c->stream));   // host vectors die here
    return CUDE_OK;
}

int32_t check_times(int32_t n_obs, const double* tp) {
    if (n_obs < 2 || n_obs > cude::kMaxObs) return fail(CUDE_ERR_ARG, "n_obs must be in [2, 32]");
    for (int t = 1; t < n_obs; t++)
        if (!(tp[t] > tp[t - 1])) return fail(CUDE_ERR_ARG, "timepoints must be strictly increasing");
    return CUDE_OK;
}

}  // namespace

extern "C" {

const char* cude_last_error(void) { return g_err.c_str(); }

int32_t cude_device_count(int32_t* count) {
    if (!count) return fail(CUDE_ERR_ARG, "null count");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    *count = n;
    return CUDE_OK;
}

int32_t cude_n_params(int32_t nn_in, int32_t nn_width, int32_t nn_depth) {
    if (nn_width == 0 && nn_depth == 0) return 1;   // analytic production model: [p0]
    if (nn_in < 1 || nn_width < 1 || nn_depth < 1) return fail(CUDE_ERR_ARG, "bad network shape");
    cude::NetShape n{nn_in, nn_width, nn_depth};
    return n.n_params();
}

int32_t cude_create(const cude_config* cfg, cude_ctx** out) {
    if (!cfg || !out) return fail(CUDE_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->n_steps < 0 || cfg->n_steps > 100000) return fail(CUDE_ERR_ARG, "n_steps out of range");
    if (cfg->n_steps == 0 && cfg->model != CUDE_MODEL_SUPP && cfg->n_state != 2)
        return fail(CUDE_ERR_UNSUPPORTED, "adaptive mode (n_steps = 0) integrates the reference's 2-state c-peptide model");
    cude::NetShape net{cfg->nn_in, cfg->nn_width, cfg->nn_depth};
    if (cfg->model == CUDE_MODEL_CPEP_SYM) {
        if (cfg->nn_width != 0 || cfg->nn_depth != 0)
            return fail(CUDE_ERR_ARG, "the symbolic model has no network: nn_width and nn_depth must be 0");
        if (cfg->cond_space != CUDE_COND_LOG && cfg->cond_space != CUDE_COND_RAW)
            return fail(CUDE_ERR_ARG, "cond_space must be CUDE_COND_LOG or CUDE_COND_RAW");
        if (cfg->n_state != 2 && cfg->n_state != 3) return fail(CUDE_ERR_UNSUPPORTED, "n_state must be 2 or 3");
        net = cude::NetShape{1, 0, 0};
    } else if (cfg->model == CUDE_MODEL_CPEP || cfg->model == CUDE_MODEL_SUPP) {
        if (cfg->nn_in < 1 || cfg->nn_width < 1 || cfg->nn_depth < 1) return fail(CUDE_ERR_ARG, "bad network shape");
        if (cfg->cond_space != CUDE_COND_LOG)
            return fail(CUDE_ERR_ARG, "cond_space must be CUDE_COND_LOG for the network models");
        if (cfg->model == CUDE_MODEL_CPEP && cfg->n_state != 2 && cfg->n_state != 3) return fail(CUDE_ERR_UNSUPPORTED, "n_state must be 2 or 3");
        if (cfg->model == CUDE_MODEL_SUPP && cfg->n_state != 3) return fail(CUDE_ERR_UNSUPPORTED, "the suppression model has 3 states");
        const bool tuned = cfg->model == CUDE_MODEL_CPEP ? cude::cpep_shape_supported(net, cfg->n_state)
                                                         : cude::supp_shape_supported(net);
        if (!tuned) {       // no tuned kernel for this width / depth: the fallback kernel (cude_generic.hip), tanh / softplus
            std::vector<int32_t> w((size_t)cfg->nn_depth, cfg->nn_width), act((size_t)cfg->nn_depth + 1, cude::kGenActTanh);
            act.back() = cude::kGenActSoftplus;
            int32_t rc = make_general_network(cfg->model, cfg->nn_in, cfg->nn_depth, w.data(), act.data(), &net.gen);
            if (rc) return rc;
        }
    } else {
        return fail(CUDE_ERR_ARG, "unknown model id");
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(CUDE_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));
    cude_ctx* c = new (std::nothrow) cude_ctx();
    if (!c) return fail(CUDE_ERR_ARG, "out of host memory");
    c->cfg = *cfg;
    c->net = net;
    options_from_environment(c);
    c->P = net.n_params();
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(CUDE_ERR_HIP, hipGetErrorString(e)); }
    if (alloc_network_buffers(c)) {
        cude_destroy(c);
        return fail(CUDE_ERR_HIP, "hipMalloc failed");
    }
    *out = c;
    return CUDE_OK;
}

int32_t cude_set_network(cude_ctx* c, int32_t n_hidden, const int32_t* widths, const int32_t* activations) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (c->cfg.model != CUDE_MODEL_CPEP && c->cfg.model != CUDE_MODEL_SUPP) return fail(CUDE_ERR_ARG, "the symbolic model has no network");
    if (c->have_pop) return fail(CUDE_ERR_STATE, "set the network before the population");
    if (n_hidden < 1 || !widths || !activations) return fail(CUDE_ERR_ARG, "a network needs at least one hidden layer");
    cude::NetShape net{c->cfg.nn_in, widths[0], n_hidden};
    // one width, tanh in every hidden layer, softplus at the output, and a kernel compiled for it: the tuned path
    bool plain = activations[n_hidden] == cude::kGenActSoftplus;
    for (int l = 0; l < n_hidden; l++) plain = plain && widths[l] == widths[0] && activations[l] == cude::kGenActTanh;
    const bool tuned = plain && !c->opt.force_fallback &&
                       (c->cfg.model == CUDE_MODEL_CPEP ? cude::cpep_shape_supported(net, c->cfg.n_state)
                                                        : cude::supp_shape_supported(net));
    if (!tuned && (rc = make_general_network(c->cfg.model, c->cfg.nn_in, n_hidden, widths, activations, &net.gen))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->net = net;
    c->cfg.nn_width = net.generic() ? net.gen.max_width() : widths[0];
    c->cfg.nn_depth = n_hidden;
    c->P = net.n_params();
    c->have_nn = false;
    HIP_TRY(c->param_mask.resize(0));
    return alloc_network_buffers(c);
}

int32_t cude_network_info(cude_ctx* c, int32_t* n_params, int32_t* fallback_kernel) {
    if (!c || !n_params || !fallback_kernel) return fail(CUDE_ERR_ARG, "null argument");
    *n_params = c->P;
    *fallback_kernel = c->net.generic() ? 1 : 0;
    return CUDE_OK;
}

int32_t cude_destroy(cude_ctx* c) {
    if (!c) return CUDE_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    comm_release(c);
    xchg_release(c);
    for (auto& pr : c->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_pairs) (void)hipHostFree(c->pinned_pairs);
    if (c->tr_pinned) (void)hipHostFree(c->tr_pinned);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CUDE_OK;
}

int32_t cude_set_population_cpep(cude_ctx* c, int64_t N, int32_t n_obs, const double* timepoints,
                                 const double* glucose, const double* cpeptide, int64_t ld_subject, int64_t ld_time,
                                 const double* age, const uint8_t* t2dm) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!is_cpep(c)) return fail(CUDE_ERR_STATE, "context is not a c-peptide model");
    if (N < 1 || !timepoints || !glucose || !cpeptide || !age || !t2dm) return fail(CUDE_ERR_ARG, "null/empty input");
    if ((rc = check_times(n_obs, timepoints))) return rc;
    const int T = n_obs;
    c->have_pop = false;
    c->N = N;
    c->T = T;
    c->tp.assign(timepoints, timepoints + T);
    // stage as [T][N] (subject fastest) so every device access is coalesced
    std::vector<double> g((size_t)T * N), cp((size_t)T * N);
    for (int t = 0; t < T; t++)
        for (int64_t i = 0; i < N; i++) {
            g[(size_t)t * N + i] = glucose[i * ld_subject + t * ld_time];
            cp[(size_t)t * N + i] = cpeptide[i * ld_subject + t * ld_time];
        }
    DevBuf<double> gdev;
    DevBuf<uint8_t> t2dev;
    HIP_TRY(gdev.resize((size_t)T * N));
    HIP_TRY(t2dev.resize(N));
    HIP_TRY(c->obs.resize((size_t)T * N));
    HIP_TRY(c->dG.resize((size_t)T * N));
    HIP_TRY(c->k0.resize(N)); HIP_TRY(c->k1.resize(N)); HIP_TRY(c->k2.resize(N)); HIP_TRY(c->c0.resize(N));
    HIP_TRY(c->age.resize(N));
    HIP_TRY(c->auc.resize(c->cfg.n_state == 3 ? N : 0));
    HIP_TRY(hipMemcpyAsync(gdev.p, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->obs.p, cp.data(), cp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->age.p, age, N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t2dev.p, t2dm, N * sizeof(uint8_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(cude::launch_prepare_cpep(N, T, gdev.p, c->obs.p, c->age.p, t2dev.p, c->k0.p, c->k1.p, c->k2.p, c->c0.p,
                                      c->dG.p, c->stream));
    if ((rc = alloc_common(c))) return rc;
    HIP_TRY(c->act.resize(cpep_keep_activations(c) ? cpep_act_doubles(c) : 0));
    if ((rc = upload_tables(c, true))) return rc;
    if ((rc = setup_chunks(c))) return rc;
    c->n_global = (double)N;
    if (distributed(c)) {
        double v[1] = {(double)N};
        if ((rc = cude_comm_allreduce_host(c, v, 1))) return rc;
        c->n_global = v[0];
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_pop = true;
    return CUDE_OK;
}

int32_t cude_set_population_supp(cude_ctx* c, int64_t N, int32_t n_obs, const double* timepoints, const double* data) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (c->cfg.model != CUDE_MODEL_SUPP) return fail(CUDE_ERR_STATE, "context is not a suppression model");
    if (N < 1 || !timepoints || !data) return fail(CUDE_ERR_ARG, "null/empty input");
    if ((rc = check_times(n_obs, timepoints))) return rc;
    const int T = n_obs;
    c->have_pop = false;
    c->N = N;
    c->T = T;
    c->tp.assign(timepoints, timepoints + T);
    std::vector<double> d((size_t)3 * T * N);
    double ssum[4] = {0, 0, 0, (double)N};
    for (int64_t i = 0; i < N; i++)
        for (int s = 0; s < 3; s++) {
            double m = -std::numeric_limits<double>::infinity();
            for (int t = 0; t < T; t++) {
                const double v = data[s + 3 * (t + (int64_t)T * i)];
                d[((size_t)s * T + t) * N + i] = v;
                if (v > m) m = v;
            }
            ssum[s] += m;
        }
    if (distributed(c) && (rc = cude_comm_allreduce_host(c, ssum, 4))) return rc;
    c->n_global = ssum[3];
    for (int s = 0; s < 3; s++) c->scale[s] = ssum[s] / ssum[3];
    HIP_TRY(c->data.resize(d.size()));
    HIP_TRY(c->ckpt.resize((size_t)cude::supp_ckpt_rows(c->cfg.n_steps, c->T) * N));   // every stage input + residuals
    HIP_TRY(c->act.resize(supp_keep_activations(c, 1) ? supp_act_doubles(c) : 0));   // kept activations (small N)
    HIP_TRY(hipMemcpyAsync(c->data.p, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = alloc_common(c))) return rc;
    if ((rc = upload_tables(c, false))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_pop = true;
    return CUDE_OK;
}

int32_t cude_set_params(cude_ctx* c, const double* nn, const double* cond) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (nn) {
        HIP_TRY(hipMemcpyAsync(c->nn.p, nn, c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->have_nn = true;
    }
    if (cond) {
        if (!c->have_pop) return fail(CUDE_ERR_STATE, "set the population before the conditional parameters");
        HIP_TRY(hipMemcpyAsync(c->cond.p, cond, c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->have_cond = true;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_get_params(cude_ctx* c, double* nn, double* cond) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (nn) HIP_TRY(hipMemcpyAsync(nn, c->nn.p, c->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (cond) {
        if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
        HIP_TRY(hipMemcpyAsync(cond, c->cond.p, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_set_param_mask(cude_ctx* c, const double* mask) {
    int32_t rc = bind(c);
    if (rc) return rc;
    drop_graph(c);                                  // the mask pointer is baked into captured launches
    if (!mask) {
        HIP_TRY(c->param_mask.resize(0));
        c->mask_host.clear();
        return CUDE_OK;
    }
    for (int q = 0; q < c->P; q++)
        if (!std::isfinite(mask[q])) return fail(CUDE_ERR_ARG, "mask entries must be finite");
    c->mask_host.assign(mask, mask + c->P);
    HIP_TRY(c->param_mask.resize((size_t)c->P));
    HIP_TRY(hipMemcpyAsync(c->param_mask.p, c->mask_host.data(), c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // Adam moments gathered before the mask was set would keep moving a frozen entry (lr * m_hat / (sqrt(v_hat) + eps)
    // while m decays): they are multiplied by the mask as well
    if (c->adam_ready && c->m_nn.p && c->v_nn.p) {
        std::vector<double> mv(2 * (size_t)c->P);
        HIP_TRY(hipMemcpyAsync(mv.data(), c->m_nn.p, c->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(mv.data() + c->P, c->v_nn.p, c->P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int q = 0; q < c->P; q++) { mv[q] *= mask[q]; mv[c->P + q] *= mask[q] * mask[q]; }
        HIP_TRY(hipMemcpyAsync(c->m_nn.p, mv.data(), c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->v_nn.p, mv.data() + c->P, c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_set_rng(cude_ctx* c, uint64_t seed, int64_t subject_offset) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    if (subject_offset < 0) return fail(CUDE_ERR_ARG, "subject_offset must be >= 0");
    c->rng_seed = seed;
    c->rng_offset = subject_offset;
    c->rng_step = 0;
    return CUDE_OK;
}

int32_t cude_rng_draws(cude_ctx* c, int64_t first_step, int32_t n_steps, double* normals, double* uniforms) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (first_step < 0 || n_steps < 1 || (!normals && !uniforms)) return fail(CUDE_ERR_ARG, "bad argument");
    const int64_t N = c->N;
    DevBuf<double> d_z, d_u;
    if (normals) HIP_TRY(d_z.resize((size_t)n_steps * N));
    if (uniforms) HIP_TRY(d_u.resize((size_t)n_steps * N));
    for (int k = 0; k < n_steps; k++)
        HIP_TRY(cude::launch_rng_draws(N, cude::RngKey{c->rng_seed, c->rng_offset, first_step + k},
                                       normals ? d_z.p + (size_t)k * N : nullptr, uniforms ? d_u.p + (size_t)k * N : nullptr,
                                       c->stream));
    if (normals) HIP_TRY(hipMemcpyAsync(normals, d_z.p, (size_t)n_steps * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (uniforms) HIP_TRY(hipMemcpyAsync(uniforms, d_u.p, (size_t)n_steps * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_set_global_subjects(cude_ctx* c, double n_global, const double* scale3) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    if (!(n_global >= (double)c->N)) return fail(CUDE_ERR_ARG, "global subject count smaller than the local one");
    drop_graph(c);                              // 1/n_global and the scale are baked into the captured launches
    c->n_global = n_global;
    if (scale3) {
        for (int s = 0; s < 3; s++) {
            if (!(scale3[s] > 0)) return fail(CUDE_ERR_ARG, "scale must be positive");
            c->scale[s] = scale3[s];
        }
    }
    return CUDE_OK;
}

int32_t cude_get_scale(cude_ctx* c, double* scale3, double* n_global) {
    if (!c || !scale3 || !n_global) return fail(CUDE_ERR_ARG, "null argument");
    for (int s = 0; s < 3; s++) scale3[s] = c->scale[s];
    *n_global = c->n_global;
    return CUDE_OK;
}

int32_t cude_set_option(cude_ctx* c, const char* name, const char* value) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    return apply_option(c, name, value);
}

int32_t cude_set_tolerances(cude_ctx* c, double abstol, double reltol) {
    if (!(abstol > 0) || !(reltol > 0) || !std::isfinite(abstol) || !std::isfinite(reltol))
        return fail(CUDE_ERR_ARG, "tolerances must be positive");
    int32_t rc = bind(c);
    if (rc) return rc;
    drop_graph(c);      // a captured optimiser iteration carries the tolerances by value in its kernel arguments
    c->abstol = abstol;
    c->reltol = reltol;
    return CUDE_OK;
}

int32_t cude_grad_occupancy(cude_ctx* c, int32_t* waves_per_cu) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!waves_per_cu) return fail(CUDE_ERR_ARG, "null output");
    if (!c->have_pop) return fail(CUDE_ERR_STATE, "population not set");
    *waves_per_cu = is_cpep(c) ? cude::cpep_grad_waves_per_cu(c->net, c->cfg.n_state, c->T)
                               : cude::supp_grad_waves_per_cu(c->net);
    return CUDE_OK;
}

int32_t cude_synchronize(cude_ctx* c) {
    int32_t rc = bind(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CUDE_OK;
}

int32_t cude_set_kernel_timing(cude_ctx* c, int32_t enabled) {
    if (!c) return fail(CUDE_ERR_ARG, "null context");
    c->timing = enabled != 0;
    c->timing_period = enabled > 1 ? enabled : 1;
    c->timing_count = 0;
    c->ev_used = 0;
    return CUDE_OK;
}

int32_t cude_kernel_time_stats(cude_ctx* c, double* avg_ms, double* median_ms, double* min_ms, int64_t* launches) {
    int32_t rc = bind(c);
    if (rc) return rc;
    if (!avg_ms) return fail(CUDE_ERR_ARG, "null output");
    HIP_TRY(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    std::vector<double> all(c->ev_used);
    for (size_t k = 0; k < c->ev_used; k++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[k].first, c->ev_pool[k].second));
        tot += ms;
        all[k] = ms;
    }
    std::sort(all.begin(), all.end());
    *avg_ms = c->ev_used ? tot / (double)c->ev_used : 0.0;
    if (median_ms) *median_ms = all.empty() ? 0.0 : 0.5 * (all[(all.size() - 1) / 2] + all[all.size() / 2]);
    if (min_ms) *min_ms = all.empty() ? 0.0 : all.front();
    if (launches) *launches = (int64_t)c->ev_used;
    c->ev_used = 0;
    c->timing_count = 0;            // (the first launch after a query is a timed one, whatever the period)
    return CUDE_OK;
}

int32_t cude_kernel_time_ms(cude_ctx* c, double* avg_ms, int64_t* launches) {
    return cude_kernel_time_stats(c, avg_ms, nullptr, nullptr, launches);
}

}  // extern "C"
