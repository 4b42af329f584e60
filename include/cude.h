/* cude.h -- C ABI of libcude_hip.so: the MI355X (gfx950) population ensemble ODE solve +
 * discrete adjoint for conditional universal differential equations.
 *
 * This is the drop-in boundary for ONE hot path of Computational-Biology-TUe/conditional-ude:
 * everything Optimization.jl calls as `f(theta, p)` / its gradient while training a cUDE.
 * The reference has no FFI layer (it is pure Julia); each entry point below names the
 * reference function(s) (paths relative to the reference repo root) whose arithmetic it
 * replaces, so a maintainer can bind it with `ccall` (INTEGRATION.md has the Julia stubs).
 *
 * Conventions
 *   - every function returns int32 status: 0 = ok, <0 = error (CUDE_ERR_*); the message is
 *     available from cude_last_error() (thread-local).  No exceptions, no abort().
 *   - fp64 throughout.  The caller owns every host buffer; the library copies on set_* and
 *     never retains host pointers.  All device memory is owned by the context.
 *   - a context is bound to one HIP device and one stream; it is not thread-safe.
 *   - solver failure convention of the reference (src/parameter-estimation.jl:61-64,134-136):
 *     a non-finite trajectory in any subject makes the returned loss +Inf with status 0;
 *     cude_n_failed() tells how many subjects failed.
 *   - discretisation: fixed-step Tsit5, n_steps uniform steps over [t[0], t[T-1]],
 *     observations by the Tsit5 dense-output interpolant (DESIGN.md "numerical contract"); n_steps = 0 selects the
 *     reference's own adaptive Tsit5 (every entry point; gradients = adjoint of the accepted steps).
 */
#ifndef CUDE_H
#define CUDE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CUDE_OK 0
#define CUDE_ERR_ARG (-1)         /* bad argument / size */
#define CUDE_ERR_HIP (-2)         /* HIP runtime error (message has hipGetErrorString) */
#define CUDE_ERR_STATE (-3)       /* call order (e.g. loss before set_population) */
#define CUDE_ERR_UNSUPPORTED (-4) /* network shape / model not compiled in */
#define CUDE_ERR_COMM (-5)        /* RCCL error or librccl not loadable */

#define CUDE_MODEL_CPEP 0 /* c-peptide cUDE: src/c-peptide-models.jl:7-14,86-94,170-194 */
#define CUDE_MODEL_SUPP 1 /* suppression cUDE: suppression/src/suppression_model.jl:88-95 */
/* c-peptide model with the analytic (symbolic-regression) production term instead of the network:
 * production(dG, k) = dG >= 0 ? p0*dG/(dG + k) : 0  with p0 = 1.78 in the reference
 * (c-peptide/03-symreg.jl:37-40 via CPeptideODEModel / analytic_production, src/c-peptide-models.jl:68-75,
 * 118-142; src/saem-symreg.jl:23-29 writes the same term).  The shared parameter vector is [p0] (P = 1,
 * cude_n_params(.,0,0)); the per-subject conditional parameter is k.  Population data, loss, gradient,
 * multi-start, Metropolis E-step, Adam and sharding are those of CUDE_MODEL_CPEP. */
#define CUDE_MODEL_CPEP_SYM 2

/* how the per-subject conditional parameter enters the symbolic model (cude_config.cond_space) */
#define CUDE_COND_LOG 0 /* k = exp(conditional): the cUDE convention; km_pop*exp(eta) of saem-symreg.jl:57-59 */
#define CUDE_COND_RAW 1 /* k = conditional: the box-constrained fit of 03-symreg.jl:99-106 (p.ode[1] in [0,1000]) */

#define CUDE_UNIQUE_ID_BYTES 128
#define CUDE_XCHG_HANDLE_BYTES 128 /* a rank's exchange mailbox, as cude_xchg_export describes it to its peers */
#define CUDE_XCHG_MAX_RANKS 16    /* ranks of one node (the exchange writes into peers' memory over xGMI) */

typedef struct cude_ctx cude_ctx;

/* Replaces the per-script constants + `chain(width, depth, tanh; input_dims)`
 * (src/neural-network.jl:105-107; suppression_model.jl:78-85) that fix the model shape. */
typedef struct cude_config {
    int32_t model;    /* CUDE_MODEL_* */
    int32_t n_state;  /* CPEP: 2, or 3 = +cumulative-secretion quadrature state (zero loss weight); SUPP: 3 */
    int32_t nn_in;    /* CPEP: 2 = [dG, exp(beta)], 3 = [dG, exp(beta), age] (covariate model,
                         src/c-peptide-models.jl:96-104); SUPP: 4 = [u1,u2,u3,exp(theta)] */
    int32_t nn_width; /* hidden width (0 for CUDE_MODEL_CPEP_SYM) */
    int32_t nn_depth; /* number of hidden layers (tanh unless cude_set_option says otherwise); output layer: 1 unit, softplus
                         unless cude_set_option says otherwise (0 for CPEP_SYM) */
    int32_t n_steps;  /* fixed Tsit5 steps over the time span; 0 = ADAPTIVE Tsit5 as the reference runs it
                         (OrdinaryDiffEq defaults abstol 1e-6 / reltol 1e-3, PI controller, cude_set_tolerances), in
                         every entry point.  Gradients in this mode are those of the accepted step sequence taken as
                         fixed arithmetic -- what AutoForwardDiff through `solve` yields, dt carrying no partials
                         (src/parameter-estimation.jl:165, suppression_model.jl:155); CPEP: n_state = 2 only */
    int32_t device;   /* HIP device ordinal */
    int32_t cond_space; /* CUDE_COND_*; must be CUDE_COND_LOG (0) for the network models */
    double lambda;    /* L2 weight on the network parameters (suppression_loss :128); 0 for CPEP */
} cude_config;

const char* cude_last_error(void);
int32_t cude_device_count(int32_t* count);
/* Number of network parameters for a shape: SimpleChains layout, per layer
 * [vec_colmajor(W out x in); b] (src/neural-network.jl:52-56).  width = depth = 0 names the analytic production
 * model (CUDE_MODEL_CPEP_SYM): 1 shared parameter. */
int32_t cude_n_params(int32_t nn_in, int32_t nn_width, int32_t nn_depth);

/* Any (nn_in, nn_width, nn_depth) is accepted for the network models: shapes a tuned kernel is compiled for (24 c-peptide
 * + 16 suppression shapes, widths <= 8) run on it, every other on the fallback kernel of cude_set_network below. */
int32_t cude_create(const cude_config* cfg, cude_ctx** out);
/* The general form of `chain(widths, activation_functions; input_dims, output_dims = 1, output_activation)`,
 * src/neural-network.jl:42-58 (the docstring's example: chain([10, 20, 30], [tanh, relu, softplus]; input_dims = 4)):
 * n_hidden layer widths and n_hidden + 1 activation codes, one per hidden layer and the output layer's last.  Replaces the
 * network of the context (cfg.nn_in stays; call it after cude_create and before the population is uploaded); the parameter
 * vector is SimpleChains' [vec_colmajor(W); b] per layer and cude_network_info tells its length.  One width, tanh in every
 * hidden layer, softplus at the output and a tuned kernel compiled for it: that kernel; anything else runs on the
 * fallback kernel (csrc/cude_generic.hip: run-time shape, weights staged once per workgroup in LDS, one lane per
 * subject, fixed-step and adaptive, forward and adjoint, both network models) -- every entry point of this header works
 * on it except cude_adaptive_regroup; built for generality, not for speed.  CUDE_ERR_UNSUPPORTED only when the weights
 * and one activation column per lane exceed the 160 KB of LDS of a workgroup (sum of widths around 250) or for more
 * than 8 hidden layers. */
#define CUDE_ACT_TANH 0
#define CUDE_ACT_RELU 1
#define CUDE_ACT_SIGMOID 2
#define CUDE_ACT_SOFTPLUS 3
#define CUDE_ACT_IDENTITY 4
int32_t cude_set_network(cude_ctx* ctx, int32_t n_hidden, const int32_t* widths, const int32_t* activations);
/* Length of the context's shared parameter vector and whether its network runs on the fallback kernel (1) or a tuned one (0). */
int32_t cude_network_info(cude_ctx* ctx, int32_t* n_params, int32_t* fallback_kernel);
int32_t cude_destroy(cude_ctx* ctx);
/* Tolerances of the adaptive mode (cude_config.n_steps = 0): `solve(...; abstol, reltol)`; the defaults 1e-6 / 1e-3
 * are OrdinaryDiffEq's, which every solve call of the reference uses (src/parameter-estimation.jl:59, src/saem.jl:52,
 * suppression/src/suppression_model.jl:113,123). */
int32_t cude_set_tolerances(cude_ctx* ctx, double abstol, double reltol);
/* Adaptive mode: the accepted steps (t_n, dt_n) `subject` took in the last gradient evaluation of the context's own
 * parameters (cude_loss_grad, cude_loss_grad_partial, cude_adam_step ...) -- `sol.t` of the reference's solve.
 * *n_steps = number of accepted steps; at most `cap` of them are written to t_out / dt_out (either may be NULL).
 * CUDE_ERR_STATE before the first such evaluation, outside the adaptive mode, and when any other adaptive launch
 * (cude_forward, a fit, a likelihood profile, a multi-set evaluation, a re-ordering) has run on the context since: those
 * overwrite the step counts the tape is read with. */
int32_t cude_adaptive_steps(cude_ctx* ctx, int64_t subject, int32_t cap, double* t_out, double* dt_out, int32_t* n_steps);
/* Adaptive mode, large populations: order the launch by accepted-step count.  A wave's 64 lanes run until the slowest of
 * them has finished, and subjects differ in how many steps the controller grants them (8 ... 23 on the c-peptide data).
 * Every adaptive launch (forward or gradient) leaves each subject's accepted-step count behind; this call sorts the
 * launch by them (stable, most steps first) so that the lanes of a wave finish together: lane k of the next launches
 * works on the subject that came k-th, inputs are gathered through the permutation, the tape stays in lane order
 * (coalesced), every per-subject output lands at its subject's own index.  Per-subject results are unchanged bit for
 * bit; the shared gradient is summed in the new order (rounding).  spread_* (optional): mean over the waves of (max -
 * min accepted steps within the wave), before and after.  Until the next gradient evaluation cude_adaptive_steps returns
 * CUDE_ERR_STATE (the tape on the device is in the old order).  A new population resets the order.
 * WHO CALLS IT BY ITSELF (populations of >= 8192 subjects; option "auto_regroup" = 0 / CUDE_NO_AUTO_REGROUP=1 turns all
 * of this off): cude_adam_run -- after its 1st, 200th, 400th ... iteration on the population, counted over calls, so
 * that adam_run(400) and 2 x adam_run(200) produce the same bits (single cude_adam_step calls never re-order);
 * cude_multistart_loss_grad / cude_train_restarts -- at the first call that finds counts, then after every 200th
 * evaluation; cude_fit_conditional -- after its first probe.  A caller who needs one summation order across entry
 * points switches the option off and calls this function where it wants the order to change.  (No reference line:
 * the reference solves its subjects one after the other, suppression_model.jl:113,123; parameter-estimation.jl:126-140.) */
int32_t cude_adaptive_regroup(cude_ctx* ctx, int32_t* spread_before, int32_t* spread_after);

/* --- population (replaces the CPeptideConditionalUDEModel constructor loop,
 * src/c-peptide-models.jl:170-194 incl. van_cauter_parameters :30-42, u0, tspan and the
 * LinearInterpolation of glucose :181; c-peptide/02-conditional.jl:26-28).
 * glucose / cpeptide are N x T matrices addressed as base[i*ld_subject + t*ld_time]
 * (Julia column-major N x T: ld_subject = 1, ld_time = N; numpy C-order: ld_subject = T, ld_time = 1).
 * age[N], t2dm[N] (0/1).  timepoints[T] are shared by all subjects and are also the glucose knots.
 * If a communicator is attached (cude_comm_init) the global subject count is all-reduced here. */
int32_t cude_set_population_cpep(cude_ctx* ctx, int64_t n_subjects, int32_t n_obs, const double* timepoints,
                                 const double* glucose, const double* cpeptide, int64_t ld_subject,
                                 int64_t ld_time, const double* age, const uint8_t* t2dm);

/* suppression model population: data is the Julia array individual_data[3 x T x N] in
 * column-major order (data[s + 3*(t + T*i)]); u0 = data[:,1,:]; scale = mean_i max_t data
 * (suppression_model.jl:119,126). */
int32_t cude_set_population_supp(cude_ctx* ctx, int64_t n_subjects, int32_t n_obs, const double* timepoints,
                                 const double* data);

/* theta = ComponentArray(neural = nn[P], conditional = cond[N]) (parameter-estimation.jl:354-357).
 * Either pointer may be NULL to leave that part unchanged. */
int32_t cude_set_params(cude_ctx* ctx, const double* nn, const double* cond);
int32_t cude_get_params(cude_ctx* ctx, double* nn, double* cond);

/* Forward-only population loss: `loss(theta, (models, timepoints, data))`
 * (src/parameter-estimation.jl:126-140, per-subject :56-68) / `suppression_loss` without AD
 * (suppression_model.jl:117-130) / `simul` (:107-115).
 * per_subject_sse[N] (optional) is the un-normalised SSE of each subject (the quantity
 * loss_sigma :70-75 and individual_log_likelihood saem.jl:55-66 are built from).
 * traj (optional) receives the interpolated states as [n_state x T x N] column-major
 * (Julia Array(sol) layout of simul). */
int32_t cude_forward(cude_ctx* ctx, double* loss, double* per_subject_sse, double* traj);

/* Dense output: the states of every subject at n_times arbitrary, non-decreasing times inside [t[0], t[T-1]] (Tsit5
 * interpolant of the same fixed-step solve), at the context's current parameters.  Replaces
 * `simulate(p_neural, p_individual, individual, network; timepoints = t0:0.1:tend)` (src/saem.jl:31-53, called with
 * dense grids at c-peptide/06-saem.jl:221-240) and `solve(model.problem, p=..., saveat=sol_timepoints)` of the
 * model-fit figures (c-peptide/02-conditional.jl, 03-symreg.jl:113).  traj receives [n_state x n_times x N]
 * column-major.  c-peptide models only (the reference simulates the suppression model at its data times: cude_forward). */
int32_t cude_simulate(cude_ctx* ctx, int32_t n_times, const double* times, double* traj);

/* Multi-start screening: forward-only loss of n_sets candidate parameter sets over the resident
 * population in one launch (first phase of `train`, src/parameter-estimation.jl:351-366;
 * fit_suppression_model suppression_model.jl:135; c-peptide/06-saem.jl:41-42).
 * nn_sets[n_sets][P], cond_sets[n_sets][N] row-major; losses[n_sets] (+Inf for a set with a failed subject).
 * Does not touch the context's current parameters. */
int32_t cude_multistart_forward(cude_ctx* ctx, int32_t n_sets, const double* nn_sets, const double* cond_sets,
                                double* losses);

/* The whole screening phase with the selection on the device: `losses_initial = [loss(p, ...) for p in initials]` followed
 * by `partialsortperm(losses_initial, 1:selected_initials)` (src/parameter-estimation.jl:359-372; suppression_model.jl:
 * 135-145; c-peptide/06-saem.jl:41-45).  Candidates are STREAMED: `gen` is asked for one chunk at a time (candidates
 * first ... first+count-1 as rows nn_sets[count][P], cond_sets[count][N]; return < 0 to abort), so the host never holds
 * the 25 000 x (P + N) candidate table; each chunk is evaluated in one multi-start launch, its losses stay on the device
 * and are merged into a running top-n_keep (ties broken by the lower candidate index, as partialsortperm does), whose
 * parameter sets are kept on the device as well.  Out: index_out / loss_out [n_keep] in increasing order of loss, and
 * the selected parameter sets nn_out[n_keep][P], cond_out[n_keep][N] -- ready for cude_train_restarts. */
typedef int32_t (*cude_candidate_fn)(int64_t first, int32_t count, double* nn_sets, double* cond_sets, void* user);
int32_t cude_screen_candidates(cude_ctx* ctx, int64_t n_candidates, int32_t n_keep, cude_candidate_fn gen, void* user,
                               int64_t* index_out, double* loss_out, double* nn_out, double* cond_out);

/* Restarts trained side by side: loss AND gradient of n_sets independent parameter sets over the resident
 * population in one launch (the grid's second dimension is the set).  The reference trains its selected
 * initial guesses one after the other (`for p in initials[selected] ... _optimize(...)`,
 * src/parameter-estimation.jl:372-383; suppression_model.jl:140-170), each a serial chain of thousands of
 * loss+gradient evaluations over a few dozen subjects -- one wave of work per evaluation; evaluating all restarts'
 * current points together turns that latency-bound loop into a throughput-bound one.
 * nn_sets[n_sets][P], cond_sets[n_sets][N] row-major in; losses[n_sets] (+Inf for a set with a failed subject,
 * L2 term included), g_nn_sets[n_sets][P], g_cond_sets[n_sets][N] out -- exactly what cude_loss_grad returns for
 * each set.  Does not touch the context's current parameters. */
int32_t cude_multistart_loss_grad(cude_ctx* ctx, int32_t n_sets, const double* nn_sets, const double* cond_sets,
                                  double* losses, double* g_nn_sets, double* g_cond_sets);

/* The second half of `train` (src/parameter-estimation.jl:368-383; fit_suppression_model suppression_model.jl:140-170)
 * for n_sets selected initial guesses SIDE BY SIDE: `adam_iters` iterations of Optimisers.Adam(learning_rate), then
 * up to `lbfgs_iters` iterations of Optim's L-BFGS (m = 10) with LineSearches.BackTracking -- every optimiser
 * iteration evaluates all restarts' current points with one cude_multistart_loss_grad launch; each restart follows
 * the path it would follow alone.  A restart whose loss becomes non-finite during Adam is dropped (objective +Inf),
 * as the reference skips a failed optimisation.  In: nn_sets[n_sets][P], cond_sets[n_sets][N]; out: the trained
 * nn_out / cond_out (same layout) and objective_out[n_sets] (`OptimizationSolution.objective`); loss_trace (optional,
 * [n_sets][adam_iters + lbfgs_iters], NaN where a run had already stopped) receives what the reference's callbacks
 * collect: the loss of every Adam iteration, then the loss after every successful L-BFGS iteration.  On a sharded
 * population (cude_comm_init) cond_sets / cond_out hold this rank's subjects; losses and network gradients are
 * all-reduced inside the launch sequence and the conditional part of every L-BFGS inner product is summed over the
 * ranks, so all ranks return the same networks and objectives.  The optimiser
 * bookkeeping runs on the host (vectors of P+N doubles); every loss and gradient comes from the device. */
int32_t cude_train_restarts(cude_ctx* ctx, int32_t n_sets, const double* nn_sets, const double* cond_sets,
                            int32_t adam_iters, double learning_rate, int32_t lbfgs_iters, double* nn_out,
                            double* cond_out, double* objective_out, double* loss_trace);

/* The same L-BFGS + BackTracking for any objective (host only, needs no GPU): `Optimization.solve(prob,
 * LBFGS(linesearch = BackTracking()), maxiters)` with Optim.jl's defaults (memory 10, g_tol 1e-8, c_1 1e-4,
 * rho in [0.1, 0.5], cubic interpolation).  fn fills *f and g[n] at x and returns >= 0 (a non-finite f is a failed
 * solve: the line search backs away from it).  iterations / f_calls / converged may be NULL. */
typedef int32_t (*cude_objective_fn)(const double* x, int32_t n, double* f, double* g, void* user);
int32_t cude_lbfgs_minimize(int32_t n, const double* x0, int32_t maxiters, cude_objective_fn fn, void* user,
                            double* x_out, double* f_out, int32_t* iterations, int32_t* f_calls, int32_t* converged);

/* The same optimiser for a SHARDED vector: x = [shared (n_shared entries, replicated on every rank); local (this
 * rank's n - n_shared conditional parameters)].  fn returns the GLOBAL f and this rank's gradient entries; `reduce`
 * sums (op 0) or takes the maximum (op 1) of `count` doubles over all ranks in place (MPI.Allreduce!, gloo, ...) and
 * returns >= 0.  Every inner product / max-norm of the L-BFGS recursion takes its local part through `reduce`, so all
 * ranks walk the same iterates as one process holding the whole vector would (up to the summation order of the inner
 * products).  This is the second stage of `_optimize` (src/parameter-estimation.jl:179-180) on a population sharded
 * over GPUs; cude_train_restarts does the same internally over its RCCL communicator. */
typedef int32_t (*cude_reduce_fn)(double* values, int32_t count, int32_t op, void* user);
int32_t cude_lbfgs_minimize_sharded(int32_t n, int32_t n_shared, const double* x0, int32_t maxiters,
                                    cude_objective_fn fn, cude_reduce_fn reduce, void* user, double* x_out,
                                    double* f_out, int32_t* iterations, int32_t* f_calls, int32_t* converged);

/* Likelihood profiles of ALL subjects in one launch: sse_out[n_points][N] (row-major) = SSE_i(values[k]) with the
 * shared parameters frozen -- the grid's second dimension is the scan value.  Replaces
 * `likelihood_profile(beta, nn, model, timepoints, data, lower, upper, sigma; steps)` (src/likelihood-profiles.jl:4-17:
 * `loss_values = [loss(b, (model, ...)) for b in range(lower, upper, length = steps)]`, called per subject with
 * 1000-10000 points at c-peptide/02-conditional.jl:186-188); the profile is sse / (2 sigma^2). */
int32_t cude_profile_conditional(cude_ctx* ctx, int32_t n_points, const double* values, double* sse_out);

/* Per-subject fits of the conditional parameter with the shared parameters frozen, for all subjects at once:
 * every subject i minimises  SSE_i(x) + penalty_weight * (x - penalty_center)^2  over [lower, upper] by a coarse scan
 * of n_grid points followed by n_iters golden-section steps inside the best bracket; the whole search is queued on
 * the stream (one forward launch per probe, search state on the device, one synchronisation at the end).
 * Replaces the per-subject loops  `for (i, model) in enumerate(models) ... Optimization.solve(..., LBFGS, Fminbox)`:
 *   train(models, timepoints, data, neural_network_parameters) / train_with_sigma   src/parameter-estimation.jl:272-307
 *   evaluate_model                                                                   :406-433
 *   validate_suppression_model (loss separates per subject for a frozen network)     suppression_model.jl:179-222
 *   the (k, sigma) fits of the symbolic model                                        c-peptide/03-symreg.jl:94-106
 *   MAP (penalty_weight = sigma^2/Omega^2, penalty_center = prior mean) and MLE      c-peptide/06-saem.jl:114-126
 * (for fixed x the sigma of the reference's 2-parameter objectives has the closed form sigma^2 = SSE/n).
 * cond_out[N]; objective_out[N] and sse_out[N] optional.  Uses the context's current shared parameters and leaves
 * its conditional parameters untouched. */
int32_t cude_fit_conditional(cude_ctx* ctx, double lower, double upper, int32_t n_grid, int32_t n_iters,
                             double penalty_weight, double penalty_center, double* cond_out, double* objective_out,
                             double* sse_out);

/* SAEM E-step on the device: n_mc Metropolis-Hastings steps of every subject's conditional parameter
 * (`mcmc_step` src/saem.jl:86-108, applied n_mcmc_steps times with the stochastic-approximation update of the
 * chain state :177-186).  The chain state is the context's conditional parameters (updated in place); the
 * network is the context's current one.  normals / uniforms are the host's randn() / rand() draws, [n_mc][N]
 * row-major.  log-likelihood = -(T/2) log sigma^2 - SSE/(2 sigma^2) (:55-66), -Inf on a failed solve; prior
 * Normal(prior_mean, prior_sd); accept iff log u < prior ratio + (ll_new - ll_cur)/temperature; state <-
 * (1-gamma) state + gamma (accepted ? proposal : state).  The reference re-evaluates the current state's likelihood
 * at every step (:96-97); here that happens only for gamma != 1 (the state is then a blend that was never solved
 * at).  For gamma == 1 -- its burn-in phase and the posterior sampling loop -- the next state is exactly the
 * accepted proposal or the unchanged state and the solve is deterministic, so the known SSE is carried over: same
 * bits, n_mc + 1 ensemble solves instead of 2 n_mc.  accepted[N] (optional) receives per-subject acceptance
 * counts.  All launches are queued on the stream; the call synchronises once at the end.
 * normals == uniforms == NULL: the draws are generated on the device (cude_set_rng) -- nothing crosses PCIe. */
int32_t cude_mh_estep(cude_ctx* ctx, int32_t n_mc, const double* normals, const double* uniforms, double sigma,
                      double prior_mean, double prior_sd, double proposal_std, double temperature, double gamma,
                      int64_t* accepted);

/* The same chain with every state kept: samples[n_mc][N] (row-major) receives each subject's conditional parameter
 * after every step -- the per-individual posterior sampling loop that follows SAEM (`for _ in 1:3000; p_individual,
 * accepted = mcmc_step(...); push!(individual_samples, p_individual)`, c-peptide/06-saem.jl:107-112) for all
 * individuals at once.  samples may be NULL (then identical to cude_mh_estep). */
int32_t cude_mh_chain(cude_ctx* ctx, int32_t n_mc, const double* normals, const double* uniforms, double sigma,
                      double prior_mean, double prior_sd, double proposal_std, double temperature, double gamma,
                      int64_t* accepted, double* samples);

/* Freeze shared parameters: every network-gradient entry the library forms -- cude_loss_grad, the Adam updates, the
 * restarts of cude_train_restarts / cude_multistart_loss_grad, the partial rows of cude_loss_grad_partial -- is
 * multiplied by mask[q] (0 = frozen, 1 = free; mask[P], or NULL to lift it).  Frozen entries keep exactly the value
 * cude_set_params gave them under Adam and L-BFGS.  Use: `chain([w1, w2, ...], tanh)` with unequal widths
 * (src/neural-network.jl:42-58) is the equal-width network of width max(w) whose extra units have zero weights; frozen at
 * zero they stay that network (the host mirror's chain(widths) does exactly this). */
int32_t cude_set_param_mask(cude_ctx* ctx, const double* mask);

/* Device-side random draws for cude_mh_estep / cude_mh_chain (the `randn()` / `rand()` of mcmc_step, src/saem.jl:88,103)
 * when the caller passes no draw arrays: counter-based Philox4x32-10, key = seed, counter = (global subject index,
 * index of the Metropolis step since this call, kind); standard normals by Box-Muller on two 53-bit uniforms, uniforms
 * from 53 bits, both strictly inside (0, 1).  A draw depends only on (seed, subject_offset + local index, step): the
 * chain of a subject is the same whatever the launch shape and however the population is sharded (subject_offset =
 * global index of this context's first subject).  Successive calls continue the stream; this call rewinds it.
 * Default: seed 0x243F6A8885A308D3, offset 0. */
int32_t cude_set_rng(cude_ctx* ctx, uint64_t seed, int64_t subject_offset);

/* The draws steps first_step ... first_step + n_steps - 1 of that stream use, [n_steps][N] row-major each (either
 * pointer may be NULL): lets a caller replay a device-generated chain elsewhere (the parity tests feed them to the
 * oracle chain and to the host-supplied-draws path). */
int32_t cude_rng_draws(cude_ctx* ctx, int64_t first_step, int32_t n_steps, double* normals, double* uniforms);

/* Loss and gradient: replaces ForwardDiff.gradient(loss, theta) under AutoForwardDiff()
 * (parameter-estimation.jl:370; suppression_model.jl:155; saem.jl:120) by a discrete adjoint
 * of the same fixed-step map.  g_nn[P]; g_cond[N] may be NULL (stays on the device). */
int32_t cude_loss_grad(cude_ctx* ctx, double* loss, double* g_nn, double* g_cond);
int32_t cude_n_failed(cude_ctx* ctx, int64_t* n_failed);

/* Optimisers.Adam(eta) state (parameter-estimation.jl:176; suppression_model.jl:164; saem.jl:128).
 * Resets the moments and the step counter. */
int32_t cude_adam_init(cude_ctx* ctx, double lr, double beta1, double beta2, double eps);
/* One fused optimiser iteration on device-resident parameters: loss + gradient + (all-reduce of
 * [g_nn; sum loss; n_failed] when a communicator is attached) + Adam update of nn (replicated)
 * and cond (sharded).  loss may be NULL: then nothing is copied back and the call does not
 * synchronise (use cude_synchronize). The reported loss is the one BEFORE the update. */
int32_t cude_adam_step(cude_ctx* ctx, double* loss);
int32_t cude_synchronize(cude_ctx* ctx);
/* Diagnostic: resident waves per compute unit (4 SIMDs) the one-lane gradient kernel of this context gets from the
 * runtime (hipOccupancyMaxActiveBlocksPerMultiprocessor with the launch's LDS size) -- registers and LDS as compiled. */
int32_t cude_grad_occupancy(cude_ctx* ctx, int32_t* waves_per_cu);
/* n_iters optimiser iterations in one call (the `maxiters` loop of Optimization.solve(prob, Adam, maxiters),
 * src/parameter-estimation.jl:176): iterations are captured into hipGraphs (eight per graph, single ones for the
 * remainder) and replayed without host round trips; losses[n_iters] (optional) receives the loss BEFORE each update,
 * read back once at the end.  The peer-write exchange (cude_xchg_*) is part of the captured reduction kernels; with an
 * RCCL communicator as the transport the iterations are queued as plain launches (RCCL calls are not captured).
 * Adaptive populations of >= 8192 subjects are re-ordered by accepted-step count on a schedule that counts this entry
 * point's iterations (see cude_adaptive_regroup). */
int32_t cude_adam_run(cude_ctx* ctx, int32_t n_iters, double* losses);

/* --- bring-your-own collective (MPI.jl, gloo, ...) instead of the built-in RCCL path.
 * cude_loss_grad_partial returns this rank's un-reduced [g_nn(P); sum_i sse_i; n_failed] (gradient entries
 * already carry the 1/N_global factor, the L2 term is NOT included); the host sums the vectors of all
 * ranks and hands the result to cude_adam_apply, which adds the L2 term, forms the loss and runs the Adam
 * update.  cude_set_global_subjects tells the context the global subject count (and, for the suppression
 * model, the globally averaged `scale`, suppression_model.jl:126) when no communicator is attached. */
int32_t cude_set_global_subjects(cude_ctx* ctx, double n_global, const double* scale3 /* or NULL */);
int32_t cude_get_scale(cude_ctx* ctx, double* scale3, double* n_global);
int32_t cude_loss_grad_partial(cude_ctx* ctx, double* partial /* P+2 */, double* g_cond /* N or NULL */);
int32_t cude_adam_apply(cude_ctx* ctx, const double* reduced /* P+2 */, double* loss);
/* The same exchange WITHOUT the host round trip, for a collective that reduces device memory in place (RCCL through
 * another binding, GPU-aware MPI): cude_partial_buffer hands out the device address of the context's P+2 doubles,
 * cude_loss_grad_partial_device fills them with this rank's un-reduced vector and synchronises the context's stream,
 * the caller all-reduces them in place on a stream of its own and waits for that, cude_adam_apply_device continues as
 * cude_adam_apply does.  (No reference line: the reference has no collective; this is the sharded form of
 * ForwardDiff.gradient + Optimisers.update, src/parameter-estimation.jl:170-176.) */
int32_t cude_partial_buffer(cude_ctx* ctx, double** device_ptr, int32_t* count /* P+2 */);
int32_t cude_loss_grad_partial_device(cude_ctx* ctx);
int32_t cude_adam_apply_device(cude_ctx* ctx, double* loss);

/* Average device time (ms) of the ensemble launches (forward, or forward+adjoint: whatever the calls since the last
 * query ran) measured with HIP events on the context's stream; resets the accumulator. */
int32_t cude_kernel_time_ms(cude_ctx* ctx, double* avg_ms, int64_t* launches);
/* The same samples with their median and minimum (the median is the sturdier figure when a few launches of the timed
 * region ran before the GPU clock had settled); resets the accumulator as cude_kernel_time_ms does. */
int32_t cude_kernel_time_stats(cude_ctx* ctx, double* avg_ms, double* median_ms, double* min_ms, int64_t* launches);
/* Enable / disable (0) the event timing used by cude_kernel_time_ms: 1 = a pair of HIP events around every ensemble
 * launch, n > 1 = around every n-th launch (a pair costs ~4.5 us of stream time and keeps queued iterations from being
 * replayed as graphs: sampling keeps a live kernel time at a fraction of that). */
int32_t cude_set_kernel_timing(cude_ctx* ctx, int32_t enabled);

/* --- multi-GPU: subjects are sharded, one context (process) per GPU; the only exchange is one
 * sum all-reduce of P+2 doubles per optimiser step (the reference has no distributed path; this
 * is the data-parallel form of EnsembleThreads, suppression_model.jl:113,123).
 * rank 0 creates the id, the host distributes its bytes, every rank calls cude_comm_init.
 * Multi-process RCCL on hosts without legacy IPC needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment before the
 * process first touches the GPU; cude_comm_init with n_ranks > 1 returns CUDE_ERR_COMM if it is not set. */
int32_t cude_comm_unique_id(uint8_t id[CUDE_UNIQUE_ID_BYTES]);
int32_t cude_comm_init(cude_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t id[CUDE_UNIQUE_ID_BYTES]);
/* Sum all-reduce of a host vector through the communicator (used for population statistics
 * such as SAEM's mean/var of the random effects, saem.jl:204-205). */
int32_t cude_comm_allreduce_host(cude_ctx* ctx, double* values, int32_t count);
/* What the attached communicator itself reports: ncclCommCount, ncclCommUserRank, ncclGetVersion (1, 0, 0 without a
 * communicator) -- lets a launcher prove which transport and how many ranks a multi-GPU run really used. */
int32_t cude_comm_info(cude_ctx* ctx, int32_t* n_ranks, int32_t* rank, int32_t* version);

/* --- multi-GPU without a collective library: the peer-write exchange.  Same role as the communicator above (the
 * data-parallel form of EnsembleThreads, suppression_model.jl:113,123; the one sum of P+2 doubles per optimiser step
 * that replaces ForwardDiff's accumulation over all subjects, src/parameter-estimation.jl:126-140,370), built for this
 * message size: every rank owns a mailbox in its GPU's memory, mapped into its peers through HIP IPC; the kernel that
 * finishes a rank's [g_nn; sum loss; n_failed] writes each double straight into a slot of EVERY rank's mailbox over
 * xGMI (two 8-byte words carrying the step's sequence number: a word is valid when its sequence matches, so no fence
 * and no flag ordering is needed), waits for the other ranks' words in its own mailbox and adds the n_ranks values IN
 * RANK ORDER.  Every rank therefore forms bit-identical sums, run to run and whatever the arrival order; there is no
 * extra launch in the step, and cude_adam_run keeps replaying captured graphs (it cannot with RCCL calls).  A wait
 * that sees no peer for `timeout_s` gives up: the step's loss becomes NaN and the next call that synchronises returns
 * CUDE_ERR_COMM (a kernel never spins for ever).
 *   every rank:  cude_xchg_export(ctx, n_ranks, rank, mine)        -> 128 bytes describing its mailbox
 *   the host:    all-gather of those bytes (any channel: torch.distributed, MPI.jl, a file)
 *   every rank:  cude_xchg_attach(ctx, all[n_ranks][128], timeout_s)   (collective: ends with a self-test)
 *   the host:    do ALL ranks report CUDE_OK?  (one logical AND over the ranks, same channel)
 *                no -> every rank, the successful ones too: cude_xchg_detach(ctx), and from the top: the next export
 *                      offers the next kind of mailbox memory (uncached device memory, then fine-grained, then ordinary
 *                      device memory -- the last only when all ranks share one device, or with option
 *                      "xchg_allow_plain"); after the third kind cude_xchg_export fails and the caller falls back to
 *                      cude_comm_init (RCCL).  cude/parallel.py `attach_exchange` and julia/CUDEHip.jl `attach_exchange!`
 *                      are this loop.
 * A kind is given up by the exporting rank itself when its allocation, clearing or hipIpcGetMemHandle fails; by all
 * ranks when any rank cannot open a peer's handle or fails the self-test (4 sum rounds over every column and both
 * parities with values that change per round, one max round) -- whether a peer ON ANOTHER DEVICE can write a given
 * kind is only known then.  A rank whose export failed contributes 128 zero bytes, which fails everybody's attach.
 * Ranks may live in one process (several contexts; at most 4 per device: their spinning reduction kernels share the
 * device's hardware queues -- a rehearsal configuration) or in several; all on ONE node.  Attach before the population is
 * uploaded, as with cude_comm_init (the global subject count is summed there).  With an exchange attached every
 * reduction of the library goes through it and no RCCL communicator is needed; when both are attached the exchange
 * is used while enabled (cude_xchg_enable(ctx, 0) switches to the communicator, for comparisons).  Every entry point
 * that synchronises behind an exchange launch reads the exchange's status word (page-locked host memory) and returns
 * CUDE_ERR_COMM if a wait gave up in ANY column. */
int32_t cude_xchg_export(cude_ctx* ctx, int32_t n_ranks, int32_t rank, uint8_t handle[CUDE_XCHG_HANDLE_BYTES]);
int32_t cude_xchg_attach(cude_ctx* ctx, const uint8_t* handles /* [n_ranks][CUDE_XCHG_HANDLE_BYTES] */, double timeout_s);
/* Releases this context's exported or attached exchange (no-op when a failed cude_xchg_attach already has); the next
 * cude_xchg_export starts one kind further down the order of preference than the last one did (the ranks move through
 * these levels together). */
int32_t cude_xchg_detach(cude_ctx* ctx);
int32_t cude_xchg_enable(cude_ctx* ctx, int32_t enabled);
/* n_ranks / rank of the attached exchange (1, 0 without), how its mailbox memory was allocated (3 = uncached, 1 =
 * fine-grained, 0 = ordinary device memory) and how many device-side waits have run out of time so far. */
int32_t cude_xchg_info(cude_ctx* ctx, int32_t* n_ranks, int32_t* rank, int32_t* memory_kind, int32_t* timeouts);

/* --- run-time options.
 * The network's activation functions -- `chain(widths, activations; output_activation)`, src/neural-network.jl:42-58 --:
 * "hidden_activation" = "tanh" (default) | "relu" | "sigmoid", "output_activation" = "softplus" (default) | "identity";
 * set them before the population is uploaded.  tanh / softplus (every network of the reference's scripts) run on the
 * tuned kernels; the other combinations on the general evaluation path of the same kernels, compiled for the shapes
 * 2-4-4-1, 2-6-6-1, 3-4-4-1 (c-peptide) and 4-3x5-1, 4-3x3-1 (suppression): CUDE_ERR_UNSUPPORTED otherwise.
 * Implementation switches (cude_ctx.h `Options` lists them): launch-path override of the tests ("cpep_path" = "1" |
 * "2:L" | "3:B:L"), "cpep_keep", "supp_store", "supp_ckpt", "tape_steps", "exp_table", "ms_split", "auto_regroup",
 * "poll_pinned", "debug_selector", "xchg_allow_plain", "xchg_fail_kinds" (tests), "force_fallback" (tests: cude_set_network takes the fallback kernel for tuned shapes too), "adaptive_team" (adaptive mode, small c-peptide
 * populations: a step's five network evaluations on five waves; 0 = the one-wave kernels), "fit_spec" (cude_fit_conditional: probes per forward launch, 0 = one, 1 ... 4 = golden-section steps per launch, -1 = by size), "mh_spec" (speculative Metropolis steps per
 * launch pair in cude_mh_estep / cude_mh_chain: 0 off, 2 ... 4, -1 = by population size; the chain is the same bit for bit).  Values are decimal integers as text unless noted.  Every option is also read once
 * at cude_create from its environment variable (CUDE_CPEP_PATH, CUDE_CPEP_KEEP, CUDE_SUPP_STORE, CUDE_SUPP_CKPT,
 * CUDE_TAPE_STEPS, CUDE_NO_EXPTAB, CUDE_NO_MS_SPLIT, CUDE_NO_AUTO_REGROUP, CUDE_NO_POLL_PINNED, CUDE_DEBUG_SELECTOR, CUDE_ALLOW_PLAIN_MAILBOX, CUDE_XCHG_FAIL_KINDS, CUDE_MH_SPEC, CUDE_FIT_SPEC, CUDE_NO_ADAPTIVE_TEAM).
 * Options that shape the launch path take effect at the next cude_set_population_*.  No reference line: these are
 * properties of this implementation. */
int32_t cude_set_option(cude_ctx* ctx, const char* name, const char* value);

#ifdef __cplusplus
}
#endif
#endif /* CUDE_H */
