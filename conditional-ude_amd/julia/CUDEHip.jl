# CUDEHip.jl -- the reference's Julia API for the accelerated path, on top of the C ABI of libcude_hip.so.
#
# Drop-in for the hot-path functions of Computational-Biology-TUe/conditional-ude with THEIR signatures:
#   chain(width, depth, tanh; input_dims)                                        src/neural-network.jl:105-107
#   CPeptideConditionalUDEModel(glucose, timepoints, age, chain, cpeptide, t2dm) src/c-peptide-models.jl:170-194
#   loss(θ, (models, timepoints, cpeptide_data))   loss(θ, (model, timepoints, data))   loss(β, (model, t, data, nn))
#   loss_sigma(θ, (...))                                                         src/parameter-estimation.jl:56-140
#   train(models, timepoints, cpeptide_data, rng; ...)  train(models, t, data, nn; ...)  train_with_sigma  evaluate_model
#                                                                                src/parameter-estimation.jl:272-433
#   likelihood_profile(β, nn, model, timepoints, data, lower, upper, sigma; steps)   src/likelihood-profiles.jl:4-17
#   suppression_loss(p, (prob, data, timepoints, λ))   simul(p, prob, data, timepoints)   fit_suppression_model
#                                                                                suppression/src/suppression_model.jl:107-177
#   simulate / individual_log_likelihood / SAEM(individuals, nn0, network; ...)  src/saem.jl:31-66,134-237
# A script of the reference switches by replacing `include("src/parameter-estimation.jl")` (etc.) with
# `include("CUDEHip.jl"); using .CUDEHip` -- models, timepoints and data are passed exactly as before; a device
# population is built once per (models, timepoints, data) and cached.
#
# Written against include/cude.h, not executed in this pipeline (Julia is not installed in the build image): the same
# ABI is exercised end to end by the Python host (conditional-ude_amd/cude/), and tests/test_julia_shim.py parses every
# `ccall` below and checks symbol, arity and argument widths against the header.
module CUDEHip

using Random: AbstractRNG, randn, rand, randperm
using Statistics: mean, var

export chain, neural_network_model, CPeptideConditionalUDEModel, CPeptideCUDEModel, CPeptideConditionalCovariateUDEModel,
       CPeptideUDEModel,
       loss, loss_sigma, loss_and_gradient!, train, train_with_sigma, evaluate_model, likelihood_profile,
       SuppressionProblem, suppression_loss, simul, fit_suppression_model, simulate, individual_log_likelihood, SAEM

const LIB = get(ENV, "CUDE_HIP_LIB", "libcude_hip.so")
const MODEL_CPEP, MODEL_SUPP, MODEL_CPEP_SYM = Int32(0), Int32(1), Int32(2)
const ADAPTIVE = 0                       # n_steps = ADAPTIVE: the reference's own adaptive Tsit5 (the default, default_steps)
const DEFAULT_STEPS = 30

struct Config
    model::Int32; n_state::Int32; nn_in::Int32; nn_width::Int32; nn_depth::Int32
    n_steps::Int32; device::Int32; cond_space::Int32; lambda::Float64
end

check(st) = st < 0 ? error(unsafe_string(ccall((:cude_last_error, LIB), Cstring, ()))) : st

# ----------------------------------------------------------------------------------------------- low-level context
mutable struct Ctx
    h::Ptr{Cvoid}; P::Int; N::Int; T::Int; n_state::Int
end

function Ctx(cfg::Config)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cude_create, LIB), Int32, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, h))
    P = ccall((:cude_n_params, LIB), Int32, (Int32, Int32, Int32), cfg.nn_in, cfg.nn_width, cfg.nn_depth)
    c = Ctx(h[], P, 0, 0, cfg.n_state)
    finalizer(x -> ccall((:cude_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), c)
    c
end

function device_count()
    n = Ref{Int32}(0)
    check(ccall((:cude_device_count, LIB), Int32, (Ref{Int32},), n))
    Int(n[])
end

set_tolerances!(c::Ctx, abstol, reltol) =
    check(ccall((:cude_set_tolerances, LIB), Int32, (Ptr{Cvoid}, Float64, Float64), c.h, abstol, reltol))

# accepted steps (t_n, dt_n) of `subject` (1-based) in the last gradient evaluation of the adaptive mode: `sol.t`
function adaptive_steps(c::Ctx, subject::Integer)
    n = Ref{Int32}(0)
    check(ccall((:cude_adaptive_steps, LIB), Int32, (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                c.h, subject - 1, 0, C_NULL, C_NULL, n))
    t = Vector{Float64}(undef, n[]); dt = similar(t)
    GC.@preserve t dt check(ccall((:cude_adaptive_steps, LIB), Int32,
                                  (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                                  c.h, subject - 1, n[], t, dt, n))
    return t, dt
end

# glucose / cpeptide as Julia column-major N×T matrices: ld_subject = 1, ld_time = N (no host copy)
function set_population!(c::Ctx, timepoints::Vector{Float64}, glucose::Matrix{Float64}, cpeptide::Matrix{Float64},
                         ages::Vector{Float64}, t2dm::Vector{UInt8})
    N, T = size(glucose)
    GC.@preserve timepoints glucose cpeptide ages t2dm check(ccall((:cude_set_population_cpep, LIB), Int32,
        (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{UInt8}),
        c.h, N, T, timepoints, glucose, cpeptide, 1, N, ages, t2dm))
    c.N = N; c.T = T
    c
end

# individual_data[3 × T × N] exactly as the reference holds it (column-major)
function set_population_supp!(c::Ctx, timepoints::Vector{Float64}, data::Array{Float64,3})
    _, T, N = size(data)
    GC.@preserve timepoints data check(ccall((:cude_set_population_supp, LIB), Int32,
        (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}), c.h, N, T, timepoints, data))
    c.N = N; c.T = T
    c
end

function set_params!(c::Ctx, nn, cond)
    nnv = nn === nothing ? nothing : Vector{Float64}(vec(nn))
    cv = cond === nothing ? nothing : Vector{Float64}(vec(cond))
    GC.@preserve nnv cv check(ccall((:cude_set_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        c.h, nnv === nothing ? C_NULL : pointer(nnv), cv === nothing ? C_NULL : pointer(cv)))
end

function get_params(c::Ctx)
    nn = Vector{Float64}(undef, c.P); cond = Vector{Float64}(undef, c.N)
    GC.@preserve nn cond check(ccall((:cude_get_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, nn, cond))
    nn, cond
end

# population loss at the context's parameters; optionally the per-subject SSEs and the states [n_state × T × N]
function forward(c::Ctx; want_sse = false, want_traj = false)
    l = Ref{Float64}()
    sse = want_sse ? Vector{Float64}(undef, c.N) : nothing
    traj = want_traj ? Array{Float64,3}(undef, c.n_state, c.T, c.N) : nothing
    GC.@preserve sse traj check(ccall((:cude_forward, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, l, sse === nothing ? C_NULL : pointer(sse), traj === nothing ? C_NULL : pointer(traj)))
    l[], sse, traj
end

function loss_grad(c::Ctx)
    l = Ref{Float64}(); gnn = Vector{Float64}(undef, c.P); gcond = Vector{Float64}(undef, c.N)
    GC.@preserve gnn gcond check(ccall((:cude_loss_grad, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, l, gnn, gcond))
    l[], gnn, gcond
end

function n_failed(c::Ctx)
    n = Ref{Int64}(0)
    check(ccall((:cude_n_failed, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}), c.h, n))
    Int(n[])
end

# states of every subject at arbitrary times inside the time span: [n_state × n_times × N]
function simulate_dense(c::Ctx, times::Vector{Float64})
    out = Array{Float64,3}(undef, c.n_state, length(times), c.N)
    GC.@preserve times out check(ccall((:cude_simulate, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}),
        c.h, length(times), times, out))
    out
end

# screening: nn_sets is P×K, cond_sets N×K (column-major = the row-major [K][P] / [K][N] of the ABI)
function multistart_forward(c::Ctx, nn_sets::Matrix{Float64}, cond_sets::Matrix{Float64})
    K = size(nn_sets, 2); losses = Vector{Float64}(undef, K)
    GC.@preserve nn_sets cond_sets losses check(ccall((:cude_multistart_forward, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), c.h, K, nn_sets, cond_sets, losses))
    losses
end

# screening with the selection on the device: gen!(nn P×count, cond N×count, first) fills one chunk of candidates
# (first is 0-based); returns (indices (1-based), losses, neural P×n_keep, conditional N×n_keep) of the best n_keep
function screen_candidates(c::Ctx, gen!, n_candidates::Integer, n_keep::Integer)
    P, N = c.P, c.N
    function thunk(first::Int64, count::Int32, nnp::Ptr{Float64}, cp::Ptr{Float64}, ::Ptr{Cvoid})::Int32
        gen!(unsafe_wrap(Array, nnp, (P, Int(count))), unsafe_wrap(Array, cp, (N, Int(count))), first); Int32(0)
    end
    cb = @cfunction($thunk, Int32, (Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
    k = min(n_keep, n_candidates)
    idx = Vector{Int64}(undef, k); losses = Vector{Float64}(undef, k)
    nn = Matrix{Float64}(undef, P, k); cond = Matrix{Float64}(undef, N, k)
    GC.@preserve cb idx losses nn cond check(ccall((:cude_screen_candidates, LIB), Int32,
        (Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, n_candidates, k, cb, C_NULL, idx, losses, nn, cond))
    idx .+ 1, losses, nn, cond
end

function multistart_loss_grad(c::Ctx, nn_sets::Matrix{Float64}, cond_sets::Matrix{Float64})
    K = size(nn_sets, 2)
    losses = Vector{Float64}(undef, K); g_nn = similar(nn_sets); g_cond = similar(cond_sets)
    GC.@preserve nn_sets cond_sets losses g_nn g_cond check(ccall((:cude_multistart_loss_grad, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, K, nn_sets, cond_sets, losses, g_nn, g_cond))
    losses, g_nn, g_cond
end

# Adam(η) × adam_iters then LBFGS(BackTracking) × lbfgs_iters for all columns side by side; returns
# (neural P×K, conditional N×K, objectives K, loss traces (adam_iters + lbfgs_iters)×K with NaN after a run stopped)
function train_restarts(c::Ctx, nn_sets::Matrix{Float64}, cond_sets::Matrix{Float64}; adam_iters = 1000, η = 1e-2,
                        lbfgs_iters = 1000)
    K = size(nn_sets, 2); nn = similar(nn_sets); cond = similar(cond_sets); obj = Vector{Float64}(undef, K)
    trace = Matrix{Float64}(undef, adam_iters + lbfgs_iters, K)
    GC.@preserve nn_sets cond_sets nn cond obj trace check(ccall((:cude_train_restarts, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}), c.h, K, nn_sets, cond_sets, adam_iters, η, lbfgs_iters, nn, cond, obj, trace))
    nn, cond, obj, trace
end

# per-subject minimisers of SSE_i(β) + w (β - μ)^2 over [lower, upper]; returns (β, objective, SSE)
function fit_conditional(c::Ctx, lower, upper; n_grid = 41, n_iters = 48, penalty_weight = 0.0, penalty_center = 0.0)
    β = Vector{Float64}(undef, c.N); obj = similar(β); sse = similar(β)
    GC.@preserve β obj sse check(ccall((:cude_fit_conditional, LIB), Int32,
        (Ptr{Cvoid}, Float64, Float64, Int32, Int32, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, lower, upper, n_grid, n_iters, penalty_weight, penalty_center, β, obj, sse))
    β, obj, sse
end

# SSE_i(values[k]) as an N×K matrix
function profile_conditional(c::Ctx, values::Vector{Float64})
    sse = Matrix{Float64}(undef, c.N, length(values))
    GC.@preserve values sse check(ccall((:cude_profile_conditional, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), c.h, length(values), values, sse))
    sse
end

adam_init!(c::Ctx, η; β1 = 0.9, β2 = 0.999, ϵ = 1e-8) =
    check(ccall((:cude_adam_init, LIB), Int32, (Ptr{Cvoid}, Float64, Float64, Float64, Float64), c.h, η, β1, β2, ϵ))

function adam_step!(c::Ctx)
    l = Ref{Float64}()
    check(ccall((:cude_adam_step, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}), c.h, l))
    l[]
end

function adam_run!(c::Ctx, iters::Integer)
    losses = Vector{Float64}(undef, iters)
    GC.@preserve losses check(ccall((:cude_adam_run, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}), c.h, iters, losses))
    losses
end

synchronize(c::Ctx) = check(ccall((:cude_synchronize, LIB), Int32, (Ptr{Cvoid},), c.h))

# E-step of SAEM: draws are N×n_mc matrices (= [n_mc][N] row-major)
function mh_estep!(c::Ctx, normals::Matrix{Float64}, uniforms::Matrix{Float64}, σ, prior_η, Ω, proposal_std;
                   temperature = 1.0, γ = 1.0)
    N, n_mc = size(normals); accepted = zeros(Int64, N)
    GC.@preserve normals uniforms accepted check(ccall((:cude_mh_estep, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Int64}),
        c.h, n_mc, normals, uniforms, σ, prior_η, Ω, proposal_std, temperature, γ, accepted))
    accepted
end

function mh_estep!(c::Ctx, n_mc::Integer, σ, prior_η, Ω, proposal_std; temperature = 1.0, γ = 1.0)
    accepted = zeros(Int64, c.N)
    GC.@preserve accepted check(ccall((:cude_mh_estep, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Int64}),
        c.h, Int32(n_mc), C_NULL, C_NULL, σ, prior_η, Ω, proposal_std, temperature, γ, accepted))
    accepted
end

function mh_chain!(c::Ctx, normals::Matrix{Float64}, uniforms::Matrix{Float64}, σ, prior_η, Ω, proposal_std;
                   temperature = 1.0, γ = 1.0)
    N, n_mc = size(normals); accepted = zeros(Int64, N); samples = similar(normals)
    GC.@preserve normals uniforms accepted samples check(ccall((:cude_mh_chain, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Int64},
         Ptr{Float64}), c.h, n_mc, normals, uniforms, σ, prior_η, Ω, proposal_std, temperature, γ, accepted, samples))
    accepted, samples
end

# freeze shared parameters (mask of 0 / 1, length P; `nothing` lifts it)
function set_param_mask!(c::Ctx, mask)
    m = mask === nothing ? nothing : Vector{Float64}(vec(mask))
    GC.@preserve m check(ccall((:cude_set_param_mask, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}),
        c.h, m === nothing ? C_NULL : pointer(m)))
end

# device-side draws of the Metropolis steps: mh_estep!(c, n_mc, σ, ...) without draw matrices
set_rng!(c::Ctx, seed::Integer, subject_offset::Integer = 0) =
    check(ccall((:cude_set_rng, LIB), Int32, (Ptr{Cvoid}, UInt64, Int64), c.h, UInt64(seed), Int64(subject_offset)))

function rng_draws(c::Ctx, first_step::Integer, n_steps::Integer)
    z = Matrix{Float64}(undef, c.N, n_steps); u = Matrix{Float64}(undef, c.N, n_steps)
    GC.@preserve z u check(ccall((:cude_rng_draws, LIB), Int32, (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}),
        c.h, Int64(first_step), Int32(n_steps), z, u))
    z, u
end

# L-BFGS + BackTracking of the library for any Julia objective fg!(g, x) -> f (host only)
function lbfgs_minimize(fg!, x0::Vector{Float64}; maxiters = 1000)
    n = length(x0)
    function thunk(xp::Ptr{Float64}, nn::Int32, fp::Ptr{Float64}, gp::Ptr{Float64}, ::Ptr{Cvoid})::Int32
        x = unsafe_wrap(Array, xp, Int(nn)); g = unsafe_wrap(Array, gp, Int(nn))
        unsafe_store!(fp, fg!(g, x)); Int32(0)
    end
    cb = @cfunction($thunk, Int32, (Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
    x = similar(x0); f = Ref{Float64}(); it = Ref{Int32}(); calls = Ref{Int32}(); conv = Ref{Int32}()
    GC.@preserve x0 x cb check(ccall((:cude_lbfgs_minimize, LIB), Int32,
        (Int32, Ptr{Float64}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ref{Int32}, Ref{Int32}, Ref{Int32}),
        n, x0, maxiters, cb, C_NULL, x, f, it, calls, conv))
    (x = x, f = f[], iterations = Int(it[]), f_calls = Int(calls[]), converged = conv[] != 0)
end

# the same for a vector sharded over ranks: x = [shared (n_shared); local]; reduce!(values, op) sums (op 0) / maximises
# (op 1) over the ranks in place, e.g.  (v, op) -> MPI.Allreduce!(v, op == 0 ? MPI.SUM : MPI.MAX, comm)
function lbfgs_minimize_sharded(fg!, reduce!, x0::Vector{Float64}, n_shared::Integer; maxiters = 1000)
    n = length(x0)
    function thunk(xp::Ptr{Float64}, nn::Int32, fp::Ptr{Float64}, gp::Ptr{Float64}, ::Ptr{Cvoid})::Int32
        x = unsafe_wrap(Array, xp, Int(nn)); g = unsafe_wrap(Array, gp, Int(nn))
        unsafe_store!(fp, fg!(g, x)); Int32(0)
    end
    function red(vp::Ptr{Float64}, count::Int32, op::Int32, ::Ptr{Cvoid})::Int32
        reduce!(unsafe_wrap(Array, vp, Int(count)), Int(op)); Int32(0)
    end
    cb = @cfunction($thunk, Int32, (Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
    rcb = @cfunction($red, Int32, (Ptr{Float64}, Int32, Int32, Ptr{Cvoid}))
    x = similar(x0); f = Ref{Float64}(); it = Ref{Int32}(); calls = Ref{Int32}(); conv = Ref{Int32}()
    GC.@preserve x0 x cb rcb check(ccall((:cude_lbfgs_minimize_sharded, LIB), Int32,
        (Int32, Int32, Ptr{Float64}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ref{Int32},
         Ref{Int32}, Ref{Int32}), n, n_shared, x0, maxiters, cb, rcb, C_NULL, x, f, it, calls, conv))
    (x = x, f = f[], iterations = Int(it[]), f_calls = Int(calls[]), converged = conv[] != 0)
end

# ---- multi-GPU: one process per GPU, subjects sharded (INTEGRATION.md "Multi-GPU launch")
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id check(ccall((:cude_comm_unique_id, LIB), Int32, (Ptr{UInt8},), id))
    id
end

comm_init!(c::Ctx, n_ranks, rank, id::Vector{UInt8}) = GC.@preserve id check(ccall((:cude_comm_init, LIB), Int32,
    (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), c.h, n_ranks, rank, id))

function comm_info(c::Ctx)
    n = Ref{Int32}(); r = Ref{Int32}(); v = Ref{Int32}()
    check(ccall((:cude_comm_info, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}), c.h, n, r, v))
    (ranks = Int(n[]), rank = Int(r[]), rccl_version = Int(v[]))
end

function comm_allreduce!(c::Ctx, values::Vector{Float64})
    GC.@preserve values check(ccall((:cude_comm_allreduce_host, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32),
        c.h, values, length(values)))
    values
end

# ---- multi-GPU without a collective library: the peer-write exchange (include/cude.h "cude_xchg_*").  Every rank
# exports its mailbox, the 128-byte handles travel over any channel (Distributed.jl: `handles = fetch.(...)`; MPI.jl:
# `MPI.Allgather`), every rank attaches all of them (collective: ends with a self-test).  Afterwards `loss`, the Adam
# steps, `train` and SAEM sum over the ranks inside the reduction kernels: no RCCL, captured graphs keep working.
function xchg_export(c::Ctx, n_ranks, rank)
    h = Vector{UInt8}(undef, 128)
    GC.@preserve h check(ccall((:cude_xchg_export, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), c.h, n_ranks, rank, h))
    h
end

function xchg_attach!(c::Ctx, handles::Vector{Vector{UInt8}}; timeout_s = 20.0)
    all = reduce(vcat, handles)
    GC.@preserve all check(ccall((:cude_xchg_attach, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Float64), c.h, all, timeout_s))
end

# releases the exported / attached exchange; the next xchg_export offers the next kind of mailbox memory
xchg_detach!(c::Ctx) = check(ccall((:cude_xchg_detach, LIB), Int32, (Ptr{Cvoid},), c.h))

# The attach protocol of include/cude.h (mirror of cude/parallel.py `attach_exchange`): export, all-gather the handles,
# attach, AGREE on the outcome; if any rank failed all detach and go again with the next kind of mailbox memory.
# `allgather(bytes) -> Vector{Vector{UInt8}}` and `anyfailed(flag::Bool) -> Bool` are the caller's channel (MPI.Allgather /
# MPI.Allreduce(|), Distributed.jl fetches).  Returns true with the exchange attached on every rank, false with it
# released on every rank (fall back to comm_init!).
function attach_exchange!(c::Ctx, n_ranks, rank, allgather, anyfailed; timeout_s = 20.0)
    for attempt in 1:3
        mine = zeros(UInt8, 128)
        failed = false
        exhausted = false
        try
            mine = xchg_export(c, n_ranks, rank)
        catch
            failed = true
            exhausted = true
        end
        handles = allgather(mine)
        if !failed
            try
                xchg_attach!(c, handles; timeout_s = timeout_s)
            catch
                failed = true
            end
        end
        anyfailed(failed) || return true
        xchg_detach!(c)
        anyfailed(exhausted) && break
    end
    false
end

xchg_enable!(c::Ctx, on::Bool) = check(ccall((:cude_xchg_enable, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, on ? 1 : 0))

function xchg_info(c::Ctx)
    n = Ref{Int32}(); r = Ref{Int32}(); k = Ref{Int32}(); t = Ref{Int32}()
    check(ccall((:cude_xchg_info, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}), c.h, n, r, k, t))
    (ranks = Int(n[]), rank = Int(r[]), memory_kind = Int(k[]), timed_out_waits = Int(t[]))
end

# the general form of `chain(widths, activation_functions; ...)` (src/neural-network.jl:42-58): per-layer widths and one
# activation code per hidden layer + the output layer's (include/cude.h CUDE_ACT_*); before the population is uploaded
const ACTIVATION_CODES = Dict(:tanh => 0, :relu => 1, :sigmoid => 2, :softplus => 3, :identity => 4)
function set_network!(c::Ctx, widths::AbstractVector{<:Integer}, activations::AbstractVector{Symbol})
    length(activations) == length(widths) + 1 || throw(ArgumentError("one activation per hidden layer and one for the output layer"))
    w = Int32.(widths); a = Int32[ACTIVATION_CODES[f] for f in activations]
    GC.@preserve w a check(ccall((:cude_set_network, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}), c.h, length(w), w, a))
    c.P = network_info(c).n_params
    c
end
function network_info(c::Ctx)
    p = Ref{Int32}(); g = Ref{Int32}()
    check(ccall((:cude_network_info, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), c.h, p, g))
    (n_params = Int(p[]), fallback_kernel = g[] != 0)
end

# run-time options of a context (launch-path override, auto_regroup, poll_pinned ...: include/cude.h)
set_option!(c::Ctx, name::AbstractString, value) =
    check(ccall((:cude_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Cstring), c.h, name, string(value)))

# bring-your-own collective (MPI.jl): partial = loss_grad_partial(c); MPI.Allreduce!(partial, +, comm); adam_apply!(c, partial)
function set_global_subjects!(c::Ctx, n_global; scale = nothing)
    sc = scale === nothing ? nothing : Vector{Float64}(scale)
    GC.@preserve sc check(ccall((:cude_set_global_subjects, LIB), Int32, (Ptr{Cvoid}, Float64, Ptr{Float64}),
        c.h, n_global, sc === nothing ? C_NULL : pointer(sc)))
end

function get_scale(c::Ctx)
    sc = Vector{Float64}(undef, 3); n = Ref{Float64}()
    GC.@preserve sc check(ccall((:cude_get_scale, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), c.h, sc, n))
    sc, n[]
end

function loss_grad_partial(c::Ctx; want_cond_grad = false)
    part = Vector{Float64}(undef, c.P + 2); gc = want_cond_grad ? Vector{Float64}(undef, c.N) : nothing
    GC.@preserve part gc check(ccall((:cude_loss_grad_partial, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        c.h, part, gc === nothing ? C_NULL : pointer(gc)))
    part, gc
end

function adam_apply!(c::Ctx, reduced::Vector{Float64})
    l = Ref{Float64}()
    GC.@preserve reduced check(ccall((:cude_adam_apply, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), c.h, reduced, l))
    l[]
end

# the same exchange without the host round trip: the P+2 doubles stay on the device, a GPU-aware collective (RCCL through
# another binding, MPI.jl built against a ROCm-aware MPI) reduces them in place between the two calls
function partial_buffer(c::Ctx)
    p = Ref{Ptr{Float64}}(C_NULL); n = Ref{Int32}(0)
    check(ccall((:cude_partial_buffer, LIB), Int32, (Ptr{Cvoid}, Ref{Ptr{Float64}}, Ref{Int32}), c.h, p, n))
    p[], Int(n[])
end
loss_grad_partial_device!(c::Ctx) = check(ccall((:cude_loss_grad_partial_device, LIB), Int32, (Ptr{Cvoid},), c.h))
function adam_apply_device!(c::Ctx)
    l = Ref{Float64}()
    check(ccall((:cude_adam_apply_device, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}), c.h, l))
    l[]
end

function adaptive_regroup!(c::Ctx)
    b = Ref{Int32}(0); a = Ref{Int32}(0)
    check(ccall((:cude_adaptive_regroup, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), c.h, b, a))
    Int(b[]), Int(a[])
end

function grad_occupancy(c::Ctx)
    n = Ref{Int32}(0)
    check(ccall((:cude_grad_occupancy, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}), c.h, n))
    Int(n[])
end

# on = true / 1: events around every ensemble launch; n > 1: around every n-th; false / 0: off
set_kernel_timing!(c::Ctx, on::Integer) = check(ccall((:cude_set_kernel_timing, LIB), Int32, (Ptr{Cvoid}, Int32), c.h, Int32(on)))

function kernel_time_ms(c::Ctx)
    ms = Ref{Float64}(); n = Ref{Int64}()
    check(ccall((:cude_kernel_time_ms, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Int64}), c.h, ms, n))
    ms[], Int(n[])
end
function kernel_time_stats(c::Ctx)
    ms = Ref{Float64}(); med = Ref{Float64}(); mn = Ref{Float64}(); n = Ref{Int64}()
    check(ccall((:cude_kernel_time_stats, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ref{Int64}), c.h, ms, med, mn, n))
    (mean = ms[], median = med[], minimum = mn[], launches = Int(n[]))
end

# ----------------------------------------------------------------------------------------------- reference API: models
# chain(width, depth, act; input_dims = 2, output_activation = softplus): what the kernels compile -- equal widths, ONE
# hidden activation out of tanh / relu / sigmoid, one output unit with softplus or identity (recognised by name)
struct Chain
    input_dims::Int; width::Int; depth::Int
    widths::Vector{Int}     # empty = equal widths; otherwise the network is carried zero-padded to width = maximum(widths)
    activation::String; output_activation::String
    # non-empty: the general form -- one activation name per hidden layer, any of ACTIVATION_CODES, `widths` holds every
    # layer's own width and parameter vectors have SimpleChains' own (unpadded) layout; runs on the library's fallback kernel
    layer_activations::Vector{String}
end
Chain(input_dims::Integer, width::Integer, depth::Integer, widths::Vector{Int}, a::String, o::String) =
    Chain(input_dims, width, depth, widths, a, o, String[])
Chain(input_dims::Integer, width::Integer, depth::Integer, widths::Vector{Int} = Int[]) =
    Chain(input_dims, width, depth, widths, "tanh", "softplus")
is_general(c::Chain) = !isempty(c.layer_activations)
softplus(x) = log(1 + exp(x))                                   # src/neural-network.jl:13-15
act_name(f) = (n = string(nameof(f)); n in ("σ", "sigmoid_fast") ? "sigmoid" : n == "tanh_fast" ? "tanh" : n)
function chain(width::Integer, depth::Integer, activation = tanh; input_dims::Integer = 2, output_dims::Integer = 1,
               output_activation = softplus)
    a, o = act_name(activation), act_name(output_activation)
    (output_dims == 1 && a in ("tanh", "relu", "sigmoid") && o in ("softplus", "identity")) ||
        error("compiled into the HIP kernels: hidden tanh / relu / sigmoid, one output with softplus / identity")
    Chain(input_dims, width, depth, Int[], a, o)
end
# the activation functions travel to the library as options of the context, before the population is uploaded
function configure!(c, net::Chain)
    if is_general(net)
        return set_network!(c, net.widths, Symbol.(vcat(net.layer_activations, net.output_activation)))
    end
    net.activation == "tanh" || set_option!(c, "hidden_activation", net.activation)
    net.output_activation == "softplus" || set_option!(c, "output_activation", net.output_activation)
    c
end
# chain(widths, activation_functions; input_dims, output_activation) in its general form (src/neural-network.jl:42-58; the
# docstring's chain([10, 20, 30], [tanh, relu, softplus]; input_dims = 4)): one function per hidden layer
function chain(widths::AbstractVector{<:Integer}, activation_functions::AbstractVector; input_dims::Integer = 2,
               output_dims::Integer = 1, output_activation = softplus)
    isempty(widths) && throw(ArgumentError("Input widths must be non-empty."))
    length(widths) == length(activation_functions) ||
        throw(ArgumentError("The number of widths must match the number of activation functions."))
    output_dims == 1 || error("the models of the reference use ONE network output")
    names = [act_name(f) for f in activation_functions]; o = act_name(output_activation)
    all(n -> haskey(ACTIVATION_CODES, Symbol(n)), vcat(names, o)) || error("activation functions of the library: $(keys(ACTIVATION_CODES))")
    Chain(input_dims, maximum(widths), length(widths), collect(Int, widths), "tanh", o, names)
end
# chain(widths, tanh) (src/neural-network.jl:42-58): unequal widths = the equal-width network of width maximum(widths)
# whose extra units have zero weights, frozen through cude_set_param_mask; parameter vectors carry THAT layout
function chain(widths::AbstractVector{<:Integer}, activation = tanh; input_dims::Integer = 2, output_dims::Integer = 1)
    isempty(widths) && throw(ArgumentError("Input widths must be non-empty."))
    (output_dims == 1 && activation === tanh) || error("only tanh hidden layers with one softplus output are compiled into the HIP kernels")
    all(==(widths[1]), widths) ? Chain(input_dims, widths[1], length(widths)) :
                                 Chain(input_dims, maximum(widths), length(widths), collect(Int, widths))
end
neural_network_model(depth::Integer, width::Integer; input_dims::Integer = 2) = Chain(input_dims, width, depth)
function n_params(c::Chain)
    if is_general(c)
        p = 0; fan = c.input_dims
        for w in vcat(c.widths, 1); p += w * fan + w; fan = w; end
        return p
    end
    Int(ccall((:cude_n_params, LIB), Int32, (Int32, Int32, Int32), c.input_dims, c.width, c.depth))
end

# SimpleChains' layout of an unequal-width network -> the padded layout (and the 0 / 1 mask of its live entries)
function pad_network(c::Chain, p::AbstractVector{<:Real})
    out = zeros(n_params(c)); at = 0; bt = 0; fan = c.input_dims; W = c.width
    for (k, w) in enumerate(vcat(c.widths, 1))
        wb = k <= c.depth ? W : 1; fb = k == 1 ? c.input_dims : W
        M = zeros(wb, fb); M[1:w, 1:fan] .= reshape(p[at+1:at+w*fan], w, fan)
        out[bt+1:bt+wb*fb] .= vec(M); out[bt+wb*fb+1:bt+wb*fb+w] .= p[at+w*fan+1:at+w*fan+w]
        at += w * fan + w; bt += wb * fb + wb; fan = w
    end
    out
end
param_mask(c::Chain) = isempty(c.widths) ? nothing :
    Float64.(pad_network(c, ones(sum(w * f + w for (w, f) in zip(vcat(c.widths, 1), vcat(c.input_dims, c.widths))))) .!= 0)

# SimpleChains.init_params restated: a TurboDense{true} layer is ONE out × (in + 1) matrix [W b], all of it Glorot-normal
function init_params(c::Chain; rng::AbstractRNG)
    p = Float64[]; fan = c.input_dims
    for out in vcat(isempty(c.widths) ? fill(c.width, c.depth) : c.widths, 1)
        append!(p, randn(rng, out * (fan + 1)) .* sqrt(2 / (out + fan + 1))); fan = out
    end
    isempty(c.widths) ? p : pad_network(c, p)
end

abstract type CPeptideModel end
struct CPeptideConditionalUDEModel <: CPeptideModel
    glucose::Vector{Float64}; timepoints::Vector{Float64}; age::Float64; chain::Chain
    cpeptide::Vector{Float64}; t2dm::Bool
end
CPeptideConditionalUDEModel(glucose_data::AbstractVector{<:Real}, glucose_timepoints::AbstractVector{<:Real}, age::Real,
                            network::Chain, cpeptide_data::AbstractVector{<:Real}, t2dm::Bool) =
    CPeptideConditionalUDEModel(Vector{Float64}(glucose_data), Vector{Float64}(glucose_timepoints), Float64(age),
                                network, Vector{Float64}(cpeptide_data), t2dm)
const CPeptideCUDEModel = CPeptideConditionalUDEModel          # the name of the reference's docstrings / stale script
CPeptideConditionalCovariateUDEModel(g, t, age, network::Chain, c, t2dm) =
    network.input_dims == 3 ? CPeptideConditionalUDEModel(g, t, age, network, c, t2dm) :
    error("the covariate model takes a network with input_dims = 3")

# CPeptideUDEModel (src/c-peptide-models.jl:144-168): the non-conditional UDE, production = network([ΔG]) - network([0])
# (:76-84).  On the device it rides on the conditional model's kernels: the 1-input network is carried as the 2-input
# network whose first-layer weights of the second input are zero and frozen (cude_set_param_mask), so that
# network([ΔG; e^β]) == network([ΔG]) exactly; parameter vectors in and out have the 1-input SimpleChains layout.
struct CPeptideUDEModel <: CPeptideModel
    glucose::Vector{Float64}; timepoints::Vector{Float64}; age::Float64; chain::Chain
    cpeptide::Vector{Float64}; t2dm::Bool
end
CPeptideUDEModel(glucose_data::AbstractVector{<:Real}, glucose_timepoints::AbstractVector{<:Real}, age::Real,
                 network::Chain, cpeptide_data::AbstractVector{<:Real}, t2dm::Bool) =
    network.input_dims == 1 ? CPeptideUDEModel(Vector{Float64}(glucose_data), Vector{Float64}(glucose_timepoints),
                                               Float64(age), network, Vector{Float64}(cpeptide_data), t2dm) :
    error("the non-conditional model takes a network with input_dims = 1")
embed_single_input(W::Integer, p::AbstractVector{<:Real}) = vcat(p[1:W], zeros(W), p[W+1:end])
extract_single_input(W::Integer, q::AbstractVector{<:Real}) = vcat(q[1:W], q[2W+1:end])
carrier(c::Chain) = Chain(2, c.width, c.depth, c.widths, c.activation, c.output_activation)
function carrier_mask(c::Chain)            # 1 = live entry of the carrier's parameter vector
    n1 = n_params(Chain(1, c.width, c.depth))
    embed_single_input(c.width, isempty(c.widths) ? ones(n1) : param_mask(c))
end

struct Solution{U}               # the fields of Optimization.jl's solution that the reference's scripts read
    u::U; objective::Float64
end

# ----------------------------------------------------------------------------------------------- population cache
# Discretisation of every method below whose caller passes no `n_steps` (cude/api.py `default_steps`, the same rule):
# ADAPTIVE -- the reference's `solve(prob, Tsit5())` at OrdinaryDiffEq's default tolerances (src/parameter-estimation.jl:59),
# the mode that reproduces the reference's stored numbers.  The fixed-step mode (smooth loss, exact discrete adjoint,
# time-split kernels for small populations; 1e-4 ... 1e-3 away from the reference's values) is asked for explicitly:
# `n_steps = fixed_steps(timepoints)` per call or `set_default_steps!(:fixed)` for the module.
const DEFAULT_MODE = Ref{Union{Symbol,Int}}(ADAPTIVE)
function fixed_steps(timepoints; per_interval = 8)
    d = diff(Vector{Float64}(timepoints))
    length(d) >= 1 && all(x -> isapprox(x, d[1]; rtol = 1e-12, atol = 0.0), d) ? per_interval * length(d) : DEFAULT_STEPS
end
function default_steps(timepoints = nothing; per_interval = 8)
    DEFAULT_MODE[] === :fixed || return Int(DEFAULT_MODE[])
    timepoints === nothing ? DEFAULT_STEPS : fixed_steps(timepoints; per_interval = per_interval)
end
function set_default_steps!(mode)
    (mode === :fixed || (mode isa Integer && mode >= 0)) || error("set_default_steps!: ADAPTIVE (0), :fixed or a positive step count")
    prev = DEFAULT_MODE[]
    DEFAULT_MODE[] = mode === :fixed ? :fixed : Int(mode)
    prev
end
const POPULATIONS = Dict{UInt64,Ctx}()

function population(models::AbstractVector{CPeptideConditionalUDEModel}, timepoints, cpeptide_data; n_steps = nothing)
    S = n_steps === nothing ? default_steps(timepoints) : n_steps
    data = cpeptide_data isa AbstractVector ? reshape(Vector{Float64}(cpeptide_data), 1, :) : Matrix{Float64}(cpeptide_data)
    key = hash((S, Vector{Float64}(timepoints), data, [(m.glucose, m.age, m.t2dm, m.chain) for m in models]))
    get!(POPULATIONS, key) do
        net = models[1].chain
        all(m -> m.timepoints == timepoints, models) || error("timepoints must equal the models' own timepoints")
        c = configure!(Ctx(Config(MODEL_CPEP, 2, net.input_dims, net.width, net.depth, S, 0, 0, 0.0)), net)
        G = Matrix{Float64}(undef, length(models), length(timepoints))
        for (i, m) in enumerate(models); G[i, :] .= m.glucose; end
        set_population!(c, Vector{Float64}(timepoints), G, data, [m.age for m in models], UInt8[m.t2dm for m in models])
        (isempty(net.widths) || is_general(net)) || set_param_mask!(c, param_mask(net))
        c
    end
end
function population(models::AbstractVector{CPeptideUDEModel}, timepoints, cpeptide_data; n_steps = nothing)
    S = n_steps === nothing ? default_steps(timepoints) : n_steps
    data = cpeptide_data isa AbstractVector ? reshape(Vector{Float64}(cpeptide_data), 1, :) : Matrix{Float64}(cpeptide_data)
    key = hash((:ude, S, Vector{Float64}(timepoints), data, [(m.glucose, m.age, m.t2dm, m.chain) for m in models]))
    get!(POPULATIONS, key) do
        net = models[1].chain
        all(m -> m.timepoints == timepoints, models) || error("timepoints must equal the models' own timepoints")
        c = Ctx(Config(MODEL_CPEP, 2, 2, net.width, net.depth, S, 0, 0, 0.0))
        G = Matrix{Float64}(undef, length(models), length(timepoints))
        for (i, m) in enumerate(models); G[i, :] .= m.glucose; end
        set_population!(c, Vector{Float64}(timepoints), G, data, [m.age for m in models], UInt8[m.t2dm for m in models])
        set_param_mask!(c, carrier_mask(net))
        c
    end
end
clear_populations!() = empty!(POPULATIONS)

# ----------------------------------------------------------------------------------------------- reference API: loss
# loss(θ, (models, timepoints, cpeptide_data)): mean over subjects of the SSE; Inf when a solve fails (:126-140)
function loss(θ, (models, timepoints, cpeptide_data)::Tuple{AbstractVector{CPeptideConditionalUDEModel},AbstractVector{T},AbstractMatrix{T}}) where T<:Real
    c = population(models, timepoints, cpeptide_data)
    set_params!(c, θ.neural, θ.conditional[1:length(models)])
    forward(c)[1]
end
# loss(θ, (model, timepoints, cpeptide_data)): one subject's SSE (:56-68)
function loss(θ, (model, timepoints, cpeptide_data)::Tuple{CPeptideConditionalUDEModel,AbstractVector{T},AbstractVector{T}}) where T<:Real
    c = population([model], timepoints, cpeptide_data)
    set_params!(c, θ.neural, [θ.conditional[1]])
    forward(c)[1]
end
# loss(θ, (model::CPeptideUDEModel, timepoints, cpeptide_data)): θ = the network's parameters (:56-68)
function loss(θ, (model, timepoints, cpeptide_data)::Tuple{CPeptideUDEModel,AbstractVector{T},AbstractVector{T}}) where T<:Real
    c = population([model], timepoints, cpeptide_data)
    set_params!(c, embed_single_input(model.chain.width, θ), [0.0])
    forward(c)[1]
end
# loss(β, (model, timepoints, cpeptide_data, neural_network_parameters)): frozen network (:93-99)
loss(θ, (model, timepoints, cpeptide_data, nn)::Tuple{CPeptideConditionalUDEModel,AbstractVector{T},AbstractVector{T},AbstractVector{T}}) where T<:Real =
    loss((neural = nn, conditional = θ), (model, timepoints, cpeptide_data))
function loss_sigma(θ, (model, timepoints, cpeptide_data)::Tuple{CPeptideConditionalUDEModel,AbstractVector{T},AbstractVector{T}}) where T<:Real
    n = length(timepoints)
    (n / 2) * log(θ.sigma^2) + loss(θ, (model, timepoints, cpeptide_data)) / (2 * θ.sigma^2)
end
function loss_sigma(θ, (model, timepoints, cpeptide_data, nn)::Tuple{CPeptideConditionalUDEModel,AbstractVector{T},AbstractVector{T},AbstractVector{T}}) where T<:Real
    n = length(timepoints)
    (n / 2) * log(θ.sigma^2) + loss(θ.ode, (model, timepoints, cpeptide_data, nn)) / (2 * θ.sigma^2)
end

# gradient hook for Optimization.jl in place of AutoForwardDiff():
#   OptimizationFunction(loss; grad = (G, θ, p) -> loss_and_gradient!(G, θ, p))
function loss_and_gradient!(G, θ, (models, timepoints, cpeptide_data))
    c = population(models, timepoints, cpeptide_data)
    set_params!(c, θ.neural, θ.conditional[1:length(models)])
    l, gnn, gcond = loss_grad(c)
    G.neural .= gnn; G.conditional .= reshape(gcond, size(G.conditional))
    l
end

# ----------------------------------------------------------------------------------------------- reference API: train
# QuasiMonteCarlo.LatinHypercubeSample restated: one stratified draw per interval and dimension, shuffled
function initial_parameters(n_models::Integer, lb::Real, ub::Real, n_initials::Integer, rng::AbstractRNG)
    u = Matrix{Float64}(undef, n_models, n_initials)
    for i in 1:n_models; u[i, :] .= (randperm(rng, n_initials) .- rand(rng, n_initials)) ./ n_initials; end
    lb .+ (ub - lb) .* u
end
initial_parameters(c::Chain, n_initials::Integer; rng::AbstractRNG) = [init_params(c; rng = rng) for _ in 1:n_initials]

# train(models, timepoints, cpeptide_data, rng; ...) (:340-386): screening of `initial_guesses` candidates in one
# multi-start launch, the best `selected_initials` trained side by side (Adam then L-BFGS) in one call
function train(models::AbstractVector{CPeptideConditionalUDEModel}, timepoints::AbstractVector{T},
               cpeptide_data::AbstractVecOrMat{T}, rng::AbstractRNG;
               initial_guesses::Int = 25_000, selected_initials::Int = 25, lhs_lower_bound = -2.0, lhs_upper_bound = 0.0,
               n_conditional_parameters::Int = 1, number_of_iterations_adam::Int = 1000,
               number_of_iterations_lbfgs::Int = 1000, learning_rate_adam::Real = 1e-2) where T<:Real
    c = population(models, timepoints, cpeptide_data)
    nn0 = reduce(hcat, initial_parameters(models[1].chain, initial_guesses; rng = rng))           # P × K
    cond0 = initial_parameters(length(models), lhs_lower_bound, lhs_upper_bound, initial_guesses, rng)   # N × K
    losses_initial = multistart_forward(c, nn0, cond0)
    println("Initial parameters evaluated. Optimizing for the best $(selected_initials) initial parameters.")
    best = partialsortperm(losses_initial, 1:selected_initials)
    nn, cond, obj, _ = train_restarts(c, nn0[:, best], cond0[:, best]; adam_iters = number_of_iterations_adam,
                                      η = learning_rate_adam, lbfgs_iters = number_of_iterations_lbfgs)
    optsols = Solution[]
    for k in eachindex(obj)
        isfinite(obj[k]) || (println("Optimization failed... Skipping"); continue)
        push!(optsols, Solution((neural = nn[:, k], conditional = repeat(cond[:, k], 1, n_conditional_parameters)), obj[k]))
    end
    optsols
end

# train(model::CPeptideUDEModel, timepoints, cpeptide_data, rng; ...) (:205-247): the conventional UDE on one (mean)
# subject -- the same two device calls on a population of one
function train(model::CPeptideUDEModel, timepoints::AbstractVector{T}, cpeptide_data::AbstractVector{T}, rng::AbstractRNG;
               initial_guesses::Int = 10_000, selected_initials::Int = 10, number_of_iterations_adam::Int = 1000,
               number_of_iterations_lbfgs::Int = 1000, learning_rate_adam::Real = 1e-2) where T<:Real
    c = population([model], timepoints, cpeptide_data)
    W = model.chain.width
    nn0 = reduce(hcat, [embed_single_input(W, p) for p in initial_parameters(model.chain, initial_guesses; rng = rng)])
    cond0 = zeros(1, initial_guesses)
    losses_initial = multistart_forward(c, nn0, cond0)
    best = partialsortperm(losses_initial, 1:selected_initials)
    nn, _, obj, _ = train_restarts(c, nn0[:, best], cond0[:, best]; adam_iters = number_of_iterations_adam,
                                   η = learning_rate_adam, lbfgs_iters = number_of_iterations_lbfgs)
    optsols = Solution[]
    for k in eachindex(obj)
        isfinite(obj[k]) || (println("Optimization failed... Skipping"); continue)
        push!(optsols, Solution(extract_single_input(W, nn[:, k]), obj[k]))
    end
    optsols
end

# train(models, timepoints, cpeptide_data, neural_network_parameters; ...) (:272-288): every subject's β at once
function train(models::AbstractVector{CPeptideConditionalUDEModel}, timepoints::AbstractVector{T},
               cpeptide_data::AbstractMatrix{T}, neural_network_parameters::AbstractVector{T};
               initial_beta = -2.0, lbfgs_lower_bound = -4.0, lbfgs_upper_bound = 1.0, lbfgs_iterations::Int = 1000) where T<:Real
    c = population(models, timepoints, cpeptide_data)
    set_params!(c, neural_network_parameters, nothing)
    lo = isfinite(lbfgs_lower_bound) ? lbfgs_lower_bound : initial_beta - 6.0
    hi = isfinite(lbfgs_upper_bound) ? lbfgs_upper_bound : initial_beta + 6.0
    β, _, sse = fit_conditional(c, lo, hi)
    [Solution([β[i]], sse[i]) for i in eachindex(β)]
end

# train_with_sigma (:290-307): for fixed β the optimum is σ² = SSE / n, so the 2-D problem separates
function train_with_sigma(models::AbstractVector{CPeptideConditionalUDEModel}, timepoints::AbstractVector{T},
                          cpeptide_data::AbstractMatrix{T}, neural_network_parameters::AbstractVector{T};
                          initial_beta = -2.0, lbfgs_lower_bound = -4.0, lbfgs_upper_bound = 1.0,
                          lbfgs_iterations::Int = 1000) where T<:Real
    sols = train(models, timepoints, cpeptide_data, neural_network_parameters; initial_beta = initial_beta,
                 lbfgs_lower_bound = lbfgs_lower_bound, lbfgs_upper_bound = lbfgs_upper_bound)
    n = length(timepoints)
    map(sols) do s
        σ = sqrt(max(s.objective, 1e-300) / n)
        Solution((ode = s.u, sigma = σ), (n / 2) * log(σ^2) + s.objective / (2 * σ^2))
    end
end

# evaluate_model (:406-433): objectives of every candidate network on the validation subjects (n_subjects × n_networks)
function evaluate_model(models::AbstractVector{CPeptideConditionalUDEModel}, timepoints::AbstractVector{T},
                        cpeptide_data::AbstractMatrix{T}, neural_network_parameters,
                        betas_train::AbstractVector{<:AbstractVector{T}}) where T<:Real
    cols = map(zip(betas_train, neural_network_parameters)) do (betas, p_nn)
        try
            [s.objective for s in train(models, timepoints, cpeptide_data, Vector{Float64}(p_nn); initial_beta = mean(betas),
                                        lbfgs_lower_bound = -Inf, lbfgs_upper_bound = Inf)]
        catch
            fill(Inf, length(models))
        end
    end
    reduce(hcat, cols)
end

# likelihood_profile (src/likelihood-profiles.jl:4-17): all `steps` values of β in one launch
function likelihood_profile(β, neural_network_parameters, model::CPeptideConditionalUDEModel, timepoints, cpeptide_data,
                            lower_bound, upper_bound, sigma; steps = 1000)
    c = population([model], timepoints, cpeptide_data)
    set_params!(c, neural_network_parameters, [β[1]])
    nll_minimum = forward(c; want_sse = true)[2][1] / (2 * sigma^2)
    parameter_values = range(lower_bound, stop = upper_bound, length = steps)
    nll_values = vec(profile_conditional(c, collect(Float64, parameter_values))) ./ (2 * sigma^2)
    nll_values, nll_minimum, parameter_values
end

# ----------------------------------------------------------------------------------------------- suppression model
struct SuppressionProblem          # stands in for ODEProblem(ude_lsup!, u0, tspan) with the network closed over
    network::Chain
end
const SUPP_POPULATIONS = Dict{UInt64,Ctx}()
function supp_population(prob::SuppressionProblem, data::AbstractArray{<:Real,3}, timepoints, λ; n_steps = nothing)
    n_steps = n_steps === nothing ? default_steps() : n_steps       # (fixed mode: DEFAULT_STEPS = 30)
    d = Array{Float64,3}(data)
    key = hash((n_steps, prob.network, Vector{Float64}(timepoints), d, Float64(λ)))
    get!(SUPP_POPULATIONS, key) do
        c = configure!(Ctx(Config(MODEL_SUPP, 3, 4, prob.network.width, prob.network.depth, n_steps, 0, 0, Float64(λ))),
                       prob.network)
        set_population_supp!(c, Vector{Float64}(timepoints), d)
    end
end

# suppression_loss(p, (prob, individual_data, timepoints, λ)) with p.theta[N], p.neural[P] (:117-130)
function suppression_loss(p, (prob, individual_data, timepoints, λ))
    c = supp_population(prob, individual_data, timepoints, λ)
    set_params!(c, p.neural, p.theta)
    forward(c)[1]
end
function suppression_loss_and_gradient!(G, p, (prob, individual_data, timepoints, λ))
    c = supp_population(prob, individual_data, timepoints, λ)
    set_params!(c, p.neural, p.theta)
    l, gnn, gθ = loss_grad(c)
    G.neural .= gnn; G.theta .= gθ
    l
end
# simul(p, prob, individual_data, timepoints) -> 3 × T × N (:107-115)
function simul(p, prob::SuppressionProblem, individual_data, timepoints)
    c = supp_population(prob, individual_data, timepoints, 0.0)
    set_params!(c, p.neural, p.theta)
    forward(c; want_traj = true)[3]
end
# fit_suppression_model(p_init, prob, data, timepoints, λ; select_best_n) (:132-177): Adam() [η = 1e-3] × 2000, L-BFGS × 2000
function fit_suppression_model(p_init, prob::SuppressionProblem, data, timepoints, λ; select_best_n = 1,
                               adam_iters = 2000, lbfgs_iters = 2000)
    c = supp_population(prob, data, timepoints, λ)
    nn0 = reduce(hcat, [Vector{Float64}(p.neural) for p in p_init]); θ0 = reduce(hcat, [Vector{Float64}(p.theta) for p in p_init])
    initial_losses = multistart_forward(c, nn0, θ0)
    best = select_best_n > 1 ? partialsortperm(initial_losses, 1:select_best_n) : [argmin(initial_losses)]
    println("Selected best $(length(best)) initials")
    nn, θ, obj, trace = train_restarts(c, nn0[:, best], θ0[:, best]; adam_iters = adam_iters, η = 1e-3, lbfgs_iters = lbfgs_iters)
    optsols = Solution[]; loss_traces = Vector{Float64}[]
    for k in eachindex(obj)
        isfinite(obj[k]) ? push!(optsols, Solution((theta = θ[:, k], neural = nn[:, k]), obj[k])) : println("Optimization failed")
        push!(loss_traces, filter(!isnan, trace[:, k]))
    end
    optsols, loss_traces
end

# validate_suppression_model_sigma(p_init, prob, data, timepoints, network_params) (suppression_model.jl:224-275): θ and one
# noise level per state for a test subject (data: 3 × T), minimising Σ_s (n/2) log σ_s² + SSE_s / (2σ_s²) on the unscaled
# residuals with the network frozen.  For given θ the optimal σ_s² is SSE_s / n: a 1-D search in θ (fine grid, then a
# golden-section refinement around the three deepest local minima), one forward launch per probe.
function validate_suppression_model_sigma(p_init, prob::SuppressionProblem, data::AbstractMatrix{<:Real}, timepoints,
                                          network_params; lower = -8.0, upper = 5.0, n_grid = 261, iters = 50)
    d3 = reshape(Array{Float64}(data), 3, :, 1)
    c = supp_population(prob, d3, timepoints, 0.0)
    n = size(d3, 2)
    lo = isempty(p_init) ? lower : min(lower, minimum(p_init)); hi = isempty(p_init) ? upper : max(upper, maximum(p_init))
    function nll(θ)
        set_params!(c, network_params, [θ])
        traj = forward(c; want_traj = true)[3]                       # 3 × T × 1
        sse = [sum(abs2, traj[s, :, 1] .- d3[s, :, 1]) for s in 1:3]
        v = sum(0.5 * n * (log(x / n) + 1) for x in sse)
        (isfinite(v) ? v : Inf), sse
    end
    grid = collect(range(lo, hi; length = n_grid))
    vals = [nll(g)[1] for g in grid]
    mins = [k for k in 1:n_grid if isfinite(vals[k]) && (k == 1 || vals[k] <= vals[k-1]) && (k == n_grid || vals[k] <= vals[k+1])]
    sort!(mins; by = k -> vals[k])
    r = (sqrt(5) - 1) / 2
    best_θ, best_v = grid[argmin(vals)], minimum(vals)
    for k in mins[1:min(3, length(mins))]
        a, b = grid[max(k - 1, 1)], grid[min(k + 1, n_grid)]
        x1, x2 = b - r * (b - a), a + r * (b - a)
        f1, f2 = nll(x1)[1], nll(x2)[1]
        for _ in 1:iters
            if f1 < f2
                b, x2, f2 = x2, x1, f1
                x1 = b - r * (b - a); f1 = nll(x1)[1]
            else
                a, x1, f1 = x1, x2, f2
                x2 = a + r * (b - a); f2 = nll(x2)[1]
            end
        end
        θc, vc = f1 < f2 ? (x1, f1) : (x2, f2)
        vals[k] < vc && ((θc, vc) = (grid[k], vals[k]))
        vc < best_v && ((best_θ, best_v) = (θc, vc))
    end
    v, sse = nll(best_θ)
    (ode = best_θ, sigma = sqrt.(sse ./ n)), v
end

# ----------------------------------------------------------------------------------------------- SAEM
# individuals: NamedTuples with glucose, timepoints, cpeptide, age, condition ("T2DM" or not), as c-peptide/06-saem.jl builds
function individuals_population(individuals, network::Chain; n_steps = nothing)
    models = [CPeptideConditionalUDEModel(i.glucose, i.timepoints, i.age, network, i.cpeptide, i.condition == "T2DM")
              for i in individuals]
    data = reduce(vcat, [reshape(Vector{Float64}(i.cpeptide), 1, :) for i in individuals])
    population(models, Vector{Float64}(individuals[1].timepoints), data; n_steps = n_steps)
end

# simulate(p_neural, p_individual, individual, network; timepoints) (src/saem.jl:31-53): plasma c-peptide at `timepoints`
function simulate(p_neural, p_individual, individual, network::Chain; timepoints = individual.timepoints)
    c = individuals_population([individual], network)
    set_params!(c, p_neural, [p_individual[1]])
    vec(simulate_dense(c, Vector{Float64}(timepoints))[1, :, 1])
end

# individual_log_likelihood(p_individual, p_neural, individual, network, σ) (:55-66); -Inf on a failed solve
function individual_log_likelihood(p_individual, p_neural, individual, network::Chain, σ)
    c = individuals_population([individual], network)
    set_params!(c, p_neural, [p_individual[1]])
    sse = forward(c; want_sse = true)[2][1]
    isfinite(sse) ? -(length(individual.timepoints) / 2) * log(σ^2) - sse / (2 * σ^2) : -Inf
end

# map_objective / compute_individual_maps (src/saem.jl:68-84): -(ll + log N(p; prior, Ω)); the MAPs of all individuals are one
# penalised per-subject search on the device: argmin SSE(x) + (σ/Ω)² (x - prior)²  (cude_fit_conditional)
function map_objective(p_individual, p_neural, individual, σ, Ω, network::Chain; prior_individual = 0.0)
    ll = individual_log_likelihood(p_individual, p_neural, individual, network, σ)
    prior = -0.5 * ((p_individual - prior_individual) / Ω)^2 - log(Ω) - 0.5 * log(2π)
    -(ll + prior)
end
function compute_individual_maps(p_individuals, p_neural, individuals, σ, Ω, network::Chain; prior_individual = 0.0,
                                 lower = -6.0, upper = 4.0)
    c = individuals_population(individuals, network)
    set_params!(c, p_neural, Vector{Float64}(p_individuals))
    x, _, _ = fit_conditional(c, min(lower, minimum(p_individuals)), max(upper, maximum(p_individuals)); n_grid = 81,
                              penalty_weight = (σ / Ω)^2, penalty_center = prior_individual)
    x
end

# SAEM(individuals, initial_neural_params, network; ...) (:134-237): the E-step of all individuals is one call
# (cude_mh_estep), the M-step's 5 Adam(1e-2) iterations on (neural, σ) use the device gradient
function SAEM(individuals, initial_neural_params, network::Chain;
              σ = 1.0, prior_η = 0.0, prior_Ω = 1.0, iterations = 500, n_burnin_iterations = 100, proposal_std = 0.1,
              proposal_std_bounds = (1e-3, 1.0), α = 0.7, n_mcmc_steps = 1, initial_mcmc_steps = n_mcmc_steps,
              target_acceptance_rate = 0.25, initial_temperature = 10.0, temperature_decay = 0.05, Ω_learning_rate = 0.04)
    println("Initializing the SAEM algorithm...")
    c = individuals_population(individuals, network)
    N, T = c.N, c.T
    p_individuals = fill(Float64(prior_η), N); p_neural = Vector{Float64}(initial_neural_params); Ω = Float64(prior_Ω)
    total_nll_values = Float64[]; acceptance_rates = Float64[]
    for iteration in 1:iterations
        gamma = iteration <= n_burnin_iterations ? 1.0 : 1.0 / (iteration - n_burnin_iterations)^α
        steps = iteration <= n_burnin_iterations ? initial_mcmc_steps : n_mcmc_steps
        temperature = max(1, initial_temperature * exp(-temperature_decay * iteration))
        set_params!(c, p_neural, p_individuals)
        accepted = mh_estep!(c, randn(N, steps), rand(N, steps), σ, prior_η, Ω, proposal_std; temperature = temperature, γ = gamma)
        p_individuals = get_params(c)[2]
        sse = forward(c; want_sse = true)[2]
        loglikelihood = sum(s -> isfinite(s) ? -(T / 2) * log(σ^2) - s / (2 * σ^2) : -Inf, sse)
        # M-step (update_population_parameters :118-131): Optimisers.Adam(1e-2), maxiters = 5 on total_nll(neural, σ)
        x = vcat(p_neural, σ); m = zero(x); v = zero(x)
        for t in 1:5
            set_params!(c, x[1:end-1], p_individuals)
            mean_sse, g_nn, _ = loss_grad(c)
            s = x[end]
            g = vcat(g_nn .* (N / (2 * s^2)), N * T / s - mean_sse * N / s^3)
            m .= 0.9 .* m .+ 0.1 .* g; v .= 0.999 .* v .+ 0.001 .* g .^ 2
            x .-= 1e-2 .* (m ./ (1 - 0.9^t)) ./ (sqrt.(v ./ (1 - 0.999^t)) .+ 1e-8)
        end
        σ = x[end]
        p_neural = (1 - gamma) .* p_neural .+ gamma .* x[1:end-1]
        Ω = (1 - Ω_learning_rate) * Ω + Ω_learning_rate * var(p_individuals)
        prior_η = (1 - Ω_learning_rate) * prior_η + Ω_learning_rate * mean(p_individuals)
        acceptance_rate = sum(accepted) / (N * steps)
        push!(total_nll_values, -loglikelihood); push!(acceptance_rates, acceptance_rate)
        if iteration > n_burnin_iterations
            proposal_std = clamp(exp(log(proposal_std) + gamma * (acceptance_rate - target_acceptance_rate)),
                                 proposal_std_bounds[1], proposal_std_bounds[2])
        end
    end
    (p_neural = p_neural, p_individuals = p_individuals, Ω = Ω, σ = σ, η = prior_η,
     total_nll_values = total_nll_values, acceptance_rates = acceptance_rates)
end

end # module
