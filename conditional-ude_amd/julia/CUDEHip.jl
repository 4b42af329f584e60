# Julia shim over the C ABI of libcude_hip.so (see INTEGRATION.md).  Written, not executed in this pipeline
# (Julia is not installed in the build image); mirrors conditional-ude_amd/cude/engine.py 1:1.
module CUDEHip
const LIB = "libcude_hip.so"
struct Config
    model::Int32; n_state::Int32; nn_in::Int32; nn_width::Int32; nn_depth::Int32
    n_steps::Int32; device::Int32; cond_space::Int32; lambda::Float64
end
check(st) = st < 0 ? error(unsafe_string(ccall((:cude_last_error, LIB), Cstring, ()))) : st

mutable struct Ctx; h::Ptr{Cvoid}; P::Int; N::Int; end
function Ctx(cfg::Config)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:cude_create, LIB), Int32, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, h))
    P = ccall((:cude_n_params, LIB), Int32, (Int32, Int32, Int32), cfg.nn_in, cfg.nn_width, cfg.nn_depth)
    c = Ctx(h[], P, 0); finalizer(x -> ccall((:cude_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), c); c
end

# models' data as Julia column-major N×T matrices: ld_subject = 1, ld_time = N (no host copy)
function set_population!(c::Ctx, timepoints::Vector{Float64}, glucose::Matrix{Float64}, cpeptide::Matrix{Float64},
                         ages::Vector{Float64}, t2dm::Vector{UInt8})
    N, T = size(glucose)
    GC.@preserve timepoints glucose cpeptide ages t2dm check(ccall((:cude_set_population_cpep, LIB), Int32,
        (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{UInt8}),
        c.h, N, T, timepoints, glucose, cpeptide, 1, N, ages, t2dm))
    c.N = N
end

# drop-in for `loss(θ, (models, timepoints, cpeptide_data))`  (src/parameter-estimation.jl:126-140)
function loss(c::Ctx, θ)
    nn = Vector{Float64}(θ.neural); cond = vec(Matrix{Float64}(θ.conditional)); l = Ref{Float64}()
    GC.@preserve nn cond begin
        check(ccall((:cude_set_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, nn, cond))
        check(ccall((:cude_forward, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}), c.h, l, C_NULL, C_NULL))
    end
    l[]
end

# gradient hook:  OptimizationFunction((θ,p)->loss(ctx,θ); grad = (G,θ,p)->grad!(G,ctx,θ))
function grad!(G, c::Ctx, θ)
    nn = Vector{Float64}(θ.neural); cond = vec(Matrix{Float64}(θ.conditional))
    gnn = Vector{Float64}(undef, c.P); gcond = Vector{Float64}(undef, c.N); l = Ref{Float64}()
    GC.@preserve nn cond gnn gcond begin
        check(ccall((:cude_set_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, nn, cond))
        check(ccall((:cude_loss_grad, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}), c.h, l, gnn, gcond))
    end
    G.neural .= gnn; G.conditional .= reshape(gcond, size(G.conditional)); G
end

# fast path replacing Optimization.solve(prob, Optimisers.Adam(η), maxiters=K): parameters stay on the GPU
function adam!(c::Ctx, θ0, η, iters; callback = (l)->false)
    loss(c, θ0); check(ccall((:cude_adam_init, LIB), Int32, (Ptr{Cvoid}, Float64, Float64, Float64, Float64), c.h, η, 0.9, 0.999, 1e-8))
    l = Ref{Float64}()
    for _ in 1:iters
        check(ccall((:cude_adam_step, LIB), Int32, (Ptr{Cvoid}, Ref{Float64}), c.h, l)); callback(l[]) && break
    end
    nn = Vector{Float64}(undef, c.P); cond = Vector{Float64}(undef, c.N)
    check(ccall((:cude_get_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), c.h, nn, cond)); (nn, cond, l[])
end
end

# --- additional entry points (same module; appended for readability) -----------------------------------------
module CUDEHipExtras
import ..CUDEHip: LIB, Ctx, check

# screening loop of `train` (src/parameter-estimation.jl:362-366): K candidate parameter sets in one launch.
# nn_sets is P×K, cond_sets is N×K (Julia column-major = the row-major [K][P] / [K][N] the ABI expects).
function multistart_forward(c::Ctx, nn_sets::Matrix{Float64}, cond_sets::Matrix{Float64})
    K = size(nn_sets, 2); losses = Vector{Float64}(undef, K)
    GC.@preserve nn_sets cond_sets losses check(ccall((:cude_multistart_forward, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), c.h, K, nn_sets, cond_sets, losses))
    losses
end

# restart loop of `train` (src/parameter-estimation.jl:372-383) with all restarts evaluated per optimiser iteration:
# losses (K), ∂/∂neural (P×K) and ∂/∂conditional (N×K) of the K current points in one launch.
function multistart_loss_grad(c::Ctx, nn_sets::Matrix{Float64}, cond_sets::Matrix{Float64})
    K = size(nn_sets, 2); N = size(cond_sets, 1)
    losses = Vector{Float64}(undef, K); g_nn = similar(nn_sets); g_cond = similar(cond_sets)
    GC.@preserve nn_sets cond_sets losses g_nn g_cond check(ccall((:cude_multistart_loss_grad, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, K, nn_sets, cond_sets, losses, g_nn, g_cond))
    losses, g_nn, g_cond
end

# second half of `train` (src/parameter-estimation.jl:368-383): the selected initial guesses (columns of nn_sets P×K and
# cond_sets N×K) trained side by side with Adam(η) × adam_iters then LBFGS(BackTracking) × lbfgs_iters;
# returns (neural P×K, conditional N×K, objectives K)
function train_restarts(c::Ctx, nn_sets::Matrix{Float64}, cond_sets::Matrix{Float64}; adam_iters = 1000, η = 1e-2,
                        lbfgs_iters = 1000)
    K = size(nn_sets, 2); nn = similar(nn_sets); cond = similar(cond_sets); obj = Vector{Float64}(undef, K)
    GC.@preserve nn_sets cond_sets nn cond obj check(ccall((:cude_train_restarts, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Float64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}), c.h, K, nn_sets, cond_sets, adam_iters, η, lbfgs_iters, nn, cond, obj, C_NULL))
    nn, cond, obj
end

# `train(models, timepoints, data, neural_network_parameters)` (src/parameter-estimation.jl:272-288) for all models at
# once: per-subject minimisers of SSE_i(β) + w (β - μ)^2 over [lower, upper]; returns (β, objective, SSE)
function fit_conditional(c::Ctx, lower, upper; n_grid = 41, n_iters = 48, penalty_weight = 0.0, penalty_center = 0.0)
    β = Vector{Float64}(undef, c.N); obj = similar(β); sse = similar(β)
    GC.@preserve β obj sse check(ccall((:cude_fit_conditional, LIB), Int32,
        (Ptr{Cvoid}, Float64, Float64, Int32, Int32, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, lower, upper, n_grid, n_iters, penalty_weight, penalty_center, β, obj, sse))
    β, obj, sse
end

# likelihood profiles (src/likelihood-profiles.jl:4-17) of all models at once: SSE_i(values[k]) as an N×K matrix
function profile_conditional(c::Ctx, values::Vector{Float64})
    sse = Matrix{Float64}(undef, c.N, length(values))
    GC.@preserve values sse check(ccall((:cude_profile_conditional, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), c.h, length(values), values, sse))
    sse
end

# `maxiters` Adam iterations in one call (hipGraph replay); returns the loss trace
function adam_run!(c::Ctx, iters::Integer)
    losses = Vector{Float64}(undef, iters)
    check(ccall((:cude_adam_run, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}), c.h, iters, losses)); losses
end

# E-step of SAEM (src/saem.jl:177-186): n_mc Metropolis steps for every subject; draws are N×n_mc matrices
function mh_estep!(c::Ctx, normals::Matrix{Float64}, uniforms::Matrix{Float64}, σ, prior_η, Ω, proposal_std;
                   temperature = 1.0, γ = 1.0)
    N, n_mc = size(normals); accepted = zeros(Int64, N)
    GC.@preserve normals uniforms accepted check(ccall((:cude_mh_estep, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Int64}),
        c.h, n_mc, normals, uniforms, σ, prior_η, Ω, proposal_std, temperature, γ, accepted))
    accepted
end

# posterior sampling after SAEM (c-peptide/06-saem.jl:107-112) for all individuals at once: every chain state is kept;
# returns (accepted, samples) with samples N×n_mc
function mh_chain!(c::Ctx, normals::Matrix{Float64}, uniforms::Matrix{Float64}, σ, prior_η, Ω, proposal_std;
                   temperature = 1.0, γ = 1.0)
    N, n_mc = size(normals); accepted = zeros(Int64, N); samples = similar(normals)
    GC.@preserve normals uniforms accepted samples check(ccall((:cude_mh_chain, LIB), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Int64}, Ptr{Float64}),
        c.h, n_mc, normals, uniforms, σ, prior_η, Ω, proposal_std, temperature, γ, accepted, samples))
    accepted, samples
end
end
