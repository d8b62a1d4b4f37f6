# NLLSsolverAMD.jl -- the reference-side binding of libnlls_amd.so (include/nlls_amd.h).
#
# NOT EXECUTED IN THIS REPOSITORY: no Julia toolchain exists in the build image (SURVEY.md F6).  It is the
# `ccall` shim a maintainer of NLLSsolver.jl would add.  What it assumes about the library is pinned without Julia:
# tests/abi/abi_replay.c (plain C, built and run by tests/test_abi_replay.py) static_asserts the two struct layouts
# mirrored below (CostGroup, NllsInfo), resolves every symbol named in a ccall here, and replays on the GPU the exact
# call sequence and argument types of the device-resident Levenberg-Marquardt loop at the bottom of this file.
# Nothing in the reference tree is modified: the shim adds a linear-system type and method overloads of the generic
# functions the optimizer already calls (src/iterators.jl:139-172, src/optimize.jl:109-180,207-214).
module NLLSsolverAMD

import NLLSsolver
using NLLSsolver: NLLSProblem, NLLSOptions, NLLSInternal, EuclideanVector, ZeroToInfScalar, ZeroToOneScalar,
                  ContaminatedGaussian, NoRobust, HuberKernel, GemanMcclureKernel, Scaled, SimpleError2
using StaticArrays

const lib = get(ENV, "NLLS_AMD_LIB", "libnlls_amd.so")

# ---- kinds (include/nlls_amd.h) ------------------------------------------------------------------------
const VAR_EUCLIDEAN, VAR_ZERO_TO_INF, VAR_ZERO_TO_ONE, VAR_CONTAMINATED_GAUSSIAN = Int32(1), Int32(2), Int32(3), Int32(4)
const VAR_DYNAMIC = Int32(6)                                       # DynamicVector{Float64}: run-time length (dynamic-size residual kinds 11, 12)
const ROBUST_NONE, ROBUST_HUBER, ROBUST_HUBER2O, ROBUST_GEMAN_MCCLURE, ROBUST_SCALED = Int32(0), Int32(1), Int32(2), Int32(3), Int32(0x10)

struct CostGroup              # nlls_cost_group
    res_kind::Int32
    robust_kind::Int32
    robust_params::NTuple{4, Float64}
    ncost::Int64
    varind::Ptr{Int64}
    data::Ptr{Float64}
end

struct NllsInfo               # nlls_info (112 bytes: offsets pinned by tests/abi/abi_replay.c)
    is_sparse::Int32; has_schur::Int32
    nvar::Int64; nblocks::Int64; ndof::Int64; nnz_data::Int64; nblocks_stored::Int64; ncost::Int64; var_storage::Int64
    nschur_blocks::Int64; nreduced_dof::Int64; owner_path::Int64; solve_mode::Int64; bandwidth::Int64; nborder_dof::Int64
end

check(ctx, rc) = rc == 0 ? nothing : error("nlls_amd error $rc: " * unsafe_string(ccall((:nlls_last_error, lib), Cstring, (Ptr{Cvoid},), ctx)))

# variable kind/dim + packed storage of a reference variable
varkind(::Number) = (VAR_EUCLIDEAN, Int32(1));                    pack!(out, v::Number) = push!(out, Float64(v))
varkind(::EuclideanVector{N}) where N = (VAR_EUCLIDEAN, Int32(N)); pack!(out, v::EuclideanVector) = append!(out, v)
varkind(::ZeroToInfScalar) = (VAR_ZERO_TO_INF, Int32(1));         pack!(out, v::ZeroToInfScalar) = push!(out, v.val)
varkind(::ZeroToOneScalar) = (VAR_ZERO_TO_ONE, Int32(1));         pack!(out, v::ZeroToOneScalar) = push!(out, v.val)
varkind(::ContaminatedGaussian) = (VAR_CONTAMINATED_GAUSSIAN, Int32(3))
pack!(out, v::ContaminatedGaussian) = append!(out, (v.invsigma1.val, v.invsigma2.val, v.w.val))
varkind(v::NLLSsolver.DynamicVector{Float64}) = (VAR_DYNAMIC, Int32(length(v))); pack!(out, v::NLLSsolver.DynamicVector{Float64}) = append!(out, v)
varkind(::Any) = nothing                                          # unregistered: decline

robustspec(::NoRobust) = (ROBUST_NONE, (0.0, 0.0, 0.0, 0.0))
robustspec(k::HuberKernel) = (NLLSsolver.dynamic(k.secondorder) ? ROBUST_HUBER2O : ROBUST_HUBER, (Float64(k.width), 0.0, 0.0, 0.0))
robustspec(k::GemanMcclureKernel) = (ROBUST_GEMAN_MCCLURE, (sqrt(Float64(k.width_squared)), 0.0, 0.0, 0.0))
function robustspec(k::Scaled)
    inner = robustspec(k.robust); inner === nothing && return nothing
    (inner[1] & ROBUST_SCALED) != 0 && return nothing
    return (inner[1] | ROBUST_SCALED, (inner[2][1], Float64(k.height), 0.0, 0.0))
end
robustspec(::Any) = nothing

# Users register the residual kind of their cost type (closed world, SURVEY.md F3), e.g. for the bundle
# adjustment of test/optimizeba.jl:   NLLSsolverAMD.reskind(::Type{<:SimpleError2{2,Float64,EuclideanVector{6,Float64},EuclideanVector{3,Float64}}}) = Int32(1)
reskind(::Type) = nothing
costdata(c::SimpleError2) = c.measurement

# ---- the device-resident linear system (replaces MultiVariateLSsparse/dense, src/linearsystem.jl:44-87) ----
mutable struct MultiVariateLSgpu
    ctx::Ptr{Cvoid}
    blockindices::Vector{UInt}
    boffsets::Vector{UInt}
    ndof::Int
    x::Vector{Float64}          # host mirror of the step: callbacks read it (src/callbacks.jl:47,105)
    b::Vector{Float64}
    kinds::Vector{Tuple{Int32, Int32}}
    packed::Vector{Float64}
    resident::Bool              # the device's NLLS_VARS_CURRENT holds problem.variables (the device-resident LM loop keeps it so)
    havebest::Bool              # NLLS_VARS_BEST has been written once (the reference's `length(problem.variables) == length(problem.varbest)`)
end

"makesymmvls replacement (src/linearsystem.jl:91-124); returns `nothing` to decline (caller keeps the CPU system)."
function makesymmvls_gpu(problem::NLLSProblem, unfixed, nblocks)
    kinds = [varkind(v) for v in problem.variables]
    any(isnothing, kinds) && return nothing
    blockindices = zeros(UInt, length(problem.variables)); nb = 0
    for (i, u) in enumerate(unfixed); if u; nb += 1; blockindices[i] = nb; end; end
    groups = CostGroup[]; keep = Any[]
    for costs in values(problem.costs)
        isempty(costs) && continue
        rk = reskind(eltype(costs)); rs = robustspec(NLLSsolver.robustkernel(costs[1]))
        (rk === nothing || rs === nothing) && return nothing
        nd = length(NLLSsolver.varindices(costs[1]))
        vi = Matrix{Int64}(undef, nd, length(costs)); da = reduce(hcat, [collect(Float64, costdata(c)) for c in costs])
        for (k, c) in enumerate(costs); vi[:, k] .= NLLSsolver.varindices(c); end      # 1-based, as stored
        push!(keep, vi, da)
        push!(groups, CostGroup(rk, rs[1], rs[2], length(costs), pointer(vi), pointer(da)))
    end
    ctxref = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:nlls_ctx_create, lib), Cint, (Ptr{Int32}, Int32, Ptr{Ptr{Cvoid}}), C_NULL, 0, ctxref) == 0 || return nothing
    ctx = ctxref[]
    vk = Int32[k[1] for k in kinds]; vd = Int32[k[2] for k in kinds]
    rc = GC.@preserve keep groups vk vd blockindices ccall((:nlls_upload_structure, lib), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{UInt64}, Int32, Ptr{CostGroup}, Int32),
        ctx, length(kinds), vk, vd, blockindices, length(groups), groups, 0)
    if rc != 0; ccall((:nlls_ctx_destroy, lib), Cint, (Ptr{Cvoid},), ctx); return nothing; end   # NLLS_ERR_UNSUPPORTED: decline
    info = Ref{NllsInfo}(); check(ctx, ccall((:nlls_get_info, lib), Cint, (Ptr{Cvoid}, Ptr{NllsInfo}), ctx, info))
    ndof = Int(info[].ndof)
    boff = zeros(Int64, nb); ccall((:nlls_get_bsm_index, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), ctx, C_NULL, C_NULL, C_NULL, boff)
    ls = MultiVariateLSgpu(ctx, blockindices, UInt.(boff), ndof, zeros(ndof), zeros(ndof), kinds, Float64[], false, false)
    finalizer(l -> ccall((:nlls_ctx_destroy, lib), Cint, (Ptr{Cvoid},), l.ctx), ls)
    return ls
end

function setvariables!(ls::MultiVariateLSgpu, vars::Vector, which::Integer)
    empty!(ls.packed); foreach(v -> pack!(ls.packed, v), vars)
    check(ls.ctx, ccall((:nlls_set_variables, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), ls.ctx, which, ls.packed))
end

# ---- generic functions the iterators call (SURVEY.md 8b) ------------------------------------------------
NLLSsolver.zero!(::MultiVariateLSgpu) = nothing                                # fused into the sweep
function NLLSsolver.costgradhess!(ls::MultiVariateLSgpu, vars::Vector, costs)   # src/optimize.jl:118,167-170
    ls.resident || setvariables!(ls, vars, 0); c = Ref(0.0)
    check(ls.ctx, ccall((:nlls_sweep_gradhess, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, c)); return c[]
end
function gpucost(ls::MultiVariateLSgpu, vars::Vector)                           # src/cost.jl:10-13, for host-side iterators (dogleg, ...)
    setvariables!(ls, vars, 1); c = Ref(0.0)
    check(ls.ctx, ccall((:nlls_sweep_cost, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), ls.ctx, 1, c)); return c[]
end
# The reference's OTHER iterators (Newton, dogleg, gradient descent: src/iterators.jl:15-27,47-115,187-208) evaluate their trial points with
# cost(problem.varnext, problem.costs) (src/iterators.jl:24,100,191,203), which does not see the linear system.  NLLSInternal is parametric in the
# linear-system type (src/structs.jl:81-104), so -- exactly as for Levenberg-Marquardt below -- iterate! methods on NLLSInternal{MultiVariateLSgpu}
# take their place and call `gpucost` (the device cost sweep, src/cost.jl:10-13) instead: see "the other iterators" further down.
struct GpuHessian; ls::MultiVariateLSgpu; end                                  # what gethessgrad hands to the iterator
function NLLSsolver.gethessgrad(ls::MultiVariateLSgpu)                          # src/linearsystem.jl:190
    check(ls.ctx, ccall((:nlls_get_grad, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, ls.b)); return GpuHessian(ls), ls.b
end
NLLSsolver.gethessian(ls::MultiVariateLSgpu) = GpuHessian(ls)
function NLLSsolver.initlambda(h::GpuHessian)                                   # src/iterators.jl:131-137
    m = Ref(0.0); check(h.ls.ctx, ccall((:nlls_max_abs_diag, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ls.ctx, m)); return m[] * 1e-6
end
NLLSsolver.uniformscaling!(h::GpuHessian, k) = check(h.ls.ctx, ccall((:nlls_damp, lib), Cint, (Ptr{Cvoid}, Float64), h.ls.ctx, k))
function NLLSsolver.solve!(ls::MultiVariateLSgpu, options)                      # src/iterators.jl:152 (x comes back ALREADY negated)
    check(ls.ctx, ccall((:nlls_solve, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, ls.x)); ls.x .= .-ls.x; return ls.x   # negate!() of the caller restores the sign
end
function NLLSsolver.fast_bAb(h::GpuHessian, x::Vector)                          # src/iterators.jl:163
    check(h.ls.ctx, ccall((:nlls_set_step, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ls.ctx, x))
    q = Ref(0.0); g = Ref(0.0); check(h.ls.ctx, ccall((:nlls_quadform, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h.ls.ctx, q, g)); return q[]
end
# update!(to, from, linsystem) keeps the reference's host implementation (src/linearsystem.jl:206-213): it only
# needs blockindices / boffsets / x, which this type carries, so arbitrary user `update` methods keep working.
NLLSsolver.getoffsets(block, ls::MultiVariateLSgpu) = @inbounds(ls.blockindices[NLLSsolver.varindices(block)])

# ---- Levenberg-Marquardt with the variables resident on the device ------------------------------------------------
# optimizeinternal! (src/optimize.jl:109-180) and iterate!(::LevMarData) (src/iterators.jl:139-172) for the GPU linear system,
# statement for statement, with
#   update!(varnext, variables, linsystem) + cost(varnext, costs)  ->  ONE nlls_lm_trial (damp, solve, retract, cost sweep)
#   updatefromnext! / updatefrombest! / updatetobest!              ->  nlls_swap_variables / nlls_copy_variables (pointer swaps)
#   zero! + costgradhess!                                          ->  nlls_sweep_gradhess(ctx, NULL): enqueue only
# No variable crosses PCIe inside the loop; problem.variables is fetched once at the end (and before a user callback).
# tests/abi/abi_replay.c replays exactly this sequence, with these argument types, against the library on the GPU.
const VARS_CURRENT, VARS_NEXT, VARS_BEST = Int32(0), Int32(1), Int32(2)

"device variable set `which` -> the host vector `dest` (problem.variables, problem.varnext or problem.varbest): the inverse of pack!"
function fetchvariables!(dest::Vector, ls::MultiVariateLSgpu, which::Int32)
    resize!(ls.packed, sum(k -> ccall((:nlls_var_storage, lib), Cint, (Int32, Int32), k[1], k[2]), ls.kinds))
    check(ls.ctx, ccall((:nlls_get_variables, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), ls.ctx, which, ls.packed))
    o = 0
    for (i, v) in enumerate(dest)
        dest[i], o = unpack(v, ls.packed, o)
    end
end
fetchvariables!(problem::NLLSProblem, ls::MultiVariateLSgpu, which::Int32 = VARS_CURRENT) = fetchvariables!(problem.variables, ls, which)
unpack(::Number, p, o) = (p[o+1], o + 1)
unpack(::EuclideanVector{N, T}, p, o) where {N, T} = (EuclideanVector{N, T}(ntuple(k -> p[o+k], N)), o + N)
unpack(v::NLLSsolver.DynamicVector{Float64}, p, o) = (p[o+1:o+length(v)], o + length(v))
unpack(::ZeroToInfScalar{T}, p, o) where T = (ZeroToInfScalar{T}(p[o+1]), o + 1)
unpack(::ZeroToOneScalar{T}, p, o) where T = (ZeroToOneScalar{T}(p[o+1]), o + 1)
unpack(::ContaminatedGaussian{T}, p, o) where T = (ContaminatedGaussian(ZeroToInfScalar{T}(p[o+1]), ZeroToInfScalar{T}(p[o+2]), ZeroToOneScalar{T}(p[o+3])), o + 3)

function lm_trial!(ls::MultiVariateLSgpu, dlambda::Float64)      # src/iterators.jl:149-157 in one call, one synchronisation
    c = Ref(0.0)
    check(ls.ctx, ccall((:nlls_lm_trial, lib), Cint, (Ptr{Cvoid}, Float64, Int32, Int32, Ptr{Float64}), ls.ctx, dlambda, VARS_NEXT, VARS_CURRENT, c))
    return c[]
end
function stepmaxabs(ls::MultiVariateLSgpu)                       # maximum(abs, linsystem.x), src/optimize.jl:149
    m = Ref(0.0); check(ls.ctx, ccall((:nlls_step_maxabs, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, m)); return m[]
end

function NLLSsolver.iterate!(levmardata::NLLSsolver.LevMarData, data::NLLSInternal{MultiVariateLSgpu}, problem::NLLSProblem, options::NLLSOptions)::Float64
    ls = data.linsystem
    @assert levmardata.lambda >= 0.
    if levmardata.lambda == 0
        levmardata.lambda = NLLSsolver.initlambda(GpuHessian(ls))
    end
    lastlambda = 0.; mu = 2.
    while true
        data.timesolver += NLLSsolver.@elapsed_ns cost_ = lm_trial!(ls, levmardata.lambda - lastlambda)
        lastlambda = levmardata.lambda
        data.linearsolvers += 1; data.costcomputations += 1
        if !(cost_ > data.bestcost) || stepmaxabs(ls) < options.dstep
            NLLSsolver.uniformscaling!(GpuHessian(ls), -lastlambda)
            q = Ref(0.0); g = Ref(0.0)
            check(ls.ctx, ccall((:nlls_quadform, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), ls.ctx, q, g))
            stepquality = (cost_ - data.bestcost) / (0.5 * q[] + g[])
            levmardata.lambda *= stepquality < 0.983 ? 1 - (2 * stepquality - 1) ^ 3 : 0.1
            return cost_
        end
        levmardata.lambda *= mu; mu *= 2.
    end
end

# ---- the other iterators (src/iterators.jl:15-27 Newton, :47-115 dogleg, :187-208 gradient descent) on the GPU linear system ----------------
# Same quantities, same order of evaluation as the reference; the one substitution is the trial cost: gpucost(ls, problem.varnext) -- an upload of
# the trial point and the device cost sweep -- where the reference calls cost(problem.varnext, problem.costs).  Gradient, Hessian products and the
# solve already go through the overloads above (gethessgrad, fast_bAb, solve!); update! stays the reference's (host: arbitrary `update` methods).
trialcost!(data, problem) = (t0 = Base.time_ns(); c = gpucost(data.linsystem, problem.varnext); data.timecost += Base.time_ns() - t0; data.costcomputations += 1; c)

function NLLSsolver.iterate!(::NLLSsolver.NewtonData, data::NLLSInternal{MultiVariateLSgpu}, problem::NLLSProblem, options::NLLSOptions)::Float64
    NLLSsolver.gethessian(data.linsystem)
    data.timesolver += NLLSsolver.@elapsed_ns NLLSsolver.negate!(NLLSsolver.solve!(data.linsystem, options))
    data.linearsolvers += 1
    NLLSsolver.update!(problem.varnext, problem.variables, data.linsystem)
    return trialcost!(data, problem)
end

function NLLSsolver.iterate!(gd::NLLSsolver.GradientDescentData, data::NLLSInternal{MultiVariateLSgpu}, problem::NLLSProblem, options::NLLSOptions)::Float64
    ls = data.linsystem; g = NLLSsolver.getgrad(ls)
    trial(step) = (ls.x .= (-step) .* g; NLLSsolver.update!(problem.varnext, problem.variables, ls); trialcost!(data, problem))
    c = trial(gd.stepsize)
    while c > data.bestcost                                          # quadratic fit through the last trial (src/iterators.jl:196-204)
        slope = ls.x' * g
        gd.stepsize *= 0.5 * slope / (data.bestcost + slope - c)
        c = trial(gd.stepsize)
    end
    gd.stepsize *= 2
    return c
end
NLLSsolver.getgrad(ls::MultiVariateLSgpu) = (check(ls.ctx, ccall((:nlls_get_grad, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, ls.b)); ls.b)

function NLLSsolver.iterate!(dl::NLLSsolver.DoglegData, data::NLLSInternal{MultiVariateLSgpu}, problem::NLLSProblem, options::NLLSOptions)::Float64
    ls = data.linsystem
    H, g = NLLSsolver.gethessgrad(ls)
    local a, gg, alpha, alpha2, beta
    data.timesolver += NLLSsolver.@elapsed_ns begin
        gg = g' * g
        a = gg / (NLLSsolver.fast_bAb(H, g) + floatmin(Float64))      # Cauchy step length along -g
        dl.cauchy .= (-a) .* g
        alpha2 = a * a * gg; alpha = sqrt(alpha2)
        dl.trustradius == 0 && (dl.trustradius = alpha)               # the first step is the Cauchy point
        beta = Inf
        if alpha < dl.trustradius                                     # the Gauss-Newton step is needed
            NLLSsolver.negate!(NLLSsolver.solve!(ls, options)); beta = sqrt(ls.x' * ls.x); data.linearsolvers += 1
        end
    end
    c = data.bestcost
    while true
        if !(alpha < dl.trustradius)                                  # first leg: along the Cauchy direction
            ls.x .= (dl.trustradius / alpha) .* dl.cauchy
            predicted = dl.trustradius * (2 * alpha - dl.trustradius) / (2 * a)
        elseif beta <= dl.trustradius                                 # the whole Gauss-Newton step fits
            predicted = c
        else                                                          # second leg: Cauchy -> Newton, cut at the trust-region boundary
            ls.x .-= dl.cauchy
            leg2 = ls.x' * ls.x; cl = dl.cauchy' * ls.x; room = dl.trustradius^2 - alpha2
            t = sqrt(cl * cl + leg2 * room)
            t = cl <= 0 ? (t - cl) / leg2 : room / (cl + t)
            ls.x .*= t; ls.x .+= dl.cauchy
            predicted = 0.5 * (a * (1 - t)^2 * gg) + t * (2 - t) * c
        end
        NLLSsolver.update!(problem.varnext, problem.variables, ls)
        c = trialcost!(data, problem)
        gain = (data.bestcost - c) / predicted
        if gain > 0.375
            dl.trustradius = max(dl.trustradius, 3 * sqrt(ls.x' * ls.x))
        elseif gain < 0.125
            dl.trustradius *= 0.5
        end
        (!(c > data.bestcost) || maximum(abs, ls.x) < options.dstep) && return c
    end
end

swapvars!(ls, a, b) = check(ls.ctx, ccall((:nlls_swap_variables, lib), Cint, (Ptr{Cvoid}, Int32, Int32), ls.ctx, a, b))
copyvars!(ls, dst, src) = check(ls.ctx, ccall((:nlls_copy_variables, lib), Cint, (Ptr{Cvoid}, Int32, Int32), ls.ctx, dst, src))
# Two callers reach these three (src/optimize.jl:207-214).  The Levenberg-Marquardt overload of optimizeinternal! below keeps the variables ON THE DEVICE
# (ls.resident): the three device sets are swapped, the host vectors are fetched once at the end.  Every other iterator (Newton, dogleg, gradient descent)
# runs the reference's generic optimizeinternal!, whose costgradhess!(linsystem, problem.variables, costs) uploads the HOST vector every iteration
# (ls.resident == false): there the host vectors are the authority and must advance exactly as the reference's own methods advance them.
function NLLSsolver.updatefromnext!(problem::NLLSProblem, data::NLLSInternal{MultiVariateLSgpu})
    if data.linsystem.resident; swapvars!(data.linsystem, VARS_CURRENT, VARS_NEXT)
    else; problem.variables, problem.varnext = problem.varnext, problem.variables; end
end
function NLLSsolver.updatefrombest!(problem::NLLSProblem, data::NLLSInternal{MultiVariateLSgpu})
    if data.linsystem.resident; swapvars!(data.linsystem, VARS_CURRENT, VARS_BEST)
    else; problem.variables, problem.varbest = problem.varbest, problem.variables; end
end
# src/optimize.jl:138-142,214.  Device-resident: a swap once varbest exists (what CURRENT then holds is stale, and updatefromnext! replaces it at once), a copy
# the first time (the reference's deepcopy) -- nlls_lm.cpp does the same.  Host-resident: the reference's swap (its caller has checked the lengths).
function NLLSsolver.updatetobest!(problem::NLLSProblem, data::NLLSInternal{MultiVariateLSgpu})
    ls = data.linsystem
    if !ls.resident; problem.variables, problem.varbest = problem.varbest, problem.variables; return; end
    if ls.havebest; swapvars!(ls, VARS_CURRENT, VARS_BEST); else; copyvars!(ls, VARS_BEST, VARS_CURRENT); ls.havebest = true; end
end

# nlls_lm_options / nlls_lm_state (include/nlls_amd.h; sizes and offsets pinned by tests/abi/abi_replay.c): the library's own outer loop
struct LmOptions; reldcost::Float64; absdcost::Float64; dstep::Float64; maxfails::Int64; maxiters::Int64; stoptime_ns::Int64; end
mutable struct LmState
    lambda::Float64; bestcost::Float64; cost::Float64
    iternum::Int64; fails::Int64; have_best::Int64; converged::Int64
    linearsolvers::Int64; costcomputations::Int64; gradientcomputations::Int64; singulartrials::Int64
    timesolver_ns::Int64; timegradient_ns::Int64
    timecost_ns::Int64                     # (round 6, appended: device-timed buckets, nlls_get_time_buckets)
end

function NLLSsolver.optimizeinternal!(problem::NLLSProblem, options::NLLSOptions, data::NLLSInternal{MultiVariateLSgpu}, iteratedata::NLLSsolver.LevMarData, callback)
    ls = data.linsystem
    if callback === NLLSsolver.nullcallback
        # no user code between two iterations: the whole while-loop below runs inside the library (nlls_lm_iterations, csrc/nlls_lm.cpp --
        # the same statements, in terms of the same entry points), so that the GPU does not wait for this process between two trials
        data.startcost = NLLSsolver.preoptimization(iteratedata, problem, options, data)::Float64
        data.iternum = 0
        data.timeinit += Base.time_ns() - data.starttime
        setvariables!(ls, problem.variables, VARS_CURRENT); copyvars!(ls, VARS_NEXT, VARS_CURRENT); ls.resident = true
        data.timegradient += NLLSsolver.@elapsed_ns cost = NLLSsolver.costgradhess!(ls, problem.variables, problem.costs)
        data.gradientcomputations += 1
        data.bestcost = cost; data.startcost = max(cost, data.startcost)
        big = typemax(Int64) >> 1
        opts = Ref(LmOptions(options.reldcost, options.absdcost, options.dstep, min(Int64(options.maxfails), big), min(Int64(options.maxiters), big),
                             min(Int64(data.starttime + options.maxtime), big)))           # Base.time_ns() is CLOCK_MONOTONIC, as the library's clock
        st = LmState(iteratedata.lambda, cost, cost, 0, 0, 0, 0, data.linearsolvers, data.costcomputations, data.gradientcomputations, 0, 0, 0, 0)
        GC.@preserve st check(ls.ctx, ccall((:nlls_lm_iterations, lib), Cint, (Ptr{Cvoid}, Ptr{LmOptions}, Ptr{LmState}, Int64), ls.ctx, opts,
                                            Ptr{LmState}(pointer_from_objref(st)), big))
        iteratedata.lambda = st.lambda; data.bestcost = st.bestcost; data.iternum = st.iternum; data.converged = st.converged
        data.linearsolvers = st.linearsolvers; data.costcomputations = st.costcomputations; data.gradientcomputations = st.gradientcomputations
        data.timesolver += st.timesolver_ns; data.timegradient += st.timegradient_ns; data.timecost += st.timecost_ns
        !(st.bestcost >= st.cost) && NLLSsolver.updatefrombest!(problem, data)
        fetchvariables!(problem, ls, VARS_CURRENT); ls.resident = false
        data.timetotal += Base.time_ns() - data.starttime
        return data
    end
    data.startcost = NLLSsolver.preoptimization(iteratedata, problem, options, data)::Float64
    fails = 0; data.iternum = 0
    stoptime = data.starttime + options.maxtime
    data.timeinit += Base.time_ns() - data.starttime
    setvariables!(ls, problem.variables, VARS_CURRENT); copyvars!(ls, VARS_NEXT, VARS_CURRENT); ls.resident = true
    data.timegradient += NLLSsolver.@elapsed_ns cost = NLLSsolver.costgradhess!(ls, problem.variables, problem.costs)
    data.gradientcomputations += 1
    data.bestcost = cost; data.startcost = max(cost, data.startcost)
    while true
        data.iternum += 1
        cost = NLLSsolver.iterate!(iteratedata, data, problem, options)::Float64
        # a user callback sees the trial point where the reference puts it -- problem.varnext (src/optimize.jl:128; src/callbacks.jl) -- and the step in
        # linsystem.x; what it WRITES there counts: the reference's own EM pattern edits varnext inside the callback (src/robustadaptive.jl:48-73,
        # test/adaptivecost.jl:54), and updatefromnext! then makes it the current point.  So: next variables down before, and up again after.
        length(problem.varnext) == length(problem.variables) || (problem.varnext = deepcopy(problem.variables))
        fetchvariables!(problem.varnext, ls, VARS_NEXT); check(ls.ctx, ccall((:nlls_get_step, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, ls.x))
        cost, terminate = callback(cost, problem, data, iteratedata)::Tuple{Float64, Int}
        setvariables!(ls, problem.varnext, VARS_NEXT)                # nlls_set_variables(ctx, 1, ...): host edits of varnext reach the device
        dcost = data.bestcost - cost
        if dcost >= 0
            data.bestcost = cost; fails = 0
        else
            dcost = cost; fails += 1
            fails == 1 && NLLSsolver.updatetobest!(problem, data)
        end
        NLLSsolver.updatefromnext!(problem, data)
        maxstep = stepmaxabs(ls)
        converged = 0
        converged |= isinf(cost) << 0; converged |= isnan(cost) << 1
        converged |= (dcost < data.bestcost * options.reldcost) << 2; converged |= (dcost < options.absdcost) << 3
        converged |= isinf(maxstep) << 4; converged |= isnan(maxstep) << 5; converged |= (maxstep < options.dstep) << 6
        converged |= (fails > options.maxfails) << 7; converged |= (data.iternum >= options.maxiters) << 8
        converged |= (Base.time_ns() > stoptime) << 9; converged |= terminate << 16
        data.converged = converged
        converged != 0 && break
        data.timegradient += NLLSsolver.@elapsed_ns check(ls.ctx, ccall((:nlls_sweep_gradhess, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, C_NULL))
        data.gradientcomputations += 1
    end
    !(data.bestcost >= cost) && NLLSsolver.updatefrombest!(problem, data)
    fetchvariables!(problem, ls, VARS_CURRENT); ls.resident = false
    data.timetotal += Base.time_ns() - data.starttime
    return data
end

end # module
