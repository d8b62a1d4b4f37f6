# NLLSsolverAMD.jl -- the reference-side binding of libnlls_amd.so (include/nlls_amd.h).
#
# NOT TESTED IN THIS REPOSITORY: no Julia toolchain exists in the build image (SURVEY.md F6).  It is the
# `ccall` shim a maintainer of NLLSsolver.jl would add; the Python mirror in nllssolver.jl_amd/ exercises the
# same C entry points in the test-suite.  Nothing in the reference tree is modified: the shim only adds a new
# linear-system type and method overloads of the generic functions the iterators already call
# (src/iterators.jl:139-172, src/optimize.jl:109-180).
module NLLSsolverAMD

import NLLSsolver
using NLLSsolver: NLLSProblem, NLLSOptions, NLLSInternal, EuclideanVector, ZeroToInfScalar, ZeroToOneScalar,
                  ContaminatedGaussian, NoRobust, HuberKernel, GemanMcclureKernel, Scaled, SimpleError2
using StaticArrays

const lib = get(ENV, "NLLS_AMD_LIB", "libnlls_amd.so")

# ---- kinds (include/nlls_amd.h) ------------------------------------------------------------------------
const VAR_EUCLIDEAN, VAR_ZERO_TO_INF, VAR_ZERO_TO_ONE, VAR_CONTAMINATED_GAUSSIAN = Int32(1), Int32(2), Int32(3), Int32(4)
const ROBUST_NONE, ROBUST_HUBER, ROBUST_HUBER2O, ROBUST_GEMAN_MCCLURE, ROBUST_SCALED = Int32(0), Int32(1), Int32(2), Int32(3), Int32(0x10)

struct CostGroup              # nlls_cost_group
    res_kind::Int32
    robust_kind::Int32
    robust_params::NTuple{4, Float64}
    ncost::Int64
    varind::Ptr{Int64}
    data::Ptr{Float64}
end

check(ctx, rc) = rc == 0 ? nothing : error("nlls_amd error $rc: " * unsafe_string(ccall((:nlls_last_error, lib), Cstring, (Ptr{Cvoid},), ctx)))

# variable kind/dim + packed storage of a reference variable
varkind(::Number) = (VAR_EUCLIDEAN, Int32(1));                    pack!(out, v::Number) = push!(out, Float64(v))
varkind(::EuclideanVector{N}) where N = (VAR_EUCLIDEAN, Int32(N)); pack!(out, v::EuclideanVector) = append!(out, v)
varkind(::ZeroToInfScalar) = (VAR_ZERO_TO_INF, Int32(1));         pack!(out, v::ZeroToInfScalar) = push!(out, v.val)
varkind(::ZeroToOneScalar) = (VAR_ZERO_TO_ONE, Int32(1));         pack!(out, v::ZeroToOneScalar) = push!(out, v.val)
varkind(::ContaminatedGaussian) = (VAR_CONTAMINATED_GAUSSIAN, Int32(3))
pack!(out, v::ContaminatedGaussian) = append!(out, (v.invsigma1.val, v.invsigma2.val, v.w.val))
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
    info = zeros(Int64, 16); ccall((:nlls_get_info, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), ctx, info)
    ndof = Int(info[4])          # nlls_info: 2 x int32, then nvar, nblocks, ndof, ...
    boff = zeros(Int64, nb); ccall((:nlls_get_bsm_index, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), ctx, C_NULL, C_NULL, C_NULL, boff)
    ls = MultiVariateLSgpu(ctx, blockindices, UInt.(boff), ndof, zeros(ndof), zeros(ndof), kinds, Float64[])
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
    setvariables!(ls, vars, 0); c = Ref(0.0)
    check(ls.ctx, ccall((:nlls_sweep_gradhess, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ls.ctx, c)); return c[]
end
function gpucost(ls::MultiVariateLSgpu, vars::Vector)                           # src/cost.jl:10-13
    setvariables!(ls, vars, 1); c = Ref(0.0)
    check(ls.ctx, ccall((:nlls_sweep_cost, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), ls.ctx, 1, c)); return c[]
end
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

end # module
