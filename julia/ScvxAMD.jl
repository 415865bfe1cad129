# ScvxAMD.jl — ccall binding of libscvx_hip.so behind the reference's own API.
#
# NOT EXECUTED IN THE BUILD CONTAINER (no Julia there, SURVEY.md F6): this is the binding a maintainer of
# BenChung/SuccessiveConvexification adds next to master.jl.  It keeps RocketlandDefns' types and replaces
#   Dynamics.linearize_dynamics (dynamics.jl:321), Dynamics.predict_state (:315),
#   Rocketland.create_initial (rocketland.jl:34), solve_step (:226), solve_problem (:432)
# with calls through include/scvx.h.  Every symbol used below is exercised by the Python ctypes host layer
# (successiveconvexification_amd/_lib.py), which binds the identical C signatures.
module ScvxAMD
using ..RocketlandDefns
using LinearAlgebra

const LIB = get(ENV, "SCVX_HIP_LIB", joinpath(@__DIR__, "..", "successiveconvexification_amd", "libscvx_hip.so"))

# struct scvx_problem (include/scvx.h) — field order and types must match exactly
struct CProblem
    g::Cdouble; mdry::Cdouble; mwet::Cdouble; Tmin::Cdouble; Tmax::Cdouble
    deltaMax::Cdouble; thetaMax::Cdouble; gammaGs::Cdouble; omMax::Cdouble; dpMax::Cdouble
    jB::NTuple{9,Cdouble}
    alpha::Cdouble; rho::Cdouble; sos::Cdouble
    rTB::NTuple{3,Cdouble}; rFB::NTuple{3,Cdouble}
    rIi::NTuple{3,Cdouble}; rIf::NTuple{3,Cdouble}; vIi::NTuple{3,Cdouble}; vIf::NTuple{3,Cdouble}
    qBIi::NTuple{4,Cdouble}; qBIf::NTuple{4,Cdouble}
    wBi::NTuple{3,Cdouble}; wBf::NTuple{3,Cdouble}
    wNu::Cdouble; wID::Cdouble; wDS::Cdouble; wCst::Cdouble; wTviol::Cdouble; nuTol::Cdouble; delTol::Cdouble; tf_guess::Cdouble
    ri::Cdouble; rh0::Cdouble; rh1::Cdouble; rh2::Cdouble; alph::Cdouble; bet::Cdouble
    force_scalar::Cdouble; length_scalar::Cdouble
    K::Int32; imax::Int32; aero_kind::Int32; reserved::Int32
end

t3(v) = (Float64(v[1]), Float64(v[2]), Float64(v[3]))
t4(v) = (Float64(v[1]), Float64(v[2]), Float64(v[3]), Float64(v[4]))

function CProblem(p::DescentProblem)
    aero = p.aero isa AtmosphericData
    CProblem(p.g, p.mdry, p.mwet, p.Tmin, p.Tmax, p.deltaMax, p.thetaMax, p.gammaGs, p.omMax, p.dpMax,
             Tuple(Float64.(vec(p.jB))), p.alpha, p.rho, p.sos, t3(p.rTB), t3(p.rFB), t3(p.rIi), t3(p.rIf), t3(p.vIi), t3(p.vIf),
             t4(p.qBIi), t4(p.qBIf), t3(p.wBi), t3(p.wBf),
             p.wNu, p.wID, p.wDS, p.wCst, p.wTviol, p.nuTol, p.delTol, p.tf_guess, p.ri, p.rh0, p.rh1, p.rh2, p.alph, p.bet,
             aero ? p.aero.force_scalar : 1.0, aero ? p.aero.length_scalar : 1.0,
             Int32(p.K), Int32(p.imax), Int32(aero ? 1 : 0), Int32(0))
end

check(ctx, rc, what) = rc == 0 || error("$what failed ($rc): " * unsafe_string(ccall((:scvx_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx)))

# IntegratorCache (dynamics.jl:258) becomes the owner of the device context
mutable struct Cache
    ctx::Ptr{Cvoid}
    problem::DescentProblem
end
function Cache(prob::DescentProblem; device::Int=0, npts::Int=10)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    cp = Ref(CProblem(prob))
    rc = ccall((:scvx_ctx_create, LIB), Cint, (Ref{CProblem}, Cint, Ref{Ptr{Cvoid}}), cp, device, ref)
    rc == 0 || error("scvx_ctx_create failed ($rc)")
    c = Cache(ref[], prob)
    check(c.ctx, ccall((:scvx_set_nsub, LIB), Cint, (Ptr{Cvoid}, Cint), c.ctx, npts), "scvx_set_nsub")
    finalizer(x -> ccall((:scvx_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.ctx), c)
    return c
end

# Dynamics.linearize_dynamics(states, tf_guess, base_dt, cache) -> Array{LinRes,1}   (dynamics.jl:321-334)
function linearize_dynamics(states::Array{LinPoint,1}, tf_guess::Float64, base_dt::Float64, cache::Cache)
    K = length(states) - 1
    x = hcat((s.state for s in states)...)       # 14 x (K+1), column-major == [K+1][14]
    u = hcat((s.control for s in states)...)     # 3 x (K+1)
    endpoint = Matrix{Float64}(undef, 14, K)
    deriv = Array{Float64,3}(undef, 14, 21, K)   # column-major 14x21 per segment == [K][21][14]
    check(cache.ctx, ccall((:scvx_linearize_f64_host, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}),
        cache.ctx, 1, K, x, u, [tf_guess], base_dt, endpoint, deriv), "scvx_linearize_f64_host")
    return [LinRes(endpoint[:, k], deriv[:, :, k]) for k = 1:K]
end

# Dynamics.predict_state(initial_state, uk, up, sigma, dt, pinfo, cache)   (dynamics.jl:315-317)
function predict_state(initial_state, uk, up, sigma, dt, pinfo, cache::Cache)
    x = hcat(initial_state, zeros(14)); u = hcat(uk, up); out = Matrix{Float64}(undef, 14, 1)
    check(cache.ctx, ccall((:scvx_propagate_f64_host, LIB), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}),
        cache.ctx, 1, 1, x, u, [Float64(sigma)], dt, out), "scvx_propagate_f64_host")
    return out[:, 1]
end

# The batched ProblemIteration.  B = 1 reproduces the reference's single-trajectory API.
mutable struct Batch
    h::Ptr{Cvoid}
    cache::Cache
    B::Int
end
function create_initial(problem::DescentProblem, cache::Cache; ics::Union{Nothing,Matrix{Float64}}=nothing)
    B = ics === nothing ? 1 : size(ics, 2)       # ics: 6 x B = (rIi; vIi) per trajectory
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(cache.ctx, ccall((:scvx_batch_create, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), cache.ctx, B, ref), "scvx_batch_create")
    b = Batch(ref[], cache, B)
    finalizer(x -> ccall((:scvx_batch_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), b)
    check(cache.ctx, ccall((:scvx_batch_init, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), b.h, ics === nothing ? C_NULL : ics), "scvx_batch_init")
    return b
end

# solve_step(iteration, cache) -> (iteration, ||nu||, dJ)     (rocketland.jl:226-321)
function solve_step(b::Batch, cache::Cache=b.cache)
    st = Vector{Int32}(undef, b.B); nu = Vector{Float64}(undef, b.B); dj = Vector{Float64}(undef, b.B)
    check(cache.ctx, ccall((:scvx_solve_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}), b.h, st, nu, dj), "scvx_solve_step")
    any(st .== 3) && error("Non-optimal result exiting")            # rocketland.jl:273-276
    return b.B == 1 ? (b, nu[1], dj[1]) : (b, nu, dj)
end

# solve_problem(iprob, cache) -> (iteration, cnu, cdel)        (rocketland.jl:432-443)
function solve_problem(iprob::DescentProblem, cache::Cache; ics=nothing)
    b = create_initial(iprob, cache; ics=ics)
    st = Vector{Int32}(undef, b.B); it = Vector{Int32}(undef, b.B); nu = Vector{Float64}(undef, b.B); dj = Vector{Float64}(undef, b.B)
    check(cache.ctx, ccall((:scvx_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}), b.h, st, it, nu, dj), "scvx_solve")
    return b.B == 1 ? (b, nu[1], dj[1]) : (b, nu, dj)
end

# iterate access: about::Array{LinPoint,1}, sigma  (master.jl:122-134)
function about(b::Batch)
    K = b.cache.problem.K; nrec = (K + 1) * 17 + 1
    rec = Matrix{Float64}(undef, nrec, b.B)
    check(b.cache.ctx, ccall((:scvx_batch_get_trajectory, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), b.h, rec), "scvx_batch_get_trajectory")
    map(1:b.B) do t
        x = reshape(rec[1:14(K+1), t], 14, K + 1); u = reshape(rec[14(K+1)+1:17(K+1), t], 3, K + 1)
        ([LinPoint(x[:, k], u[:, k]) for k = 1:K+1], rec[end, t])
    end
end
end
