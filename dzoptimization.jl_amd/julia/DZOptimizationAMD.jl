# DZOptimizationAMD.jl -- Julia host module over the C ABI of include/dzo.h (libdzo_hip.so).
#
# Drop-in for the in-place BFGS / L-BFGS `step!()` path of dzhang314/DZOptimization.jl on an
# AMD MI355X: the same constructors, the same public fields, `step!`, and the termination flag
# under its three historical names (`is_stuck` live src/DZOptimization.jl:327, `has_terminated`
# legacy/DZOptimization.jl:478,738, `has_converged` README.md:38).  No CUDA.jl, no AMDGPU.jl:
# device memory and kernels live behind `ccall`.
#
# NOT EXERCISED IN THE BUILD CONTAINER (there is no Julia there); the tested twin of this file
# is the Python module next to it, which binds the very same symbols with ctypes.  Keep this
# file thin enough to be correct by inspection: every method is one ccall.
#
#   using DZOptimizationAMD
#   x   = HipVector(rand(10_000_000))
#   opt = LBFGSOptimizer(nothing, RosenbrockChain(length(x)), nothing, x, 1.0, 20)
#   while !opt.is_stuck[]; step!(opt); end
module DZOptimizationAMD

using LinearAlgebra
import LinearAlgebra: axpy!, axpby!, dot, norm, rmul!

export HipVector, LBFGSOptimizer, BFGSOptimizer, AdGDOptimizer, GradientDescentOptimizer, QuadraticLineSearch,
       UniformBoxConstraint, with_l2!, with_box_gradient!, with_box_constraint!, step!,
       set_safeguards!, set_line_search!, BACKTRACKING, STRONG_WOLFE,
       RosenbrockChain, Rosenbrock2D, DenseQuadratic, LogSumExp, QuadraticChain, BuiltinProblem,
       BatchedBFGSOptimizer, count_active, LineSearchEvaluator, compute_lbfgs_step_direction!,
       update_inverse_hessian!, reset_inverse_hessian!, synchronize,
       norm2, inv_norm, negate!, scale!, HipBackend, install_state!, ShardComm, all_done, read_field

const libdzo = get(ENV, "DZO_LIB", joinpath(@__DIR__, "..", "libdzo_hip.so"))

const DZO_F32, DZO_F64 = Cint(0), Cint(1)
dtype_code(::Type{Float32}) = DZO_F32
dtype_code(::Type{Float64}) = DZO_F64

struct DzoError <: Exception
    code::Int
    msg::String
end

function check(rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:dzo_last_error, libdzo), Cstring, ()))
    rc == 3 && throw(AssertionError(msg))          # the reference's @assert
    throw(DzoError(rc, msg))
end

const _initialised = Ref(false)
function init(device::Integer=0)
    check(ccall((:dzo_init, libdzo), Cint, (Cint,), device))
    _initialised[] = true
end
ensure_init() = _initialised[] || init(parse(Int, get(ENV, "DZO_DEVICE", "0")))
"""Wait for every stream of the library (step functions return before their last kernels finish)."""
synchronize() = check(ccall((:dzo_synchronize, libdzo), Cint, ()))
"""Diagnostic (include/dzo.h): how often a host wait found a kernel's published result not yet matching its seal."""
function unsealed_first_reads()
    r = Ref{Int64}(0)
    check(ccall((:dzo_unsealed_first_reads, libdzo), Cint, (Ref{Int64},), r))
    return r[]
end

################################################################################ HipVector

"""Dense device vector: the array type `A<:AbstractArray{T}` the reference's optimizers are
generic over (src/DZOptimization.jl:101,179,321)."""
mutable struct HipVector{T<:Union{Float32,Float64}} <: AbstractVector{T}
    ptr::Ptr{Cvoid}
    len::Int
    owner::Bool
    function HipVector{T}(::UndefInitializer, n::Integer) where {T}
        ensure_init()
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dzo_malloc, libdzo), Cint, (Ref{Ptr{Cvoid}}, Int64), p, max(n * sizeof(T), 16)))
        v = new{T}(p[], n, true)
        finalizer(v) do w
            w.owner && w.ptr != C_NULL && ccall((:dzo_free, libdzo), Cint, (Ptr{Cvoid},), w.ptr)
            w.ptr = C_NULL
        end
        return v
    end
    HipVector{T}(p::Ptr{Cvoid}, n::Integer) where {T} = new{T}(p, n, false)   # borrowed view
end

function HipVector(a::AbstractArray{T}) where {T<:Union{Float32,Float64}}
    v = HipVector{T}(undef, length(a))
    h = Array{T}(vec(a))
    check(ccall((:dzo_memcpy_h2d, libdzo), Cint, (Ptr{Cvoid}, Ptr{T}, Int64), v.ptr, h, sizeof(h)))
    return v
end

Base.size(v::HipVector) = (v.len,)
Base.length(v::HipVector) = v.len
Base.similar(v::HipVector{T}) where {T} = HipVector{T}(undef, v.len)
Base.getindex(v::HipVector, i::Int) = Array(v)[i]          # debugging only: one D2H per call
function Base.Array(v::HipVector{T}) where {T}
    h = Vector{T}(undef, v.len)
    check(ccall((:dzo_memcpy_d2h, libdzo), Cint, (Ptr{T}, Ptr{Cvoid}, Int64), h, v.ptr, sizeof(h)))
    return h
end
function Base.copy!(dst::HipVector{T}, src::HipVector{T}) where {T}
    check(ccall((:dzo_copy, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}), src.len, dtype_code(T), src.ptr, dst.ptr))
    return dst
end
Base.copy(v::HipVector) = copy!(similar(v), v)
function Base.fill!(v::HipVector{T}, a) where {T}
    check(ccall((:dzo_fill, libdzo), Cint, (Int64, Cint, Cdouble, Ptr{Cvoid}), v.len, dtype_code(T), Float64(a), v.ptr))
    return v
end
function Base.isequal(a::HipVector{T}, b::HipVector{T}) where {T}
    r = Ref{Cint}(0)
    check(ccall((:dzo_isequal, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cint}), a.len, dtype_code(T), a.ptr, b.ptr, r))
    return r[] != 0
end

# the L1 method set the reference calls (src/DZOptimization.jl:4)
function axpy!(a::Number, x::HipVector{T}, y::HipVector{T}) where {T}
    check(ccall((:dzo_axpy, libdzo), Cint, (Int64, Cint, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}), x.len, dtype_code(T), Float64(a), x.ptr, y.ptr))
    return y
end
function axpby!(a::Number, x::HipVector{T}, b::Number, y::HipVector{T}) where {T}
    check(ccall((:dzo_axpby, libdzo), Cint, (Int64, Cint, Cdouble, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}), x.len, dtype_code(T), Float64(a), x.ptr, Float64(b), y.ptr))
    return y
end
function rmul!(x::HipVector{T}, a::Number) where {T}
    check(ccall((:dzo_scal, libdzo), Cint, (Int64, Cint, Cdouble, Ptr{Cvoid}), x.len, dtype_code(T), Float64(a), x.ptr))
    return x
end
function dot(x::HipVector{T}, y::HipVector{T}) where {T}
    r = Ref{Cdouble}(0)
    check(ccall((:dzo_dot, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}), x.len, dtype_code(T), x.ptr, y.ptr, r))
    return T(r[])
end
function norm(x::HipVector{T}) where {T}
    r = Ref{Cdouble}(0)
    check(ccall((:dzo_nrm2, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Ref{Cdouble}), x.len, dtype_code(T), x.ptr, r))
    return T(r[])
end

# legacy/Kernels.jl primitives without a LinearAlgebra twin (SURVEY.md a14)
"""`norm2(x)`: the sum of squares, NOT its root (legacy/Kernels.jl:49-55,139)."""
function norm2(x::HipVector{T}) where {T}
    r = Ref{Cdouble}(0)
    check(ccall((:dzo_norm2, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Ref{Cdouble}), x.len, dtype_code(T), x.ptr, r))
    return T(r[])
end
"""`inv_norm(x) = rsqrt(norm2(x))` (legacy/Kernels.jl:141)."""
function inv_norm(x::HipVector{T}) where {T}
    r = Ref{Cdouble}(0)
    check(ccall((:dzo_inv_norm, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Ref{Cdouble}), x.len, dtype_code(T), x.ptr, r))
    return T(r[])
end
"""`negate!(x)` (legacy/Kernels.jl:76-83,143)."""
function negate!(x::HipVector{T}) where {T}
    check(ccall((:dzo_negate, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}), x.len, dtype_code(T), x.ptr))
    return x
end
"""`scale!(x, alpha)` in place (legacy/Kernels.jl:87-94,145-146), `scale!(dst, alpha, x)` out of place (:96-104)."""
scale!(x::HipVector, alpha::Number) = rmul!(x, alpha)
function scale!(dst::HipVector{T}, alpha::Number, x::HipVector{T}) where {T}
    check(ccall((:dzo_scal_oop, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}), x.len, dtype_code(T), dst.ptr, Float64(alpha), x.ptr))
    return x                                                   # (the reference returns x, :103)
end

# KernelAbstractions.get_backend(::HipVector): the reference's constructors call it on every array and
# @assert the backends equal (src/DZOptimization.jl:3,44-54,216-226,363-378,410,420).  KernelAbstractions
# is a dependency of the REFERENCE, not of this module: when it is loadable (it is whenever the
# reference itself is), HipVector gets a singleton backend, so the reference's own generic
# constructors and `step!` run unmodified on HipVector (INTEGRATION.md route A).
const _KA = try
    Base.require(Base.PkgId(Base.UUID("63c18a36-062a-441e-b654-da1e3ab1ce7c"), "KernelAbstractions"))
catch
    nothing
end
if _KA !== nothing
    @eval begin
        """Singleton backend of `HipVector` (device memory behind libdzo_hip.so)."""
        struct HipBackend <: $(_KA).GPU end
        $(_KA).get_backend(::HipVector) = HipBackend()
    end
else
    struct HipBackend end                                       # KernelAbstractions absent: route B only
end

################################################################################ built-in objectives

mutable struct BuiltinProblem{T}
    handle::Ptr{Cvoid}
    n::Int
    keep::Any
    function BuiltinProblem{T}(h::Ptr{Cvoid}, n::Integer, keep) where {T}
        # optimizers built from a problem hold it in a field, so it outlives their handles
        return finalizer(q -> ccall((:dzo_problem_destroy, libdzo), Cint, (Ptr{Cvoid},), q.handle), new{T}(h, n, keep))
    end
end
function _problem(kind, n, ::Type{T}; A=nothing, c=nothing, lambda=0.0) where {T}
    ensure_init()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_problem_create, libdzo), Cint,
                (Cint, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}),
                kind, n, dtype_code(T), A === nothing ? C_NULL : A.ptr, c === nothing ? C_NULL : c.ptr, lambda, h))
    return BuiltinProblem{T}(h[], n, (A, c))
end
Rosenbrock2D(::Type{T}=Float64) where {T} = _problem(0, 2, T)
RosenbrockChain(n::Integer, ::Type{T}=Float64) where {T} = _problem(1, n, T)
DenseQuadratic(A::HipVector{T}, n::Integer) where {T} = _problem(2, n, T; A=A)      # A column-major n*n
LogSumExp(c::HipVector{T}, lambda) where {T} = _problem(3, length(c), T; c=c, lambda=lambda)
# sum 1/2 (x[i+1]-x[i])^2 + lambda/2 (x[i]-1)^2: the large-n convex quadratic (tridiagonal Hessian); like RosenbrockChain it runs on
# the L-BFGS point pass (DZO_PROBLEM_QUADRATIC_CHAIN)
QuadraticChain(n::Integer, lambda, ::Type{T}=Float64) where {T} = _problem(4, n, T; lambda=lambda)
function (p::BuiltinProblem{T})(x::HipVector{T}) where {T}                           # objective_function(x)
    f = Ref{Cdouble}(0)
    check(ccall((:dzo_problem_eval, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}), p.handle, x.ptr, f))
    return T(f[])
end
function gradient!(p::BuiltinProblem{T}, g::HipVector{T}, x::HipVector{T}) where {T}
    check(ccall((:dzo_problem_grad, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), p.handle, g.ptr, x.ptr))
    return g
end

# decorators of legacy/DZOptimization.jl:219-296, applied on the device
"""`with_l2!(p, lambda)`: L2RegularizationWrapper + L2GradientWrapper (legacy :225-249)."""
with_l2!(p::BuiltinProblem, lambda::Real) = (check(ccall((:dzo_problem_set_l2, libdzo), Cint, (Ptr{Cvoid}, Cdouble), p.handle, lambda)); p)
"""`with_box_gradient!(p, lo, hi)`: UniformBoxGradientWrapper (legacy :275-296)."""
with_box_gradient!(p::BuiltinProblem, lo::Real, hi::Real) =
    (check(ccall((:dzo_problem_set_box_gradient, libdzo), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble), p.handle, 1, lo, hi)); p)
"""`with_box_constraint!(p, lo, hi)`: UniformBoxConstraint (legacy :258-272) as constraint_function!."""
with_box_constraint!(p::BuiltinProblem, lo::Real, hi::Real) =
    (check(ccall((:dzo_problem_set_box_constraint, libdzo), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble), p.handle, 1, lo, hi)); p)
"""`UniformBoxConstraint(lo, hi)(x)` on a device vector (legacy :264-272)."""
struct UniformBoxConstraint{T}; lower_bound::T; upper_bound::T; end
function (c::UniformBoxConstraint)(x::HipVector{T}) where {T}
    check(ccall((:dzo_box_clamp, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Cdouble, Cdouble), x.len, dtype_code(T), x.ptr, c.lower_bound, c.upper_bound))
    return true
end

################################################################################ callbacks

# C trampolines: ctx is a pointer to a Julia Ref holding (constraint!, objective, gradient!, T, n)
struct Callbacks{C,F,G,T}
    constraint!::C
    objective::F
    gradient!::G
    n::Int
end
function _c_constraint(ctx::Ptr{Cvoid}, x::Ptr{Cvoid})::Cint
    cb = unsafe_pointer_to_objref(ctx)[]
    return Cint(cb.constraint!(_view(cb, x)) ? 1 : 0)
end
function _c_objective(ctx::Ptr{Cvoid}, x::Ptr{Cvoid})::Cdouble
    cb = unsafe_pointer_to_objref(ctx)[]
    return Cdouble(cb.objective(_view(cb, x)))
end
function _c_gradient(ctx::Ptr{Cvoid}, g::Ptr{Cvoid}, x::Ptr{Cvoid})::Cvoid
    cb = unsafe_pointer_to_objref(ctx)[]
    cb.gradient!(_view(cb, g), _view(cb, x))
    return nothing
end
_view(cb::Callbacks{C,F,G,T}, p::Ptr{Cvoid}) where {C,F,G,T} = HipVector{T}(p, cb.n)

################################################################################ L-BFGS

abstract type AbstractOptimizer{T,A} end
function step! end

"""`LBFGSOptimizer(constraint!, objective, gradient!, x0, step, m)` (src/DZOptimization.jl:400-407)
or the full form with `f0, g0` (:347-356).  `objective` may be a `BuiltinProblem`, in which case
the whole step runs on the device.  Aliases `x0` as `current_point` (:393)."""
mutable struct LBFGSOptimizer{T,A,C,F,G} <: AbstractOptimizer{T,A}
    handle::Ptr{Cvoid}
    constraint_function!::C
    objective_function::F
    gradient_function!::G
    current_point::A
    history_length::Int
    keep::Any
    function LBFGSOptimizer{T,A,C,F,G}(h, c, f, g, x, m, keep) where {T,A,C,F,G}
        return finalizer(o -> ccall((:dzo_lbfgs_destroy, libdzo), Cint, (Ptr{Cvoid},), getfield(o, :handle)),
                         new{T,A,C,F,G}(h, c, f, g, x, m, keep))
    end
end

function LBFGSOptimizer(constraint!::C, objective::F, gradient!::G, x0::HipVector{T},
                        initial_step_length::Real, history_length::Int) where {T,C,F,G}
    ensure_init()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if objective isa BuiltinProblem && constraint! === nothing
        check(ccall((:dzo_lbfgs_create_problem, libdzo), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}),
                    objective.handle, history_length, x0.ptr, initial_step_length, h))
        return LBFGSOptimizer{T,HipVector{T},C,F,G}(h[], constraint!, objective, gradient!, x0, history_length, nothing)
    end
    cb = Ref(Callbacks{C,F,G,T}(constraint!, objective, gradient!, length(x0)))
    cf = constraint! === nothing ? C_NULL : @cfunction(_c_constraint, Cint, (Ptr{Cvoid}, Ptr{Cvoid}))
    check(ccall((:dzo_lbfgs_create_callbacks, libdzo), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}),
                cf, @cfunction(_c_objective, Cdouble, (Ptr{Cvoid}, Ptr{Cvoid})),
                @cfunction(_c_gradient, Cvoid, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid})),
                pointer_from_objref(cb), length(x0), history_length, dtype_code(T), x0.ptr, initial_step_length, h))
    return LBFGSOptimizer{T,HipVector{T},C,F,G}(h[], constraint!, objective, gradient!, x0, history_length, cb)
end

function LBFGSOptimizer(constraint!::C, objective::F, gradient!::G, x0::HipVector{T}, f0::Real, g0::HipVector{T},
                        initial_step_length::Real, history_length::Int) where {T,C,F,G}
    ensure_init()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_lbfgs_create, libdzo), Cint, (Int64, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ref{Ptr{Cvoid}}),
                length(x0), history_length, dtype_code(T), x0.ptr, g0.ptr, f0, initial_step_length, h))
    cb = Ref(Callbacks{C,F,G,T}(constraint!, objective, gradient!, length(x0)))
    cf = constraint! === nothing ? C_NULL : @cfunction(_c_constraint, Cint, (Ptr{Cvoid}, Ptr{Cvoid}))
    check(ccall((:dzo_lbfgs_set_callbacks, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                h[], cf, @cfunction(_c_objective, Cdouble, (Ptr{Cvoid}, Ptr{Cvoid})),
                @cfunction(_c_gradient, Cvoid, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid})), pointer_from_objref(cb)))
    return LBFGSOptimizer{T,HipVector{T},C,F,G}(h[], constraint!, objective, gradient!, x0, history_length, (cb, g0))
end

"""`step!(opt)` (src/DZOptimization.jl:454-509)."""
step!(opt::LBFGSOptimizer) = (check(ccall((:dzo_lbfgs_step, libdzo), Cint, (Ptr{Cvoid},), getfield(opt, :handle))); opt)

"""`compute_lbfgs_step_direction!(opt)` (src/DZOptimization.jl:430-451): the two-loop recursion alone,
`opt.step_direction = -H_k * opt.current_gradient`."""
compute_lbfgs_step_direction!(opt::LBFGSOptimizer) =
    (check(ccall((:dzo_lbfgs_direction, libdzo), Cint, (Ptr{Cvoid},), getfield(opt, :handle))); opt)

const BACKTRACKING, STRONG_WOLFE = Cint(0), Cint(1)

"""`set_safeguards!(opt; descent_check=false, steepest_descent_fallback=false)`: the legacy
optimizer's descent check (legacy/DZOptimization.jl:682-692) and steepest-descent fallback with
history reset (:588-610) as options of the live `step!`; both off = the reference."""
set_safeguards!(opt::LBFGSOptimizer; descent_check::Bool=false, steepest_descent_fallback::Bool=false) =
    (check(ccall((:dzo_lbfgs_set_safeguards, libdzo), Cint, (Ptr{Cvoid}, Cint, Cint), getfield(opt, :handle),
                 descent_check, steepest_descent_fallback)); opt)

"""`set_line_search!(opt, STRONG_WOLFE; c1=1e-4, c2=0.9, max_evals=40)`: strong-Wolfe search on the
`LineSearchEvaluator` quotients (src/DZOptimization.jl:84, :88-89); `BACKTRACKING` restores
`take_backtracking_step!` (:107-154)."""
set_line_search!(opt::LBFGSOptimizer, kind::Integer; c1::Real=0.0, c2::Real=0.0, max_evals::Integer=0) =
    (check(ccall((:dzo_lbfgs_set_line_search, libdzo), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cint),
                 getfield(opt, :handle), kind, c1, c2, max_evals)); opt)

_lb_i(o, w) = (v = Ref{Int64}(0); check(ccall((:dzo_lbfgs_get_i, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}), getfield(o, :handle), w, v)); v[])
_lb_s(o, w) = (v = Ref{Cdouble}(0); check(ccall((:dzo_lbfgs_get_s, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), getfield(o, :handle), w, v)); v[])
function _lb_p(o::LBFGSOptimizer{T}, w, idx=0) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_lbfgs_get_ptr, libdzo), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}), getfield(o, :handle), w, idx, p))
    return HipVector{T}(p[], length(getfield(o, :current_point)))
end
"""`read_field(opt, :current_point)` (also `:delta_point`, `:current_gradient`, `:delta_gradient`, `:step_direction`): the field as
a host `Vector{T}`, copied by the library itself (`dzo_lbfgs_read`).  No device pointer is handed out, so the step behind the read
does not have to check the aliased arrays for host writes (src/DZOptimization.jl:393; after `opt.current_point`, which hands the
pointer out, it does) -- the way to watch a run."""
function read_field(o::LBFGSOptimizer{T}, s::Symbol) where {T}
    w = findfirst(==(s), (:current_point, :delta_point, :current_gradient, :delta_gradient, :step_direction))
    w === nothing && throw(ArgumentError("read_field: unknown field $s"))
    out = Vector{T}(undef, length(getfield(o, :current_point)))
    check(ccall((:dzo_lbfgs_read, libdzo), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), getfield(o, :handle), w - 1, 0, out))
    return out
end
function _lb_hist(o::LBFGSOptimizer{T}, rho::Bool) where {T}   # (ccall needs a literal symbol name)
    buf = Vector{Cdouble}(undef, 64); cnt = Ref{Cint}(0)
    if rho
        check(ccall((:dzo_lbfgs_get_rho, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ref{Cint}), getfield(o, :handle), buf, 64, cnt))
    else
        check(ccall((:dzo_lbfgs_get_alpha, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ref{Cint}), getfield(o, :handle), buf, 64, cnt))
    end
    return T.(buf[1:cnt[]])
end

# Public fields of the reference (src/DZOptimization.jl:321-344); scalars come back boxed in
# 0-dim arrays like the reference's Array{T,0}, so `opt.is_stuck[]` reads the same.
function Base.getproperty(o::LBFGSOptimizer{T}, s::Symbol) where {T}
    s in (:is_stuck, :has_terminated, :has_converged) && return fill(_lb_i(o, 0) != 0)
    s === :iteration_count && return fill(Int(_lb_i(o, 1)))
    s === :current_objective_value && return fill(T(_lb_s(o, 0)))
    s === :delta_objective_value && return fill(T(_lb_s(o, 1)))
    s === :delta_point && return _lb_p(o, 1)
    s === :current_gradient && return _lb_p(o, 2)
    s === :delta_gradient && return _lb_p(o, 3)
    s === :step_direction && return _lb_p(o, 4)
    s === :delta_point_history && return [_lb_p(o, 5, i - 1) for i in 1:_lb_i(o, 4)]
    s === :delta_gradient_history && return [_lb_p(o, 6, i - 1) for i in 1:_lb_i(o, 4)]
    s === :rho_history && return _lb_hist(o, true)
    s === :alpha_history && return _lb_hist(o, false)
    s === :last_step_length && return fill(T(_lb_s(o, 2)))
    s === :history_resets && return Int(_lb_i(o, 8))
    s === :descent_resets && return Int(_lb_i(o, 9))
    # informational (include/dzo.h): steps taken as one sweep, their rejected first trials / second passes, and how the
    # history lives in HBM (0 slabs, 1 tiles of pairs, 2 tiles of points; tile arrangement 1 tile-major, 2 stream-major)
    s === :single_pass_steps && return Int(_lb_i(o, 11))
    s === :single_pass_rejections && return Int(_lb_i(o, 12))
    s === :single_pass_retries && return Int(_lb_i(o, 13))
    s === :ring_layout && return Int(_lb_i(o, 14))
    s === :tile_arrangement && return Int(_lb_i(o, 15))
    s === :pass_recomputes_gradients && return _lb_i(o, 16) != 0
    s === :pass_register_sets && return Int(_lb_i(o, 17))
    return getfield(o, s)
end

################################################################################ dense BFGS

"""`BFGSOptimizer(objective, gradient!, [constraint!,] x0, initial_step_length)` (README.md:33-36,
legacy/DZOptimization.jl:753-766).  Copies `x0` (:769)."""
mutable struct BFGSOptimizer{T,F,G,C}
    handle::Ptr{Cvoid}
    objective_function::F
    gradient_function!::G
    constraint_function!::C
    n::Int
    keep::Any
    function BFGSOptimizer{T,F,G,C}(h, f, g, c, n, keep) where {T,F,G,C}
        return finalizer(o -> ccall((:dzo_bfgs_destroy, libdzo), Cint, (Ptr{Cvoid},), getfield(o, :handle)),
                         new{T,F,G,C}(h, f, g, c, n, keep))
    end
end
BFGSOptimizer(objective, gradient!, x0::HipVector, step::Real) = BFGSOptimizer(objective, gradient!, nothing, x0, step)
# README.md:33-36 passes a host array (`rand(2)`): upload it (the optimizer copies x0 anyway, :769)
BFGSOptimizer(objective, gradient!, x0::Array{T}, step::Real) where {T<:Union{Float32,Float64}} =
    BFGSOptimizer(objective, gradient!, nothing, HipVector(x0), step)
BFGSOptimizer(objective, gradient!, constraint!, x0::Array{T}, step::Real) where {T<:Union{Float32,Float64}} =
    BFGSOptimizer(objective, gradient!, constraint!, HipVector(x0), step)
function BFGSOptimizer(objective::F, gradient!::G, constraint!::C, x0::HipVector{T}, step::Real) where {T,F,G,C}
    ensure_init()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if objective isa BuiltinProblem && constraint! === nothing
        check(ccall((:dzo_bfgs_create_problem, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), objective.handle, x0.ptr, step, h))
        return BFGSOptimizer{T,F,G,C}(h[], objective, gradient!, constraint!, length(x0), nothing)
    end
    cb = Ref(Callbacks{C,F,G,T}(constraint!, objective, gradient!, length(x0)))
    cf = constraint! === nothing ? C_NULL : @cfunction(_c_constraint, Cint, (Ptr{Cvoid}, Ptr{Cvoid}))
    check(ccall((:dzo_bfgs_create_callbacks, libdzo), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}),
                @cfunction(_c_objective, Cdouble, (Ptr{Cvoid}, Ptr{Cvoid})),
                @cfunction(_c_gradient, Cvoid, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid})), cf,
                pointer_from_objref(cb), length(x0), dtype_code(T), x0.ptr, step, h))
    return BFGSOptimizer{T,F,G,C}(h[], objective, gradient!, constraint!, length(x0), cb)
end

"""`BFGSOptimizer(T, opt, objective_in_T)` (legacy/DZOptimization.jl:812-862): the same optimizer state in
another precision (fp32 warm start, fp64 finish); objective value, gradient and `H*g` are recomputed
in `T` (:825-836).  `objective_in_T` is the built-in problem of the target precision."""
function BFGSOptimizer(::Type{T}, opt::BFGSOptimizer, objective::BuiltinProblem{T}) where {T}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_bfgs_convert_problem, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                getfield(opt, :handle), objective.handle, h))
    return BFGSOptimizer{T,typeof(objective),Nothing,Nothing}(h[], objective, nothing, nothing, getfield(opt, :n), nothing)
end

"""`update_inverse_hessian!(H, step_length, d, delta_gradient, scratch)` (legacy/DZOptimization.jl:864-889) on
raw device arrays (`H` column-major n*n); rescales `d` in place (:874)."""
function update_inverse_hessian!(H::HipVector{T}, step_length::Real, d::HipVector{T}, delta_gradient::HipVector{T},
                                 scratch::HipVector{T}) where {T}
    check(ccall((:dzo_bfgs_update, libdzo), Cint,
                (Int64, Cint, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                length(d), dtype_code(T), H.ptr, step_length, d.ptr, delta_gradient.ptr, scratch.ptr, C_NULL, C_NULL))
    return H
end

"""`reset_inverse_hessian!(opt)`: `approximate_inverse_hessian = I`, `next_step_direction = gradient`
(the reset of legacy/DZOptimization.jl:981-986, `identity_matrix!` :712-720)."""
reset_inverse_hessian!(opt::BFGSOptimizer) = (check(ccall((:dzo_bfgs_reset, libdzo), Cint, (Ptr{Cvoid},), getfield(opt, :handle))); opt)

"""`step!(opt)` (legacy/DZOptimization.jl:891-994)."""
step!(opt::BFGSOptimizer) = (check(ccall((:dzo_bfgs_step, libdzo), Cint, (Ptr{Cvoid},), getfield(opt, :handle))); opt)

_bf_i(o, w) = (v = Ref{Int64}(0); check(ccall((:dzo_bfgs_get_i, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}), getfield(o, :handle), w, v)); v[])
_bf_s(o, w) = (v = Ref{Cdouble}(0); check(ccall((:dzo_bfgs_get_s, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), getfield(o, :handle), w, v)); v[])
function _bf_p(o::BFGSOptimizer{T}, w, len=getfield(o, :n)) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_bfgs_get_ptr, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), getfield(o, :handle), w, p))
    return HipVector{T}(p[], len)
end
function Base.getproperty(o::BFGSOptimizer{T}, s::Symbol) where {T}
    s in (:has_converged, :has_terminated, :is_stuck) && return fill(_bf_i(o, 0) != 0)
    s === :iteration_count && return fill(Int(_bf_i(o, 1)))
    s === :last_step_type && return fill(Int(_bf_i(o, 3)))       # 0 Null, 1 GradientDescent, 2 BFGS (:727-731)
    s === :current_objective_value && return fill(T(_bf_s(o, 0)))
    s === :last_step_length && return fill(T(_bf_s(o, 1)))
    s === :current_point && return _bf_p(o, 0)
    s === :delta_point && return _bf_p(o, 1)
    s === :current_gradient && return _bf_p(o, 2)
    s === :delta_gradient && return _bf_p(o, 3)
    s === :next_step_direction && return _bf_p(o, 4)
    s === :approximate_inverse_hessian && return _bf_p(o, 5, getfield(o, :n)^2)   # column-major n*n
    return getfield(o, s)
end

"""`install_state!(opt; x, g, H, d, f, last_step_length, iteration_count, last_step_type, delta_point, delta_gradient)`:
overwrite the optimizer's state from host arrays ("save/load data in the middle of optimization",
README.md:11); `H` column-major n*n.  The device arrays are the state itself (`opt.current_point` etc. are
views of them); the host-side fields go through `dzo_bfgs_set_s / set_i`."""
function install_state!(opt::BFGSOptimizer{T}; x=nothing, g=nothing, H=nothing, d=nothing, f=nothing, last_step_length=nothing,
                        iteration_count=nothing, last_step_type=nothing, delta_point=nothing, delta_gradient=nothing) where {T}
    up(dst::HipVector, src) = (h = Array{T}(vec(src));
                               check(ccall((:dzo_memcpy_h2d, libdzo), Cint, (Ptr{Cvoid}, Ptr{T}, Int64), dst.ptr, h, sizeof(h))))
    x === nothing || up(opt.current_point, x)
    g === nothing || up(opt.current_gradient, g)
    H === nothing || up(opt.approximate_inverse_hessian, H)
    d === nothing || up(opt.next_step_direction, d)
    delta_point === nothing || up(opt.delta_point, delta_point)
    delta_gradient === nothing || up(opt.delta_gradient, delta_gradient)
    f === nothing || check(ccall((:dzo_bfgs_set_s, libdzo), Cint, (Ptr{Cvoid}, Cint, Cdouble), getfield(opt, :handle), 0, f))
    last_step_length === nothing || check(ccall((:dzo_bfgs_set_s, libdzo), Cint, (Ptr{Cvoid}, Cint, Cdouble), getfield(opt, :handle), 1, last_step_length))
    iteration_count === nothing || check(ccall((:dzo_bfgs_set_i, libdzo), Cint, (Ptr{Cvoid}, Cint, Int64), getfield(opt, :handle), 1, iteration_count))
    last_step_type === nothing || check(ccall((:dzo_bfgs_set_i, libdzo), Cint, (Ptr{Cvoid}, Cint, Int64), getfield(opt, :handle), 3, last_step_type))
    return opt
end

################################################################################ legacy gradient descent

"""`GradientDescentOptimizer(objective, gradient!, QuadraticLineSearch(), x0, step)`
(legacy/DZOptimization.jl:377-390); built-in objectives only in this thin binding.  The handle is a
BFGS-family handle, so the `BFGSOptimizer` getters apply (no `approximate_inverse_hessian`)."""
struct QuadraticLineSearch; max_increases::Int; end
QuadraticLineSearch() = QuadraticLineSearch(0)
function GradientDescentOptimizer(objective::BuiltinProblem{T}, ::Any, ls::QuadraticLineSearch, x0::HipVector{T}, step::Real) where {T}
    ensure_init()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_gd_create_problem, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), objective.handle, x0.ptr, step, h))
    check(ccall((:dzo_bfgs_set_max_increases, libdzo), Cint, (Ptr{Cvoid}, Cint), h[], ls.max_increases))
    return GDHandle{T}(h[], length(x0), objective)
end
mutable struct GDHandle{T}
    handle::Ptr{Cvoid}
    n::Int
    keep::Any
    GDHandle{T}(h, n, keep) where {T} =
        finalizer(o -> ccall((:dzo_bfgs_destroy, libdzo), Cint, (Ptr{Cvoid},), getfield(o, :handle)), new{T}(h, n, keep))
end
step!(opt::GDHandle) = (check(ccall((:dzo_gd_step, libdzo), Cint, (Ptr{Cvoid},), getfield(opt, :handle))); opt)
function Base.getproperty(o::GDHandle{T}, s::Symbol) where {T}      # the BFGS-family getters (include/dzo.h)
    s in (:has_terminated, :has_converged, :is_stuck) && return fill(_bf_i(o, 0) != 0)
    s === :iteration_count && return fill(Int(_bf_i(o, 1)))
    s === :current_objective_value && return fill(T(_bf_s(o, 0)))
    s === :last_step_length && return fill(T(_bf_s(o, 1)))
    if s in (:current_point, :delta_point, :current_gradient, :delta_gradient, :next_step_direction)
        w = findfirst(==(s), (:current_point, :delta_point, :current_gradient, :delta_gradient, :next_step_direction)) - 1
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dzo_bfgs_get_ptr, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), getfield(o, :handle), w, p))
        return HipVector{T}(p[], getfield(o, :n))
    end
    return getfield(o, s)
end

################################################################################ AdGD

"""`AdGDOptimizer(constraint!, objective, gradient!, x0, initial_step_length)` (src/DZOptimization.jl:245-272) or the full
form with `f0, g0` (:203-243).  The struct of :179-200: the three callbacks are fields, the state fields are read
through `getproperty`.  `objective` may be a `BuiltinProblem` (then the step runs on the device's fused pass).  Aliases
`x0` as `current_point` and `g0` as `current_gradient` (:232, :235)."""
mutable struct AdGDOptimizer{T,A,C,F,G} <: AbstractOptimizer{T,A}
    handle::Ptr{Cvoid}
    constraint_function!::C
    objective_function::F
    gradient_function!::G
    current_point::A
    keep::Any
    AdGDOptimizer{T,A,C,F,G}(h, c, f, g, x, keep) where {T,A,C,F,G} =
        finalizer(o -> ccall((:dzo_adgd_destroy, libdzo), Cint, (Ptr{Cvoid},), getfield(o, :handle)), new{T,A,C,F,G}(h, c, f, g, x, keep))
end
# full form (:203-243)
function AdGDOptimizer(constraint!::C, objective::F, gradient!::G, x0::HipVector{T}, f0::Real, g0::HipVector{T},
                       initial_step_length::Real) where {T,C,F,G}
    ensure_init()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_adgd_create, libdzo), Cint, (Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ref{Ptr{Cvoid}}),
                length(x0), dtype_code(T), x0.ptr, g0.ptr, f0, initial_step_length, h))
    cb = Ref(Callbacks{C,F,G,T}(constraint!, objective, gradient!, length(x0)))
    cf = constraint! === nothing ? C_NULL : @cfunction(_c_constraint, Cint, (Ptr{Cvoid}, Ptr{Cvoid}))
    check(ccall((:dzo_adgd_set_callbacks, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                h[], cf, @cfunction(_c_objective, Cdouble, (Ptr{Cvoid}, Ptr{Cvoid})),
                @cfunction(_c_gradient, Cvoid, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid})), pointer_from_objref(cb)))
    return AdGDOptimizer{T,HipVector{T},C,F,G}(h[], constraint!, objective, gradient!, x0, (cb, g0))
end
# short form (:245-272)
function AdGDOptimizer(constraint!::C, objective::F, gradient!::G, x0::HipVector{T}, initial_step_length::Real) where {T,C,F,G}
    ensure_init()
    if objective isa BuiltinProblem && constraint! === nothing
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dzo_adgd_create_problem, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}), objective.handle, x0.ptr, initial_step_length, h))
        return AdGDOptimizer{T,HipVector{T},C,F,G}(h[], constraint!, objective, gradient!, x0, objective)
    end
    if constraint! !== nothing
        @assert constraint!(x0)                                                    # :256-258
    end
    f0 = objective(x0)                                                             # :260
    g0 = similar(x0)                                                               # :262
    objective isa BuiltinProblem ? gradient!(objective, g0, x0) : gradient!(g0, x0)   # :265
    return AdGDOptimizer(constraint!, objective, objective isa BuiltinProblem ? ((g, x) -> gradient!(objective, g, x)) : gradient!,
                         x0, f0, g0, initial_step_length)
end
step!(opt::AdGDOptimizer) = (check(ccall((:dzo_adgd_step, libdzo), Cint, (Ptr{Cvoid},), getfield(opt, :handle))); opt)
_ad_i(o, w) = (v = Ref{Int64}(0); check(ccall((:dzo_adgd_get_i, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}), getfield(o, :handle), w, v)); v[])
_ad_s(o, w) = (v = Ref{Cdouble}(0); check(ccall((:dzo_adgd_get_s, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), getfield(o, :handle), w, v)); v[])
# public fields of src/DZOptimization.jl:179-200
function Base.getproperty(o::AdGDOptimizer{T,A,C,F,G}, s::Symbol) where {T,A,C,F,G}
    s in (:is_stuck, :has_terminated, :has_converged) && return fill(_ad_i(o, 0) != 0)
    s === :iteration_count && return fill(Int(_ad_i(o, 1)))
    s === :current_objective_value && return fill(T(_ad_s(o, 0)))
    s === :delta_objective_value && return fill(T(_ad_s(o, 1)))
    s === :current_step_size && return fill(T(_ad_s(o, 2)))
    s === :previous_step_size && return fill(T(_ad_s(o, 3)))
    if s in (:delta_point, :current_gradient, :delta_gradient)
        # re-fetched on every access: delta_gradient moves between two buffers from step to step, and the
        # library re-reads what it cached about an array whose pointer was handed out (include/dzo.h)
        w = findfirst(==(s), (:current_point, :delta_point, :current_gradient, :delta_gradient)) - 1
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dzo_adgd_get_ptr, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), getfield(o, :handle), w, p))
        return HipVector{T}(p[], length(getfield(o, :current_point)))
    end
    return getfield(o, s)
end

function read_field(o::AdGDOptimizer{T}, s::Symbol) where {T}       # (dzo_adgd_read: the same, for AdGD's four vector fields)
    w = findfirst(==(s), (:current_point, :delta_point, :current_gradient, :delta_gradient))
    w === nothing && throw(ArgumentError("read_field: unknown field $s"))
    out = Vector{T}(undef, length(getfield(o, :current_point)))
    check(ccall((:dzo_adgd_read, libdzo), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), getfield(o, :handle), w - 1, out))
    return out
end

################################################################################ batched dense BFGS

"""`BatchedBFGSOptimizer(kind, x0, n, initial_step_length)`: `length(x0) / n` independent `BFGSOptimizer`s
(legacy/DZOptimization.jl:733-994 each) on one device, `x0` instance-major; `kind` = 1 for the chained
Rosenbrock objective.  `step!(b, k)` runs k synchronous steps of every live instance;
`count_active(b) == 0` is the shard's "everyone has_terminated".  One shard per process / GPU."""
mutable struct BatchedBFGSOptimizer{T}
    handle::Ptr{Cvoid}
    batch::Int
    n::Int
    function BatchedBFGSOptimizer(kind::Union{Integer,BuiltinProblem}, x0::HipVector{T}, n::Integer, step::Real; device::Union{Nothing,Integer}=nothing,
                                  matrices::Union{Nothing,HipVector{T}}=nothing) where {T}
        ensure_init()
        h = Ref{Ptr{Cvoid}}(C_NULL)
        batch = div(length(x0), n)
        if matrices !== nothing         # one symmetric n x n matrix per instance (instance-major); the caller keeps `matrices` alive
            check(ccall((:dzo_bfgs_batch_create_problem_matrices, libdzo), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cdouble, Cint, Ref{Ptr{Cvoid}}),
                        kind.handle, batch, matrices.ptr, n * n, x0.ptr, step, device === nothing ? -1 : device, h))
        elseif kind isa BuiltinProblem      # objective, shared A of the quadratic and the decorators from the problem handle
            check(ccall((:dzo_bfgs_batch_create_problem, libdzo), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cdouble, Cint, Ref{Ptr{Cvoid}}),
                        kind.handle, batch, x0.ptr, step, device === nothing ? -1 : device, h))
        elseif device === nothing
            check(ccall((:dzo_bfgs_batch_create, libdzo), Cint, (Cint, Int64, Int64, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}),
                        kind, batch, n, dtype_code(T), x0.ptr, step, h))
        else   # a shard on an explicit GPU (x0 must live there: `init(device); x0 = HipVector(...)`)
            check(ccall((:dzo_bfgs_batch_create_on, libdzo), Cint, (Cint, Cint, Int64, Int64, Cint, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}),
                        device, kind, batch, n, dtype_code(T), x0.ptr, step, h))
        end
        return finalizer(o -> ccall((:dzo_bfgs_batch_destroy, libdzo), Cint, (Ptr{Cvoid},), getfield(o, :handle)),
                         new{T}(h[], batch, n))
    end
end

"""`ShardComm(devices)`: RCCL communicator over the GPUs of one node for ONE host process
(`ncclCommInitAll` behind `dzo_comm_init_all`); `ShardComm(id, nranks, rank)` joins a
process-per-GPU communicator whose 128-byte `id = ShardComm.unique_id()` rank 0 created.  The only
collective of the design is the convergence flag: `all_done(comm, shards)` is true when every instance of
every shard `has_terminated` (`dzo_bfgs_batch_all_done`: local counts + one 4-byte all-reduce(MIN) over
xGMI).  "run multiple optimizers in parallel" (README.md:12):

    comm   = ShardComm(0:7)
    shards = [ (init(d); BatchedBFGSOptimizer(1, HipVector(x0[d]), 256, 1.0; device=d)) for d in 0:7 ]
    while !all_done(comm, shards); foreach(s -> step!(s, 10), shards); end
"""
mutable struct ShardComm
    handle::Ptr{Cvoid}
    ShardComm(h::Ptr{Cvoid}) = finalizer(c -> ccall((:dzo_comm_destroy, libdzo), Cint, (Ptr{Cvoid},), c.handle), new(h))
end
function ShardComm(devices::AbstractVector{<:Integer})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    devs = Cint.(collect(devices))
    check(ccall((:dzo_comm_init_all, libdzo), Cint, (Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}), devs, length(devs), h))
    _initialised[] = true
    return ShardComm(h[])
end
function unique_id()
    ensure_init()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:dzo_comm_unique_id, libdzo), Cint, (Ptr{UInt8},), id))
    return id
end
function ShardComm(id::Vector{UInt8}, nranks::Integer, rank::Integer)
    ensure_init()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_comm_init_rank, libdzo), Cint, (Ptr{UInt8}, Cint, Cint, Ref{Ptr{Cvoid}}), id, nranks, rank, h))
    return ShardComm(h[])
end
function all_done(comm::Union{ShardComm,Nothing}, shards::AbstractVector{<:BatchedBFGSOptimizer})
    hs = Ptr{Cvoid}[s.handle for s in shards]
    r = Ref{Cint}(0)
    check(ccall((:dzo_bfgs_batch_all_done, libdzo), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cint, Ref{Cint}),
                comm === nothing ? C_NULL : comm.handle, hs, length(hs), r))
    return r[] != 0
end
"""`allreduce_min(comm, local_flags)`: the raw 4-byte collective (one flag per local rank)."""
function allreduce_min(comm::ShardComm, local_flags::AbstractVector{<:Integer})
    fl = Cint.(collect(local_flags)); r = Ref{Cint}(0)
    check(ccall((:dzo_flag_allreduce_min_n, libdzo), Cint, (Ptr{Cvoid}, Ptr{Cint}, Cint, Ref{Cint}), comm.handle, fl, length(fl), r))
    return Int(r[])
end
step!(b::BatchedBFGSOptimizer, steps::Integer=1) =
    (check(ccall((:dzo_bfgs_batch_step, libdzo), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}), b.handle, steps, C_NULL)); b)
"""`set_max_increases!(b, k)`: `QuadraticLineSearch.max_increases` (legacy/DZOptimization.jl:181-188) of every instance."""
set_max_increases!(b::BatchedBFGSOptimizer, k::Integer) =
    (check(ccall((:dzo_bfgs_batch_set_max_increases, libdzo), Cint, (Ptr{Cvoid}, Cint), b.handle, k)); b)
function count_active(b::BatchedBFGSOptimizer)
    v = Ref{Int64}(0)
    check(ccall((:dzo_bfgs_batch_count_active, libdzo), Cint, (Ptr{Cvoid}, Ref{Int64}), b.handle, v))
    return Int(v[])
end
"""`current_point(b)`: the batch x n points (instance-major) as one device vector."""
function current_point(b::BatchedBFGSOptimizer{T}) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dzo_bfgs_batch_get_ptr, libdzo), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), b.handle, 0, p))
    return HipVector{T}(p[], b.batch * b.n)
end

################################################################################ LineSearchEvaluator

"""`LineSearchEvaluator(constraint!, objective, gradient!, x, f, d, overlap)` and its call
`ev(t, compute_gradient)` -> `(trial_objective_value, improvement_ratio, slope_ratio)`
(src/DZOptimization.jl:12-62, :65-92).  `trial_point` / `trial_gradient` are the evaluator's own."""
struct LineSearchEvaluator{T,CB}
    callbacks::CB
    current_point::HipVector{T}
    current_objective_value::T
    step_direction::HipVector{T}
    overlap::T
    trial_point::HipVector{T}
    trial_gradient::HipVector{T}
end
function LineSearchEvaluator(constraint!::C, objective::F, gradient!::G, x::HipVector{T}, f::Real, d::HipVector{T},
                             overlap::Real) where {T,C,F,G}
    cb = Ref(Callbacks{C,F,G,T}(constraint!, objective, gradient!, length(x)))
    return LineSearchEvaluator{T,typeof(cb)}(cb, x, T(f), d, T(overlap), similar(x), similar(x))
end
function (ev::LineSearchEvaluator{T})(step_size::Real, compute_gradient::Bool) where {T}
    cb = ev.callbacks
    cf = cb[].constraint! === nothing ? C_NULL : @cfunction(_c_constraint, Cint, (Ptr{Cvoid}, Ptr{Cvoid}))
    f, imp, slope = Ref{Cdouble}(0), Ref{Cdouble}(0), Ref{Cdouble}(0)
    GC.@preserve cb check(ccall((:dzo_line_search_eval, libdzo), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Cdouble, Ptr{Cvoid}, Cdouble, Cdouble,
                 Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cdouble}),
                cf, @cfunction(_c_objective, Cdouble, (Ptr{Cvoid}, Ptr{Cvoid})),
                @cfunction(_c_gradient, Cvoid, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid})), pointer_from_objref(cb),
                length(ev.current_point), dtype_code(T), ev.current_point.ptr, ev.current_objective_value,
                ev.step_direction.ptr, ev.overlap, step_size, compute_gradient, ev.trial_point.ptr,
                ev.trial_gradient.ptr, f, imp, slope))
    return T(f[]), T(imp[]), T(slope[])
end

end # module
