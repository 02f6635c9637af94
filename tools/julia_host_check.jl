# Runs where a `julia` exists (the build container and the GPU boxes seen so far have none: tests/test_julia_host.py and
# bench.py record that).  Drives dzoptimization.jl_amd/julia/DZOptimizationAMD.jl -- the ccall host module with the
# reference's constructors / fields / step! -- on the golden L-BFGS trajectory and prints what it got, one line per step:
#     julia tools/julia_host_check.jl <x0.txt> <m> <steps>
# x0.txt: one Float64 per line.  Output lines:  "lbfgs <k> <f>"  (built-in problem),  "lbfgs_cb <k> <f>"  (the same run with the
# objective and gradient handed over as Julia CLOSURES, src/DZOptimization.jl:400-407),  "adgd_cb <k> <f>"  (AdGDOptimizer with
# closures, :245-272),  "rate <step!()/s>"  (config 3 through this host module, when a 4th argument n is given).
include(joinpath(@__DIR__, "..", "dzoptimization.jl_amd", "julia", "DZOptimizationAMD.jl"))
using .DZOptimizationAMD
using Printf

x0 = [parse(Float64, l) for l in readlines(ARGS[1]) if !isempty(strip(l))]
m, steps = parse(Int, ARGS[2]), parse(Int, ARGS[3])
n = length(x0)
prob = RosenbrockChain(n)

opt = LBFGSOptimizer(nothing, prob, nothing, HipVector(x0), 1.0, m)
for k in 1:steps
    step!(opt)
    @printf("lbfgs %d %.17g\n", k, opt.current_objective_value[])
end

f_cb = x -> prob(x)
g_cb! = (g, x) -> DZOptimizationAMD.gradient!(prob, g, x)
opt2 = LBFGSOptimizer(nothing, f_cb, g_cb!, HipVector(x0), 1.0, m)
@assert opt2 isa DZOptimizationAMD.AbstractOptimizer{Float64,HipVector{Float64}}
for k in 1:steps
    step!(opt2)
    @printf("lbfgs_cb %d %.17g\n", k, opt2.current_objective_value[])
end

ad = AdGDOptimizer(nothing, f_cb, g_cb!, HipVector(x0), 0.1)
@assert ad isa DZOptimizationAMD.AbstractOptimizer{Float64,HipVector{Float64}}
@assert ad.objective_function === f_cb && ad.gradient_function! === g_cb! && ad.constraint_function! === nothing
for k in 1:steps
    step!(ad)
    @printf("adgd_cb %d %.17g\n", k, ad.current_objective_value[])
end

if length(ARGS) >= 4
    nn = parse(Int, ARGS[4])
    xs = [isodd(i) ? -1.2 : 1.0 for i in 1:nn]
    big = LBFGSOptimizer(nothing, RosenbrockChain(nn), nothing, HipVector(xs), 1.0, 20)
    for _ in 1:25; step!(big); end
    synchronize()
    t0 = time_ns()
    for _ in 1:50; step!(big); end
    synchronize()
    @printf("rate %.2f\n", 50 / ((time_ns() - t0) * 1e-9))
end
