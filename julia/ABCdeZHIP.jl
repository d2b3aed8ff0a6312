# ABCdeZHIP.jl -- thin `ccall` shim that puts libabcdez_hip.so behind ABCdeZ.jl's own
# entry points.  NOT EXECUTED IN THE BUILD CONTAINER (no Julia there); kept small and
# literal so a maintainer can check it against src/abcdez_smc.jl / src/abcdez_mc.jl line by
# line.  The Python package `abcdez_amd` is the tested host; this file is the same host
# loop in the reference's language.
#
# Usage:
#     using ABCdeZ, Distributions
#     include("ABCdeZHIP.jl"); using .ABCdeZHIP
#     r = abcdesmc!(Normal(0, sqrt(10)), Normal1D(3.0), 0.3, nothing; nparticles = 1_000_000)
#
# Dispatch: `abcdesmc!` / `abcdemc!` get a new method on a `DeviceSimulator` in the `dist!`
# position; every other `dist!` (an ordinary closure) keeps hitting ABCdeZ.jl's CPU methods.
module ABCdeZHIP

using ABCdeZ, Distributions
import ABCdeZ: abcdesmc!, abcdemc!

export DeviceSimulator, Normal1D, MVNormalSim, abcdesmc!, abcdemc!

const LIB = get(ENV, "ABCDEZ_HIP_LIB", joinpath(@__DIR__, "..", "abcdez.jl_amd", "lib", "libabcdez_hip.so"))

# ---- include/abcdez_spec.h: abz_prior_dim (48 B), abz_model ---------------------------------
struct AbzPriorDim
    family::Int32; discrete::Int32
    p0::Float64; p1::Float64; c0::Float64; c1::Float64; reserved::Float64
end
struct AbzModel
    d::Int32; ld::Int32; sim_id::Int32; abck::Int32
    seed::UInt64
    n_data::Int32; n_blob::Int32          # n_blob = 0: blobs off (abcdez_blob_eval not bound by this shim)
    sim_p::NTuple{8,Float64}
    data::Ptr{Float64}
    prior::NTuple{64,AbzPriorDim}
end

abstract type DeviceSimulator end
struct Normal1D <: DeviceSimulator; data::Float64; sigma::Float64; end       # ABZ_SIM_NORMAL1D
Normal1D(data) = Normal1D(data, 1.0)
struct MVNormalSim <: DeviceSimulator; y::Vector{Float64}; sigma::Float64; end  # ABZ_SIM_MVN
MVNormalSim(y) = MVNormalSim(collect(Float64, y), 1.0)
simid(::Normal1D) = Int32(0);  simid(::MVNormalSim) = Int32(1)
simparams(s::Normal1D) = (s.sigma,);  simparams(s::MVNormalSim) = (s.sigma,)
simdata(s::Normal1D) = [s.data];  simdata(s::MVNormalSim) = s.y

factors(p::Factored) = collect(p.p)
factors(p::UnivariateDistribution) = [p]
const PAD = AbzPriorDim(0, 0, 0.0, 0.0, 0.0, 0.0, 0.0)
descriptor(p::Normal) = AbzPriorDim(1, 0, p.μ, p.σ, -log(p.σ) - 0.5 * log(2π), 1 / p.σ, 0.0)
descriptor(p::Uniform) = AbzPriorDim(2, 0, p.a, p.b, -log(p.b - p.a), 0.0, 0.0)
descriptor(p::DiscreteUniform) = AbzPriorDim(3, 1, p.a, p.b, -log(p.b - p.a + 1), 0.0, 0.0)
kernelid(::Type{ABCdeZ.Indicator0toϵ}) = Int32(0);  kernelid(::Type{ABCdeZ.IndicatorStrict0toϵ}) = Int32(1)
kernelid(::Type{ABCdeZ.Epa0toϵ}) = Int32(2);        kernelid(::Type{ABCdeZ.EpaStrict0toϵ}) = Int32(3)

check(rc) = rc == 0 || error(unsafe_string(ccall((:abcdez_last_error, LIB), Cstring, ())))

mutable struct Engine
    ctx::Ptr{Cvoid}; N::Int; ld::Int; d::Int
    theta::Vector{Ptr{Cvoid}}; logpi::Vector{Ptr{Cvoid}}; delta::Vector{Ptr{Cvoid}}   # ping-pong, smc:337-350
    wns::Ptr{Cvoid}; alive::Ptr{Cvoid}; alive_idx::Ptr{Cvoid}; arank::Ptr{Cvoid}; inds::Ptr{Cvoid}
    order::Ptr{Cvoid}; sorted::Ptr{Cvoid}
    cur::Int; sweep::UInt32; draw::UInt32; n_alive::Int; dead_synced::Bool
end

devalloc(bytes) = (p = Ref{Ptr{Cvoid}}(); check(ccall((:abcdez_dev_alloc, LIB), Cint, (Csize_t, Ptr{Ptr{Cvoid}}), bytes, p)); p[])

function Engine(prior, sim::DeviceSimulator, ABCk, seed::Integer, N::Int)
    fs = factors(prior); d = length(fs); ld = nextpow(2, d)
    data = simdata(sim); sp = simparams(sim)
    m = AbzModel(d, ld, simid(sim), kernelid(ABCk), UInt64(seed), length(data), 0,
                 ntuple(i -> i <= length(sp) ? Float64(sp[i]) : 0.0, 8), pointer(data),
                 ntuple(k -> k <= d ? descriptor(fs[k]) : PAD, 64))
    ctx = Ref{Ptr{Cvoid}}()
    GC.@preserve data check(ccall((:abcdez_ctx_create, LIB), Cint, (Ref{AbzModel}, Cint, Ptr{Ptr{Cvoid}}), m, 0, ctx))
    Engine(ctx[], N, ld, d, [devalloc(8N * ld) for _ in 1:2], [devalloc(8N) for _ in 1:2], [devalloc(8N) for _ in 1:2],
           devalloc(8N), devalloc(N), devalloc(4N), devalloc(4N), devalloc(4N), devalloc(4N), devalloc(8N),
           1, 0, 0, N, true)
end
other(e) = 3 - e.cur

# ---- one ccall per reference function (include/abcdez_hip.h) --------------------------------
init!(e) = check(ccall((:abcdez_init, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
                       e.ctx, e.theta[e.cur], e.logpi[e.cur], e.delta[e.cur], 0, e.N))                    # init.jl:2-22
function reset_weights!(e)                                                                                # smc:266-270: Wns = 1/N, alive = true
    w = fill(1.0 / e.N, e.N); a = ones(UInt8, e.N)
    check(ccall((:abcdez_memcpy_h2d, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, e.wns, w, 8e.N))
    check(ccall((:abcdez_memcpy_h2d, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, e.alive, a, e.N))
    e.n_alive = e.N; e.dead_synced = true
end
function extrema_dev(e)                                                                                   # smc:286,364
    lo = Ref(0.0); hi = Ref(0.0)
    check(ccall((:abcdez_extrema, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Float64}), e.ctx, e.delta[e.cur], e.N, lo, hi))
    (lo[], hi[])
end
function quantile_alive(e, α)                                                                             # smc:301
    q = Ref(0.0)
    check(ccall((:abcdez_quantile_alive, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Ref{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
                e.ctx, e.delta[e.cur], e.alive, e.N, e.n_alive, α, q, C_NULL, C_NULL)); q[]
end
function reweight!(e, ϵ_old, ϵ_new)                                                                       # smc:59-83, :308-311, :323
    wnorm = Ref(0.0); ess = Ref(0.0); na = Ref(Int64(0))
    check(ccall((:abcdez_smc_reweight, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Float64, Ref{Float64}, Ref{Float64}, Ref{Int64}),
                e.ctx, e.delta[e.cur], e.wns, e.alive, e.N, ϵ_old, ϵ_new, wnorm, ess, na))
    e.n_alive = na[]; e.dead_synced = false
    (wnorm[], ess[], Int(na[]))
end
function get_ess(e)                                                                                       # smc:8
    ess = Ref(0.0)
    check(ccall((:abcdez_get_ess, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}), e.ctx, e.wns, e.N, ess)); ess[]
end
function resample!(e)                                                                                     # smc:85-104
    check(ccall((:abcdez_wsample_stratified, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt32, Ptr{Cvoid}), e.ctx, e.wns, e.N, e.draw, e.inds))
    e.draw += 1; o = other(e)
    check(ccall((:abcdez_smc_resample_gather, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                e.ctx, e.inds, e.N, 0, e.N, e.theta[e.cur], e.logpi[e.cur], e.delta[e.cur], e.theta[o], e.logpi[o], e.delta[o], e.wns, e.alive))
    e.cur = o; e.n_alive = e.N; e.dead_synced = true
end
function compact!(e)                                                                                      # smc:121,125
    na = Ref(Int64(0))
    check(ccall((:abcdez_alive_compact, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}), e.ctx, e.alive, e.N, e.alive_idx, e.arank, na))
    e.n_alive = na[]
end
function smc_swarm!(e, ϵ, γ0, γσ)                                                                         # smc:106-153 + :337-340
    nacc = Ref(Int64(0)); nsim = Ref(Int64(0)); o = other(e)
    check(ccall((:abcdez_smc_swarm, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Float64, Float64, Float64, Int64, Int64, Cint, Ptr{Cvoid}, UInt32, Ref{Int64}, Ref{Int64}),
                e.ctx, e.alive_idx, e.arank, e.n_alive, 0, e.n_alive, e.theta[e.cur], e.logpi[e.cur], e.delta[e.cur],
                e.theta[o], e.logpi[o], e.delta[o], ϵ, γ0, γσ, 0, e.N, (!e.dead_synced && e.n_alive < e.N) ? 1 : 0, C_NULL, e.sweep, nacc, nsim))
    e.sweep += 1; e.dead_synced = true; e.cur = o
    (Int(nacc[]), Int(nsim[]))
end
function download(e)
    th = Matrix{Float64}(undef, e.ld, e.N); pushed = devalloc(8 * e.N * e.ld)
    check(ccall((:abcdez_push_p, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}), e.ctx, e.theta[e.cur], e.N, pushed))   # types.jl:20-23
    check(ccall((:abcdez_memcpy_d2h, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, th, pushed, sizeof(th)))
    Δ = Vector{Float64}(undef, e.N); W = Vector{Float64}(undef, e.N)
    check(ccall((:abcdez_memcpy_d2h, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, Δ, e.delta[e.cur], 8e.N))
    check(ccall((:abcdez_memcpy_d2h, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, W, e.wns, 8e.N))
    ccall((:abcdez_dev_free, LIB), Cint, (Ptr{Cvoid},), pushed)
    P = e.d == 1 ? th[1, :] : [Tuple(th[1:e.d, i]) for i in 1:e.N]
    (P, W, Δ)
end

# ---- abcdesmc!: the host loop of src/abcdez_smc.jl:215-394, each ★ call replaced by one ccall ----
function abcdesmc!(prior, dist!::DeviceSimulator, ϵ_target, varexternal;
                   nparticles::Int=100, α=0.95, δess=0.5, nsims_max::Int=10^7, Kmcmc::Int=3, Kmcmc_min=1.0,
                   ABCk=ABCdeZ.IndicatorStrict0toϵ, facc_stop=0.0, facc_min=0.0, facc_tune=0.975,
                   verbose::Bool=true, verboseout::Bool=true, rng::Integer=1, parallel::Bool=true)
    0.0 ≤ α < 1.0 || error("α must be in 0 <= α < 1")                                  # smc:223-235
    0.0 ≤ δess ≤ 1.0 || error("δess must be in 0 <= δess <= 1")
    0.0 ≤ facc_stop ≤ 1.0 || error("facc_stop must be in 0 <= facc_stop <= 1")
    0.0 ≤ facc_min ≤ 1.0 || error("facc_min must be in 0 <= facc_min <= 1")
    0.0 ≤ facc_tune ≤ 1.0 || error("facc_tune must be in 0 <= facc_tune <= 1")
    0.0 ≤ ϵ_target || error("ϵ_target must be non-negative")
    1 ≤ Kmcmc || error("Kmcmc must be at least 1")
    0.0 ≤ Kmcmc_min ≤ Inf || error("Kmcmc_min must be in 0 <= Kmcmc_min <= Inf")
    1 ≤ nsims_max || error("nsims_max must be at least 1")
    Kmcmc_min > facc_min || @warn("Kmcmc_min should be larger than facc_min")
    nparticles_min = ceil(Int, 3 * length(prior) / (min(α, δess)))
    nparticles_min ≤ nparticles || error("nparticles must be at least $(nparticles_min)")

    e = Engine(prior, dist!, ABCk, rng, nparticles)
    init!(e)                                                                            # smc:242-252
    reset_weights!(e)                                                                   # smc:266-270
    ϵ = Inf; ϵ_k = Inf; logZ = 0.0; ess = 0.0; nsims = 0; facc = 1.0; Ki = Kmcmc        # smc:255-276
    ess_min = nparticles * δess
    γ0 = 2.38 / sqrt(2 * length(prior)); γσ = 1e-5                                      # smc:280-281
    ϵs = [ϵ]; ranges_ϵ = [extrema_dev(e)]; logZs = [logZ]; esss = [get_ess(e)]; faccs = [facc]; γ0s = [γ0]; Kmcmcs = [Ki]
    iters = 0
    while true                                                                          # smc:295
        iters += 1
        ϵ = max(min(quantile_alive(e, α), ϵ), ϵ_target)                                 # smc:301
        ABCk(ϵ)
        wnorm, ess, n_alive = reweight!(e, ϵ_k, ϵ)                                      # smc:305-311
        logZ += log(wnorm)                                                              # smc:315
        naccs = 0; Ki = Kmcmc
        facc < facc_min && (γ0 *= facc_tune)                                            # smc:320
        if n_alive > 0 && ess < ess_min                                                 # smc:323-326
            resample!(e); ess = get_ess(e); n_alive = nparticles
        end
        if n_alive ≥ 3
            compact!(e)
            for i in 1:Kmcmc                                                            # smc:336-353
                nacc, nsim = smc_swarm!(e, ϵ, γ0, γσ)
                naccs += nacc; nsims += nsim
                (naccs / n_alive ≥ Kmcmc_min) && (Ki = i; break)                        # smc:352
            end
        end
        facc = naccs / (n_alive * Ki); ϵ_k = ϵ                                          # smc:357-360
        push!(ϵs, ϵ); push!(ranges_ϵ, extrema_dev(e)); push!(logZs, logZ); push!(esss, ess)
        push!(faccs, facc); push!(γ0s, γ0); push!(Kmcmcs, Ki)
        verbose && (@info "Finished run:" iteration = iters nsim = nsims ϵ = ϵ ess = ess facc = facc logZ = logZ)
        n_alive ≥ 3 || (@warn("No alive particles"); break)                             # smc:375
        (ϵ ≤ ϵ_target || nsims ≥ nsims_max || facc < facc_stop) && break                # smc:376
    end
    P, Wns, Δs = download(e)                                                            # smc:382
    ccall((:abcdez_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), e.ctx)
    blobs = fill(nothing, nparticles)
    verboseout ? (P = P, Wns = Wns, C = Δs, ϵ = ϵ, logZ = logZ, blobs = blobs, ϵs = ϵs, ranges_ϵ = ranges_ϵ,
                  logZs = logZs, esss = esss, faccs = faccs, γ0s = γ0s, Kmcmcs = Kmcmcs) :
                 (P = P, Wns = Wns, C = Δs, ϵ = ϵ, logZ = logZ, blobs = blobs)         # smc:388-393
end

# ---- abcdemc!: the host loop of src/abcdez_mc.jl:102-172 --------------------------------------------
function count_gt(e, thr)                                                                                 # mc:133,156
    c = Ref(Int64(0))
    check(ccall((:abcdez_count_gt, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Ref{Int64}), e.ctx, e.delta[e.cur], e.N, thr, c)); Int(c[])
end
rank_prepare!(e) = check(ccall((:abcdez_mc_rank_prepare, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                               e.ctx, e.delta[e.cur], e.N, e.order, e.sorted))                            # mc:23
function mc_swarm!(e, ϵ_pop, ϵ_target, γ0, γσ)                                                            # mc:5-61 + :140-143
    nsim = Ref(Int64(0)); o = other(e)
    check(ccall((:abcdez_mc_swarm, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Float64, Float64, Float64, Float64, Int64, Int64, UInt32, Ref{Int64}),
                e.ctx, e.order, e.sorted, e.N, e.theta[e.cur], e.logpi[e.cur], e.delta[e.cur], e.theta[o], e.logpi[o], e.delta[o],
                ϵ_pop, ϵ_target, γ0, γσ, 0, e.N, e.sweep, nsim))
    e.sweep += 1; e.cur = o
    Int(nsim[])
end

function abcdemc!(prior, dist!::DeviceSimulator, ϵ_target, varexternal;
                  nparticles::Int=50, generations::Int=20, verbose=true, rng::Integer=1, parallel::Bool=true)
    α = 0.0                                                                              # mc:107
    0.0 ≤ ϵ_target || error("ϵ_target must be non-negative")
    5 ≤ nparticles || error("nparticles must be at least 5")
    1 ≤ generations || error("generations must be at least 1")
    e = Engine(prior, dist!, ABCdeZ.IndicatorStrict0toϵ, rng, nparticles)
    init!(e)                                                                             # mc:117-125
    nsims = 0; γ0 = 2.38 / sqrt(2 * length(prior)); γσ = 1e-5; iters = 0                 # mc:128-131
    complete = 1 - count_gt(e, ϵ_target) / nparticles                                    # mc:133
    while iters < generations                                                            # mc:134
        iters += 1
        ϵ_l, ϵ_h = extrema_dev(e)                                                        # mc:146
        ϵ_pop = max(ϵ_target, ϵ_l + α * (ϵ_h - ϵ_l))                                     # mc:147
        ϵ_h > ϵ_target && rank_prepare!(e)
        nsims += mc_swarm!(e, ϵ_pop, ϵ_target, γ0, γσ)                                   # mc:149
        ncomplete = 1 - count_gt(e, ϵ_target) / nparticles                               # mc:156
        verbose && (ncomplete != complete || complete >= (nparticles - 1) / nparticles) &&
            (@info "Finished run:" completion = ncomplete nsim = nsims range_ϵ = extrema_dev(e))
        complete = ncomplete
    end
    conv = extrema_dev(e)[2] <= ϵ_target                                                 # mc:163
    P, _, Δs = download(e)                                                               # mc:166
    ccall((:abcdez_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), e.ctx)
    (P = P, C = Δs, reached_ϵ = conv, blobs = fill(nothing, nparticles))                # mc:171
end

end # module
