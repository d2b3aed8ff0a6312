# ABCdeZHIP.jl -- thin `ccall` shim that puts libabcdez_hip.so behind ABCdeZ.jl's own
# entry points.  NOT EXECUTED IN THE BUILD CONTAINER (no Julia there); kept small and
# literal so a maintainer can check it against src/abcdez_smc.jl / src/abcdez_mc.jl line by
# line.  The Python package `abcdez_amd` is the tested host; this file is the same host
# loop in the reference's language.  Its call sequence -- every entry point, argument order
# and type it binds -- is pinned by tests/test_gpu_shim_sequence.py (a raw-ctypes
# transliteration, no torch), its struct layout by abcdez_abi_layout() (asserted below and
# in tests/test_host_api.py).
#
# Usage:
#     using ABCdeZ, Distributions
#     include("ABCdeZHIP.jl"); using .ABCdeZHIP
#     r = abcdesmc!(Normal(0, sqrt(10)), Normal1D(3.0), 0.3, nothing; nparticles = 1_000_000)
#
# Dispatch: `abcdesmc!` / `abcdemc!` get a new method on a `DeviceSimulator` in the `dist!`
# position; every other `dist!` (an ordinary closure) keeps hitting ABCdeZ.jl's CPU methods.
#
# Multi-GPU (one Julia process per GPU): the collectives are behind the C ABI too (RCCL over xGMI on the library's own
# stream, csrc/abz_comm.hip).  The host only carries the 128-byte rendezvous id from rank 0 to the others, e.g.
#     id = rank == 0 ? comm_unique_id() : Vector{UInt8}(undef, 128);  MPI.Bcast!(id, 0, MPI.COMM_WORLD)
#     r = abcdesmc!(prior, sim, ϵ, nothing; nparticles = N, rng = 1, comm = (id = id, rank = rank, world = world))
# A host without RCCL (or with ranks that share GPUs) hands the library its own transport instead -- the library stages every
# piece through page-locked host memory and calls back (abcdez_comm_init_host; no rendezvous id):
#     ag!(buf::Vector{UInt8}, piece::Int) = MPI.Allgather!(MPI.IN_PLACE, MPI.UBuffer(buf, piece), MPI.COMM_WORLD)   # in place, rank order
#     r = abcdesmc!(prior, sim, ϵ, nothing; nparticles = N, rng = 1, comm = (rank = rank, world = world, allgather! = ag!))
# Either way every rank returns the same result, bit for bit what one GPU returns (random numbers are keyed by position).  Per sweep one
# byte per alive position crosses the fabric, per generation the distances; abcdez_smc_sweeps_sharded is the whole sharded
# generation in one ccall (abcdez_amd/engine.py drives the same entry points and is the tested reference for their order).
module ABCdeZHIP

using ABCdeZ, Distributions, LinearAlgebra, Random
import ABCdeZ: abcdesmc!, abcdemc!

export DeviceSimulator, Normal1D, MVNormalSim, DiracSquare, Quad2D, Mixture01, NormalTimesDU, WienerRMS,
       LotkaVolterraRK4, Socks, UserSimulator, abcdesmc!, abcdemc!, comm_unique_id

const LIB = get(ENV, "ABCDEZ_HIP_LIB", joinpath(@__DIR__, "..", "abcdez.jl_amd", "lib", "libabcdez_hip.so"))

# ---- include/abcdez_spec.h: abz_prior_dim (48 B), abz_model ---------------------------------
struct AbzPriorDim
    family::Int32; discrete::Int32
    p0::Float64; p1::Float64; c0::Float64; c1::Float64; reserved::Float64
end
struct AbzModel
    d::Int32; ld::Int32; sim_id::Int32; abck::Int32
    seed::UInt64
    n_data::Int32; n_blob::Int32          # n_blob = 0: blobs off
    sim_p::NTuple{8,Float64}
    data::Ptr{Float64}
    prior::NTuple{256,AbzPriorDim}
    mv::Ptr{Float64}                      # C_NULL, or [μ | L⁻¹ | L] of an MvNormal prior (include/abcdez_spec.h)
    ext::Ptr{Float64}                     # C_NULL, or the records of truncated(...) / MixtureModel factors (ABZ_PRIOR_TRUNCATED / _MIXTURE)
    n_ext::Int32; reserved0::Int32
end
# the library reports sizeof / offsetof of both structs; a mismatch is a build mix-up, not a run-time condition
const MIN_VERSION = 610        # abz_model.ext (wrapper priors), abcdez_comm_init_host, abcdez_smc_generation_packed (include/abcdez_hip.h)
function check_abi()
    # an older library would link -- C has no signature check -- and misread arguments added since
    v = ccall((:abcdez_version, LIB), Cint, ())
    v >= MIN_VERSION || error("ABCdeZHIP: libabcdez_hip.so reports version $v, this shim needs >= $MIN_VERSION (rebuild the library)")
    ccall((:abcdez_rng_rounds, LIB), Cint, ()) == 10 || @warn("libabcdez_hip.so was built with a non-default Philox round count: results differ from a default build's")
    lay = Vector{Int32}(undef, 32)
    n = ccall((:abcdez_abi_layout, LIB), Cint, (Ptr{Int32}, Cint), lay, length(lay))
    mine = Int32[sizeof(AbzPriorDim), fieldoffset.(AbzPriorDim, 1:7)...,
                 sizeof(AbzModel), fieldoffset.(AbzModel, 1:14)...]
    (n == length(mine) && lay[1:n] == mine) || error("ABCdeZHIP: struct layout differs from libabcdez_hip.so (abcdez_abi_layout)")
end

# ---- device simulators = dist!(θ, ve) on the GPU (include/abcdez_spec.h, ABZ_SIM_*) -------------------------------
abstract type DeviceSimulator end
struct Normal1D <: DeviceSimulator; data::Float64; sigma::Float64; blobs::Bool; end                 # ABZ_SIM_NORMAL1D
Normal1D(data; sigma=1.0, blobs=false) = Normal1D(data, sigma, blobs)
struct MVNormalSim <: DeviceSimulator; y::Vector{Float64}; sigma::Float64; blobs::Bool; end          # ABZ_SIM_MVN
MVNormalSim(y; sigma=1.0, blobs=false) = MVNormalSim(collect(Float64, y), sigma, blobs)
struct DiracSquare <: DeviceSimulator; target::Float64; blobs::Bool; end                             # test/runtests.jl:495-497
DiracSquare(target; blobs=false) = DiracSquare(target, blobs)
struct Quad2D <: DeviceSimulator; p_inf::Float64; blobs::Bool; end                                   # test/runtests.jl:603,614
Quad2D(p_inf; blobs=false) = Quad2D(p_inf, blobs)
struct Mixture01 <: DeviceSimulator; data::Float64; blobs::Bool; end                                 # test/runtests.jl:582-583
Mixture01(data; blobs=false) = Mixture01(data, blobs)
struct NormalTimesDU <: DeviceSimulator; data::Float64; blobs::Bool; end                             # test/runtests.jl:524-525
NormalTimesDU(data; blobs=false) = NormalTimesDU(data, blobs)
struct WienerRMS <: DeviceSimulator; tdata::Vector{Float64}; blobs::Bool; end                        # test/runtests.jl:537-546
WienerRMS(tdata; blobs=false) = WienerRMS(collect(Float64, tdata), blobs)
struct LotkaVolterraRK4 <: DeviceSimulator                                                           # BASELINE.json configs[3]
    obs::Vector{Float64}; x0::Float64; y0::Float64; dt::Float64; steps_per_obs::Int; noise::Float64; blobs::Bool
end
LotkaVolterraRK4(obs; x0=1.0, y0=0.5, dt=0.01, steps_per_obs=100, noise=0.1, blobs=false) =
    LotkaVolterraRK4(collect(Float64, obs), x0, y0, dt, steps_per_obs, noise, blobs)
struct Socks <: DeviceSimulator; pairs::Float64; odd::Float64; n_picked::Int; blobs::Bool; end       # test/runtests.jl:427-437
Socks(pairs, odd; n_picked=11, blobs=false) = Socks(pairs, odd, n_picked, blobs)
# a device function given as HIP source text (abcdez_ctx_create_user): defines abz_user_dist (length(prior) <= 16), abz_user_dist_lanes
# (17 .. 256: the row spread over the lanes of a wavefront) or the staged abz_user_round (ABZ_USER_ROUNDS: proposals whose running lower
# bound of the distance has passed ϵ leave the simulation early), and abz_user_blob when n_blob > 0 (INTEGRATION.md section 1)
struct UserSimulator <: DeviceSimulator; source::String; params::Vector{Float64}; data::Vector{Float64}; n_blob::Int; end
UserSimulator(source; params=Float64[], data=Float64[], n_blob=0) = UserSimulator(source, collect(Float64, params), collect(Float64, data), n_blob)

simid(::Normal1D) = Int32(0); simid(::MVNormalSim) = Int32(1); simid(::DiracSquare) = Int32(2); simid(::Quad2D) = Int32(3)
simid(::Mixture01) = Int32(4); simid(::NormalTimesDU) = Int32(5); simid(::WienerRMS) = Int32(6); simid(::LotkaVolterraRK4) = Int32(7)
simid(::Socks) = Int32(8); simid(::UserSimulator) = Int32(9)
simparams(s::Normal1D) = (s.sigma,);              simdata(s::Normal1D) = [s.data]
simparams(s::MVNormalSim) = (s.sigma,);           simdata(s::MVNormalSim) = s.y
simparams(s::DiracSquare) = (s.target,);          simdata(s::DiracSquare) = Float64[]
simparams(s::Quad2D) = (s.p_inf,);                simdata(s::Quad2D) = Float64[]
simparams(s::Mixture01) = (s.data,);              simdata(s::Mixture01) = Float64[]
simparams(s::NormalTimesDU) = (s.data,);          simdata(s::NormalTimesDU) = Float64[]
simparams(s::WienerRMS) = ();                     simdata(s::WienerRMS) = s.tdata
simparams(s::LotkaVolterraRK4) = (s.x0, s.y0, s.dt, Float64(s.steps_per_obs), s.noise); simdata(s::LotkaVolterraRK4) = s.obs
simparams(s::Socks) = (s.pairs, s.odd, Float64(s.n_picked)); simdata(s::Socks) = Float64[]
simparams(s::UserSimulator) = Tuple(s.params);    simdata(s::UserSimulator) = s.data
# doubles per blob = the simulated data behind a distance (abz_sim_blob_size)
blobsize(s::Union{Normal1D,DiracSquare,Mixture01,NormalTimesDU}, d) = 1
blobsize(s::Union{Quad2D,Socks}, d) = 2
blobsize(s::MVNormalSim, d) = d
blobsize(s::WienerRMS, d) = length(s.tdata)
blobsize(s::LotkaVolterraRK4, d) = length(s.obs)
nblob(s::UserSimulator, d) = s.n_blob
nblob(s::DeviceSimulator, d) = s.blobs ? blobsize(s, d) : 0

factors(p::Factored) = collect(p.p)
factors(p::UnivariateDistribution) = [p]
# product_distribution([...]) in the prior position (test/runtests.jl:45): the same descriptors; push_p broadcasts the WHOLE
# distribution over the vector (types.jl:21), so every component follows the product's value support
factors(p::Distributions.Product) = collect(p.v)
# MvNormal(μ, Σ) in the prior position -- not a product: Σ = L Lᵀ, θ = μ + L z; the device takes the per-dimension Normal tree
# over the whitened components z = L⁻¹(θ - μ): descriptors Normal(0, 1) with c0 = -log L_kk - log(2π)/2, maps in AbzModel.mv
struct WhitenedNormal; c0::Float64; end
descriptor(p::WhitenedNormal) = AbzPriorDim(1, 0, 0.0, 1.0, p.c0, 1.0, 0.0)
function factors(p::Distributions.AbstractMvNormal)
    L = Matrix(cholesky(Symmetric(Matrix(cov(p)))).L)
    [WhitenedNormal(-log(L[k, k]) - 0.5 * log(2π)) for k in 1:length(p)]
end
mvmaps(p, ld) = Float64[]
function mvmaps(p::Distributions.AbstractMvNormal, ld)
    d = length(p); L = Matrix(cholesky(Symmetric(Matrix(cov(p)))).L); W = inv(LowerTriangular(L))
    out = zeros(ld + 2ld^2); out[1:d] = mean(p)
    for k in 1:d, m in 1:k                                   # row-major ld x ld blocks, zero above the diagonal
        out[ld + (k - 1) * ld + m] = W[k, m]; out[ld + ld^2 + (k - 1) * ld + m] = L[k, m]
    end
    out
end
pushrule(p, k, dflt) = dflt
pushrule(p::Distributions.Product, k, dflt) = Int32(p isa DiscreteDistribution ? 1 : 0)
const PAD = AbzPriorDim(0, 0, 0.0, 0.0, 0.0, 0.0, 0.0)
const lgam = Distributions.SpecialFunctions.loggamma
descriptor(p::Normal) = AbzPriorDim(1, 0, p.μ, p.σ, -log(p.σ) - 0.5 * log(2π), 1 / p.σ, 0.0)
descriptor(p::Uniform) = AbzPriorDim(2, 0, p.a, p.b, -log(p.b - p.a), 0.0, 0.0)
descriptor(p::DiscreteUniform) = AbzPriorDim(3, 1, p.a, p.b, -log(p.b - p.a + 1), 0.0, 0.0)
descriptor(p::Beta) = AbzPriorDim(4, 0, p.α, p.β, -(lgam(p.α) + lgam(p.β) - lgam(p.α + p.β)), 0.0, 0.0)
descriptor(p::NegativeBinomial) = AbzPriorDim(5, 1, p.r, p.p, p.r * log(p.p) - lgam(p.r), p.p < 1 ? log1p(-p.p) : -Inf, 0.0)
# the further univariate families of Distributions.jl the device knows (include/abcdez_spec.h, ABZ_PRIOR_EXPONENTIAL ...), in
# Distributions' own parametrisations; (family, discrete, p0, p1, c0, c1, reserved) as laid out there
descriptor(p::Exponential) = (θ = scale(p); AbzPriorDim(6, 0, θ, 0.0, -log(θ), 1 / θ, 0.0))
descriptor(p::Gamma) = ((α, θ) = params(p); AbzPriorDim(7, 0, α, θ, -lgam(α) - α * log(θ), 1 / θ, 0.0))
descriptor(p::Chisq) = descriptor(Gamma(dof(p) / 2, 2.0))
descriptor(p::Erlang) = descriptor(Gamma(Float64(shape(p)), scale(p)))
descriptor(p::LogNormal) = ((μ, σ) = params(p); AbzPriorDim(8, 0, μ, σ, -log(σ) - 0.5 * log(2π), 1 / σ, 0.0))
descriptor(p::Cauchy) = ((μ, σ) = params(p); AbzPriorDim(9, 0, μ, σ, -log(π * σ), 1 / σ, 0.0))
descriptor(p::Laplace) = ((μ, θ) = params(p); AbzPriorDim(10, 0, μ, θ, -log(2θ), 1 / θ, 0.0))
descriptor(p::Weibull) = ((α, θ) = params(p); AbzPriorDim(11, 0, α, θ, log(α / θ), 1 / θ, 0.0))
descriptor(p::Rayleigh) = descriptor(Weibull(2.0, sqrt(2.0) * scale(p)))
descriptor(p::InverseGamma) = ((α, θ) = params(p); AbzPriorDim(12, 0, α, θ, α * log(θ) - lgam(α), 0.0, 0.0))
function descriptor(p::Truncated{<:Normal})                  # truncated(Normal(μ, σ), lo, hi); either bound may be missing
    (μ, σ) = params(p.untruncated)
    lo = p.lower === nothing ? -Inf : Float64(p.lower); hi = p.upper === nothing ? Inf : Float64(p.upper)
    mass = cdf(p.untruncated, hi) - cdf(p.untruncated, lo)
    mass >= 0.01 || error("truncated(Normal): [lo, hi] holds $mass of the Normal's mass; the device draws the initial " *
                          "population by rejection from the parent and needs at least 0.01")
    AbzPriorDim(13, 0, μ, σ, -log(σ) - 0.5 * log(2π) - log(mass), lo, hi)
end
descriptor(p::Logistic) = ((μ, θ) = params(p); AbzPriorDim(14, 0, μ, θ, -log(θ), 1 / θ, 0.0))
descriptor(p::TDist) = (ν = dof(p); AbzPriorDim(15, 0, ν, (ν + 1) / 2, lgam((ν + 1) / 2) - lgam(ν / 2) - 0.5 * log(ν * π), 1 / ν, 0.0))
descriptor(p::Pareto) = ((α, θ) = params(p); AbzPriorDim(16, 0, α, θ, log(α) + α * log(θ), 0.0, 0.0))
function descriptor(p::Poisson)
    λ = rate(p); 0 < λ <= 700 || error("Poisson: the device's inversion sampler needs 0 < λ <= 700")
    AbzPriorDim(17, 1, λ, 0.0, -λ, log(λ), 0.0)
end
function descriptor(p::Binomial)
    (n, q) = params(p)
    0 < q < 1 || error("Binomial: a degenerate prior (p = 0 or 1) has no device descriptor")
    n * log1p(-min(q, 1 - q)) > -700 || error("Binomial: n too large for the device's inversion sampler")
    AbzPriorDim(18, 1, Float64(n), q, lgam(n + 1.0) + n * log1p(-q), log(q) - log1p(-q), 0.0)
end
descriptor(p::Geometric) = descriptor(NegativeBinomial(1.0, succprob(p)))
# wrappers: truncated(d, lo, hi) of any other parent and MixtureModel of univariate components keep their records in the model's ext
# table (include/abcdez_spec.h, ABZ_PRIOR_TRUNCATED / ABZ_PRIOR_MIXTURE); the descriptor points at them by offset (0-based, in doubles)
asdoubles(q::AbzPriorDim) = Float64[q.family, q.discrete, q.p0, q.p1, q.c0, q.c1, q.reserved]
isbase(p) = !(p isa Truncated && !(p.untruncated isa Normal)) && !(p isa MixtureModel) && !(p isa Distributions.AffineDistribution)
function descriptor!(ext::Vector{Float64}, p::Truncated)
    p.untruncated isa Normal && return descriptor(p)          # keeps its own family (13)
    isbase(p.untruncated) || error("truncated(): the parent must be one of the base univariate families")
    lo = p.lower === nothing ? -Inf : Float64(p.lower); hi = p.upper === nothing ? Inf : Float64(p.upper)
    mass = exp(p.logtp)                                       # Distributions' own log(cdf(hi) - cdf(lo⁻))
    mass >= 0.01 || error("truncated(): [lo, hi] holds $mass of the parent's mass; the device draws the initial population by " *
                          "rejection from the parent and needs at least 0.01")
    par = descriptor(p.untruncated); off = length(ext)
    append!(ext, [lo, hi, p.logtp]); append!(ext, asdoubles(par))
    AbzPriorDim(19, par.discrete, Float64(off), 0.0, 0.0, 0.0, 0.0)
end
function descriptor!(ext::Vector{Float64}, p::MixtureModel)
    cs = components(p); w = probs(p); K = length(cs)
    1 <= K <= 16 || error("MixtureModel: 1 .. 16 components")
    all(isbase, cs) || error("MixtureModel: components must be base univariate families (no nesting)")
    ds = descriptor.(cs)
    all(q -> q.discrete == ds[1].discrete, ds) || error("MixtureModel: the components must be all continuous or all discrete")
    off = length(ext); cum = 0.0
    for j in 1:K
        cum = j == K ? 1.0 : cum + w[j]
        append!(ext, [log(w[j]), cum]); append!(ext, asdoubles(ds[j]))
    end
    AbzPriorDim(20, ds[1].discrete, Float64(K), Float64(off), 0.0, 0.0, 0.0)
end
# μ + σ * d (Distributions.LocationScale / AffineDistribution) of a continuous base family, σ > 0: family 21
function descriptor!(ext::Vector{Float64}, p::Distributions.AffineDistribution)
    μ, σ, ρ = Float64(p.μ), Float64(p.σ), p.ρ
    (σ > 0 && isfinite(σ) && isfinite(μ)) || error("μ + σ * d: need a finite μ and σ > 0")
    (isbase(ρ) && !(ρ isa Distributions.AffineDistribution) && ρ isa ContinuousUnivariateDistribution) ||
        error("μ + σ * d: the parent must be one of the continuous base univariate families")
    off = length(ext)
    append!(ext, [μ, σ, 1 / σ, log(σ)]); append!(ext, asdoubles(descriptor(ρ)))
    AbzPriorDim(21, 0, Float64(off), 0.0, 0.0, 0.0, 0.0)
end
descriptor!(ext::Vector{Float64}, p) = descriptor(p)
descriptor(p) = error("no device descriptor for a prior of type $(typeof(p)): the univariate families of include/abcdez_spec.h " *
                      "(ABZ_PRIOR_*), truncated(...) and MixtureModel of them, Factored / product_distribution of those and MvNormal are supported")
kernelid(::Type{ABCdeZ.Indicator0toϵ}) = Int32(0);  kernelid(::Type{ABCdeZ.IndicatorStrict0toϵ}) = Int32(1)
kernelid(::Type{ABCdeZ.Epa0toϵ}) = Int32(2);        kernelid(::Type{ABCdeZ.EpaStrict0toϵ}) = Int32(3)

check(rc) = rc == 0 || error(unsafe_string(ccall((:abcdez_last_error, LIB), Cstring, ())))

# ---- the population the reference driver owns (smc:242-275), device-resident, PACKED layout (include/abcdez_hip.h):
# two row slots per position + one bit per position naming the current one; (logπ, Δ, stamps) ping-pong at resamplings;
# abcdemc uses slot[1] / slot[2] as the reference's (θs, nθs) double buffer
mutable struct Engine
    ctx::Ptr{Cvoid}; N::Int; ld::Int; d::Int; nb::Int
    slot::Vector{Ptr{Cvoid}}; logpi::Vector{Ptr{Cvoid}}; delta::Vector{Ptr{Cvoid}}; bits::Vector{Ptr{Cvoid}}
    stamp::Vector{Ptr{Cvoid}}
    wns::Ptr{Cvoid}; alive::Ptr{Cvoid}; inds::Ptr{Cvoid}; order::Ptr{Cvoid}; sorted::Ptr{Cvoid}; cnt::Ptr{Cvoid}
    cur::Int; bc::Int; sweep::UInt32; draw::UInt32; n_alive::Int; n_prev::Int
    rank::Int; world::Int; flags::Ptr{Cvoid}        # multi-GPU: this process's rank, the number of ranks, one flag byte per position
    transport::Any                                  # host transport (abcdez_comm_init_host): kept alive as long as the context
end

# the host-supplied transport of a sharded run: `allgather!(buf, piece)` gathers IN PLACE over `world` pieces of `piece` bytes of
# host memory (this rank's piece at buf[rank * piece + 1 : (rank + 1) * piece] on entry), e.g. MPI.Allgather!(MPI.IN_PLACE, ...)
mutable struct HostTransport
    allgather!::Function
    world::Int
end
function host_allgather_cb(user::Ptr{Cvoid}, buf::Ptr{UInt8}, piece::Int64)::Cint
    t = unsafe_pointer_to_objref(user)::HostTransport
    try
        t.allgather!(unsafe_wrap(Array, buf, Int(piece) * t.world; own = false), Int(piece))
        return Cint(0)
    catch err                                       # nothing may unwind through the C frames: the library reports status -4
        @error "ABCdeZHIP host transport: all-gather failed" err
        return Cint(1)
    end
end

devalloc(bytes) = (p = Ref{Ptr{Cvoid}}(); check(ccall((:abcdez_dev_alloc, LIB), Cint, (Csize_t, Ptr{Ptr{Cvoid}}), bytes, p)); p[])
devfree(p) = p == C_NULL || ccall((:abcdez_dev_free, LIB), Cint, (Ptr{Cvoid},), p)
h2d(e, dst, src, bytes) = check(ccall((:abcdez_memcpy_h2d, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, dst, src, bytes))
d2h(e, dst, src, bytes) = check(ccall((:abcdez_memcpy_d2h, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, dst, src, bytes))
# the bulk result download goes through page-locked staging memory: the copy engine then runs at the link's rate, and the arrays
# of a result travel back to back behind one wait (include/abcdez_hip.h, abcdez_host_alloc / abcdez_memcpy_d2h_async)
hostalloc(bytes) = (p = Ref{Ptr{Cvoid}}(); check(ccall((:abcdez_host_alloc, LIB), Cint, (Csize_t, Ptr{Ptr{Cvoid}}), bytes, p)); p[])
hostfree(p) = p == C_NULL || ccall((:abcdez_host_free, LIB), Cint, (Ptr{Cvoid},), p)
d2h_async(e, dst, src, bytes) = check(ccall((:abcdez_memcpy_d2h_async, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), e.ctx, dst, src, bytes))

# `rng` of the reference signatures (src/abcdez_smc.jl:220, src/abcdez_mc.jl:104: `rng=Random.default_rng()`): the device stream
# is a counter-based Philox4x32-10 keyed by 64 bits.  An AbstractRNG -- the reference's default included -- gives the key with one
# draw, so seeding that rng makes the run reproducible exactly as it does for the CPU methods; an Integer is used as the key itself
# (the Python host's convention, convenient for bit-for-bit comparisons with it).
philox_key(rng::Integer) = rng % UInt64
philox_key(rng::AbstractRNG) = rand(rng, UInt64)

# rank 0 of a multi-GPU run: the rendezvous id every rank passes to abcdesmc!(...; comm = (id = ..., rank = ..., world = ...))
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:abcdez_comm_unique_id, LIB), Cint, (Ptr{Cvoid}, Csize_t), id, 128)); id
end

function Engine(prior, sim::DeviceSimulator, ABCk, seed::Integer, N::Int; comm=nothing)
    check_abi()
    fs = factors(prior); d = length(fs); ld = nextpow(2, d)
    data = simdata(sim); sp = simparams(sim); nb = nblob(sim, d); mv = mvmaps(prior, ld)
    ext = Float64[]                                           # records of truncated(...) / MixtureModel factors
    qs = [descriptor!(ext, fs[k]) for k in 1:d]
    m = AbzModel(d, ld, simid(sim), kernelid(ABCk), UInt64(seed), length(data), nb,
                 ntuple(i -> i <= length(sp) ? Float64(sp[i]) : 0.0, 8), isempty(data) ? C_NULL : pointer(data),
                 ntuple(k -> k <= d ? (q = qs[k]; AbzPriorDim(q.family, pushrule(prior, k, q.discrete), q.p0, q.p1, q.c0, q.c1, q.reserved)) : PAD, 256),
                 isempty(mv) ? C_NULL : pointer(mv), isempty(ext) ? C_NULL : pointer(ext), Int32(length(ext)), Int32(0))
    ctx = Ref{Ptr{Cvoid}}()
    GC.@preserve data mv ext begin
        if sim isa UserSimulator
            check(ccall((:abcdez_ctx_create_user, LIB), Cint, (Ref{AbzModel}, Cstring, Cint, Ptr{Ptr{Cvoid}}), m, sim.source, 0, ctx))
        else
            check(ccall((:abcdez_ctx_create, LIB), Cint, (Ref{AbzModel}, Cint, Ptr{Ptr{Cvoid}}), m, 0, ctx))
        end
    end
    check(ccall((:abcdez_ctx_reserve, LIB), Cint, (Ptr{Cvoid}, Int64), ctx[], N))
    nw = cld(N, 32)
    rank, world = comm === nothing ? (0, 1) : (Int(comm.rank), Int(comm.world))
    N % world == 0 || error("nparticles must be divisible by the number of ranks")
    transport = nothing
    if comm !== nothing && hasproperty(comm, :allgather!)     # the host's own transport (MPI ...): the library stages through pinned memory
        transport = HostTransport(getproperty(comm, :allgather!), world)
        cb = @cfunction(host_allgather_cb, Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64))
        # no all-reduce callback: the library all-gathers the words and reduces them in rank order (the same bits on every rank)
        check(ccall((:abcdez_comm_init_host, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    ctx[], rank, world, cb, C_NULL, pointer_from_objref(transport)))
    elseif comm !== nothing                          # every rank: join the RCCL communicator on this context's device
        check(ccall((:abcdez_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint, Cint), ctx[], comm.id, length(comm.id), rank, world))
    end
    Np = N + 64 * world                              # sharded runs exchange chunks of the per-position arrays: room for `world` of them
    e = Engine(ctx[], N, ld, d, nb, [devalloc(8N * ld) for _ in 1:2], [devalloc(8N) for _ in 1:2], [devalloc(8Np) for _ in 1:2],
               [devalloc(4nw) for _ in 1:2], nb > 0 ? [devalloc(8N) for _ in 1:2] : Ptr{Cvoid}[],
               devalloc(8N), devalloc(N), devalloc(4N), devalloc(4N), devalloc(8N), devalloc(4N), 1, 1, 0, 0, N, N,
               rank, world, world > 1 ? devalloc(Np) : C_NULL, transport)
    z = zeros(UInt32, nw)                       # every position's current row is slot 1
    h2d(e, e.bits[1], z, 4nw); h2d(e, e.bits[2], z, 4nw)
    finalizer(free!, e)
end
# releases the population and the context; safe to call twice (ADVICE r1: every run used to leak its arrays)
function free!(e::Engine)
    e.ctx == C_NULL && return
    foreach(devfree, vcat(e.slot, e.logpi, e.delta, e.bits, e.stamp, [e.wns, e.alive, e.inds, e.order, e.sorted, e.cnt, e.flags]))
    ccall((:abcdez_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), e.ctx)
    e.ctx = C_NULL
end
other(e) = 3 - e.cur
bind_stamps!(e) = e.nb > 0 && check(ccall((:abcdez_ctx_set_stamps, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), e.ctx, e.stamp[e.cur], e.stamp[other(e)]))

# ---- one ccall per reference function (include/abcdez_hip.h) --------------------------------
allgather!(e, buf, piece_bytes) = check(ccall((:abcdez_comm_allgather, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), e.ctx, buf, piece_bytes))
function init!(e)                                                                                         # init.jl:2-22
    bind_stamps!(e)
    nl = e.N ÷ e.world                               # this rank draws the particles [rank nl, (rank + 1) nl) ...
    check(ccall((:abcdez_init, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64),
                e.ctx, e.slot[1], e.logpi[e.cur], e.delta[e.cur], e.rank * nl, nl))
    if e.world > 1                                   # ... and every rank ends up with the whole population (any particle may be a donor)
        allgather!(e, e.slot[1], 8nl * e.ld); allgather!(e, e.logpi[e.cur], 8nl); allgather!(e, e.delta[e.cur], 8nl)
        e.nb > 0 && allgather!(e, e.stamp[e.cur], 8nl)
    end
end
function reset_weights!(e)                                                                                # smc:266-270: Wns = 1/N, alive = true
    w = fill(1.0 / e.N, e.N); a = ones(UInt8, e.N)
    h2d(e, e.wns, w, 8e.N); h2d(e, e.alive, a, e.N)
    # the weights are uniform now (and stay so under an indicator kernel): the prologue may take the closed forms of the reweight
    check(ccall((:abcdez_ctx_set_uniform_weights, LIB), Cint, (Ptr{Cvoid}, Cint), e.ctx, 1))
    e.n_alive = e.n_prev = e.N
end
function extrema_dev(e)                                                                                   # smc:286,364
    lo = Ref(0.0); hi = Ref(0.0)
    check(ccall((:abcdez_extrema, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Float64}), e.ctx, e.delta[e.cur], e.N, lo, hi))
    (lo[], hi[])
end
# smc:301 (ϵ), :305-311 (weights, alive), :323 (ESS), the extrema of :364 for the generation before, and the partition of
# the packed population: ONE call, ONE host synchronisation.  The ϵ schedule stays here: ϵ and ϵ_target go in.
function prologue!(e, α, ϵ, ϵ_target, ϵ_k, ess_min)
    bind_stamps!(e)
    ϵn = Ref(0.0); q = Ref(0.0); wnorm = Ref(0.0); ess = Ref(0.0); lo = Ref(0.0); hi = Ref(0.0); na = Ref(Int64(0)); part = Ref(Int32(0))
    check(ccall((:abcdez_smc_prologue_packed, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Float64, Float64, Float64, Float64,
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Ref{Float64}, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ref{Int64}, Ref{Int32}, Ref{Float64}, Ref{Float64}),
                e.ctx, e.delta[e.cur], e.wns, e.alive, e.N, e.n_prev, α, ϵ, ϵ_target, ϵ_k, ess_min,
                e.bits[e.bc], e.bits[3 - e.bc], e.slot[1], e.slot[2], e.logpi[e.cur], ϵn, q, wnorm, ess, na, part, lo, hi))
    e.n_alive = na[]
    part[] != 0 && (e.n_prev = e.n_alive)
    (ϵn[], wnorm[], ess[], Int(na[]), (lo[], hi[]))
end
function get_ess(e)                                                                                       # smc:8
    ess = Ref(0.0)
    check(ccall((:abcdez_get_ess, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}), e.ctx, e.wns, e.N, ess)); ess[]
end
function resample!(e)                                                                                     # smc:85-104
    check(ccall((:abcdez_wsample_stratified, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt32, Ptr{Cvoid}), e.ctx, e.wns, e.N, e.draw, e.inds))
    e.draw += 1; o = other(e); bind_stamps!(e)
    check(ccall((:abcdez_smc_resample_gather_packed, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                e.ctx, e.inds, e.N, e.bits[e.bc], e.bits[3 - e.bc], e.slot[1], e.slot[2], e.logpi[e.cur], e.delta[e.cur],
                e.logpi[o], e.delta[o], e.wns, e.alive))
    e.cur = o; e.n_alive = e.n_prev = e.N
end
function smc_swarm!(e, ϵ, γ0, γσ)                                                                         # smc:106-153 (+ the copies of :337-340, which the packed layout does not need)
    nacc = Ref(Int64(0)); nsim = Ref(Int64(0)); bind_stamps!(e)
    check(ccall((:abcdez_smc_swarm_packed, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Float64, Float64, Float64, UInt32, Ref{Int64}, Ref{Int64}),
                e.ctx, e.bits[e.bc], e.bits[3 - e.bc], e.n_alive, 0, e.n_alive, e.slot[1], e.slot[2], e.logpi[e.cur], e.delta[e.cur],
                C_NULL, ϵ, γ0, γσ, e.sweep, nacc, nsim))
    e.sweep += 1; e.bc = 3 - e.bc
    (Int(nacc[]), Int(nsim[]))
end
# the sweeps of one generation, `for i in 1:Kmcmc ... (sum(naccs) / n_alive ≥ Kmcmc_min) && break` (smc:336-353), in one
# call: the test of :352 runs on the device between the sweeps -> (Σnaccs, Σnsims, Ki); Kmcmc ≤ 16 per call
function smc_sweeps!(e, ϵ, γ0, γσ, Kmcmc, Kmcmc_min; next_prologue=nothing)
    nacc = zeros(Int64, Kmcmc); nsim = zeros(Int64, Kmcmc); done = Ref(Int32(0)); bind_stamps!(e)
    if e.world > 1
        # sharded by position: this rank sweeps its chunk of the alive prefix; flag all-gather, replay of the other ranks' accepted
        # proposals, the test of smc:352 and the distance all-gather all happen inside this one call, on every rank alike
        chunk = cld(cld(e.n_alive, e.world), 64) * 64
        check(ccall((:abcdez_smc_sweeps_sharded, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                     Float64, Float64, Float64, UInt32, Int32, Float64, Ptr{Int64}, Ptr{Int64}, Ref{Int32}),
                    e.ctx, e.bits[e.bc], e.bits[3 - e.bc], e.n_alive, chunk, e.slot[1], e.slot[2], e.logpi[e.cur], e.delta[e.cur], e.flags,
                    ϵ, γ0, γσ, e.sweep, Kmcmc, Kmcmc_min, nacc, nsim, done))
        e.sweep += done[]; isodd(done[]) && (e.bc = 3 - e.bc)
        return (sum(nacc), sum(nsim), Int(done[]))
    end
    # (α, ϵ_target) of the next prologue!: its quantile select is enqueued behind these sweeps (it only reads Δs; same results)
    next_prologue === nothing || check(ccall((:abcdez_smc_select_ahead, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Float64),
                                             e.ctx, e.delta[e.cur], e.alive, e.N, next_prologue[1], next_prologue[2]))
    check(ccall((:abcdez_smc_sweeps_packed, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Float64, Float64, Float64, UInt32, Int32, Float64, Ptr{Int64}, Ptr{Int64}, Ref{Int32}),
                e.ctx, e.bits[e.bc], e.bits[3 - e.bc], e.n_alive, e.slot[1], e.slot[2], e.logpi[e.cur], e.delta[e.cur],
                ϵ, γ0, γσ, e.sweep, Kmcmc, Kmcmc_min, nacc, nsim, done))
    e.sweep += done[]; isodd(done[]) && (e.bc = 3 - e.bc)
    (sum(nacc), sum(nsim), Int(done[]))
end
# the loop body of smc:301-353 on an unsharded population in ONE call (abcdez_smc_generation_packed): prologue!, the resample! of
# smc:323-326 when ESS < ess_min, smc_sweeps! with the next generation's select armed -- the two decisions in between are taken
# inside the library.  -> (ϵ, wnorm, ess, n_alive the sweeps ran on, Σnaccs, Σnsims, Ki, extrema(Δs) of the generation before)
function generation!(e, α, ϵ, ϵ_target, ϵ_k, ess_min, γ0, γσ, Kmcmc, Kmcmc_min)
    bind_stamps!(e); o = other(e)
    ϵn = Ref(0.0); wnorm = Ref(0.0); ess = Ref(0.0); essr = Ref(0.0); lo = Ref(0.0); hi = Ref(0.0)
    na = Ref(Int64(0)); ns = Ref(Int64(0)); part = Ref(Int32(0)); res = Ref(Int32(0)); done = Ref(Int32(0))
    nacc = zeros(Int64, Kmcmc); nsim = zeros(Int64, Kmcmc)
    check(ccall((:abcdez_smc_generation_packed, LIB), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Float64, Float64, Float64, Float64, UInt32, UInt32,
                 Int32, Float64, Int32, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ref{Int64}, Ref{Int32}, Ref{Int32}, Ref{Float64},
                 Ref{Int64}, Ptr{Int64}, Ptr{Int64}, Ref{Int32}, Ref{Float64}, Ref{Float64}),
                e.ctx, e.N, e.n_prev, e.bits[e.bc], e.bits[3 - e.bc], e.slot[1], e.slot[2], e.logpi[e.cur], e.delta[e.cur],
                e.logpi[o], e.delta[o], e.wns, e.alive, e.inds, α, ϵ, ϵ_target, ϵ_k, ess_min, γ0, γσ, e.sweep, e.draw,
                Kmcmc, Kmcmc_min, 1, ϵn, wnorm, ess, na, part, res, essr, ns, nacc, nsim, done, lo, hi))
    e.n_alive = na[]
    part[] != 0 && (e.n_prev = e.n_alive)
    if res[] != 0                                # the other (logpi, Δ, stamp) arrays are the current ones now (smc:85-104)
        e.draw += 1; e.cur = o; e.n_alive = e.n_prev = e.N
    end
    e.sweep += done[]; isodd(done[]) && (e.bc = 3 - e.bc)
    (ϵn[], wnorm[], res[] != 0 ? essr[] : ess[], Int(ns[]), sum(nacc), sum(nsim), Int(done[]), (lo[], hi[]))
end
# P (push_p-cast, smc:382 / mc:166), Wns, C and -- with blobs on -- the simulated data behind every distance
function download(e; packed::Bool)
    rows = devalloc(8 * e.N * e.ld); pushed = devalloc(8 * e.N * e.ld)
    try
        if packed
            check(ccall((:abcdez_packed_gather, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                        e.ctx, e.bits[e.bc], e.N, e.slot[1], e.slot[2], rows))
        end
        src = packed ? rows : e.slot[e.cur]
        check(ccall((:abcdez_push_p, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}), e.ctx, src, e.N, pushed))   # types.jl:20-23
        nrow = 8 * e.ld * e.N
        stage = hostalloc(nrow + 16e.N)               # rows | Δ | Wns, pinned
        th = Matrix{Float64}(undef, e.ld, e.N); Δ = Vector{Float64}(undef, e.N); W = Vector{Float64}(undef, e.N)
        try
            d2h_async(e, stage, pushed, nrow); d2h_async(e, stage + nrow, e.delta[e.cur], 8e.N); d2h_async(e, stage + nrow + 8e.N, e.wns, 8e.N)
            check(ccall((:abcdez_sync, LIB), Cint, (Ptr{Cvoid},), e.ctx))
            unsafe_copyto!(pointer(th), Ptr{Float64}(stage), e.ld * e.N)
            unsafe_copyto!(pointer(Δ), Ptr{Float64}(stage + nrow), e.N); unsafe_copyto!(pointer(W), Ptr{Float64}(stage + nrow + 8e.N), e.N)
        finally
            hostfree(stage)
        end
        blobs = fill(nothing, e.N)
        if e.nb > 0                                # rebuild the blobs from the stamps; the re-run distance must be the stored one
            w = Ref(Int32(0)); check(ccall((:abcdez_blob_width, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), e.ctx, w))
            bl = devalloc(8 * e.N * w[]); redo = devalloc(8e.N)
            try
                check(ccall((:abcdez_blob_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                            e.ctx, src, e.stamp[e.cur], e.N, bl, redo))
                check(ccall((:abcdez_sync, LIB), Cint, (Ptr{Cvoid},), e.ctx))
                B = Matrix{Float64}(undef, w[], e.N); R = Vector{Float64}(undef, e.N)
                d2h(e, B, bl, sizeof(B)); d2h(e, R, redo, 8e.N)
                reinterpret(UInt64, R) == reinterpret(UInt64, Δ) || error("blobs: a re-run simulation does not reproduce the stored distance")
                blobs = e.nb == 1 ? B[1, :] : [B[1:e.nb, i] for i in 1:e.N]
            finally
                devfree(bl); devfree(redo)
            end
        end
        P = e.d == 1 ? th[1, :] : [Tuple(th[1:e.d, i]) for i in 1:e.N]
        return (P, W, Δ, blobs)
    finally
        devfree(rows); devfree(pushed)
    end
end

# ---- abcdesmc!: the host loop of src/abcdez_smc.jl:215-394, each ★ call replaced by one ccall ----
function abcdesmc!(prior, dist!::DeviceSimulator, ϵ_target, varexternal;
                   nparticles::Int=100, α=0.95, δess=0.5, nsims_max::Int=10^7, Kmcmc::Int=3, Kmcmc_min=1.0,
                   ABCk=ABCdeZ.IndicatorStrict0toϵ, facc_stop=0.0, facc_min=0.0, facc_tune=0.975,
                   verbose::Bool=true, verboseout::Bool=true, rng::Union{Integer,AbstractRNG}=Random.default_rng(),
                   parallel::Bool=false, comm=nothing)
    # comm = (id, rank, world): one of `world` processes, one GPU each (top of this file); every rank must pass the same integer `rng`
    # `varexternal` and `parallel` are accepted and ignored: the device simulator holds its own data (no per-task deepcopy,
    # smc:166-173) and the whole population always runs in parallel on the GPU (smc:237's executor has no counterpart)
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

    (comm === nothing || (Kmcmc ≤ 16 && rng isa Integer)) || error("comm: a sharded run needs Kmcmc <= 16 and the same integer rng on every rank")
    e = Engine(prior, dist!, ABCk, philox_key(rng), nparticles; comm = comm)
    try
        # smc:238-239 (the executor of :237 has no counterpart: the population always runs in parallel on the device)
        verbose && (@info("Running abcdesmc! with executor (libabcdez_hip.so on $(e.world) GPU rank(s)) ", typeof(dist!)))
        verbose && (@info "Running abcdesmc! with" ϵ_target nparticles α δess nsims_max Kmcmc Kmcmc_min ABCk facc_stop facc_min facc_tune rng parallel verboseout)
        init!(e)                                                                            # smc:242-252
        reset_weights!(e)                                                                   # smc:266-270
        ϵ = Inf; ϵ_k = Inf; logZ = 0.0; ess = 0.0; nsims = 0; facc = 1.0; Ki = Kmcmc        # smc:255-276
        ess_min = nparticles * δess
        γ0 = 2.38 / sqrt(2 * length(prior)); γσ = 1e-5                                      # smc:280-281
        ϵs = [ϵ]; ranges_ϵ = [extrema_dev(e)]; logZs = [logZ]; esss = [get_ess(e)]; faccs = [facc]; γ0s = [γ0]; Kmcmcs = [Ki]
        iters = 0
        while true                                                                          # smc:295
            iters += 1
            naccs = 0; Ki = Kmcmc
            facc < facc_min && (γ0 *= facc_tune)                                            # smc:320 (γ0 is only read by the sweeps)
            if e.world == 1 && Kmcmc ≤ 16                                                   # smc:301-353 in one call
                ϵ, wnorm, ess, n_alive, naccs, nsim, Kdone, range_prev = generation!(e, α, ϵ, ϵ_target, ϵ_k, ess_min, γ0, γσ, Kmcmc, Kmcmc_min)
                n_alive ≥ 3 && (Ki = Kdone)
                nsims += nsim
            else
                ϵ, wnorm, ess, n_alive, range_prev = prologue!(e, α, ϵ, ϵ_target, ϵ_k, ess_min) # smc:301-311, :323
                if n_alive > 0 && ess < ess_min                                             # smc:323-326
                    resample!(e); ess = get_ess(e); n_alive = nparticles
                end
                if n_alive ≥ 3 && Kmcmc ≤ 16
                    naccs, nsim, Ki = smc_sweeps!(e, ϵ, γ0, γσ, Kmcmc, Kmcmc_min)           # smc:336-353, one (collective) call
                    nsims += nsim
                elseif n_alive ≥ 3
                    for i in 1:Kmcmc                                                        # smc:336-353
                        nacc, nsim = smc_swarm!(e, ϵ, γ0, γσ)
                        naccs += nacc; nsims += nsim
                        (naccs / n_alive ≥ Kmcmc_min) && (Ki = i; break)                    # smc:352
                    end
                end
            end
            iters > 1 && push!(ranges_ϵ, range_prev)                                        # smc:364 of the generation before
            ABCk(ϵ)
            logZ += log(wnorm)                                                              # smc:315
            facc = naccs / (n_alive * Ki); ϵ_k = ϵ                                          # smc:357-360
            push!(ϵs, ϵ); push!(logZs, logZ); push!(esss, ess); push!(faccs, facc); push!(γ0s, γ0); push!(Kmcmcs, Ki)
            # smc:372 -- range_ϵ = extrema(Δs) of THIS generation is one more small reduction, made only for the log line
            verbose && (@info "Finished run:" iteration = iters nsim = nsims ϵ = ϵ range_ϵ = extrema_dev(e) ess = ess facc = facc logZ = logZ)
            n_alive ≥ 3 || (@warn("No alive particles"); break)                             # smc:375
            (ϵ ≤ ϵ_target || nsims ≥ nsims_max || facc < facc_stop) && break                # smc:376
        end
        check(ccall((:abcdez_smc_select_discard, LIB), Cint, (Ptr{Cvoid},), e.ctx))        # the run ends: nothing is left armed
        push!(ranges_ϵ, extrema_dev(e))                                                     # smc:364 of the last generation
        verbose && (@info "Final run:" iteration = iters nsim = nsims ϵ = ϵ range_ϵ = ranges_ϵ[end] ess = ess facc = facc logZ = logZ)   # smc:379
        P, Wns, Δs, blobs = download(e; packed=true)                                        # smc:382
        return verboseout ? (P = P, Wns = Wns, C = Δs, ϵ = ϵ, logZ = logZ, blobs = blobs, ϵs = ϵs, ranges_ϵ = ranges_ϵ,
                             logZs = logZs, esss = esss, faccs = faccs, γ0s = γ0s, Kmcmcs = Kmcmcs) :
                            (P = P, Wns = Wns, C = Δs, ϵ = ϵ, logZ = logZ, blobs = blobs)  # smc:388-393
    finally
        free!(e)
    end
end

# ---- abcdemc!: the host loop of src/abcdez_mc.jl:102-172 --------------------------------------------
function count_gt(e, thr)                                                                                 # mc:133
    c = Ref(Int64(0))
    check(ccall((:abcdez_count_gt, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Ref{Int64}), e.ctx, e.delta[e.cur], e.N, thr, c)); Int(c[])
end
rank_prepare!(e, ϵ_pop, ϵ_h) = check(ccall((:abcdez_mc_rank_prepare, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                                           e.ctx, e.delta[e.cur], e.N, ϵ_pop, ϵ_h, e.order, e.sorted, e.cnt))    # mc:23
# (the synchronous pair: better particles of mc:23 BY RANK; C_NULL for order and cnt draws them by rejection instead -- the library's
#  asynchronous generations below choose between the two themselves, include/abcdez_spec.h abz_mc_draws_by_rejection)
function mc_swarm!(e, ϵ_pop, ϵ_target, γ0, γσ)                                                            # mc:5-61 + :140-143; also mc:156, :146
    nsim = Ref(Int64(0)); above = Ref(Int64(0)); lo = Ref(0.0); hi = Ref(0.0); o = other(e); bind_stamps!(e)
    check(ccall((:abcdez_mc_swarm, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Float64, Float64, Float64, Float64, Int64, Int64, UInt32, Ref{Int64}, Ref{Int64}, Ref{Float64}, Ref{Float64}),
                e.ctx, e.order, e.cnt, e.N, e.slot[e.cur], e.logpi[e.cur], e.delta[e.cur], e.slot[o], e.logpi[o], e.delta[o],
                ϵ_pop, ϵ_target, γ0, γσ, 0, e.N, e.sweep, nsim, above, lo, hi))
    e.sweep += 1; e.cur = o
    (Int(nsim[]), Int(above[]), lo[], hi[])
end

# abcdemc!'s loop has no data-dependent exit (mc:134): a generation is ISSUED without waiting (ϵ_pop of mc:147 and the rank pass's
# window are made on the device from the extrema the sweep before left there; lo_hi only for the first one) and its reductions
# are COLLECTED later, in issue order; at most 8 tickets may be outstanding
function mc_generation_issue!(e, α, ϵ_target, γ0, γσ, lo_hi, do_rank::Bool)
    t = Ref(Int64(0)); o = other(e); bind_stamps!(e)
    lh = lo_hi === nothing ? C_NULL : pointer(lo_hi)
    if e.world > 1
        # sharded over the library's communicator: this rank sweeps its own particles, the library exchanges the new rows and the
        # generation's reductions (one group of RCCL collectives on its stream); tickets and their results as on one GPU
        GC.@preserve lo_hi check(ccall((:abcdez_mc_generation_sharded_async, LIB), Cint,
                (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Float64, Float64, Ptr{Float64}, Int32, Float64, Float64, UInt32, Ref{Int64}),
                e.ctx, e.N, e.slot[e.cur], e.logpi[e.cur], e.delta[e.cur], e.slot[o], e.logpi[o], e.delta[o], e.order, e.sorted, e.cnt,
                α, ϵ_target, lh, do_rank ? 1 : 0, γ0, γσ, e.sweep, t))
        e.sweep += 1; e.cur = o
        return t[]
    end
    GC.@preserve lo_hi check(ccall((:abcdez_mc_generation_async, LIB), Cint,
                (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                 Float64, Float64, Ptr{Float64}, Int32, Float64, Float64, UInt32, Ref{Int64}),
                e.ctx, e.N, e.slot[e.cur], e.logpi[e.cur], e.delta[e.cur], e.slot[o], e.logpi[o], e.delta[o], e.order, e.sorted, e.cnt,
                α, ϵ_target, lh, do_rank ? 1 : 0, γ0, γσ, e.sweep, t))
    e.sweep += 1; e.cur = o
    t[]
end
function mc_generation_collect!(e, ticket)
    nsim = Ref(Int64(0)); above = Ref(Int64(0)); lo = Ref(0.0); hi = Ref(0.0); ϵ_pop = Ref(0.0)
    check(ccall((:abcdez_mc_generation_wait, LIB), Cint, (Ptr{Cvoid}, Int64, Ref{Int64}, Ref{Int64}, Ref{Float64}, Ref{Float64}, Ref{Float64}),
                e.ctx, ticket, nsim, above, lo, hi, ϵ_pop))
    (Int(nsim[]), Int(above[]), lo[], hi[])
end

function abcdemc!(prior, dist!::DeviceSimulator, ϵ_target, varexternal;
                  nparticles::Int=50, generations::Int=20, verbose=true, rng::Union{Integer,AbstractRNG}=Random.default_rng(),
                  parallel::Bool=false, comm=nothing)
    # comm = (id, rank, world): one of `world` processes, one GPU each (top of this file); every rank must pass the same integer `rng`
    (comm === nothing || rng isa Integer) || error("comm: a sharded run needs the same integer rng on every rank")
    α = 0.0                                                                              # mc:107
    0.0 ≤ ϵ_target || error("ϵ_target must be non-negative")
    5 ≤ nparticles || error("nparticles must be at least 5")
    1 ≤ generations || error("generations must be at least 1")
    e = Engine(prior, dist!, ABCdeZ.IndicatorStrict0toϵ, philox_key(rng), nparticles; comm = comm)
    try
        verbose && (@info("Running abcdemc! with executor (libabcdez_hip.so on $(e.world) GPU rank(s)) ", typeof(dist!)))   # mc:113
        verbose && (@info "Running abcdemc! with" ϵ_target nparticles generations α rng parallel)                           # mc:114
        init!(e)                                                                             # mc:117-125
        nsims = 0; γ0 = 2.38 / sqrt(2 * length(prior)); γσ = 1e-5; iters = 0                 # mc:128-131
        complete = 1 - count_gt(e, ϵ_target) / nparticles                                    # mc:133
        ϵ_l, ϵ_h = extrema_dev(e)                                                            # mc:146 (afterwards the sweeps keep them on the device)
        tickets = Int64[]; converged = false; ahead = 4
        function collect!()
            nsim, above, ϵ_l, ϵ_h = mc_generation_collect!(e, popfirst!(tickets))            # mc:156, next :146
            nsims += nsim
            ncomplete = 1 - above / nparticles                                               # mc:156
            verbose && (ncomplete != complete || complete >= (nparticles - 1) / nparticles) &&
                (@info "Finished run:" completion = ncomplete nsim = nsims range_ϵ = (ϵ_l, ϵ_h))
            complete = ncomplete
            converged = converged || ϵ_h <= ϵ_target                                         # stays true (mc:19): no more rank passes
        end
        while iters < generations                                                            # mc:134
            iters += 1
            # ϵ_pop = max(ϵ_target, ϵ_l + α (ϵ_h - ϵ_l)) mc:147 on the device; rank pass; abcdemc_swarm! mc:149
            push!(tickets, mc_generation_issue!(e, α, ϵ_target, γ0, γσ, iters == 1 ? [ϵ_l, ϵ_h] : nothing, !converged))
            while length(tickets) > ahead; collect!(); end
        end
        while !isempty(tickets); collect!(); end
        conv = ϵ_h <= ϵ_target                                                               # mc:163
        verbose && (@info "End:" completion = complete converged = conv nsim = nsims range_ϵ = (ϵ_l, ϵ_h))   # mc:164
        P, _, Δs, blobs = download(e; packed=false)                                          # mc:166
        return (P = P, C = Δs, reached_ϵ = conv, blobs = blobs)                              # mc:171
    finally
        free!(e)
    end
end

end # module
