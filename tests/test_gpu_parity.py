"""GPU parity: every C-ABI entry point of libabcdez_hip.so against the CPU oracle,
bit for bit, on the same seeded inputs (sizes the oracle finishes in seconds)."""
import math

import numpy as np
import pytest
import torch

import abcdez_amd as A
from abcdez_amd import _lib
from abcdez_amd.engine import HipOps, PopulationEngine

pytestmark = pytest.mark.gpu


def models():
    n32 = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(32)])
    return {
        "normal1d": (A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0), 0.3),
        "uniform1d": (A.Uniform(-10.0, 10.0), A.Normal1D(3.0), 0.3),
        # a small eps_target: fewer than 1 / 16 of the particles at or below it for many generations (abcdemc draws by rank)
        "normal1d_tight": (A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0), 0.02),
        "mvn32": (n32, A.MVNormal(tuple([1.0] * 32)), 6.0),
        "mvn8": (A.Factored(*[A.Normal(0.0, 1.0) for _ in range(8)]), A.MVNormal(tuple([1.0] * 8)), 2.5),
        "mvn3": (A.Factored(A.Normal(0, 1), A.Uniform(-3, 3), A.Normal(1, 2)), A.MVNormal((0.5, 0.2, 1.0)), 0.8),
        "quad2d_inf": (A.Factored(A.Normal(0, 5), A.Normal(0, 5)), A.Quad2D(0.5), 0.01),
        "normdu": (A.Factored(A.Normal(1, 0.5), A.DiscreteUniform(1, 10)), A.NormalTimesDU(5.5), 0.01),
        "dirac": (A.Normal(1, 0.2), A.DiracSquare(1.5), 0.1),
        "mixture": (A.Uniform(-10, 10), A.Mixture01(0.0), 0.01),
        # test/runtests.jl:439-445: NegativeBinomial(mu 30, sd 15) x Beta(15, 2), rejection samplers at init
        "socks": (A.Factored(A.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), A.Beta(15, 2)), A.Socks(0, 11), 0.01),
        # the further Distributions.jl families (abcdez_spec.h ABZ_PRIOR_EXPONENTIAL ...): one lane per particle, lane groups
        "gamma1d": (A.Gamma(2.0, 1.5), A.Normal1D(3.0), 0.3),
        "further8": (A.Factored(A.Exponential(1.5), A.Gamma(2.5, 0.6), A.LogNormal(0.0, 0.5), A.Cauchy(1.0, 0.5), A.Laplace(1.0, 1.0),
                                A.Weibull(1.8, 1.2), A.InverseGamma(3.0, 2.0), A.truncated(A.Normal(1.0, 2.0), 0.0, 4.0)),
                     A.MVNormal(tuple([1.0] * 8)), 2.5),
        "further5": (A.Factored(A.Logistic(1.0, 0.5), A.TDist(4.0), A.Pareto(3.0, 0.5), A.Poisson(2.0), A.Binomial(6, 0.3)),
                     A.MVNormal((1.0, 0.5, 0.8, 2.0, 2.0)), 2.0),
    }


def engines(name, N, seed=3, ABCk=A.IndicatorStrict0toϵ, lanes=0, oracle=None, storage="packed"):
    """(spec, product engine on the GPU, the same host logic on the oracle, eps); storage "packed" = abcdesmc's
    population, "classic" = abcdemc's double buffer"""
    prior, sim, eps = models()[name]
    spec = A.ModelSpec(prior, sim, ABCk, seed=seed)
    hip = PopulationEngine(spec, N, ops=HipOps(spec, lanes=lanes), storage=storage)
    orc = oracle.oracle_engine(spec, N, storage=storage)
    return spec, hip, orc, eps


def same(a: torch.Tensor, b: torch.Tensor):
    a = a.detach().cpu().contiguous()
    b = b.detach().cpu().contiguous()
    if a.dtype == torch.float64:
        return torch.equal(a.view(torch.int64), b.view(torch.int64))
    return torch.equal(a, b)


def assert_state_equal(hip, orc, what=""):
    for k, nm in enumerate(("theta", "logpi", "delta")):
        assert same(hip.state[k], orc.state[k]), f"{what}: {nm} differs"
    assert same(hip.wns, orc.wns), f"{what}: wns differs"
    assert same(hip.alive, orc.alive), f"{what}: alive differs"


# ---------------------------------------------------------------- spec arithmetic on the device
@pytest.mark.parametrize("fn,gen", [
    (0, "pos"), (1, "exparg"), (2, "unit"), (3, "round"), (4, "round"), (5, "pos"), (6, "pair"),
    (7, "posnormal"), (8, "unit52"), (9, "bmrange"), (10, "lgamma"),
])
def test_math_bit_exact(oracle, fn, gen):
    rng = np.random.default_rng(100 + fn)
    n = 1 << 20
    if gen == "pos":
        x = np.exp(rng.uniform(-700, 700, n))
        x[:8] = [0.0, 1.0, 5e-324, 2.2250738585072014e-308, np.inf, 0.5, 2.0, 1.0 - 2 ** -53]
    elif gen == "posnormal":
        x = np.concatenate([np.exp(rng.uniform(-700, 700, n // 2)), (rng.integers(0, 1 << 52, n // 2) + 0.5) * 2.0 ** -52])
        x[:6] = [1.0, 2.0 ** -53, 1 - 2.0 ** -53, 0.5, 2.2250738585072014e-308, 1.7e308]
    elif gen == "unit52":
        x = rng.integers(0, 1 << 52, n).astype(np.float64) * 2.0 ** -52
        x[:4] = [0.0, 0.125, 0.5, 1 - 2.0 ** -52]
    elif gen == "bmrange":
        n = 1 << 24                       # sqrt_pn replaces the compiler's expansion: check it hard
        x = np.exp(rng.uniform(-37.0, 4.4, n))     # -2 log u for u in [2^-53, 1)
        x[:4] = [2.0 ** -52, 73.5, 1.0, 2.0]
    elif gen == "lgamma":
        x = np.concatenate([rng.uniform(1e-3, 40, n // 2), np.exp(rng.uniform(0, 20, n // 2))])
    elif gen == "exparg":
        x = rng.uniform(-750, 710, n)
        x[:6] = [0.0, -np.inf, np.inf, -745.2, 709.8, 1e-300]
    elif gen == "unit":
        x = rng.integers(0, 1 << 53, n).astype(np.float64) * 2.0 ** -53
        x[:4] = [0.0, 0.125, 0.5, 1 - 2.0 ** -53]
    elif gen == "round":
        x = rng.uniform(-1e6, 1e6, n)
        x[:8] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 0.0, 4503599627370497.0]
    else:
        x = rng.normal(0, 1e3, n)
    n = x.size
    y2in = rng.uniform(0.5, 3.0, n)
    spec = A.ModelSpec(A.Normal(0, 1), A.Normal1D(0.0), seed=1)
    ops = HipOps(spec)
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros_like(xd)
    y2d = torch.from_numpy(y2in.copy()).cuda()
    ops.math_eval(fn, xd, yd, y2d)
    yh = np.zeros(n)
    y2h = y2in.copy()
    oracle.lib().orc_math_eval(fn, x.ctypes.data, yh.ctypes.data, y2h.ctypes.data, n)
    assert np.array_equal(yd.cpu().numpy().view(np.int64), yh.view(np.int64))
    if fn in (2, 8):
        assert np.array_equal(y2d.cpu().numpy().view(np.int64), y2h.view(np.int64))
    if fn == 9:
        assert np.array_equal(yh, np.sqrt(x))          # and both are the correctly rounded sqrt


# ---------------------------------------------------------------- S1
@pytest.mark.parametrize("name", list(models().keys()))
def test_init_parity(oracle, name):
    spec, hip, orc, _ = engines(name, 20000, oracle=oracle)
    hip.init_population()
    orc.init_population()
    assert_state_equal(hip, orc, "init")
    assert torch.isfinite(hip.state[2]).all() and torch.isfinite(hip.state[1]).all()   # init.jl:14


# ---------------------------------------------------------------- population passes on a synthetic state
@pytest.mark.parametrize("N", [1, 2, 777, 2048, 2049, 100003])
@pytest.mark.parametrize("abck", A.ALL_KERNELS)
def test_reweight_quantile_compact_parity(oracle, N, abck):
    prior, sim, _ = models()["normal1d"]
    spec = A.ModelSpec(prior, sim, abck, seed=5)
    hip = PopulationEngine(spec, N, ops=HipOps(spec))
    orc = oracle.oracle_engine(spec, N)
    g = torch.Generator().manual_seed(N)
    delta = torch.rand(N, generator=g, dtype=torch.float64) * 3.0
    delta[::7] = delta[0]          # ties
    for e in (hip, orc):
        e.state[2].copy_(delta)
        e.reset_weights()
    eps_old = math.inf
    for alpha in (0.95, 0.5, 0.9):
        qh, qo = hip.quantile_alive(alpha), orc.quantile_alive(alpha)
        assert qh == qo
        assert hip.extrema() == orc.extrema()
        assert hip.count_gt(qh) == orc.count_gt(qo)
        rh, ro = hip.smc_reweight(eps_old, qh), orc.smc_reweight(eps_old, qo)
        assert rh[2] == ro[2]
        if ro[2] == 0:
            break
        assert rh == ro, (rh, ro)
        assert hip.get_ess() == orc.get_ess()
        assert same(hip.wns, orc.wns) and same(hip.alive, orc.alive)
        nh, no = hip.alive_compact(), orc.alive_compact()          # partition: the alive particles become a prefix
        assert nh == no == ro[2]
        assert bool(hip.alive[:nh].all()) and not bool(hip.alive[nh:].any())
        assert_state_equal(hip, orc, "partition")
        eps_old = qh


def _quantile_inputs(shape, N, rng):
    if shape == "uniform":
        return rng.random(N) * 3.0
    if shape == "clustered":            # the annealed population: a narrow band below the last epsilon
        return 9.0 + rng.random(N) ** 0.25 * 0.04
    if shape == "heavy":                # 20 octaves and infinite distances (2-D test of runtests.jl:246-318)
        d = np.exp(rng.normal(0.0, 5.0, N))
        d[rng.random(N) < 0.1] = np.inf
        return d
    if shape == "equal":
        return np.full(N, 2.5)
    if shape == "few":                  # discrete distances (Socks, DiracSquare): long runs of equal keys
        return rng.integers(0, 5, N).astype(np.float64)
    if shape == "zeros_and_tiny":
        d = rng.random(N) * 1e-300
        d[::3] = 0.0
        return d
    raise KeyError(shape)


@pytest.mark.parametrize("shape", ["uniform", "clustered", "heavy", "equal", "few", "zeros_and_tiny"])
@pytest.mark.parametrize("N", [3, 1000, 70001, 1 << 20])
def test_quantile_select_exact_on_any_distribution(oracle, shape, N):
    """smc:301.  The select bins the IEEE keys in a window carried over from the previous call: fresh windows, windows
    that fit (driver order: quantile -> reweight), stale windows (new data behind the same pointers, larger p than
    before) all give the exact order statistics -- checked against numpy's sort and the oracle's quantile."""
    rng = np.random.default_rng(N + len(shape))
    spec = A.ModelSpec(A.Normal(0, 1), A.Normal1D(0.0), seed=1)
    ops = HipOps(spec)
    delta = torch.zeros(N, dtype=torch.float64, device="cuda")
    alive = torch.ones(N, dtype=torch.uint8, device="cuda")

    def check(d, a, p, hint):
        delta.copy_(torch.from_numpy(d))
        alive.copy_(torch.from_numpy(a))
        got = ops.quantile_alive(delta, alive, p, hint)
        x = np.sort(d[a != 0])
        n = x.size
        h = (n - 1) * p + 1.0
        j = min(max(int(math.floor(h)), 1), max(n - 1, 1))
        xj, xj1 = x[j - 1], x[min(j, n - 1)]
        assert got[1] == xj and got[2] == xj1, (shape, N, p, got, xj, xj1)
        ref = oracle.lib().orc_quantile_alive(d.ctypes.data, a.ctypes.data, N, p, None, None)
        assert got[0] == ref or (math.isnan(got[0]) and math.isnan(ref))       # inf - inf at the top of "heavy"

    d = _quantile_inputs(shape, N, rng)
    a = np.ones(N, dtype=np.uint8)
    check(d, a, 0.95, N)                                  # first call: window from the min / max pass
    for p in (0.95, 0.5, 0.99, 0.0, 1.0):                 # driver order: the survivors of the last quantile
        x = d[a != 0]
        if x.size < 3:
            break
        q = np.quantile(x, 0.9)
        a = (a != 0) & (d < q) if (d[a != 0] < q).sum() >= 3 else a
        a = a.astype(np.uint8)
        check(d, a, p, int(a.sum()))
        check(d, a, p, -1)                                # alive count not known to the caller
    a = np.ones(N, dtype=np.uint8)
    check(d, a, 0.97, N)                                  # stale window: everything is alive again
    for other in ("heavy", "clustered", "uniform"):       # stale window: unrelated data behind the same pointers
        d2 = _quantile_inputs(other, N, rng)
        a2 = (rng.random(N) < 0.7).astype(np.uint8)
        a2[:3] = 1
        check(d2, a2, 0.95, int(a2.sum()))
        check(d2, a2, 0.05, int(a2.sum()))


@pytest.mark.parametrize("N", [5, 1000, 4096, 65537])
def test_stratified_resample_parity(oracle, N):
    spec, hip, orc, _ = engines("mvn8", N, oracle=oracle)
    hip.init_population(); orc.init_population()
    g = torch.Generator().manual_seed(N)
    w = torch.rand(N, generator=g, dtype=torch.float64)
    w[torch.rand(N, generator=g) < 0.4] = 0.0
    w[0] = 1.0
    w /= w.sum()
    for e in (hip, orc):
        e.wns.copy_(w)
        e.alive.copy_((w > 0).to(torch.uint8))
    for _ in range(2):
        hip.smc_resample(); orc.smc_resample()
        assert same(hip.inds, orc.inds)
        assert_state_equal(hip, orc, "resample")
        for e in (hip, orc):
            e.wns.copy_(w)
    inds = hip.inds.cpu().numpy().astype(np.int64)
    assert (np.diff(inds) >= 0).all()                       # smc:45-54: monotone walk
    assert (w.numpy()[inds] > 0).all()                      # zero weights never chosen
    counts = np.bincount(inds, minlength=N)
    expect = N * w.numpy()
    assert (counts >= np.floor(expect) - 1).all() and (counts <= np.ceil(expect) + 1).all()


# (S2/S3 sweeps, partition, replay: tests/test_gpu_packed.py)


# ---------------------------------------------------------------- S4
@pytest.mark.parametrize("name", ["normal1d", "mvn8", "quad2d_inf", "normdu", "dirac", "socks", "gamma1d", "further8", "further5"])
def test_mc_sweep_parity(oracle, name):
    N = 5000
    spec, hip, orc, eps_target = engines(name, N, oracle=oracle, storage="classic")
    hip.init_population(); orc.init_population()
    gamma0 = 2.38 / math.sqrt(2 * spec.d)
    for gen in range(6):
        lo, hi = orc.extrema()
        assert hip.extrema() == (lo, hi)
        eps_pop = max(eps_target, lo)
        # a hint that is too small, exact, or far too large only changes the binning, never the result
        hip.mc_rank_prepare(eps_pop, (0.5 * hi, hi, 1e6 * hi + 1.0)[gen % 3]); orc.mc_rank_prepare(eps_pop, hi)
        assert same(hip.order, orc.order) and same(hip.sorted_delta, orc.sorted_delta)
        draws = orc.state[2] > eps_pop                              # candidate counts of the particles that draw (mc:19-20)
        assert torch.equal(hip.rank_cnt.cpu()[draws], orc.rank_cnt[draws])
        sd = orc.sorted_delta
        assert bool((sd[1:] >= sd[:-1]).all())                      # upper_bound(sorted_delta, Ds[i]) is well defined
        got, want = hip.mc_swarm(eps_pop, eps_target, gamma0, 1e-5), orc.mc_swarm(eps_pop, eps_target, gamma0, 1e-5)
        assert got == want                                          # (nsim, #(Ds > eps_target), min Ds, max Ds)
        assert got[1:] == (orc.count_gt(eps_target),) + orc.extrema()   # the folded-in reductions: mc:156, mc:146
        assert_state_equal(hip, orc, f"mc gen {gen}")


@pytest.mark.parametrize("name,alpha", [("normal1d", 0.0), ("mvn8", 0.0), ("mvn8", 0.3), ("normdu", 0.0), ("dirac", 0.0)])
def test_mc_generations_issued_ahead_equal_the_step_by_step_oracle(oracle, name, alpha):
    """abcdez_mc_generation_async / _wait: generations enqueued AHEAD of their results (eps_pop of mc:147 and the rank
    pass's binning window made on the device from the extrema the sweep before left there) against the oracle driven one
    synchronous generation at a time with eps_pop computed on the host -- every generation's reductions and eps_pop, and
    the final population, bit for bit; converged populations (do_rank = False) included"""
    N = 3000
    spec, hip, orc, eps_target = engines(name, N, oracle=oracle, storage="classic")
    hip.init_population(); orc.init_population()
    gamma0 = 2.38 / math.sqrt(2 * spec.d)
    lo, hi = orc.extrema()
    want, got, converged = [], [], False
    gens = 40
    for gen in range(gens):
        eps_pop = max(eps_target, lo + alpha * (hi - lo))                                  # mc:147 on the host
        nsim, ngt, lo, hi = orc.mc_generation(eps_pop, eps_target, hi, gamma0, 1e-5)
        want.append((nsim, ngt, lo, hi, eps_pop))
    first = hip.extrema()
    for gen in range(gens):
        hip.mc_generation_issue(alpha, eps_target, gamma0, 1e-5, lo_hi=first if gen == 0 else None, do_rank=not converged)
        while hip.mc_generations_in_flight() > (gen % 7):                                  # 0 .. 6 generations ahead
            got.append(hip.mc_generation_collect())
            converged = converged or got[-1][3] <= eps_target
    while hip.mc_generations_in_flight():
        got.append(hip.mc_generation_collect())
    assert got == want
    assert_state_equal(hip, orc, "after the pipelined generations")
    assert hip.sweep == orc.sweep
    # the synchronous entry points still work afterwards (one baseline for the cumulative counters)
    eps_pop = max(eps_target, lo)
    assert hip.mc_generation(eps_pop, eps_target, hi, gamma0, 1e-5) == orc.mc_generation(eps_pop, eps_target, hi, gamma0, 1e-5)


@pytest.mark.parametrize("name,N,gens", [("normal1d", 3000, 40), ("normal1d", 60000, 70), ("mvn8", 20000, 30)])
def test_mc_generations_replayed_as_graphs_equal_the_oracle(oracle, name, N, gens):
    """One abcdemc generation (rank pass + sweep + snapshot) is captured per launch shape and REPLAYED as a HIP graph
    (abcdez_ctx_set_graphs; needs a stream of its own -- the legacy default stream cannot be captured): the RNG epoch, the
    ring slot and the ticket come from the generation counter on the device.  Bit for bit the oracle's synchronous
    generations, through every shape of the rank pass (both sorts, only the LDS sort, only the radix sort, none)."""
    spec, hip, orc, eps_target = engines(name, N, oracle=oracle, storage="classic")
    hip.ops.set_graphs(True)                       # off by default (measured slower than stream launches on ROCm 7.2)
    gamma0 = 2.38 / math.sqrt(2 * spec.d)
    orc.init_population()
    lo, hi = orc.extrema()
    want = []
    for gen in range(gens):
        eps_pop = max(eps_target, lo)
        nsim, ngt, lo, hi = orc.mc_generation(eps_pop, eps_target, hi, gamma0, 1e-5)
        want.append((nsim, ngt, lo, hi, eps_pop))
    with hip.run_scope():
        assert torch.cuda.current_stream().cuda_stream != 0
        hip.init_population()
        first = hip.extrema()
        got, converged = [], False
        for gen in range(gens):
            hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, lo_hi=first if gen == 0 else None, do_rank=not converged)
            while hip.mc_generations_in_flight() > (gen % 5):
                got.append(hip.mc_generation_collect())
                converged = converged or got[-1][3] <= eps_target
        while hip.mc_generations_in_flight():
            got.append(hip.mc_generation_collect())
        replays, captures, direct = hip.ops.graph_stats()
        assert got == want
        assert_state_equal(hip, orc, "after the replayed generations")
    assert replays + direct == gens and direct >= 1 and replays >= gens // 2, (replays, captures, direct)
    assert 1 <= captures <= 24, captures          # few launch shapes: buffer parity x rank-pass path x (quantised) grid
    # the same run launch by launch gives the same population
    spec2, hip2, _, _ = engines(name, N, oracle=oracle, storage="classic")
    with hip2.run_scope():
        hip2.init_population()
        first = hip2.extrema()
        conv = False
        for gen in range(gens):
            hip2.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, lo_hi=first if gen == 0 else None, do_rank=not conv)
            conv = conv or hip2.mc_generation_collect()[3] <= eps_target
        assert hip2.ops.graph_stats()[0] == 0
        assert_state_equal(hip2, orc, "stream launches")


def test_mc_rank_pass_with_a_stale_tail_bound_fails_loudly(oracle):
    """once a chain of generations ran with eps_pop == eps_target the rank pass launches only the sort its
    proved tail bound calls for.  Distances written behind the library's back (here: a torch copy, without
    abcdez_smc_select_discard) make the bound stale -- the LDS sort alone cannot take the new tail.  The device notices and the
    ticket's redemption raises instead of handing back a sweep that drew from a stale enumeration; after a discard the
    context works again.  (A population of 4300 with 4090 particles above an eps_target nobody reaches: drawn by rank -- fewer
    than 1 / 16 at or below it -- with a tail the LDS sort takes.)"""
    N, n_in = 4300, 210
    spec, hip, orc, _ = engines("normal1d", N, oracle=oracle, storage="classic")
    eps_target = 1e-9
    hip.init_population()
    d = hip.state[2]
    d.copy_(5.0 + 1e-3 * torch.arange(N, dtype=torch.float64, device=d.device))
    d[:n_in] = 1e-10
    hip.discard_select_ahead()
    assert not hip.mc_draws_by_rejection(hip.count_gt(eps_target))
    gamma0 = 2.38 / math.sqrt(2)
    first = hip.extrema()
    for gen in range(4):
        hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, lo_hi=first if gen == 0 else None, do_rank=True)
        out = hip.mc_generation_collect()
        assert out[1] == N - n_in and out[4] == eps_target       # nobody arrives; eps_pop == eps_target bounds the tail
    assert hip.ops.mc_rank_stats()[1] >= 2                        # rank passes with only the LDS sort: the bound is 4090 <= 4096
    hip.state[2].add_(10.0)                         # every particle is in the tail now: 4300 > 4096 pairs
    hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, do_rank=True)
    with pytest.raises(_lib.AbcdezError, match="tail bound that no longer held"):
        hip.mc_generation_collect()
    # telling the library (what upload_state / reset paths do) makes the same context usable again
    hip.discard_select_ahead()
    hip._mc_pending = []
    lo_hi = hip.extrema()
    hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, lo_hi=lo_hi, do_rank=True)
    nsim, ngt, lo, hi, eps_pop = hip.mc_generation_collect()
    assert hi >= lo and 0 <= ngt <= N and 0 <= nsim <= N and eps_pop >= eps_target


def test_mc_rank_passes_stop_once_the_chain_draws_by_rejection(oracle):
    """include/abcdez_spec.h, abz_mc_draws_by_rejection: once at least 1 / 16 of the particles lie at or below eps_target the
    better particle of mc:23 is drawn by rejection.  The asynchronous path decides that on the device (the host is generations behind)
    and stops launching rank passes when a redeemed generation has shown the switch: every generation equals the oracle's
    (which counts for itself), and the number of rank passes is the number of by-rank generations plus at most the lag."""
    N, gens, ahead = 4000, 60, 3
    spec, hip, orc, eps_target = engines("normal1d_tight", N, oracle=oracle, storage="classic")
    hip.init_population(); orc.init_population()
    gamma0 = 2.38 / math.sqrt(2)
    lo, hi = orc.extrema()
    want, by_rank = [], 0
    for gen in range(gens):
        by_rank += 0 if orc.mc_draws_by_rejection(orc.count_gt(eps_target)) else 1
        eps_pop = max(eps_target, lo)
        nsim, ngt, lo, hi = orc.mc_generation(eps_pop, eps_target, hi, gamma0, 1e-5)
        want.append((nsim, ngt, lo, hi, eps_pop))
    assert 3 < by_rank < gens - 10 and hi > eps_target          # the run crosses the switch and does not converge
    first = hip.extrema()
    got = []
    for gen in range(gens):
        hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, lo_hi=first if gen == 0 else None, do_rank=True)
        while hip.mc_generations_in_flight() > ahead:
            got.append(hip.mc_generation_collect())
    while hip.mc_generations_in_flight():
        got.append(hip.mc_generation_collect())
    assert got == want
    assert_state_equal(hip, orc, "across the switch to rejection")
    ranked, skipped = sum(hip.ops.mc_rank_stats()), hip.ops.mc_draw_stats()
    assert ranked + skipped == gens
    assert by_rank <= ranked <= by_rank + ahead + 1, (by_rank, ranked, skipped)


def test_mc_rejection_draws_on_distances_written_behind_the_library_fail_loudly(oracle):
    """A chain that has switched to rejection launches no rank pass any more.  Distances written behind the library's back
    (a torch copy, without abcdez_smc_select_discard) can break the rule's premise -- candidate sets of a few particles.  A
    particle that finds no better particle in 1024 trials proves it: the ticket's redemption raises; after a discard the
    context counts again and works."""
    N = 6000
    spec, hip, orc, eps_target = engines("normal1d", N, oracle=oracle, storage="classic")
    hip.init_population()
    gamma0 = 2.38 / math.sqrt(2)
    first = hip.extrema()
    for gen in range(60):
        hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, lo_hi=first if gen == 0 else None, do_rank=True)
        hip.mc_generation_collect()
        if hip.ops.mc_draw_stats() >= 2:
            break
    skipped = hip.ops.mc_draw_stats()
    assert skipped >= 2
    d = hip.state[2]
    d.copy_(10.0 + torch.arange(N, dtype=torch.float64, device=d.device))      # particle k has k + 1 candidates; nobody converged
    hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, do_rank=True)
    with pytest.raises(_lib.AbcdezError, match="ran out of trials"):
        hip.mc_generation_collect()
    hip.discard_select_ahead()
    hip._mc_pending = []
    ranked = sum(hip.ops.mc_rank_stats())
    assert not hip.mc_draws_by_rejection(hip.count_gt(eps_target))   # (the failed generation's sweep ran: nearly all particles are still far out)
    hip.mc_generation_issue(0.0, eps_target, gamma0, 1e-5, lo_hi=hip.extrema(), do_rank=True)
    nsim, ngt, lo, hi, eps_pop = hip.mc_generation_collect()
    # counted anew: the chain draws by rank again
    assert 0 < ngt <= N and eps_pop >= eps_target and hip.ops.mc_draw_stats() == skipped + 1 and sum(hip.ops.mc_rank_stats()) == ranked + 1


def test_mc_swarm_by_rejection_where_the_rule_calls_for_ranks_is_an_error(oracle):
    """abcdez_mc_swarm with order = cnt = NULL takes the caller's word that the generation draws by rejection; on a
    population far from eps_target whose candidate sets hold a few particles only the trials run out and the call says so."""
    N = 5000
    spec, hip, orc, eps_target = engines("normal1d", N, oracle=oracle, storage="classic")
    hip.init_population()
    d = hip.state[2]
    d.copy_(10.0 + torch.arange(N, dtype=torch.float64, device=d.device))      # particle k has k + 1 candidates
    hip.discard_select_ahead()
    assert not hip.mc_draws_by_rejection(hip.count_gt(eps_target))
    lo, hi = hip.extrema()
    with pytest.raises(_lib.AbcdezError, match="ran out of trials"):
        hip.mc_swarm(max(eps_target, lo), eps_target, 2.38 / math.sqrt(2), 1e-5, reject=True)
    with pytest.raises(_lib.AbcdezError, match="both order and cnt"):
        hip.ops.mc_swarm(hip.order, None, hip.state, hip.other, 1.0, eps_target, 0.5, 1e-5, 0, N, 0)


def test_mc_generation_tickets_are_bounded_and_ordered(oracle):
    spec, hip, _, eps_target = engines("normal1d", 2000, oracle=oracle, storage="classic")
    hip.init_population()
    hip._mc_arrays()
    lo_hi = hip.extrema()
    tickets = []
    for k in range(8):
        tickets.append(hip.ops.mc_generation_async(hip.state, hip.other, hip.order, hip.sorted_delta, hip.rank_cnt, 0.0,
                                                   eps_target, lo_hi if k == 0 else None, True, 0.5, 1e-5, hip.sweep))
        hip.sweep += 1
        hip._swap()
    with pytest.raises(_lib.AbcdezError, match="too many generations in flight"):
        hip.ops.mc_generation_async(hip.state, hip.other, hip.order, hip.sorted_delta, hip.rank_cnt, 0.0, eps_target, None,
                                    True, 0.5, 1e-5, hip.sweep)
    with pytest.raises(_lib.AbcdezError, match="in the order they were issued"):
        hip.ops.mc_generation_wait(tickets[3])
    n_above_sync = hip.count_gt(eps_target)            # a synchronising call in between keeps the tickets redeemable
    res = [hip.ops.mc_generation_wait(t) for t in tickets]
    assert res[-1][1] == n_above_sync and all(r[0] > 0 for r in res)
    with pytest.raises(_lib.AbcdezError, match="in the order they were issued"):
        hip.ops.mc_generation_wait(tickets[-1] + 1)


# ---------------------------------------------------------------- whole drivers, product vs C restatement
@pytest.mark.parametrize("name,N", [("normal1d", 5000), ("uniform1d", 5000), ("mvn8", 4096), ("mvn32", 8192),
                                    ("quad2d_inf", 500), ("normdu", 100), ("dirac", 100), ("socks", 3000), ("gamma1d", 3000),
                                    ("further8", 4096), ("further5", 2000)])
@pytest.mark.parametrize("abck", [A.IndicatorStrict0toϵ, A.Indicator0toϵ, A.Epa0toϵ, A.EpaStrict0toϵ])
def test_abcdesmc_end_to_end_parity(oracle, name, N, abck):
    if name == "mvn32" and abck is not A.IndicatorStrict0toϵ:
        pytest.skip("one kernel is enough at d=32")
    prior, sim, eps = models()[name]
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, ABCk=abck, verbose=False, rng=11, nsims_max=10 ** 8)
    assert type(r.engine.ops).__name__ == "HipOps"
    c = oracle.run_abcdesmc(A.ModelSpec(prior, sim, abck, seed=11), N, eps, nsims_max=10 ** 8)
    res = r.engine.result()
    assert r.iters == c["iters"] and r.nsims == c["nsims"]
    assert np.array_equal(np.array(r.ϵs), c["eps_hist"])          # eps schedule identical
    assert r.logZ == c["logZ"] or (math.isnan(r.logZ) and math.isnan(c["logZ"]))
    assert np.array_equal(np.array(r.logZs), c["logZ_hist"], equal_nan=True)
    assert np.array_equal(np.array(r.esss), c["ess_hist"], equal_nan=True)
    assert np.array_equal(res["theta"], c["theta"])
    assert np.array_equal(res["C"], c["C"]) and np.array_equal(res["Wns"], c["Wns"], equal_nan=True)
    assert np.array_equal(res["alive"], c["alive"])


def test_abcdesmc_reuses_the_select_enqueued_ahead():
    """The rank select of generation g + 1 is enqueued behind the sweeps of generation g and the next prologue starts at the
    reweight: only the first generation and the one after each resample run it themselves.  (Binding the same stream again
    before every call, as the engine does, must not throw it away.)"""
    prior, sim, eps = models()["mvn8"]
    r = A.abcdesmc(prior, sim, eps, None, nparticles=4096, verbose=False, rng=11, nsims_max=10 ** 8)
    reused, inline = r.engine.ops.smc_select_stats()
    assert abs(reused + inline - r.iters) <= 1, (reused, inline, r.iters)
    assert reused >= (2 * r.iters) // 3, (reused, inline, r.iters)


def test_host_uploads_invalidate_only_what_they_overwrite():
    """abcdez_memcpy_h2d (the Julia shim's uploads): a copy into arrays the library holds no state about -- weights, bitmap
    words -- leaves a select enqueued ahead in place; a copy that overlaps the distances it was made from discards it.  Either way the
    results are those of a run without the copies (the discarded select is redone)."""
    import ctypes as C

    prior, sim, eps_t = models()["mvn8"]
    spec = A.ModelSpec(prior, sim, seed=5)
    g0 = 2.38 / math.sqrt(2 * spec.d)

    def run(upload):
        e = PopulationEngine(spec, 8192, ops=HipOps(spec))
        e.init_population(); e.reset_weights()
        eps, eps_k, out = math.inf, math.inf, []
        for gen in range(4):
            eps, wnorm, ess, n_alive, _ = e.smc_prologue(0.9, eps, eps_t, eps_k, 0.0)
            e.alive_compact()
            cnt = e.smc_sweeps(eps, g0, 1e-5, 2, 9.0, next_prologue=(0.9, eps_t))
            eps_k = eps
            torch.cuda.synchronize()
            if upload == "weights":                        # 64 doubles of the weights, rewritten with their own values
                host = e.wns[:64].cpu().numpy().copy()
                _lib.check(e.ops.lib, e.ops.lib.abcdez_memcpy_h2d(e.ops.ctx, e.wns.data_ptr(), host.ctypes.data, host.nbytes))
            elif upload == "distances":                    # the same for the distances the select ahead was made from
                host = e.delta[:64].cpu().numpy().copy()
                _lib.check(e.ops.lib, e.ops.lib.abcdez_memcpy_h2d(e.ops.ctx, e.delta.data_ptr(), host.ctypes.data, host.nbytes))
            out.append((eps, wnorm, n_alive, tuple(cnt[0])))
        return out, e.ops.smc_select_stats()

    base, (reused0, inline0) = run(None)
    same_w, (reused_w, inline_w) = run("weights")
    same_d, (reused_d, inline_d) = run("distances")
    assert base == same_w == same_d
    assert (reused0, inline0) == (3, 1) and (reused_w, inline_w) == (3, 1)          # untouched: every later prologue finds its select
    assert reused_d == 0 and inline_d == 4                                          # discarded every time, redone inline


@pytest.mark.parametrize("name,N,gens", [("normal1d", 5000, 60), ("mvn8", 2000, 40), ("normdu", 100, 100),
                                         ("quad2d_inf", 500, 80), ("normal1d", 60000, 70), ("normal1d_tight", 60000, 60),
                                         ("normal1d_tight", 3000, 60), ("gamma1d", 3000, 40), ("further8", 2000, 40),
                                         ("further5", 1500, 40)])
def test_abcdemc_end_to_end_parity(oracle, name, N, gens):
    """(normal1d_tight: fewer than 1 / 16 of the particles at or below eps_target for the first generations, so the rank pass is
    seen launching both sorts, then -- once eps_pop == eps_target bounds the tail -- only the radix sort (N = 60000) or only the
    LDS sort (N = 3000 <= 4096), then none at all: the better particles are drawn by rejection.  normal1d at eps 0.3 starts
    with 5 % of its particles there, 7 % after one generation: by rejection from the second generation on.)"""
    prior, sim, eps = models()[name]
    r = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=gens, verbose=False, rng=13)
    c = oracle.run_abcdemc(A.ModelSpec(prior, sim, seed=13), N, eps, gens)
    res = r.engine.result()
    assert r.nsims == c["nsims"] and r.reached_ϵ == c["reached_eps"]
    assert np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["C"], c["C"])
    both, small_only, long_only = r.engine.ops.mc_rank_stats()
    skipped = r.engine.ops.mc_draw_stats()
    if name == "normal1d_tight" and N == 60000:
        assert both >= 1 and long_only >= 3 and skipped >= 10, (both, small_only, long_only, skipped)
    if name == "normal1d_tight" and N == 3000:
        assert both >= 1 and small_only >= 3 and skipped >= 10, (both, small_only, long_only, skipped)
    if name == "normal1d":
        assert skipped >= gens - 10, (both, small_only, long_only, skipped)


def test_smoke_entry():
    import __graft_entry__ as g

    g.smoke()


# ---------------------------------------------------------------- edge cases: tiny populations, degenerate runs, ABI errors
@pytest.mark.parametrize("N", [6, 7, 64, 257])
def test_tiny_populations_end_to_end(oracle, N):
    """smallest legal populations (nparticles_min = ceil(3 d / min(alpha, delta_ess)) = 6 for d = 1, smc:234)"""
    prior, sim, eps = models()["normal1d"]
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=N)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, sim, seed=N), N, eps)
    res = r.engine.result()
    assert (r.logZ == c["logZ"] or (math.isnan(r.logZ) and math.isnan(c["logZ"]))) and r.iters == c["iters"]
    assert np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["alive"], c["alive"])
    if N >= 7:
        m = A.abcdemc(prior, sim, eps, None, nparticles=max(N, 5), generations=30, verbose=False, rng=N)
        cm = oracle.run_abcdemc(A.ModelSpec(prior, sim, seed=N), max(N, 5), eps, 30)
        assert np.array_equal(m.engine.result()["theta"], cm["theta"])


@pytest.mark.parametrize("d", [2, 3, 5, 8, 16, 17, 40, 64, 65, 96, 128, 200, 256])
def test_every_row_width(oracle, d):
    """d = 1..256 maps to ld = next power of two with zero padding (PAD prior family): every lane-group shape
    the library picks by default (beyond 64 parameters: 8 lanes of 16 or 32 components), mixed prior families across the
    components.  The reference has no upper limit on length(prior): src/abcdez_smc.jl:165,234."""
    fams = [A.Normal(0.1 * k, 1.0 + 0.05 * k) if k % 3 else A.Uniform(-4.0, 4.0) for k in range(d)]
    prior = A.Factored(*fams)
    sim = A.MVNormal(tuple(0.3 + 0.01 * k for k in range(d)))
    N = 2048
    spec = A.ModelSpec(prior, sim, seed=d)
    hip = PopulationEngine(spec, N, ops=HipOps(spec))
    orc = oracle.oracle_engine(spec, N)
    hip.init_population(); orc.init_population()
    assert_state_equal(hip, orc, f"init d={d}")
    hip.reset_weights(); orc.reset_weights()
    eps = orc.quantile_alive(0.6)
    assert hip.quantile_alive(0.6) == eps
    assert hip.smc_reweight(math.inf, eps) == orc.smc_reweight(math.inf, eps)
    hip.alive_compact(); orc.alive_compact()
    for _ in range(3):
        assert hip.smc_swarm(eps, 2.38 / math.sqrt(2 * d), 1e-5) == orc.smc_swarm(eps, 2.38 / math.sqrt(2 * d), 1e-5)
        assert_state_equal(hip, orc, f"sweep d={d}")
    assert (hip.state[0][:, d:] == 0).all()                 # padding components stay exactly zero
    hip.smc_resample(); orc.smc_resample()
    assert_state_equal(hip, orc, f"resample d={d}")


@pytest.mark.parametrize("d", [128, 256])
def test_wide_rows_end_to_end(oracle, d):
    """rows of 128 and 256 doubles (8 lanes of 16 / 32 components): whole abcdesmc and abcdemc runs of the d-dimensional Normal
    model, correlated prior included at d = 128, equal the oracle's bit for bit"""
    y = tuple(1.0 for k in range(d))
    if d == 128:
        rng = np.random.default_rng(5)
        Lm = np.tril(rng.normal(0, 0.05, (d, d)), -1) + np.diag(rng.uniform(0.8, 1.2, d))
        prior = A.MvNormal(rng.normal(0, 0.2, d), Lm @ Lm.T)
    else:
        prior = A.Factored(*[A.Normal(0.0, 1.0 + 0.001 * k) if k % 5 else A.Uniform(-5.0, 5.0) for k in range(d)])
    sim = A.MVNormal(y)
    N, eps = 2048, 0.93 * math.sqrt(3.0 * d)
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=d, nsims_max=10 ** 9)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, sim, seed=d), N, eps, nsims_max=10 ** 9)
    res = r.engine.result()
    assert r.logZ == c["logZ"] and r.iters == c["iters"] and r.nsims == c["nsims"] and r.iters >= 5
    assert int((r.Wns > 0).sum()) > 100 and math.isfinite(r.logZ)          # a run that got somewhere
    assert np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["C"], c["C"]) and np.array_equal(res["Wns"], c["Wns"])
    assert r.engine.ops.layout() == (d, 8, d // 8)
    m = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=12, verbose=False, rng=d + 1)
    cm = oracle.run_abcdemc(A.ModelSpec(prior, sim, seed=d + 1), N, eps, 12)
    assert np.array_equal(m.engine.result()["theta"], cm["theta"]) and np.array_equal(m.engine.result()["C"], cm["C"])


def test_run_that_ends_with_no_alive_particles(oracle):
    """a deterministic simulator with an unreachable target: the population dies out, the reference warns
    "No alive particles" and breaks (smc:375); GPU and oracle agree on where"""
    prior, sim = A.Normal(1, 0.2), A.DiracSquare(-5.0)      # |theta^2 + 1 + 5| >= 6 > eps
    with pytest.warns(UserWarning, match="No alive particles"):
        r = A.abcdesmc(prior, sim, 0.0, None, nparticles=200, verbose=False, rng=3, α=0.5)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, sim, seed=3), 200, 0.0, alpha=0.5)
    assert r.iters == c["iters"]
    assert np.array_equal(np.array(r.ϵs), c["eps_hist"])
    assert np.array_equal(r.engine.result()["theta"], c["theta"])


def test_c_abi_error_paths():
    """status codes + abcdez_last_error instead of exceptions or faults (include/abcdez_hip.h conventions)"""
    import ctypes as C

    from abcdez_amd import _lib

    prior, sim, _ = models()["mvn8"]
    spec = A.ModelSpec(prior, sim, seed=1)
    eng = PopulationEngine(spec, 1000, ops=HipOps(spec))
    eng.init_population()
    eng.reset_weights()
    ops, lib = eng.ops, eng.ops.lib
    s0, s1, lp, dl = eng.buf[0][0], eng.buf[1][0], eng.buf[eng.cur][1], eng.buf[eng.cur][2]
    b_in, b_out = eng.bits[eng.bc], eng.bits[1 - eng.bc]
    flags = torch.zeros(1064, dtype=torch.uint8, device="cuda")
    nacc, nsim = C.c_int64(), C.c_int64()

    def swarm(n_alive=1000, r_lo=0, r_hi=1000, out=b_out, slot1=s1, fl=flags, counters=True):
        return lib.abcdez_smc_swarm_packed(ops.ctx, b_in.data_ptr(), out.data_ptr(), n_alive, r_lo, r_hi, s0.data_ptr(),
                                           slot1.data_ptr(), lp.data_ptr(), dl.data_ptr(), fl.data_ptr() if fl is not None else None,
                                           5.0, 0.5, 1e-5, 0, C.byref(nacc) if counters else None, C.byref(nsim) if counters else None)

    assert swarm() == 0
    assert swarm(fl=None) == 0                                                          # flags are optional
    assert swarm(n_alive=2, r_hi=2) != 0 and b"3 alive" in lib.abcdez_last_error()       # donor loops need 3 (smc:119-126)
    assert swarm(r_lo=5, r_hi=2) != 0 and b"range" in lib.abcdez_last_error()
    assert swarm(r_hi=1001) != 0
    assert swarm(r_lo=32, r_hi=640) != 0 and b"multiples of 64" in lib.abcdez_last_error()   # sub-ranges own whole bitmap words
    assert swarm(r_lo=64, r_hi=640) == 0 and swarm(r_lo=64, r_hi=1000) == 0
    assert swarm(out=b_in) != 0 and b"must differ" in lib.abcdez_last_error()            # synchronous update: two bit arrays
    assert swarm(slot1=s0) != 0 and b"must differ" in lib.abcdez_last_error()
    assert swarm(counters=False) == 0                                                    # asynchronous form
    assert lib.abcdez_smc_swarm_packed(ops.ctx, None, None, 1000, 0, 1000, None, None, None, None, None, 5.0, 0.5, 1e-5, 0,
                                       C.byref(nacc), C.byref(nsim)) != 0
    assert b"null" in lib.abcdez_last_error()
    assert swarm() == 0
    assert lib.abcdez_smc_replay_packed(ops.ctx, b_in.data_ptr(), b_out.data_ptr(), 1000, 0, 1001, s0.data_ptr(), s1.data_ptr(),
                                        lp.data_ptr(), flags.data_ptr(), 0.5, 1e-5, 0, C.byref(nacc), C.byref(nsim)) != 0
    assert lib.abcdez_smc_replay_packed(ops.ctx, b_in.data_ptr(), b_out.data_ptr(), 1000, 0, 1000, s0.data_ptr(), s1.data_ptr(),
                                        lp.data_ptr(), flags.data_ptr(), 0.5, 1e-5, 0, C.byref(nacc), C.byref(nsim)) == 0
    assert nacc.value == int((flags[:1000] & 1).sum()) and nsim.value == 1000     # everything is this rank's: counted only
    # partition: the flags must describe a prefix
    assert lib.abcdez_smc_partition(ops.ctx, eng.alive.data_ptr(), 1000, 1000, 1001, b_in.data_ptr(), b_out.data_ptr(), s0.data_ptr(),
                                    s1.data_ptr(), lp.data_ptr(), dl.data_ptr(), eng.wns.data_ptr()) != 0
    eng.alive[::2] = 0                                  # 500 alive; a caller claiming n_new = 400 is caught at the next read-back
    assert lib.abcdez_smc_partition(ops.ctx, eng.alive.data_ptr(), 1000, 1000, 400, b_in.data_ptr(), b_out.data_ptr(), s0.data_ptr(),
                                    s1.data_ptr(), lp.data_ptr(), dl.data_ptr(), eng.wns.data_ptr()) == 0
    assert swarm(n_alive=400, r_hi=400) != 0 and b"prefix" in lib.abcdez_last_error()
    eng.alive.fill_(1)
    wn, es, na = C.c_double(), C.c_double(), C.c_int64()
    assert lib.abcdez_smc_reweight(ops.ctx, dl.data_ptr(), eng.wns.data_ptr(), eng.alive.data_ptr(), 1000, 1.0, -0.5,
                                   C.byref(wn), C.byref(es), C.byref(na)) != 0
    assert "ϵ ≥ 0.0".encode() in lib.abcdez_last_error()                                  # types.jl:30
    q = C.c_double()
    assert lib.abcdez_quantile_alive(ops.ctx, dl.data_ptr(), eng.alive.data_ptr(), 1000, -1, 1.5, C.byref(q), None, None) != 0
    eng.alive[::2] = 0                                  # 500 alive, but the caller claims 1000
    assert lib.abcdez_quantile_alive(ops.ctx, dl.data_ptr(), eng.alive.data_ptr(), 1000, 1000, 0.95, C.byref(q), None, None) != 0
    assert b"n_alive_hint" in lib.abcdez_last_error()
    assert lib.abcdez_quantile_alive(ops.ctx, dl.data_ptr(), eng.alive.data_ptr(), 1000, 500, 0.95, C.byref(q), None, None) == 0
    assert lib.abcdez_quantile_alive(ops.ctx, dl.data_ptr(), eng.alive.data_ptr(), 1000, -1, 0.95, C.byref(q), None, None) == 0
    eng.alive.fill_(1)
    assert lib.abcdez_ctx_set_lanes(ops.ctx, 3) != 0
    # a bad model is refused at context creation
    bad = A.ModelSpec(prior, sim, seed=1).cstruct(None)
    bad.n_data = 8
    ctx = C.c_void_p()
    assert lib.abcdez_ctx_create(C.byref(bad), 0, C.byref(ctx)) != 0 and b"data pointer" in lib.abcdez_last_error()
    with pytest.raises(_lib.AbcdezError):
        _lib.check(lib, -1)
    # the context still works after the errors
    assert swarm() == 0
    # blob entry points on a model created without blobs
    st = torch.zeros(1000, dtype=torch.int64, device="cuda")
    assert lib.abcdez_ctx_set_stamps(ops.ctx, st.data_ptr(), st.data_ptr()) != 0
    assert lib.abcdez_ctx_set_stamps(ops.ctx, st.data_ptr(), None) != 0
    st2 = torch.zeros_like(st)
    assert lib.abcdez_ctx_set_stamps(ops.ctx, st.data_ptr(), st2.data_ptr()) != 0 and b"n_blob = 0" in lib.abcdez_last_error()
    assert lib.abcdez_blob_eval(ops.ctx, s0.data_ptr(), st.data_ptr(), 1000, s1.data_ptr(), dl.data_ptr()) != 0
    assert lib.abcdez_ctx_set_stamps(ops.ctx, None, None) == 0
    host_data = np.ascontiguousarray(spec.data, dtype=np.float64)
    badblob = A.ModelSpec(prior, sim, seed=1).cstruct(host_data.ctypes.data)
    badblob.n_blob = 5                                                          # the MVN simulator's blob is d = 8 doubles
    assert lib.abcdez_ctx_create(C.byref(badblob), 0, C.byref(ctx)) != 0 and b"n_blob" in lib.abcdez_last_error()
    # the grouped sweeps and the asynchronous abcdemc generation validate their arguments like the calls they are made of
    nk, dn = (C.c_int64 * 20)(), C.c_int32()
    grp = lambda k, kmin, b1=None: lib.abcdez_smc_sweeps_packed(
        ops.ctx, b_in.data_ptr(), (b1 if b1 is not None else b_out).data_ptr(), 1000, s0.data_ptr(), s1.data_ptr(),
        lp.data_ptr(), dl.data_ptr(), 2.5, 0.5, 1e-5, 0, k, kmin, nk, nk, C.byref(dn))
    assert grp(0, 1.0) != 0 and grp(17, 1.0) != 0 and b"k_max" in lib.abcdez_last_error()
    assert grp(2, -1.0) != 0 and grp(2, 1.0, b_in) != 0
    tk = C.c_int64()
    assert lib.abcdez_mc_generation_async(ops.ctx, 1000, s0.data_ptr(), lp.data_ptr(), dl.data_ptr(), s0.data_ptr(), lp.data_ptr(),
                                          dl.data_ptr(), eng.inds.data_ptr(), dl.data_ptr(), eng.inds.data_ptr(), 0.0, 0.3, None, 1,
                                          0.5, 1e-5, 0, C.byref(tk)) != 0 and b"in/out arrays must differ" in lib.abcdez_last_error()
    assert lib.abcdez_mc_generation_wait(ops.ctx, 0, C.byref(tk), None, None, None, None) != 0
    # lane groups wider than 8 cannot own whole bitmap words
    spec32 = A.ModelSpec(*models()["mvn32"][:2], seed=1)
    e16 = PopulationEngine(spec32, 1000, ops=HipOps(spec32, lanes=16))
    e16.init_population(); e16.reset_weights()
    with pytest.raises(_lib.AbcdezError, match="at most 8 lanes"):
        e16.smc_swarm(9.0, 0.3, 1e-5)
    # 32-bit thread indices: 2^30 particles at 8 lanes per particle are 2^33 threads (the API's N <= 2^31 - 1 alone would let them
    # through and gid = tile * BLOCK + threadIdx.x would wrap).  The check precedes every launch: no allocation of that size needed,
    # the pointers are never dereferenced.
    e8 = HipOps(spec32, lanes=8)
    big, p = 1 << 30, s0.data_ptr()
    for rc in (lib.abcdez_smc_swarm_packed(e8.ctx, p, p + 8, big, 0, big, p, p + 8, p, p, None, 5.0, 0.5, 1e-5, 0, None, None),
               lib.abcdez_smc_sweeps_packed(e8.ctx, p, p + 8, big, p, p + 8, p, p, 5.0, 0.5, 1e-5, 0, 3, 1.0, nk, nk, C.byref(dn)),
               lib.abcdez_init(e8.ctx, p, p, p, 0, big), lib.abcdez_smc_group_begin(e8.ctx, big, 1.0),
               lib.abcdez_packed_gather(e8.ctx, p, big, p, p + 8, p),
               lib.abcdez_smc_partition(e8.ctx, p, big, big, big, p, p + 8, p, p + 8, p, p, p)):
        assert rc == -1 and b"2^32 threads" in lib.abcdez_last_error(), lib.abcdez_last_error()
    assert lib.abcdez_ctx_set_lanes(e8.ctx, 2) == 0          # 2^31 threads: this check passes (the arrays are checked by their owner)
    assert lib.abcdez_smc_group_begin(e8.ctx, big, 1.0) == 0 and lib.abcdez_smc_group_abort(e8.ctx) == 0


# ---------------------------------------------------------------- blobs: stamps + rebuild, product vs C restatement
@pytest.mark.parametrize("name,N", [("normal1d", 4000), ("mvn8", 3000), ("mvn32", 4096), ("mvn3", 1500), ("quad2d_inf", 500),
                                    ("normdu", 300), ("dirac", 200), ("mixture", 1500), ("socks", 3000)])
def test_blobs_parity(oracle, name, N):
    """blobs=True: the stamps the HIP kernels carry (through sweeps, partitions and resamplings) and the data
    abcdez_blob_eval rebuilds from them equal the oracle's bit for bit, for abcdesmc and for abcdemc; the rebuilt
    distances equal the stored ones (checked inside engine.result())."""
    import dataclasses
    prior, sim, eps = models()[name]
    sim = dataclasses.replace(sim, blobs=True)
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=11, nsims_max=10 ** 8)
    c = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=11, nsims_max=10 ** 8, engine=oracle.oracle_engine)
    assert r.engine.packed and type(r.engine.ops).__name__ == "HipOps" and r.blobs is not None
    assert r.iters == c.iters and np.array_equal(r.C, c.C, equal_nan=True)
    assert np.array_equal(r.blobs, c.blobs, equal_nan=True)
    assert same(r.engine.stamp[r.engine.cur], c.engine.stamp[c.engine.cur])
    m = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=12, verbose=False, rng=13)
    mo = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=12, verbose=False, rng=13, engine=oracle.oracle_engine)
    assert np.array_equal(m.C, mo.C) and np.array_equal(m.blobs, mo.blobs)


# ---------------------------------------------------------------- randomized models: every d, mixed prior families
def _further_family(k, rng):
    """one factor of the families beyond the five of the reference's tests, with random parameters"""
    u = lambda a, b: float(rng.uniform(a, b))                                                        # noqa: E731
    return [lambda: A.Exponential(u(0.3, 3.0)), lambda: A.Gamma(u(0.5, 5.0), u(0.3, 2.0)), lambda: A.LogNormal(u(-0.5, 0.5), u(0.2, 1.0)),
            lambda: A.Cauchy(u(-1.0, 1.0), u(0.3, 2.0)), lambda: A.Laplace(u(-1.0, 1.0), u(0.3, 2.0)),
            lambda: A.Weibull(u(0.7, 4.0), u(0.5, 3.0)), lambda: A.InverseGamma(u(1.5, 5.0), u(0.5, 3.0)),
            lambda: A.truncated(A.Normal(u(-1.0, 1.0), u(0.5, 2.0)), u(-2.0, -0.2), None if rng.random() < 0.4 else u(0.5, 3.0)),
            lambda: A.Logistic(u(-1.0, 1.0), u(0.3, 1.5)), lambda: A.TDist(u(1.0, 8.0)), lambda: A.Pareto(u(1.0, 4.0), u(0.2, 1.5)),
            lambda: A.Poisson(u(0.5, 12.0)), lambda: A.Binomial(int(rng.integers(1, 40)), u(0.1, 0.9)),
            # the wrapper families (ABZ_PRIOR_TRUNCATED / ABZ_PRIOR_MIXTURE): truncations of other parents, mixtures
            lambda: A.truncated(A.Gamma(u(1.0, 4.0), u(0.5, 1.5)), u(0.2, 0.8), None if rng.random() < 0.5 else u(3.0, 8.0)),
            lambda: A.truncated(A.Cauchy(u(-1.0, 1.0), u(0.5, 2.0)), u(-3.0, -1.0), u(1.0, 4.0)),
            lambda: A.truncated(A.Poisson(u(2.0, 8.0)), int(rng.integers(0, 3)), int(rng.integers(6, 14))),
            lambda: A.MixtureModel([A.Normal(u(-2.0, 0.0), u(0.3, 1.0)), A.Normal(u(0.5, 3.0), u(0.3, 1.5)), A.Laplace(u(-1.0, 1.0), u(0.5, 2.0))],
                                   [0.25, 0.45, 0.30]),
            lambda: A.MixtureModel([A.Poisson(u(0.5, 3.0)), A.Binomial(int(rng.integers(3, 20)), u(0.2, 0.8))], [0.35, 0.65]),
            lambda: A.Affine(A.TDist(u(2.0, 8.0)), u(-1.0, 2.0), u(0.3, 2.5)), lambda: A.Affine(A.Beta(u(1.0, 4.0), u(1.0, 4.0)), u(-2.0, 0.0), u(2.0, 6.0))][k]()


def _random_model(seed, nfam=5):
    rng = np.random.default_rng(seed)
    d = int(rng.integers(1, 33))
    fams = []
    for _ in range(d):
        k = int(rng.integers(0, nfam))
        if k >= 5:
            fams.append(_further_family(k - 5, rng))
            continue
        if k == 0:
            fams.append(A.Normal(float(rng.normal(0.5, 1.0)), float(rng.uniform(0.3, 2.0))))
        elif k == 1:
            a = float(rng.uniform(-3, 1))
            fams.append(A.Uniform(a, a + float(rng.uniform(1.0, 5.0))))
        elif k == 2:
            a = int(rng.integers(-3, 2))
            fams.append(A.DiscreteUniform(a, a + int(rng.integers(1, 6))))
        elif k == 3:
            fams.append(A.Beta(float(rng.uniform(0.6, 4.0)), float(rng.uniform(0.6, 4.0))))
        else:
            fams.append(A.NegativeBinomial(float(rng.uniform(0.7, 6.0)), float(rng.uniform(0.2, 0.8))))
    prior = fams[0] if d == 1 and rng.random() < 0.5 else A.Factored(*fams)
    y = tuple(float(v) for v in rng.normal(1.0, 0.5, d))
    kern = [A.IndicatorStrict0toϵ, A.Indicator0toϵ, A.Epa0toϵ, A.EpaStrict0toϵ][int(rng.integers(0, 4))]
    return d, prior, A.MVNormal(y, sigma=float(rng.uniform(0.5, 1.5)), blobs=bool(rng.random() < 0.5)), kern


def _random_model_case(oracle, seed, nfam):
    d, prior, sim, kern = _random_model(1000 + seed, nfam)
    N = max(64, int(math.ceil(3 * d / 0.5)) + 40) * 8
    rng = np.random.default_rng(seed)
    # a target the population reaches in a handful of generations: the 30 % quantile of the initial distances
    probe = oracle.oracle_engine(A.ModelSpec(prior, sim, kern, seed=seed + 1), N)
    probe.init_population()
    eps = probe.quantile_alive(0.3)
    kw = dict(nparticles=N, verbose=False, rng=seed + 1, ABCk=kern, nsims_max=10 ** 8, max_iters=25,
              Kmcmc=int(rng.integers(1, 5)), α=float(rng.uniform(0.5, 0.95)), δess=float(rng.uniform(0.2, 0.8)))
    r = A.abcdesmc(prior, sim, eps, None, **kw)
    c = A.abcdesmc(prior, sim, eps, None, engine=oracle.oracle_engine, **kw)
    assert type(r.engine.ops).__name__ == "HipOps" and r.engine.packed
    assert r.iters == c.iters and r.nsims == c.nsims
    assert r.logZ == c.logZ or (math.isnan(r.logZ) and math.isnan(c.logZ))
    assert list(r.ϵs) == list(c.ϵs) and np.array_equal(np.array(r.esss), np.array(c.esss), equal_nan=True)
    for k in ("P", "Wns", "C"):
        assert np.array_equal(getattr(r, k), getattr(c, k), equal_nan=True), (seed, d, k)
    if sim.blobs:
        assert np.array_equal(r.blobs, c.blobs, equal_nan=True)
    m = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=6, verbose=False, rng=seed + 2)
    mo = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=6, verbose=False, rng=seed + 2,
                   engine=oracle.oracle_engine)
    assert m.nsims == mo.nsims and np.array_equal(m.P, mo.P) and np.array_equal(m.C, mo.C)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_models_end_to_end_parity(oracle, seed):
    """40 seeded random models -- length(prior) 1..32 (every lane-group shape), all five prior families of the reference's tests
    mixed, all four ABC kernels, blobs on or off -- through the complete abcdesmc driver (row store) and a few abcdemc
    generations (double buffer): HIP == oracle bit for bit."""
    _random_model_case(oracle, seed, 5)


@pytest.mark.parametrize("seed", list(range(100, 124)))
def test_random_models_of_every_prior_family_end_to_end_parity(oracle, seed):
    """the same with all 18 univariate families the device knows (`prior::Distribution`, src/abcdez_smc.jl:165): heavy tails,
    half lines, truncations, counts -- their samplers at the initial population, their log-densities in every sweep"""
    _random_model_case(oracle, seed, 18)


@pytest.mark.parametrize("seed", list(range(200, 216)))
def test_random_models_with_truncated_and_mixture_priors_end_to_end_parity(oracle, seed):
    """and with the wrapper families among the factors -- truncated(d, lo, hi) of Gamma / Cauchy / Poisson parents, MixtureModel of
    continuous and of counting components, mu + sigma * d of a TDist / Beta (records in the model's ext table): rejection and inversion samplers at the initial
    population, log-sum-exp densities in every sweep, replayed log-priors -- HIP == oracle bit for bit"""
    _random_model_case(oracle, seed, 25)
