"""GPU parity of the PACKED population (abcdez_smc_partition / _swarm_packed / _replay_packed /
_resample_gather_packed / abcdez_packed_gather) against the oracle's restatement, bit for bit."""
import math

import numpy as np
import pytest
import torch

import abcdez_amd as A
from abcdez_amd import _lib
from abcdez_amd.engine import PACKED_ALIGN, HipOps, PopulationEngine

pytestmark = pytest.mark.gpu


def models():
    return {
        "normal1d": (A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0), 0.3),
        "uniform1d": (A.Uniform(-10.0, 10.0), A.Normal1D(3.0), 0.3),
        "mvn32": (A.Factored(*[A.Normal(0.0, 1.0) for _ in range(32)]), A.MVNormal(tuple([1.0] * 32)), 6.0),
        "mvn8": (A.Factored(*[A.Normal(0.0, 1.0) for _ in range(8)]), A.MVNormal(tuple([1.0] * 8)), 2.5),
        "mvn3": (A.Factored(A.Normal(0, 1), A.Uniform(-3, 3), A.Normal(1, 2)), A.MVNormal((0.5, 0.2, 1.0)), 0.8),
        "quad2d_inf": (A.Factored(A.Normal(0, 5), A.Normal(0, 5)), A.Quad2D(0.5), 0.01),
        "normdu": (A.Factored(A.Normal(1, 0.5), A.DiscreteUniform(1, 10)), A.NormalTimesDU(5.5), 0.01),
        "socks": (A.Factored(A.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), A.Beta(15, 2)), A.Socks(0, 11), 0.01),
        # further Distributions.jl families (abcdez_spec.h, ABZ_PRIOR_EXPONENTIAL ...)
        "further8": (A.Factored(A.Exponential(1.5), A.Gamma(2.5, 0.6), A.LogNormal(0.0, 0.5), A.Cauchy(1.0, 0.5), A.Poisson(2.0),
                                A.Weibull(1.8, 1.2), A.TDist(4.0), A.truncated(A.Normal(1.0, 2.0), 0.0, 4.0)),
                     A.MVNormal(tuple([1.0] * 8)), 2.5),
    }


def same(a, b):
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    if a.dtype == torch.float64:
        return torch.equal(a.view(torch.int64), b.view(torch.int64))
    return torch.equal(a, b)


def assert_equal(hip, orc, what):
    for k, nm in enumerate(("theta", "logpi", "delta")):
        assert same(hip.state[k], orc.state[k]), f"{what}: {nm} differs"
    assert same(hip.wns, orc.wns) and same(hip.alive, orc.alive), f"{what}: weights / flags differ"
    assert same(hip.bits[hip.bc], orc.bits[orc.bc]), f"{what}: slot bits differ"
    assert same(hip.buf[0][0], orc.buf[0][0]) and same(hip.buf[1][0], orc.buf[1][0]), f"{what}: a slot array differs"


@pytest.mark.parametrize("name,abck", [("normal1d", A.IndicatorStrict0toϵ), ("mvn32", A.Indicator0toϵ), ("quad2d_inf", A.IndicatorStrict0toϵ)])
def test_indicator_closed_forms_agree_with_the_general_reweight(oracle, name, abck):
    """Indicator kernel on uniform weights (abcdez_ctx_set_uniform_weights): the prologue's closed forms -- wnorm = n_new / n_old,
    Wns = 1 / n_new, ESS = n_new (smc:308-311, :8 in exact arithmetic) -- against the general path (the reference's statements with
    floating sums) on the same states: same epsilon, same survivors, same partition; weights / wnorm / ESS equal to rounding;
    and the fast path is what a run takes (reset_weights says the weights are uniform, the library keeps the flag)."""
    prior, sim, eps_target = models()[name]
    N = 20000
    spec = A.ModelSpec(prior, sim, abck, seed=11)
    fast = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    gen_ = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    for e in (fast, gen_):
        e.init_population()
        e.reset_weights()
    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps, eps_k = math.inf, math.inf
    for gen in range(12):
        gen_.ops.set_uniform_weights(False)                      # the general path, every generation
        a = fast.smc_prologue(0.9, eps, eps_target, eps_k, 0.5 * N)
        b = gen_.smc_prologue(0.9, eps, eps_target, eps_k, 0.5 * N)
        assert fast.ops.get_uniform_weights() and not gen_.ops.get_uniform_weights()
        assert a[0] == b[0] and a[3] == b[3] and a[4] == b[4]            # eps, n_alive, extrema: the very same
        assert abs(a[1] - b[1]) <= 1e-13 * b[1] and abs(a[2] - b[2]) <= 1e-10 * b[2]      # wnorm, ESS to rounding
        assert a[2] == 1.0 / (1.0 / a[3])                                 # ESS = n_new through the two divisions of the spec
        assert torch.equal(fast.alive, gen_.alive)
        wf, wg = fast.wns.cpu().numpy(), gen_.wns.cpu().numpy()
        al = fast.alive.cpu().numpy().astype(bool)
        assert (wf[al] == 1.0 / a[3]).all() and (wf[~al] == 0.0).all()
        assert np.allclose(wf, wg, rtol=1e-13, atol=0.0)
        eps, _, ess, n_alive, _ = a
        if n_alive > 0 and ess < 0.5 * N:
            fast.smc_resample(); gen_.smc_resample()
            n_alive = N
        fast.alive_compact(); gen_.alive_compact()
        for e in (fast, gen_):
            e.smc_sweeps(eps, g0, 1e-5, 2, 1.0)
        for k in range(3):
            assert same(fast.state[k], gen_.state[k])                     # the populations stay identical: only weights' last bits differ
        eps_k = eps
    assert fast.ops.fast_prologues() == 12 and gen_.ops.fast_prologues() == 0


@pytest.mark.parametrize("name", ["normal1d", "mvn32", "mvn3", "quad2d_inf", "socks"])
@pytest.mark.parametrize("abck", [A.IndicatorStrict0toϵ, A.Epa0toϵ])
def test_packed_fused_prologue_parity(oracle, name, abck):
    """abcdez_smc_prologue_packed (extrema, eps on the device, reweight, ESS, device-predicated partition: one call,
    one read-back) against the same statements issued one by one through the oracle"""
    prior, sim, eps_target = models()[name]
    N = 9001
    spec = A.ModelSpec(prior, sim, abck, seed=6)
    hip = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    orc = PopulationEngine(spec, N, ops=oracle.OracleOps(spec), storage="packed")
    for e in (hip, orc):
        e.init_population()
        e.reset_weights()
    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps, eps_k, resampled, parts = math.inf, math.inf, 0, 0
    for gen in range(16):
        want = orc.smc_prologue(0.85, eps, eps_target, eps_k, 0.5 * N)
        got = hip.smc_prologue(0.85, eps, eps_target, eps_k, 0.5 * N)
        assert got == want, (gen, got, want)
        assert hip.n_prev == orc.n_prev
        eps, _, ess, n_alive, _ = want
        assert_equal(hip, orc, f"gen {gen} prologue")
        if n_alive > 0 and ess < 0.5 * N:
            assert hip.n_prev > n_alive or n_alive == N               # not partitioned: the driver resamples
            hip.smc_resample(); orc.smc_resample()
            resampled += 1
            n_alive = N
        else:
            parts += 1
            assert bool(hip.alive[:n_alive].all()) and not bool(hip.alive[n_alive:].any())
        if n_alive < 3:
            break
        hip.alive_compact(); orc.alive_compact()
        for k in range(2):
            assert hip.smc_swarm(eps, g0, 1e-5) == orc.smc_swarm(eps, g0, 1e-5)
        assert_equal(hip, orc, f"gen {gen} sweeps")
        eps_k = eps
    assert resampled >= 1 and parts >= 3


@pytest.mark.parametrize("name", list(models()))
@pytest.mark.parametrize("abck", [A.IndicatorStrict0toϵ, A.Epa0toϵ])
def test_packed_generations_parity(oracle, name, abck):
    """every step of several generations -- quantile, reweight on the prefix, resampling when the ESS drops,
    partition, three sweeps -- leaves the packed device population equal to the oracle's, both slot arrays and
    the slot bits included"""
    prior, sim, eps_target = models()[name]
    N = 10007 if name != "mvn32" else 6151
    spec = A.ModelSpec(prior, sim, abck, seed=5)
    hip = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    orc = PopulationEngine(spec, N, ops=oracle.OracleOps(spec), storage="packed")
    for e in (hip, orc):
        e.init_population()
        e.reset_weights()
    assert_equal(hip, orc, "init")
    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps, eps_k, resampled = math.inf, math.inf, 0
    for gen in range(14):
        q = orc.quantile_alive(0.8)
        assert hip.quantile_alive(0.8) == q
        eps = max(min(q, eps), eps_target)
        rw = orc.smc_reweight(eps_k, eps)
        assert hip.smc_reweight(eps_k, eps) == rw
        n_alive = rw[2]
        if rw[1] < 0.5 * N:
            hip.smc_resample(); orc.smc_resample()
            assert same(hip.inds, orc.inds)
            assert_equal(hip, orc, f"gen {gen} resample")
            resampled += 1
            n_alive = N
        if n_alive < 3:
            break
        assert hip.alive_compact() == orc.alive_compact() == n_alive
        assert bool(hip.alive[:n_alive].all()) and not bool(hip.alive[n_alive:].any())      # the alive particles are a prefix
        assert_equal(hip, orc, f"gen {gen} partition")
        for k in range(3):
            assert hip.smc_swarm(eps, g0, 1e-5) == orc.smc_swarm(eps, g0, 1e-5), f"gen {gen} sweep {k}"
            assert_equal(hip, orc, f"gen {gen} sweep {k}")
        assert hip.extrema() == orc.extrema()
        eps_k = eps
    assert resampled >= 1


@pytest.mark.parametrize("name,N", [("normal1d", 5000), ("uniform1d", 5000), ("mvn8", 4096), ("mvn32", 8192),
                                    ("quad2d_inf", 500), ("normdu", 100), ("socks", 3000)])
@pytest.mark.parametrize("abck", [A.IndicatorStrict0toϵ, A.Indicator0toϵ, A.Epa0toϵ, A.EpaStrict0toϵ])
def test_packed_abcdesmc_end_to_end(oracle, name, N, abck):
    """the whole driver on the packed device population == the C restatement of the driver with orc_smc_partition"""
    if name == "mvn32" and abck is not A.IndicatorStrict0toϵ:
        pytest.skip("one kernel is enough at d=32")
    prior, sim, eps = models()[name]
    fac = lambda spec, n, pg, storage="packed": PopulationEngine(spec, n, pg, ops=HipOps(spec), storage="packed")
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, ABCk=abck, verbose=False, rng=11, nsims_max=10 ** 8, engine=fac)
    assert r.engine.packed and type(r.engine.ops).__name__ == "HipOps"
    c = oracle.run_abcdesmc(A.ModelSpec(prior, sim, abck, seed=11), N, eps, nsims_max=10 ** 8, packed=True)
    res = r.engine.result()
    assert r.iters == c["iters"] and r.nsims == c["nsims"]
    assert np.array_equal(np.array(r.ϵs), c["eps_hist"])
    assert r.logZ == c["logZ"] or (math.isnan(r.logZ) and math.isnan(c["logZ"]))
    for k in ("theta", "C", "Wns", "alive"):
        assert np.array_equal(res[k], c[k]), k


@pytest.mark.parametrize("d,shapes", [(32, (2, 4, 8)), (16, (1, 2, 4, 8)), (8, (1, 2, 4)), (3, (1, 2))])
def test_packed_lane_shapes(oracle, d, shapes):
    """several generations (reweight, resampling, partition, two sweeps each) for EVERY lane shape the dispatch table
    holds for this row width (up to 8 lanes: a block must own whole bitmap words): sweep counters, final population
    and weights equal the oracle's -- hence each other's"""
    prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
    spec = A.ModelSpec(prior, A.MVNormal(tuple([1.0] * d)), seed=5)
    N = 1 << 14

    def run(ops):
        e = PopulationEngine(spec, N, ops=ops, storage="packed")
        e.init_population(); e.reset_weights()
        eps, eps_k, out = math.inf, math.inf, []
        for _ in range(7):
            eps = min(e.quantile_alive(0.85), eps)
            _, ess, _ = e.smc_reweight(eps_k, eps)
            if ess < 0.5 * N:
                e.smc_resample()
            e.alive_compact()
            out += [e.smc_swarm(eps, 2.38 / math.sqrt(2 * d), 1e-5) for _ in range(2)]
            eps_k = eps
        return out, [t.cpu() for t in e.state], e.wns.cpu()

    ref = run(oracle.OracleOps(spec))
    for lanes in shapes:
        got = run(HipOps(spec, lanes=lanes))
        assert got[0] == ref[0], lanes
        for a, b in zip(got[1] + [got[2]], ref[1] + [ref[2]]):
            assert same(a, b), lanes


def wide_models():
    """rows spread over several lanes (the two-phase sweep, csrc/abz_kernels.h) with everything the d = 32 Normal-prior bench
    workload does not touch: flat priors (nothing is rejected on the prior ratio: all 64 hand-over slots of a tile fill up),
    bounded and discrete dimensions (out-of-support proposals, push_p), a correlated Normal prior"""
    rng = np.random.default_rng(3)
    Amat = rng.normal(size=(16, 16))
    cov = Amat @ Amat.T / 16 + 0.5 * np.eye(16)
    return {
        "flat16": (A.Factored(*[A.Uniform(-4.0, 4.0) for _ in range(16)]), A.MVNormal(tuple([0.5] * 16)), (2, 4, 8)),
        "mixed32": (A.Factored(*[(A.Normal(0.0, 1.0), A.Uniform(-2.5, 2.5), A.DiscreteUniform(-3, 3), A.Normal(0.5, 2.0))[k % 4]
                                 for k in range(32)]), A.MVNormal(tuple([0.25 * (k % 5) for k in range(32)])), (4, 8)),
        "mv16": (A.MvNormal(np.linspace(-0.5, 0.5, 16), cov), A.MVNormal(tuple([0.2] * 16)), (2, 4)),
        "narrow_normal32": (A.Factored(*[A.Normal(0.0, 0.3) for _ in range(32)]), A.MVNormal(tuple([1.0] * 32)), (4,)),
        # the further Distributions.jl families: half lines, heavy tails, a truncation, counts -- out-of-support proposals on
        # some components only, log-densities with logarithms and lgamma in both phases
        "further16": (A.Factored(A.Exponential(1.5), A.Gamma(2.5, 0.6), A.LogNormal(0.0, 0.5), A.Cauchy(1.0, 0.5), A.Laplace(1.0, 1.0),
                                 A.Weibull(1.8, 1.2), A.InverseGamma(3.0, 2.0), A.truncated(A.Normal(1.0, 2.0), 0.0, 4.0),
                                 A.Logistic(1.0, 0.5), A.TDist(4.0), A.Pareto(3.0, 0.5), A.Poisson(2.0), A.Binomial(6, 0.3),
                                 A.Geometric(0.4), A.Chisq(2.0), A.Normal(1.0, 1.0)),
                      A.MVNormal(tuple([1.0] * 16)), (2, 4, 8)),
        # one lane per particle, two phases all the same: the simulator is 160 RK4 steps and most proposals leave the prior's box
        "lv": (A.Factored(*[A.Uniform(0.0, 2.0)] * 4),
               A.LotkaVolterraRK4((1.0, 0.5, 1.46, 0.43, 1.77, 0.62, 1.52, 1.13, 0.95, 1.31, 0.66, 1.09, 0.61, 0.79, 0.75, 0.6, 0.9, 0.5),
                                  dt=0.05, steps_per_obs=20), (1,)),
        # the same simulator under an all-Normal prior: the PLAIN instantiations of the two launches, and parameters that go
        # negative -- trajectories that blow up (Inf / NaN running sums stay in their rounds and are rejected at the end)
        "lv_normal_prior": (A.Factored(A.Normal(1.0, 0.5), A.Normal(0.4, 0.3), A.Normal(1.0, 0.5), A.Normal(0.3, 0.3)),
                            A.LotkaVolterraRK4((1.0, 0.5, 1.46, 0.43, 1.77, 0.62, 1.52, 1.13, 0.95, 1.31, 0.66, 1.09, 0.61, 0.79, 0.75, 0.6, 0.9, 0.5),
                                               dt=0.05, steps_per_obs=20), (1,)),
    }


@pytest.mark.parametrize("name", list(wide_models()))
@pytest.mark.parametrize("abck", [A.IndicatorStrict0toϵ, A.Epa0toϵ])
def test_two_phase_sweep_on_wide_rows(oracle, name, abck):
    """The two-phase sweep (phase 1: proposal, log-prior, the acceptance ratio with the kernel term at its maximum; phase 2: the
    simulator for the proposals that may still be accepted, smc:137-145) against the oracle, which simulates every in-support
    proposal as the reference does: counters (nsims counts in-support proposals, simulated or not), both row slots, slot bits,
    log-priors, distances, weights and the flag bytes of a sharded sweep, bit for bit -- for flat priors (every proposal survives
    phase 1), bounded / discrete dimensions, a correlated Normal, a prior so narrow that almost nothing survives, with an
    indicator and an Epanechnikov kernel (K(di) != 0), in every lane shape the row width has."""
    prior, sim, shapes = wide_models()[name]
    spec = A.ModelSpec(prior, sim, seed=11, ABCk=abck)
    d = spec.d
    N = 1 << 13

    def run(ops):
        e = PopulationEngine(spec, N, ops=ops, storage="packed")
        e.init_population(); e.reset_weights()
        eps, eps_k, out = math.inf, math.inf, []
        for _ in range(6):
            eps = min(e.quantile_alive(0.85), eps)
            _, ess, _ = e.smc_reweight(eps_k, eps)
            if ess < 0.5 * N:
                e.smc_resample()
            n = e.alive_compact()
            out += [e.smc_swarm(eps, 2.38 / math.sqrt(2 * d), 1e-5) for _ in range(2)]
            eps_k = eps
        # one more sweep of a sub-range with flag bytes, as a rank of a sharded run does it
        fl = torch.zeros(N + PACKED_ALIGN, dtype=torch.uint8, device=e.device)
        lo, hi = min(64, n), n
        cur = e.buf[e.cur]
        e.ops.smc_swarm_packed(e.bits[e.bc], e.bits[1 - e.bc], n, lo, hi, e.buf[0][0], e.buf[1][0], cur[1], cur[2], fl, eps,
                               2.38 / math.sqrt(2 * d), 1e-5, e.sweep, want_counts=False)
        return out, [t.cpu() for t in (e.buf[0][0], e.buf[1][0], cur[1], cur[2], e.wns, e.bits[1 - e.bc], fl[:n])]

    ref = run(oracle.OracleOps(spec))
    assert sum(c[1] for c in ref[0]) > 0
    for lanes in shapes:
        got = run(HipOps(spec, lanes=lanes))
        assert got[0] == ref[0], (lanes, got[0], ref[0])
        for k, (a, b) in enumerate(zip(got[1], ref[1])):
            assert same(a, b), (lanes, k)


@pytest.mark.parametrize("name", ["mvn32", "mvn3", "normal1d", "further8"])
def test_packed_shard_sweep_plus_replay_equals_full_sweep(oracle, name):
    """what the ranks of a sharded run do, on one GPU: every "rank" sweeps its sub-range of the prefix on its own
    replica (flags out, no counters), the flags are merged, every rank replays the others' accepted proposals;
    all replicas must then equal the population one full sweep leaves -- rows of both slots, slot bits, log-priors
    -- and the replay's counters are the full sweep's"""
    prior, sim, eps_target = models()[name]
    N, G = 40000 + 777, 3
    spec = A.ModelSpec(prior, sim, seed=9)
    e = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    e.init_population(); e.reset_weights()
    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps = e.quantile_alive(0.7)
    e.smc_reweight(math.inf, eps)
    n = e.alive_compact()
    ops = e.ops
    chunk = -(-(-(-n // G)) // PACKED_ALIGN) * PACKED_ALIGN
    cuts = [min(r * chunk, n) for r in range(G + 1)]
    cur = e.buf[e.cur]

    def replica():
        return dict(s0=e.buf[0][0].clone(), s1=e.buf[1][0].clone(), lp=cur[1].clone(), dl=cur[2].clone(),
                    b=[e.bits[e.bc].clone(), e.bits[1 - e.bc].clone()], fl=torch.zeros(N + G * PACKED_ALIGN, dtype=torch.uint8, device="cuda"))

    for sweep in range(e.sweep, e.sweep + 3):
        full = replica()
        reps = [replica() for _ in range(G)]
        want = ops.smc_swarm_packed(full["b"][0], full["b"][1], n, 0, n, full["s0"], full["s1"], full["lp"], full["dl"], full["fl"],
                                    eps, g0, 1e-5, sweep)
        for r, rep in enumerate(reps):
            assert ops.smc_swarm_packed(rep["b"][0], rep["b"][1], n, cuts[r], cuts[r + 1], rep["s0"], rep["s1"], rep["lp"], rep["dl"],
                                        rep["fl"], eps, g0, 1e-5, sweep, want_counts=False) is None
        merged = torch.zeros_like(full["fl"])
        for r, rep in enumerate(reps):
            merged[cuts[r]:cuts[r + 1]] = rep["fl"][cuts[r]:cuts[r + 1]]
        assert torch.equal(merged[:n], full["fl"][:n])
        for r, rep in enumerate(reps):
            got = ops.smc_replay_packed(rep["b"][0], rep["b"][1], n, cuts[r], cuts[r + 1], rep["s0"], rep["s1"], rep["lp"], merged,
                                        g0, 1e-5, sweep)
            assert got == want
            for key in ("s0", "s1", "lp"):
                assert same(rep[key], full[key]), (sweep, r, key)
            assert same(rep["b"][1], full["b"][1])
            own = slice(cuts[r], cuts[r + 1])
            assert same(rep["dl"][own], full["dl"][own])
        # carry the full sweep's result into the engine for the next sweep
        e.buf[0][0].copy_(full["s0"]); e.buf[1][0].copy_(full["s1"]); cur[1].copy_(full["lp"]); cur[2].copy_(full["dl"])
        e.bits[1 - e.bc].copy_(full["b"][1])
        e.bc = 1 - e.bc


@pytest.mark.parametrize("name", ["mvn8", "normal1d"])
def test_sharded_group_of_sweeps_on_three_replicas(oracle, name):
    """abcdez_smc_group_begin / _replay / _publish / _end: three "ranks" (three contexts, three replicas on one GPU) enqueue the
    Kmcmc sweeps of a generation back to back -- own range, merged flags, replay with the device-side test of smc:352 -- and read
    back once.  Every rank must report the counters and the Ki of the single-GPU grouped sweeps (abcdez_smc_sweeps_packed) and
    hold its population, for thresholds that stop after the first sweep, in the middle, and never; misuse is an error."""
    prior, sim, eps_target = models()[name]
    N, G = 20000 + 333, 3
    spec = A.ModelSpec(prior, sim, seed=9)
    e = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    e.init_population(); e.reset_weights()
    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps = e.quantile_alive(0.7)
    e.smc_reweight(math.inf, eps)
    n = e.alive_compact()
    chunk = -(-(-(-n // G)) // PACKED_ALIGN) * PACKED_ALIGN
    cuts = [min(r * chunk, n) for r in range(G + 1)]
    cur = e.buf[e.cur]
    ranks = [HipOps(spec) for _ in range(G)]

    def replica():
        return dict(s0=e.buf[0][0].clone(), s1=e.buf[1][0].clone(), lp=cur[1].clone(), dl=cur[2].clone(),
                    b=[e.bits[e.bc].clone(), e.bits[1 - e.bc].clone()], fl=torch.zeros(N + G * PACKED_ALIGN, dtype=torch.uint8, device="cuda"))

    ops0 = ranks[0]
    with pytest.raises(_lib.AbcdezError, match="no group open"):
        ops0.smc_group_replay(e.bits[0], e.bits[1], 0, 64, e.buf[0][0], e.buf[1][0], cur[1], replica()["fl"], g0, 1e-5, 0)
    with pytest.raises(_lib.AbcdezError, match="no group open"):
        ops0.smc_group_end(3)
    ops0.smc_group_begin(n, 1.0)
    with pytest.raises(_lib.AbcdezError, match="already open"):
        ops0.smc_group_begin(n, 1.0)
    with pytest.raises(_lib.AbcdezError, match="no sweep"):
        ops0.smc_group_end(3)                                   # (closes nothing: a group without a sweep cannot be read back)
    full0 = replica()
    with pytest.raises(_lib.AbcdezError, match="group_end"):    # inside a group the counters come from the group's read-back
        ops0.smc_swarm_packed(full0["b"][0], full0["b"][1], n, 0, n, full0["s0"], full0["s1"], full0["lp"], full0["dl"], full0["fl"],
                              eps, g0, 1e-5, e.sweep)
    # an abandoned group (a collective or a launch failed between begin and end): abort closes it, and the SAME context takes a
    # new group and the counter-returning calls again (it used to stay poisoned for the life of the context)
    ops0.smc_group_abort()
    ops0.smc_group_abort()                                      # no group open: a no-op
    ab = replica()
    got = ops0.smc_swarm_packed(ab["b"][0], ab["b"][1], n, 0, n, ab["s0"], ab["s1"], ab["lp"], ab["dl"], None, eps, g0, 1e-5, e.sweep)
    ab2 = replica()
    assert got == e.ops.smc_swarm_packed(ab2["b"][0], ab2["b"][1], n, 0, n, ab2["s0"], ab2["s1"], ab2["lp"], ab2["dl"], None, eps, g0,
                                         1e-5, e.sweep)
    ops0.smc_group_begin(n, 1.0)
    ops0.smc_group_abort()

    seen = set()
    sweep0 = e.sweep
    for K, kmin in ((3, 9.0), (3, 0.0), (4, 0.6), (5, 0.3), (6, 1.0)):
        full = replica()
        want = e.ops.smc_sweeps_packed(full["b"][0], full["b"][1], n, full["s0"], full["s1"], full["lp"], full["dl"], eps, g0, 1e-5,
                                       sweep0, K, kmin)
        reps = [replica() for _ in range(G)]
        for ops in ranks:
            ops.smc_group_begin(n, kmin)
        for k in range(K):
            i, o = k & 1, 1 - (k & 1)
            for r, (ops, rep) in enumerate(zip(ranks, reps)):
                assert ops.smc_swarm_packed(rep["b"][i], rep["b"][o], n, cuts[r], cuts[r + 1], rep["s0"], rep["s1"], rep["lp"], rep["dl"],
                                            rep["fl"], eps, g0, 1e-5, sweep0 + k, want_counts=False) is None
            merged = torch.zeros_like(full["fl"])               # the all-gather of the flag chunks (every rank takes part, stopped or not)
            for r, rep in enumerate(reps):
                merged[cuts[r]:cuts[r + 1]] = rep["fl"][cuts[r]:cuts[r + 1]]
            for r, (ops, rep) in enumerate(zip(ranks, reps)):
                ops.smc_group_replay(rep["b"][i], rep["b"][o], cuts[r], cuts[r + 1], rep["s0"], rep["s1"], rep["lp"], merged, g0, 1e-5,
                                     sweep0 + k)
        for r, (ops, rep) in enumerate(zip(ranks, reps)):
            ops.smc_group_publish()
            got = ops.smc_group_end(K)
            assert got == want, (K, kmin, r, got, want)
            Ki = got[2]
            for key in ("s0", "s1", "lp"):
                assert same(rep[key], full[key]), (K, kmin, r, key)
            assert same(rep["b"][Ki & 1], full["b"][Ki & 1])
            own = slice(cuts[r], cuts[r + 1])
            assert same(rep["dl"][own], full["dl"][own])
        seen.add("first" if want[2] == 1 else "all" if want[2] == K else "middle")
    assert seen == {"first", "all", "middle"}, seen


@pytest.mark.parametrize("name", ["normal1d", "mvn32", "mvn3"])
def test_grouped_sweeps_equal_sweep_by_sweep(oracle, name):
    """abcdez_smc_sweeps_packed (the Kmcmc sweeps of a generation behind the device-side test of smc:352, one read-back)
    against the oracle swept one call at a time with the test on the host: same Ki, same per-sweep counters, same
    population -- for thresholds that stop after the first sweep, somewhere in the middle, and never"""
    prior, sim, eps_target = models()[name]
    N = 6000
    spec = A.ModelSpec(prior, sim, seed=11)
    hip = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    orc = PopulationEngine(spec, N, ops=oracle.OracleOps(spec), storage="packed")
    for e in (hip, orc):
        e.init_population()
        e.reset_weights()
    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps, eps_k, seen = math.inf, math.inf, set()
    plan = [(3, 1.0), (1, 0.0), (4, 0.0), (5, 0.25), (16, 0.6), (2, 5.0), (3, 0.05), (7, 0.9), (3, 1.0), (6, 0.4)]
    for gen, (K, kmin) in enumerate(plan):
        want = orc.smc_prologue(0.9, eps, eps_target, eps_k, 0.5 * N)
        assert hip.smc_prologue(0.9, eps, eps_target, eps_k, 0.5 * N) == want
        eps, _, ess, n_alive, _ = want
        if n_alive > 0 and ess < 0.5 * N:
            hip.smc_resample(); orc.smc_resample()
        hip.alive_compact(); orc.alive_compact()
        # next_prologue: the select of the next generation is enqueued behind the sweeps (abcdez_smc_select_ahead); the next
        # smc_prologue uses it when its arguments match (alpha = 0.9 here) and redoes it when they do not (every third generation)
        got = hip.smc_sweeps(eps, g0, 1e-5, K, kmin, next_prologue=(0.9 if gen % 3 else 0.7, eps_target))
        ref = orc.smc_sweeps(eps, g0, 1e-5, K, kmin)           # OracleOps has no grouped call: the engine's host loop
        assert got == ref, (gen, got, ref)
        assert hip.sweep == orc.sweep and hip.bc == orc.bc
        assert_equal(hip, orc, f"gen {gen} sweeps")
        seen.add("first" if got[2] == 1 and K > 1 else "all" if got[2] == K else "middle")
        eps_k = eps
    assert seen == {"first", "all", "middle"}
    # the single-sweep entry point still interleaves with groups (cumulative counters have one baseline)
    assert hip.smc_swarm(eps, g0, 1e-5) == orc.smc_swarm(eps, g0, 1e-5)
    assert hip.smc_sweeps(eps, g0, 1e-5, 2, 9.0) == orc.smc_sweeps(eps, g0, 1e-5, 2, 9.0)
    assert_equal(hip, orc, "mixed calls")


def test_plain_prior_path_is_chosen_only_for_unpadded_normal_rows(oracle):
    """all-Normal rows without padding take the sweep's short log-density path (abz_api.hip: prior_plain); a Uniform or
    discrete dimension, or a padded row (d < ld), takes the family dispatch -- both reproduce the oracle's generic
    evaluation (the parity tests above run mvn32 / mvn8 on the first path, mvn3 / socks / normdu on the second)"""
    N, g0 = 4096, 0.3
    for fams, sim in (([A.Normal(0.0, 1.0)] * 8, A.MVNormal((1.0,) * 8)),                     # plain
                      ([A.Normal(0.0, 1.0)] * 7 + [A.Uniform(-4.0, 4.0)], A.MVNormal((1.0,) * 8)),      # a Uniform: generic
                      ([A.Normal(0.0, 1.0)] * 7, A.MVNormal((1.0,) * 7))):                    # padded row: generic
        spec = A.ModelSpec(A.Factored(*fams), sim, seed=5)
        hip = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
        orc = PopulationEngine(spec, N, ops=oracle.OracleOps(spec), storage="packed")
        for e in (hip, orc):
            e.init_population(); e.reset_weights(); e.alive_compact()
        for k in range(3):
            assert hip.smc_swarm(3.0, g0, 1e-5) == orc.smc_swarm(3.0, g0, 1e-5)
        assert_equal(hip, orc, f"{len(fams)} dims")


@pytest.mark.parametrize("name", ["normal1d", "quad2d_inf", "socks"])
def test_narrow_rows_keep_one_slot_parity_over_the_prefix(oracle, name):
    """rows of one or two doubles are double-buffered (include/abcdez_hip.h): after every prologue, resampling and sweep all
    positions of the alive prefix name the same slot -- the invariant that lets the sweep take its donors' slot from the own
    position's bit -- and a rejected particle's row is present in BOTH slots right after a sweep"""
    prior, sim, eps_target = models()[name]
    N = 5000
    spec = A.ModelSpec(prior, sim, seed=21)
    assert spec.ld <= 2
    hip = PopulationEngine(spec, N, ops=HipOps(spec), storage="packed")
    orc = PopulationEngine(spec, N, ops=oracle.OracleOps(spec), storage="packed")
    for e in (hip, orc):
        e.init_population()
        e.reset_weights()

    def prefix_bits(e):
        w = e.bits[e.bc].cpu().numpy().astype(np.uint32)
        return ((w[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1)[: e.n_alive]

    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps, eps_k, resampled = math.inf, math.inf, 0
    for gen in range(14):
        want = orc.smc_prologue(0.8, eps, eps_target, eps_k, 0.5 * N)
        assert hip.smc_prologue(0.8, eps, eps_target, eps_k, 0.5 * N) == want
        eps, _, ess, n_alive, _ = want
        if n_alive > 0 and ess < 0.5 * N:
            hip.smc_resample(); orc.smc_resample()
            resampled += 1
        if hip.n_alive < 3:
            break
        hip.alive_compact(); orc.alive_compact()
        b = prefix_bits(hip)
        assert b.min() == b.max(), f"gen {gen}: the prefix is split over both slots before the sweeps"
        for k in range(2):
            assert hip.smc_swarm(eps, g0, 1e-5) == orc.smc_swarm(eps, g0, 1e-5)
            b2 = prefix_bits(hip)
            assert b2.min() == b2.max() and b2[0] != b[0], f"gen {gen} sweep {k}: every swept position moves to its other slot"
            b = b2
        assert_equal(hip, orc, f"gen {gen}")
        eps_k = eps
    assert resampled >= 1
