"""GPU tests at BASELINE.json's full sizes (where the oracle cannot replay the whole
population in seconds): size-independent properties + bit-exact spot checks of particle
ranges against the oracle, plus the remaining simulators (Lotka-Volterra RK4, Wiener) and
the committed spec vectors."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch

import abcdez_amd as A
from abcdez_amd.engine import HipOps, PopulationEngine

pytestmark = pytest.mark.gpu
GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")


def checksum(t: torch.Tensor) -> int:
    """order-independent 64-bit checksum of the raw bits"""
    v = t.contiguous().view(torch.uint8).view(-1)
    pad = (-v.numel()) % 8
    if pad:
        v = torch.cat([v, torch.zeros(pad, dtype=torch.uint8, device=v.device)])
    return int(v.view(torch.int64).sum().item())


def lv_model():
    g = json.load(open(os.path.join(GOLD_DIR, "lv_data.json")))
    prior = A.Factored(*[A.Uniform(0.0, 2.0)] * 4)
    # the fixture's own resolution = BASELINE.json configs[3] / SURVEY.md 8d-4: RK4 dt = 0.01, 100 steps between the
    # 16 observation times (T = 15, 1500 steps per particle-update)
    sim = A.LotkaVolterraRK4(tuple(g["obs"]), x0=g["x0"], y0=g["y0"], dt=g["dt"], steps_per_obs=g["steps_per_obs"],
                             noise=g["noise"])
    return prior, sim


def wiener_model():
    t = np.arange(31.0)
    tdata = np.sqrt(0.25 * t * t + 4.0 * t)
    return A.Factored(A.Uniform(0, 1), A.Uniform(0, 4)), A.WienerRMS(tuple(tdata))


@pytest.mark.parametrize("which", ["lv", "wiener"])
def test_remaining_simulators_parity(oracle, which):
    prior, sim = lv_model() if which == "lv" else wiener_model()
    N = 3000
    spec = A.ModelSpec(prior, sim, seed=17)
    hip = PopulationEngine(spec, N, ops=HipOps(spec))
    orc = oracle.oracle_engine(spec, N)
    hip.init_population(); orc.init_population()
    for k in range(3):
        assert torch.equal(hip.state[k].cpu().view(torch.int64), orc.state[k].view(torch.int64))
    assert torch.isfinite(hip.state[2]).all()           # LV blow-ups (Inf/NaN distances) were redrawn, init.jl:14
    hip.reset_weights(); orc.reset_weights()
    g0 = 2.38 / math.sqrt(2 * spec.d)
    eps_old = math.inf
    for gen in range(3):
        eps = orc.quantile_alive(0.7)
        assert hip.quantile_alive(0.7) == eps
        assert hip.smc_reweight(eps_old, eps) == orc.smc_reweight(eps_old, eps)
        hip.alive_compact(); orc.alive_compact()
        for _ in range(2):
            assert hip.smc_swarm(eps, g0, 1e-5) == orc.smc_swarm(eps, g0, 1e-5)
            for k in range(3):
                assert torch.equal(hip.state[k].cpu().view(torch.int64), orc.state[k].view(torch.int64))
        eps_old = eps
    # abcdemc path on the same model (double-buffered storage)
    hip = PopulationEngine(spec, N, ops=HipOps(spec), storage="classic")
    orc = oracle.oracle_engine(spec, N, storage="classic")
    hip.init_population(); orc.init_population()
    lo, hi = orc.extrema()
    hip.mc_rank_prepare(lo, hi); orc.mc_rank_prepare(lo, hi)
    assert torch.equal(hip.order.cpu(), orc.order)
    assert hip.mc_swarm(lo, 0.0, g0, 1e-5) == orc.mc_swarm(lo, 0.0, g0, 1e-5)
    assert torch.equal(hip.state[0].cpu().view(torch.int64), orc.state[0].view(torch.int64))


def test_lotka_volterra_end_to_end(oracle):
    """BASELINE.json configs[3] at a size the oracle replays: full driver, bit for bit, and the posterior
    concentrates near theta* = (1, 0.4, 1, 0.3)."""
    prior, sim = lv_model()
    N, eps = 4096, 1.2
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=5, nsims_max=10 ** 9)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, sim, seed=5), N, eps, nsims_max=10 ** 9)
    res = r.engine.result()
    assert r.logZ == c["logZ"] and np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["alive"], c["alive"])
    post = r.P[r.Wns > 0]
    assert np.all(np.abs(post.mean(0) - np.array([1.0, 0.4, 1.0, 0.3])) < 0.1)


def test_config4_lotka_volterra_full_size_properties_and_oracle_spot_checks(oracle):
    """BASELINE.json configs[3] at its stated workload on one GPU: Lotka-Volterra RK4, dt = 0.01, 1500 steps per
    particle-update, N = 2^20.  Whole-population properties every generation; one more sweep replayed bit for bit
    by the oracle for three 8192-position ranges from the same full input state; the initial population's first
    and last 2048 particles (prior draws, redraws of blown-up trajectories, first distances) against the oracle."""
    prior, sim = lv_model()
    assert sim.dt == 0.01 and sim.steps_per_obs == 100 and len(sim.obs) == 32
    N, d = 1 << 20, 4
    spec = A.ModelSpec(prior, sim, seed=11)
    eng = PopulationEngine(spec, N, ops=HipOps(spec))
    eng.init_population()
    th0, lp0, dl0 = (t.cpu() for t in eng.state)
    assert torch.isfinite(dl0).all() and torch.isfinite(lp0).all()                       # init.jl:14
    assert bool(((th0[:, :d] >= 0.0) & (th0[:, :d] <= 2.0)).all())                      # Uniform(0, 2) support
    m = oracle.OracleModel(spec)
    for i0 in (0, N - 2048):
        oth, olp, odl = torch.zeros_like(th0), torch.zeros_like(lp0), torch.zeros_like(dl0)
        assert oracle.lib().orc_init(m.ptr, oth.data_ptr(), olp.data_ptr(), odl.data_ptr(), i0, 2048) == 0
        sl = slice(i0, i0 + 2048)
        assert torch.equal(oth[sl].view(torch.int64), th0[sl].view(torch.int64))
        assert torch.equal(odl[sl].view(torch.int64), dl0[sl].view(torch.int64))
        assert torch.equal(olp[sl].view(torch.int64), lp0[sl].view(torch.int64))
    eng.reset_weights()
    loop = Loop(eng, d, 1.0)
    chk = Checks(oracle, spec, deep=True)
    for gen in range(3):
        loop.generation(chk)
    assert loop.eps < float(dl0.max())
    n = eng.n_alive
    nacc_all, nsim_all = oracle_spot_check_of_a_packed_sweep(oracle, spec, eng, loop.eps, loop.g0,
                                                             (0, (n // 2) // 64 * 64, (n - 8192) // 64 * 64))
    assert nsim_all < n                                                                 # bounded prior: some proposals fall outside


def further_model():
    """16 parameters over the further Distributions.jl families (include/abcdez_spec.h, ABZ_PRIOR_EXPONENTIAL ...), with scipy's
    distribution of the same parameters next to each"""
    from scipy import stats

    fams = [(A.Exponential(1.5), stats.expon(scale=1.5)), (A.Gamma(2.5, 0.6), stats.gamma(2.5, scale=0.6)),
            (A.LogNormal(0.0, 0.5), stats.lognorm(s=0.5)), (A.Cauchy(1.0, 0.5), stats.cauchy(1.0, 0.5)),
            (A.Laplace(1.0, 1.0), stats.laplace(1.0, 1.0)), (A.Weibull(1.8, 1.2), stats.weibull_min(1.8, scale=1.2)),
            (A.InverseGamma(3.0, 2.0), stats.invgamma(3.0, scale=2.0)),
            (A.truncated(A.Normal(1.0, 2.0), 0.0, 4.0), stats.truncnorm(-0.5, 1.5, 1.0, 2.0)),
            (A.Logistic(1.0, 0.5), stats.logistic(1.0, 0.5)), (A.TDist(4.0), stats.t(4.0)), (A.Pareto(3.0, 0.5), stats.pareto(3.0, scale=0.5)),
            (A.Poisson(2.0), stats.poisson(2.0)), (A.Binomial(6, 0.3), stats.binom(6, 0.3)), (A.Gamma(0.6, 2.0), stats.gamma(0.6, scale=2.0)),
            (A.Beta(2.0, 3.0), stats.beta(2.0, 3.0)), (A.Normal(1.0, 1.0), stats.norm(1.0, 1.0))]
    return fams, A.MVNormal(tuple([1.0] * 16))


def test_further_prior_families_full_size(oracle):
    """2^21 particles under a 16-parameter prior of the further families (`prior::Distribution`, src/abcdez_smc.jl:165): the device's
    initial population -- inversion / ratio / Marsaglia-Tsang / rejection samplers at particle indices up to 2^21 -- follows every
    marginal (Kolmogorov-Smirnov on 2 M draws; chi-square for the counts), first and last 2048 particles equal the oracle's bit for
    bit; generations with the whole-population properties, and one sweep replayed by the oracle on three position ranges."""
    from scipy import stats

    fams, sim = further_model()
    prior = A.Factored(*[f for f, _ in fams])
    N, d = 1 << 21, 16
    spec = A.ModelSpec(prior, sim, seed=23)
    eng = PopulationEngine(spec, N, ops=HipOps(spec))
    eng.init_population()
    th0, lp0, dl0 = (t.cpu() for t in eng.state)
    assert torch.isfinite(dl0).all() and torch.isfinite(lp0).all() and torch.isfinite(th0).all()
    m = oracle.OracleModel(spec)
    for i0 in (0, N - 2048):
        oth, olp, odl = torch.zeros_like(th0), torch.zeros_like(lp0), torch.zeros_like(dl0)
        assert oracle.lib().orc_init(m.ptr, oth.data_ptr(), olp.data_ptr(), odl.data_ptr(), i0, 2048) == 0
        sl = slice(i0, i0 + 2048)
        for a, b in ((oth, th0), (odl, dl0), (olp, lp0)):
            assert torch.equal(a[sl].view(torch.int64), b[sl].view(torch.int64))
    x = th0.numpy()
    for k, (f, ref) in enumerate(fams):
        col = x[:, k]
        if f.discrete:
            assert np.array_equal(col, np.rint(col))
            kk = col.astype(np.int64)
            obs = np.bincount(kk, minlength=40)[:40].astype(float)
            exp = ref.pmf(np.arange(40)) * N
            keep = exp > 50
            assert stats.chisquare(obs[keep] * exp[keep].sum() / obs[keep].sum(), exp[keep]).pvalue > 1e-4, type(f).__name__
        else:
            assert stats.kstest(col, ref.cdf).pvalue > 1e-4, (type(f).__name__, k)
    assert np.abs(np.corrcoef(x[:, :16].T) - np.eye(16)).max() < 0.01                   # one counter stream per component
    eng.reset_weights()
    loop = Loop(eng, d, 3.0)
    chk = Checks(oracle, spec, deep=True)
    for gen in range(3):
        loop.generation(chk)
    n = eng.n_alive
    nacc_all, nsim_all = oracle_spot_check_of_a_packed_sweep(oracle, spec, eng, loop.eps, loop.g0,
                                                             (0, (n // 2) // 64 * 64, (n - 8192) // 64 * 64))
    assert nsim_all < n                       # half lines, a truncation, counts: some proposals leave the support


def test_spec_vectors_on_gpu():
    """the committed oracle vectors (tests/golden/spec_vectors.json) reproduced by the HIP path"""
    gold = json.load(open(os.path.join(GOLD_DIR, "spec_vectors.json")))["abcdesmc_runs"]
    cases = {
        "normal1d_N2000": (A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), A.IndicatorStrict0toϵ),
        "mvn8_N2048": (A.Factored(*[A.Normal(0, 1)] * 8), A.MVNormal((1.0,) * 8), A.IndicatorStrict0toϵ),
        "mvn32_N4096": (A.Factored(*[A.Normal(0, 1)] * 32), A.MVNormal((1.0,) * 32), A.IndicatorStrict0toϵ),
        "normal1d_epa_N2000": (A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), A.Epa0toϵ),
    }
    for name, (prior, sim, K) in cases.items():
        g = gold[name]
        r = A.abcdesmc(prior, sim, g["eps_target"], None, nparticles=g["N"], ABCk=K, verbose=False, rng=g["seed"],
                       nsims_max=10 ** 9)
        assert r.logZ == float.fromhex(g["logZ"]), name
        assert r.iters == g["iters"] and r.nsims == g["nsims"]
        assert [float(v) for v in r.ϵs] == [float.fromhex(v) for v in g["eps_hist"]]
        res = r.engine.result()
        assert int(res["alive"].sum()) == g["n_alive"]
        assert float(np.sum(res["theta"][res["alive"]])) == float.fromhex(g["theta_sum"])
        assert float(np.sum(res["C"])) == float.fromhex(g["delta_sum"])


def oracle_spot_check_of_a_packed_sweep(oracle, spec, eng, eps, g0, ranges):
    """one more sweep on the device; the oracle replays the given position ranges of it from the same full input
    (both row slots, slot bits, log-priors, distances) and every touched byte of those ranges must agree"""
    b_in = eng.bits[eng.bc].cpu().contiguous()
    s0, s1 = eng.buf[0][0].cpu().contiguous(), eng.buf[1][0].cpu().contiguous()
    lp, dl = eng.buf[eng.cur][1].cpu().contiguous(), eng.buf[eng.cur][2].cpu().contiguous()
    n_alive, sweep = eng.n_alive, eng.sweep
    nacc_all, nsim_all = eng.smc_swarm(eps, g0, 1e-5)
    assert 0 < nacc_all <= nsim_all <= n_alive
    ops = oracle.OracleOps(spec)
    b_out = b_in.clone()
    flags = torch.zeros(eng.N, dtype=torch.uint8)
    for lo in ranges:
        hi = min(lo + 8192, n_alive)
        assert lo % 64 == 0 and lo < hi
        ops.smc_swarm_packed(b_in, b_out, n_alive, lo, hi, s0, s1, lp, dl, flags, eps, g0, 1e-5, sweep)
        sl = slice(lo, hi)
        for a, b in ((eng.buf[0][0], s0), (eng.buf[1][0], s1), (eng.buf[eng.cur][1], lp), (eng.buf[eng.cur][2], dl)):
            assert torch.equal(a[sl].cpu().view(torch.int64), b[sl].view(torch.int64))
        w = slice(lo // 32, (hi + 31) // 32 if hi % 32 == 0 else hi // 32)      # whole words of the range
        assert torch.equal(eng.bits[eng.bc][w].cpu(), b_out[w])
    return nacc_all, nsim_all


class Loop:
    """generation loop of smc:295-377 with hooks for the property checks"""

    def __init__(self, eng, d, eps_target):
        self.e, self.eps, self.eps_k, self.eps_target = eng, math.inf, math.inf, eps_target
        self.g0 = 2.38 / math.sqrt(2 * d)
        self.logZ = 0.0

    def generation(self, check):
        e = self.e
        q = e.quantile_alive(0.95)
        eps = max(min(q, self.eps), self.eps_target)
        assert eps <= self.eps                                           # eps sequence non-increasing (smc:301)
        wnorm, ess, n_alive = e.smc_reweight(self.eps_k, eps)
        assert 0 < wnorm <= 1.0 + 1e-12                                   # indicator kernel: fraction surviving
        self.logZ += math.log(wnorm)
        if ess < e.N * 0.5:
            w_before = e.wns.clone()
            e.smc_resample()
            n_alive = e.N
            check.resample(e, w_before)
        e.alive_compact()
        check.compaction(e, n_alive)
        for _ in range(3):
            before = [t.clone() for t in e.state] if check.deep else None
            nacc, nsim = e.smc_swarm(eps, self.g0, 1e-5)
            check.sweep(e, eps, before, nacc, nsim, n_alive)
        self.eps, self.eps_k = eps, eps


class Checks:
    def __init__(self, oracle, spec, deep=False):
        self.oracle, self.spec, self.deep = oracle, spec, deep

    def compaction(self, e, n_alive):
        """after the partition the alive particles are exactly the positions [0, n_alive)"""
        assert e.alive_indices().numel() == n_alive == e.n_prev
        assert int(e.alive.sum().item()) == n_alive
        assert bool(e.alive[:n_alive].all()) and not bool(e.alive[n_alive:].any())
        assert bool((e.wns[:n_alive] > 0).all()) and not bool((e.wns[n_alive:] != 0).any())
        assert torch.equal(e.bits[0], e.bits[1])                         # the two bit arrays agree between sweeps

    def resample(self, e, w_before):
        inds = e.inds.to(torch.int64)
        assert bool((inds[1:] >= inds[:-1]).all())                       # smc:45-54
        assert bool((w_before[inds] > 0).all())                          # zero weights never chosen
        assert bool(e.alive.all()) and float(e.wns[0]) == 1.0 / e.N

    def sweep(self, e, eps, before, nacc, nsim, n_alive):
        assert 0 <= nacc <= nsim <= n_alive
        alive = e.alive.bool()
        assert bool((e.state[2][alive] < eps).all())                     # strict indicator support (types:46)
        assert torch.isfinite(e.state[1][alive]).all()
        if before is not None:
            dead = ~alive
            for k in range(3):
                assert torch.equal(before[k][dead], e.state[k][dead])    # smc:114: dead particles untouched
            moved = (before[2] != e.state[2]) & alive
            assert abs(int(moved.sum().item()) - nacc) <= 2              # accepted <=> distance replaced (ties aside)


def test_config3_full_size_properties_and_oracle_spot_checks(oracle):
    """BASELINE.json configs[2]: d = 32 MVN, N = 2^22.  Properties on the whole population every generation;
    bit-exact comparison of three 8192-position ranges of one sweep against the oracle; identical checksums
    for two lane-group shapes."""
    d, N = 32, 1 << 22
    prior = A.Factored(*[A.Normal(0, 1)] * d)
    sim = A.MVNormal((1.0,) * d)
    spec = A.ModelSpec(prior, sim, seed=1)
    sums = []
    for lanes, deep in ((0, True), (8, False), (0, False)):
        eng = PopulationEngine(spec, N, ops=HipOps(spec, lanes=lanes))
        eng.init_population()
        eng.reset_weights()
        loop = Loop(eng, d, 6.0)
        chk = Checks(oracle, spec, deep=deep)
        for gen in range(4 if deep else 16):
            loop.generation(chk)
        if deep:
            n = eng.n_alive
            oracle_spot_check_of_a_packed_sweep(oracle, spec, eng, loop.eps, loop.g0, (0, (n // 2) // 64 * 64, (n - 8192) // 64 * 64))
        else:
            assert loop.eps < 9.9 and -2.5 < loop.logZ < 0.0
            sums.append((checksum(eng.state[0]), checksum(eng.state[2]), checksum(eng.wns), loop.logZ, loop.eps))
        del eng
        torch.cuda.empty_cache()
    assert sums[0] == sums[1]            # 8 lanes x 4 components == 4 lanes x 8 components


def test_config3_full_size_shard_sweep_plus_replay_equals_full_sweep():
    """The multi-GPU path at BASELINE.json configs[2] size (d = 32, N = 2^22), as rank 1 of 4 sees it: sweeping only
    its own chunk of the prefix and replaying the others' accepted proposals from the flag bytes must leave the
    replica exactly as the full sweep does (slot bits, both row slots, log-priors, the sweep counters), for three
    consecutive sweeps of a generation."""
    from abcdez_amd.engine import PACKED_ALIGN

    d, N, G, rank = 32, 1 << 22, 4, 1
    prior = A.Factored(*[A.Normal(0, 1)] * d)
    spec = A.ModelSpec(prior, A.MVNormal((1.0,) * d), seed=1)
    e = PopulationEngine(spec, N, ops=HipOps(spec))
    e.init_population()
    e.reset_weights()
    g0 = 2.38 / math.sqrt(2 * d)
    eps = math.inf
    for _ in range(3):
        eps = min(e.quantile_alive(0.95), eps)
        e.smc_reweight(math.inf, eps)
        e.alive_compact()
        for _ in range(2):
            e.smc_swarm(eps, g0, 1e-5)
    eps = min(e.quantile_alive(0.95), eps)
    e.smc_reweight(math.inf, eps)
    n = e.alive_compact()
    assert n < N
    ops = e.ops
    chunk = -(-(-(-n // G)) // PACKED_ALIGN) * PACKED_ALIGN
    r_lo, r_hi = rank * chunk, min((rank + 1) * chunk, n)
    cur = e.buf[e.cur]
    full = dict(s0=e.buf[0][0].clone(), s1=e.buf[1][0].clone(), lp=cur[1].clone(), dl=cur[2].clone())
    mine = dict(s0=e.buf[0][0], s1=e.buf[1][0], lp=cur[1], dl=cur[2])
    b_full = [e.bits[e.bc].clone(), e.bits[1 - e.bc].clone()]
    b_mine = [e.bits[e.bc].clone(), e.bits[1 - e.bc].clone()]
    flags = torch.zeros(N, dtype=torch.uint8, device="cuda")
    scratch = torch.zeros(N, dtype=torch.uint8, device="cuda")
    for k in range(3):
        sweep = e.sweep + k
        cnt = ops.smc_swarm_packed(b_full[0], b_full[1], n, 0, n, full["s0"], full["s1"], full["lp"], full["dl"], flags,
                                   eps, g0, 1e-5, sweep)
        assert ops.smc_swarm_packed(b_mine[0], b_mine[1], n, r_lo, r_hi, mine["s0"], mine["s1"], mine["lp"], mine["dl"],
                                    scratch, eps, g0, 1e-5, sweep, want_counts=False) is None
        assert torch.equal(scratch[r_lo:r_hi], flags[r_lo:r_hi])              # the owner's flags
        assert ops.smc_replay_packed(b_mine[0], b_mine[1], n, r_lo, r_hi, mine["s0"], mine["s1"], mine["lp"], flags, g0,
                                     1e-5, sweep) == cnt                      # global counters from the flags
        assert 0.05 * n < cnt[0] < 0.6 * n and cnt[1] == n                    # Normal prior: every proposal simulated
        assert torch.equal(b_mine[1], b_full[1])
        for key in ("s0", "s1", "lp"):                                        # both slots: current AND previous rows
            assert checksum(mine[key]) == checksum(full[key])
        assert torch.equal(mine["dl"][r_lo:r_hi], full["dl"][r_lo:r_hi])
        # next sweep: the other owners' distances arrive by all-gather in the real job
        mine["dl"].copy_(full["dl"])
        b_full.reverse(); b_mine.reverse()


def test_config2_abcdemc_one_million_particles(oracle):
    """BASELINE.json configs[1]: 1-D Normal, abcdemc, N = 2^20 (60 generations): posterior mean, never-worsening distances,
    sortedness of the per-generation order, agreement with the oracle on a strided sample of one sweep."""
    N = 1 << 20
    prior, sim = A.Normal(0, math.sqrt(10)), A.Normal1D(3.0)
    spec = A.ModelSpec(prior, sim, seed=3)
    eng = PopulationEngine(spec, N, ops=HipOps(spec), storage="classic")
    eng.init_population()
    g0 = 2.38 / math.sqrt(2)
    prev_max = math.inf
    for gen in range(60):
        lo, hi = eng.extrema()
        assert hi <= prev_max                                            # mc:54: max distance never grows
        prev_max = hi
        if hi > 0.3:
            eps_pop = max(0.3, lo)
            eng.mc_rank_prepare(eps_pop, hi)
            sd = eng.sorted_delta
            assert bool((sd[1:] >= sd[:-1]).all())
            assert torch.equal(eng.state[2][eng.order.to(torch.int64)].clamp_min(eps_pop), sd)
            assert torch.equal(eng.order.to(torch.int64).sort().values, torch.arange(N, device="cuda"))   # a permutation
            n_a = int((eng.state[2] <= eps_pop).sum())
            head = eng.order[:n_a].to(torch.int64)
            assert bool((head[1:] > head[:-1]).all()) and bool((eng.state[2][head] <= eps_pop).all())       # first block: index order
            if gen in (0, 5, 30):              # the whole enumeration against the oracle's (stable partition + qsort)
                oo, os_, oc = torch.zeros(N, dtype=torch.int32), torch.zeros(N, dtype=torch.float64), torch.zeros(N, dtype=torch.int32)
                oracle.OracleOps(spec).mc_rank_prepare(eng.state[2].cpu(), eps_pop, hi, oo, os_, oc)
                assert torch.equal(eng.order.cpu(), oo) and torch.equal(sd.cpu().view(torch.int64), os_.view(torch.int64))
                draws = eng.state[2].cpu() > eps_pop
                assert torch.equal(eng.rank_cnt.cpu()[draws], oc[draws])
        if gen == 5:
            th, lp, dl = (t.cpu().contiguous() for t in eng.state)
            order, sd_h = eng.order.cpu().contiguous(), eng.rank_cnt.cpu().contiguous()
            sweep = eng.sweep
            eps_pop = max(0.3, lo)
            eng.mc_swarm(eps_pop, 0.3, g0, 1e-5)
            nth, nlp, ndl = torch.zeros_like(th), torch.zeros_like(lp), torch.zeros_like(dl)
            m = oracle.OracleModel(spec)
            nsim = C.c_int64()
            oracle.lib().orc_mc_swarm(m.ptr, order.data_ptr(), sd_h.data_ptr(), N, th.data_ptr(), lp.data_ptr(),
                                      dl.data_ptr(), nth.data_ptr(), nlp.data_ptr(), ndl.data_ptr(), eps_pop, 0.3, g0,
                                      1e-5, 0, N, sweep, C.byref(nsim))
            assert torch.equal(eng.state[0].cpu().view(torch.int64), nth.view(torch.int64))
            assert torch.equal(eng.state[2].cpu().view(torch.int64), ndl.view(torch.int64))
        else:
            nsim_g, n_above, lo_g, hi_g = eng.mc_swarm(max(0.3, lo), 0.3, g0, 1e-5)
            assert (lo_g, hi_g) == eng.extrema() and n_above == eng.count_gt(0.3)      # reductions folded into the sweep
    assert eng.count_gt(0.3) <= 0.02 * N                                 # completion >= 98 % after 60 generations (mc:156)
    assert eng.extrema()[1] == 1.0790868611451696                        # the value the CPU oracle reaches: oracle.run_abcdemc(spec, 1 << 20, 0.3, 60)['C'].max(), seed 3
    post = eng.state[0][:, 0]
    assert abs(float(post.mean()) - 30 / 11) < 0.015     # finite-eps bias 0.0075 + Monte Carlo error
    assert abs(float(post.std()) - math.sqrt(10 / 11)) < 0.03


@pytest.mark.parametrize("logn,tol_logz,tol_bf", [(21, 0.02, 0.05), (23, 0.01, 0.025)])
def test_config5_two_model_evidence_large_n(logn, tol_logz, tol_bf):
    """BASELINE.json configs[4]: the two models of examples/minimal_example.jl (:10-56) at N = 2^21 and at the stated
    N = 2^23 = 8 M particles (on ONE GPU: at d = 1 the whole population is 0.4 GB): logZ against the exact finite-eps
    evidences, Bayes factor 2.1043, posterior means; the tolerances tighten with the population."""
    gold = json.load(open(os.path.join(GOLD_DIR, "reference_known_answers.json"), encoding="utf-8"))["analytic"]
    N = 1 << logn
    out = []
    for s2, key in ((10, "Z_exact_finite_eps_sigma2_10"), (100, "Z_exact_finite_eps_sigma2_100")):
        r = A.abcdesmc(A.Normal(0, math.sqrt(s2)), A.Normal1D(3.0), 0.3, None, nparticles=N, verbose=False, rng=s2,
                       nsims_max=10 ** 12)
        assert abs(r.logZ - gold[key]["logZ"]) < tol_logz, (s2, r.logZ)  # fp64 tolerance stated: |Δ logZ| < 0.02 / 0.01
        out.append(r.logZ)
        al = r.Wns > 0
        assert al.sum() >= 0.49 * N                                      # ESS < N / 2 resamples (smc:323-324): at least half are alive at the end
        assert abs(r.P[al].mean() - 3 * s2 / (s2 + 1)) < tol_logz
        assert float(r.C[al].max()) < 0.3                                # every alive distance inside the strict kernel
    assert abs(math.exp(out[0] - out[1]) - 2.1043) < tol_bf


def test_epanechnikov_kernel_at_scale(oracle):
    """continuous-weight kernels at scale.  Epa kernel, N = 2^20: evidence against the closed form
    (test/runtests.jl:334-336), weighted posterior through the public wsample_stratified, and the resampling
    indices of the continuous weights bit-identical to the oracle's."""
    gold = json.load(open(os.path.join(GOLD_DIR, "reference_known_answers.json"), encoding="utf-8"))["analytic"]
    N = 1 << 20
    r = A.abcdesmc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, None, nparticles=N, ABCk=A.Epa0toϵ, verbose=False,
                   rng=8, nsims_max=10 ** 12)
    Z = math.exp(r.logZ)
    assert abs(Z / gold["Z_epa_data3"]["value"] - 1) < 0.03
    w = r.Wns[r.Wns > 0]
    assert w.max() > 1.5 * w.min()                                      # genuinely continuous weights
    inds = A.wsample_stratified(r.Wns, rng=8, engine=r.engine)
    post = r.P[inds]
    assert abs(post.mean() - 30 / 11) < 0.02
    ref = np.zeros(N, dtype=np.uint32)
    wh = np.ascontiguousarray(r.Wns)
    oracle.lib().orc_wsample_stratified(8, wh.ctypes.data, N, 0, ref.ctypes.data)
    assert np.array_equal(inds, ref.astype(np.int64))


USER_NORMAL1D = """
__device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data, const double* sim_p,
                                abz_user_rng& rng) {
  double z0, z1;
  rng.normal_pair(z0, z1);
  const double x = __builtin_fma(sim_p[0], z0, theta[0]);
  return __builtin_fabs(x - data[0]);
}
"""
USER_QUAD2D = """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  double n1, n2;
  rng.normal_pair(n1, n2);
  const double u = rng.uniform();
  const double a = (th[0] + n1 * 0.01) - th[1] * th[1];
  const double b = (th[1] - 1.0) + n2 * 0.01;
  return (u < p[0]) ? ABZ_INF : 50.0 * (a * a) + b * b;
}
"""


@pytest.mark.parametrize("which", ["normal1d", "quad2d"])
def test_user_simulator_equals_builtin(oracle, which):
    """a simulator supplied as HIP source (hiprtc) must reproduce the built-in simulator it restates,
    and therefore the CPU oracle, bit for bit -- whole abcdesmc and abcdemc runs."""
    if which == "normal1d":
        prior, builtin, eps, N = A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, 5000
        user = A.UserSimulator(USER_NORMAL1D, params=(1.0,), data=(3.0,))
    else:
        prior, builtin, eps, N = A.Factored(A.Normal(0, 5), A.Normal(0, 5)), A.Quad2D(0.5), 0.01, 500
        user = A.UserSimulator(USER_QUAD2D, params=(0.5,))
    r = A.abcdesmc(prior, user, eps, None, nparticles=N, verbose=False, rng=31)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, builtin, seed=31), N, eps)
    res = r.engine.result()
    assert r.logZ == c["logZ"] and r.nsims == c["nsims"]
    assert np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["C"], c["C"])
    m = A.abcdemc(prior, user, eps, None, nparticles=N, generations=40, verbose=False, rng=32)
    cm = oracle.run_abcdemc(A.ModelSpec(prior, builtin, seed=32), N, eps, 40)
    assert np.array_equal(m.engine.result()["theta"], cm["theta"])


USER_LV = """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  const double a = th[0], b = th[1], c = th[2], e = th[3];
  double x = p[0], y = p[1];
  const double h = p[2], h2 = 0.5 * h, h6 = h / 6.0, sn = p[4];
  const int steps = (int)p[3], nobs = n_data / 2;
  double acc = 0.0;
  for (int jo = 0; jo < nobs; ++jo) {
    double z0, z1; rng.normal_pair(z0, z1);
    const double ex = abz_fma(sn, z0, x) - data[2 * jo], ey = abz_fma(sn, z1, y) - data[2 * jo + 1];
    acc = abz_fma(ex, ex, acc); acc = abz_fma(ey, ey, acc);
    if (jo + 1 == nobs) break;
    for (int s = 0; s < steps; ++s) {
      const double k1x = x * abz_fma(-b, y, a), k1y = y * abz_fma(e, x, -c);
      const double xa = abz_fma(h2, k1x, x), ya = abz_fma(h2, k1y, y);
      const double k2x = xa * abz_fma(-b, ya, a), k2y = ya * abz_fma(e, xa, -c);
      const double xb = abz_fma(h2, k2x, x), yb = abz_fma(h2, k2y, y);
      const double k3x = xb * abz_fma(-b, yb, a), k3y = yb * abz_fma(e, xb, -c);
      const double xc = abz_fma(h, k3x, x), yc = abz_fma(h, k3y, y);
      const double k4x = xc * abz_fma(-b, yc, a), k4y = yc * abz_fma(e, xc, -c);
      x = abz_fma(h6, abz_fma(2.0, k2x, k1x) + abz_fma(2.0, k3x, k4x), x);
      y = abz_fma(h6, abz_fma(2.0, k2y, k1y) + abz_fma(2.0, k3y, k4y), y);
    }
  }
  return abz_sqrt(acc);
}
"""


@pytest.mark.parametrize("sweep", ["two launches", "one kernel"])
def test_user_simulator_of_four_parameters_runs_the_two_phase_sweep(oracle, sweep, monkeypatch):
    """a user simulator with 3 to 8 parameters takes the two-phase sweep (csrc/abz_kernels.h: the simulator runs only for proposals
    that are in the prior's support and not already rejected on the prior ratio, densely packed): the Lotka-Volterra model restated
    as HIP source must equal the built-in simulator -- hence the oracle, which simulates every in-support proposal -- bit for bit.
    Rows of 4, 8 (and 16: tests/test_gpu_user_simulators.py) doubles sweep in two launches by default (phase 2 over a dense list of the surviving proposals);
    ABZ_USER_ONE_KERNEL=1, read when the context is created, keeps the one-kernel body: both must give the same bits."""
    monkeypatch.setenv("ABZ_USER_ONE_KERNEL", "1" if sweep == "one kernel" else "0")
    obs = (1.0, 0.5, 1.46, 0.43, 1.77, 0.62, 1.52, 1.13, 0.95, 1.31, 0.66, 1.09, 0.61, 0.79, 0.75, 0.6)
    prior = A.Factored(*[A.Uniform(0.0, 2.0)] * 4)
    builtin = A.LotkaVolterraRK4(obs, dt=0.05, steps_per_obs=10)
    user = A.UserSimulator(USER_LV, params=(builtin.x0, builtin.y0, builtin.dt, float(builtin.steps_per_obs), builtin.noise), data=obs)
    N, eps = 6000, 1.2
    r = A.abcdesmc(prior, user, eps, None, nparticles=N, verbose=False, rng=17)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, builtin, seed=17), N, eps)
    res = r.engine.result()
    assert r.logZ == c["logZ"] and r.nsims == c["nsims"] and r.iters == c["iters"]
    assert np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["C"], c["C"]) and np.array_equal(res["Wns"], c["Wns"])


USER_MVN6 = """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  double sq[8];
  for (int m = 0; m < 4; ++m) {
    double z[2];
    rng.normal_pair(z[0], z[1]);
    for (int c = 0; c < 2; ++c) {
      const int k = 2 * m + c;
      double v = 0.0;
      if (k < d) { const double x = abz_fma(p[0], z[c], th[k]); const double e = x - data[k]; v = e * e; }
      sq[k] = v;
    }
  }
  return abz_sqrt(abz_tree_sum_small(sq, 8));
}
"""


@pytest.mark.parametrize("sweep", ["two launches", "one kernel"])
def test_user_simulator_of_six_parameters_rows_of_eight_doubles(oracle, sweep, monkeypatch):
    """rows of eight doubles (six parameters and two padding components), one lane per particle: the d-dimensional Normal simulator
    restated as HIP source equals the built-in one -- the oracle -- bit for bit, in both forms of the sweep; a Gamma and a truncated
    Normal among the priors (out-of-support proposals; the further prior families inside the run-time-compiled kernels)"""
    monkeypatch.setenv("ABZ_USER_ONE_KERNEL", "1" if sweep == "one kernel" else "0")
    y = (1.0, 0.5, 0.8, 1.2, 0.3, 1.5)
    prior = A.Factored(A.Normal(0, 2), A.Gamma(2.0, 1.0), A.Uniform(-2, 3), A.truncated(A.Normal(1.0, 2.0), 0.0, None), A.Normal(0, 1),
                       A.LogNormal(0.0, 0.7))
    builtin = A.MVNormal(y, sigma=0.8)
    user = A.UserSimulator(USER_MVN6, params=(0.8,), data=y)
    N, eps = 8000, 1.5
    r = A.abcdesmc(prior, user, eps, None, nparticles=N, verbose=False, rng=19)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, builtin, seed=19), N, eps)
    res = r.engine.result()
    assert r.logZ == c["logZ"] and r.nsims == c["nsims"] and r.iters == c["iters"]
    assert np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["C"], c["C"]) and np.array_equal(res["Wns"], c["Wns"])


def test_user_simulator_with_blobs(oracle):
    """a user simulator that also defines abz_user_blob: its blobs equal the built-in Normal1D simulator's"""
    src = """
    __device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data, const double* sim_p,
                                    abz_user_rng& rng) {
      double z0, z1; rng.normal_pair(z0, z1);
      return __builtin_fabs(__builtin_fma(sim_p[0], z0, theta[0]) - data[0]);
    }
    __device__ void abz_user_blob(const double* theta, int d, const double* data, int n_data, const double* sim_p,
                                  abz_user_rng& rng, double* blob, int n_blob) {
      double z0, z1; rng.normal_pair(z0, z1);
      blob[0] = __builtin_fma(sim_p[0], z0, theta[0]);
    }
    """
    prior = A.Normal(0.0, math.sqrt(10.0))
    u = A.abcdesmc(prior, A.UserSimulator(src, params=(1.0,), data=(3.0,), n_blob=1), 0.3, None, nparticles=3000,
                   verbose=False, rng=4)
    b = A.abcdesmc(prior, A.Normal1D(3.0, blobs=True), 0.3, None, nparticles=3000, verbose=False, rng=4)
    assert np.array_equal(u.C, b.C) and np.array_equal(u.blobs, b.blobs)
    assert np.array_equal(np.abs(u.blobs - 3.0), u.C)
    with pytest.raises(ValueError, match="abz_user_blob"):
        A.UserSimulator("__device__ double abz_user_dist(){return 0;}", n_blob=2)


def test_user_simulator_compile_error_is_reported():
    from abcdez_amd import _lib

    bad = A.UserSimulator("__device__ double abz_user_dist(const double* t, int d, const double* a, int n, const double* p, "
                          "abz_user_rng& rng) { return undefined_symbol; }")
    with pytest.raises(_lib.AbcdezError, match="does not compile"):
        A.abcdesmc(A.Normal(0, 1), bad, 0.3, None, nparticles=100, verbose=False)


def test_user_simulator_new_model():
    """a model that is NOT built in: logistic growth observed with noise; the posterior finds the rate"""
    src = """
    __device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
      double x = p[0], acc = 0.0;
      for (int t = 0; t < n_data; ++t) {
        const double e = (x + p[1] * rng.normal()) - data[t];
        acc += e * e;
        x += th[0] * x * (1.0 - x / th[1]);
      }
      return sqrt(acc / n_data);
    }
    """
    r_true, K_true, x = 0.5, 10.0, 0.5
    data = []
    rng = np.random.default_rng(0)
    for _ in range(20):
        data.append(x + 0.05 * rng.normal())
        x += r_true * x * (1 - x / K_true)
    sim = A.UserSimulator(src, params=(0.5, 0.05), data=data)
    prior = A.Factored(A.Uniform(0, 2), A.Uniform(1, 30))
    r = A.abcdesmc(prior, sim, 0.15, None, nparticles=20000, verbose=False, rng=2, nsims_max=10 ** 9)
    post = r.P[r.Wns > 0]
    assert abs(post[:, 0].mean() - r_true) < 0.05 and abs(post[:, 1].mean() - K_true) < 0.5


def test_config3_full_run_logz_and_posterior():
    """BASELINE.json configs[2] run to its target: d = 32 MVN, N = 2^22, eps_target = 6 (about 160 generations).
    logZ against the closed form log P(chi'^2_32(16) < 18) = -8.1116 (tests/golden), posterior mean against the
    symmetry of the model (all 32 components exchangeable) and a 2^20-particle run with another seed."""
    gold = json.load(open(os.path.join(GOLD_DIR, "reference_known_answers.json"), encoding="utf-8"))["analytic"]
    prior = A.Factored(*[A.Normal(0, 1)] * 32)
    sim = A.MVNormal((1.0,) * 32)
    r = A.abcdesmc(prior, sim, 6.0, None, nparticles=1 << 22, verbose=False, rng=1, nsims_max=10 ** 12)
    assert r.ϵ == 6.0 and 120 < r.iters < 220
    assert abs(r.logZ - gold["Z_mvn32_eps6"]["logZ"]) < 0.02, r.logZ          # stated fp64 / Monte-Carlo tolerance
    al = r.Wns > 0
    m = r.P[al].mean(0)
    assert np.abs(m - m.mean()).max() < 0.015          # exchangeable components (particles are correlated: duplicates + 3 sweeps per generation)
    r2 = A.abcdesmc(prior, sim, 6.0, None, nparticles=1 << 20, verbose=False, rng=2, nsims_max=10 ** 12)
    m2 = r2.P[r2.Wns > 0].mean(0)
    assert abs(m.mean() - m2.mean()) < 0.01 and abs(r.logZ - r2.logZ) < 0.03
    print(f"config3 full run: iters={r.iters} nsims={r.nsims} logZ={r.logZ:.5f} (exact {gold['Z_mvn32_eps6']['logZ']:.5f}) "
          f"posterior mean per component={m.mean():.5f}")


def test_reference_minimal_example_script():
    """examples/minimal_example.py = the reference's examples/minimal_example.jl statement for statement: posteriors,
    evidences and model probabilities against the example's own analytic values (finite-eps evidence: within 10 %)"""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "minimal_example.py")
    spec = importlib.util.spec_from_file_location("minimal_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(nparticles=200_000, verbose=False)
    for m in ("1", "2"):
        mean, std = out["posterior" + m]
        mean_x, std_x = out["posterior" + m + "_exact"]
        assert abs(mean - mean_x) < 0.02 and abs(std - std_x) < 0.02          # eps = 0.3 widens the posterior slightly
        assert abs(out["evidence" + m] / out["evidence" + m + "_expected"] - 1.0) < 0.10
    assert abs(out["mposterior1"] - out["mposterior1_exact"]) < 0.02
    assert abs(out["mposterior1"] + out["mposterior2"] - 1.0) < 1e-12
