"""The oracle's two tiers agree: the *literal* restatements of the reference's sequential
functions (ref_*) and the order-free *spec* formulations the GPU reproduces bit for bit
(orc_*).  Exact where the formulation is exact, in law where only the consumption of
random numbers differs.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

import abcdez_amd as A
from abcdez_amd.model import ModelSpec


def _weights(N, rng, frac_zero=0.4, equal=False):
    w = np.ones(N) if equal else rng.random(N)
    w[rng.random(N) < frac_zero] = 0.0
    w[rng.integers(0, N)] = 1.0
    return w / w.sum()


@pytest.mark.parametrize("N", [1, 2, 5, 100, 1000, 4096, 50001, (1 << 24) + 3])
@pytest.mark.parametrize("equal", [False, True])
def test_stratified_fixed_point_equals_sequential_fp_walk(oracle, N, equal):
    """src/abcdez_smc.jl:15-56: the integer-cumsum pick equals the reference's sequential
    floating-point walk fed with the same uniforms (a disagreement needs a stratum draw
    within ~2^-40 of a cumulative-weight boundary)."""
    rng = np.random.default_rng(N + equal)
    L = oracle.lib()
    w = _weights(N, rng, equal=equal)
    for draw in range(3):
        inds = np.zeros(N, dtype=np.uint32)
        L.orc_wsample_stratified(77, w.ctypes.data, N, draw, inds.ctypes.data)
        u = np.zeros(N)
        L.orc_stratum_uniforms(77, N, draw, u.ctypes.data)
        ref = np.zeros(N, dtype=np.int64)
        L.ref_wsample_stratified(w.ctypes.data, N, u.ctypes.data, ref.ctypes.data)
        if N <= 100000:
            assert np.array_equal(inds.astype(np.int64), ref)
        else:
            # the sequential fp cumsum of 1.7e7 weights carries ~1e-10 of rounding error; a stratum draw
            # within that distance of a boundary picks the neighbouring (positive-weight) particle
            diff = inds.astype(np.int64) != ref
            assert diff.mean() < 1e-3      # equal weights: 1e7 sequential additions of one value drift systematically
            pos = np.flatnonzero(w > 0)
            rank = np.searchsorted(pos, np.stack([inds.astype(np.int64)[diff], ref[diff]]))
            assert (np.abs(rank[0] - rank[1]) <= 1).all()
        # invariants implied by smc:45-54
        assert (np.diff(ref) >= 0).all()
        assert (w[ref] > 0).all()
        counts = np.bincount(ref, minlength=N)
        # a weight spanning l strata widths is hit by between floor(l)-1 and ceil(l)+1 stratum draws
        # (SURVEY.md 8c-7 states {floor, ceil}; that is the systematic-resampling bound, not the stratified one)
        assert (counts >= np.floor(N * w - 1e-9) - 1).all() and (counts <= np.ceil(N * w + 1e-9) + 1).all()
        assert counts.sum() == N
    if equal:
        n_alive = int((w > 0).sum())
        counts = np.bincount(ref, minlength=N)[w > 0]
        assert counts.min() >= N // n_alive - 1 and counts.max() <= -(-N // n_alive) + 1


def test_stratified_is_unbiased(oracle):
    rng = np.random.default_rng(3)
    N = 64
    w = _weights(N, rng, frac_zero=0.3)
    tot = np.zeros(N)
    inds = np.zeros(N, dtype=np.uint32)
    D = 4000
    for draw in range(D):
        oracle.lib().orc_wsample_stratified(5, w.ctypes.data, N, draw, inds.ctypes.data)
        tot += np.bincount(inds, minlength=N)
    assert np.max(np.abs(tot / D - N * w)) < 0.05


@pytest.mark.parametrize("N", [1, 3, 2048, 2049, 100000])
def test_tree_sum_and_ess(oracle, N):
    rng = np.random.default_rng(N)
    x = rng.random(N) * rng.choice([1e-8, 1.0, 1e8], N)
    L = oracle.lib()
    s = L.orc_tree_sum(x.ctypes.data, N)
    assert abs(s - math.fsum(x)) <= 4e-16 * math.fsum(np.abs(x)) * max(1, math.log2(N + 1))
    w = x / x.sum()
    assert abs(L.orc_get_ess(w.ctypes.data, N) / L.ref_get_ess(w.ctypes.data, N) - 1) < 1e-13     # smc:8
    # permutation inside a fixed tree is not required to be invariant; zero padding is exact
    xp = np.concatenate([x, np.zeros(7)])
    assert L.orc_tree_sum(xp.ctypes.data, N + 7) == s or N + 7 > 2048 >= N or (N % 2048) + 7 > 2048


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_reweight_spec_equals_literal(oracle, kind):
    """src/abcdez_smc.jl:59-83 + :308-311"""
    rng = np.random.default_rng(kind)
    N = 5000
    L = oracle.lib()
    delta = rng.random(N) * 1.89          # alive particles lie inside the old kernel's support
    alive0 = (rng.random(N) < 0.8).astype(np.uint8)
    w0 = alive0 * rng.random(N)
    w0 /= w0.sum()
    a, wa = alive0.copy(), w0.copy()
    wn, ess, na = C.c_double(), C.c_double(), C.c_int64()
    L.orc_smc_reweight(kind, delta.ctypes.data, wa.ctypes.data, a.ctypes.data, N, 1.9, 1.2, C.byref(wn), C.byref(ess), C.byref(na))
    b, wb, ws = alive0.copy(), w0.copy(), np.ones(N)
    wn2 = C.c_double()
    L.ref_smc_reweight(kind, delta.ctypes.data, ws.ctypes.data, wb.ctypes.data, b.ctypes.data, N, 1.9, 1.2, C.byref(wn2))
    assert np.array_equal(a, b) and na.value == int(b.sum())
    assert abs(wn.value / wn2.value - 1) < 1e-13
    assert np.allclose(wa, wb, rtol=1e-13, atol=0)
    assert abs(wa.sum() - 1) < 1e-12


@pytest.mark.parametrize("N", [1, 2, 10, 1001])
@pytest.mark.parametrize("p", [0.0, 0.5, 0.95, 0.999])
def test_quantile_is_julia_type7(oracle, N, p):
    """Statistics.quantile default (type 7) == numpy's default 'linear' method."""
    rng = np.random.default_rng(N)
    d = rng.random(N)
    alive = (rng.random(N) < 0.7).astype(np.uint8)
    alive[0] = 1
    a, b = C.c_double(), C.c_double()
    q = oracle.lib().orc_quantile_alive(d.ctypes.data, alive.ctypes.data, N, p, C.byref(a), C.byref(b))
    want = np.quantile(d[alive > 0], p)
    assert abs(q - want) <= 4e-16 * max(1.0, abs(want))


def test_logprior_tree_vs_left_to_right(oracle):
    """priors.jl:40-46 sums left to right; the spec sums the same terms pairwise."""
    prior = A.Factored(*[A.Normal(0.1 * k, 1 + 0.1 * k) for k in range(32)])
    spec = ModelSpec(prior, A.MVNormal((1.0,) * 32))
    rng = np.random.default_rng(0)
    th = rng.normal(0, 2, (1000, 32))
    m = oracle.OracleModel(spec)
    a, b = np.zeros(1000), np.zeros(1000)
    oracle.lib().orc_logprior(m.ptr, th.ctypes.data, 1000, 0, a.ctypes.data)
    oracle.lib().orc_logprior(m.ptr, th.ctypes.data, 1000, 1, b.ctypes.data)
    assert np.allclose(a, b, rtol=1e-14, atol=1e-13)
    want = np.array([prior.logpdf(r) for r in th])
    assert np.allclose(b, want, rtol=1e-13, atol=1e-12)


@pytest.mark.parametrize("kernel", ["IndicatorStrict0toeps", "Epa0toeps"])
def test_sweep_spec_equals_literal_in_law(oracle, kernel):
    """abcdesmc_swarm! (smc:106-153): rank-skip donors + pairwise sums (spec) vs rejection
    loops around O(N) wsample scans + left-to-right sums (literal): same acceptance rate and
    same moments of the moved population, dead particles untouched in both.  With the Epanechnikov kernel the
    acceptance ratio of smc:140-141 carries log-kernel terms: both tiers (and the HIP kernel) evaluate it left to
    right, `((lp - lpi) + K(dp)) - K(di)`, as the reference does."""
    prior = A.Factored(*[A.Normal(0, 1)] * 4)
    spec = ModelSpec(prior, A.MVNormal((1.0,) * 4), ABCk=getattr(A, kernel), seed=9)
    N = 20000
    eng = oracle.oracle_engine(spec, N)
    eng.init_population()
    eng.reset_weights()
    eps = eng.quantile_alive(0.6)
    eng.smc_reweight(math.inf, eps)
    eng.alive_compact()
    th, lp, dl = (t.numpy().copy() for t in eng.state)
    alive = eng.alive.numpy().copy()
    g0 = 2.38 / math.sqrt(8)
    L, m = oracle.lib(), oracle.OracleModel(spec)
    aidx, arank = np.zeros(N, dtype=np.uint32), np.zeros(N, dtype=np.uint32)      # the alive list of the dense restatement
    assert L.orc_alive_compact(alive.ctypes.data, N, aidx.ctypes.data, arank.ctypes.data) == eng.n_alive
    assert np.array_equal(aidx[:eng.n_alive], np.arange(eng.n_alive))             # partitioned: alive rank r IS position r
    acc = []
    means = []
    for tier in ("spec", "literal"):
        nth, nlp, ndl = np.zeros_like(th), np.zeros_like(lp), np.zeros_like(dl)
        nacc, nsim = C.c_int64(), C.c_int64()
        if tier == "spec":
            L.orc_smc_swarm(m.ptr, aidx.ctypes.data, arank.ctypes.data, eng.n_alive, th.ctypes.data,
                            lp.ctypes.data, dl.ctypes.data, nth.ctypes.data, nlp.ctypes.data, ndl.ctypes.data, eps, g0,
                            1e-5, 0, N, 0, C.byref(nacc), C.byref(nsim))
        else:
            L.ref_smc_swarm(m.ptr, alive.ctypes.data, N, th.ctypes.data, lp.ctypes.data, dl.ctypes.data,
                            nth.ctypes.data, nlp.ctypes.data, ndl.ctypes.data, eps, g0, 1e-5, 0, C.byref(nacc),
                            C.byref(nsim))
        dead = alive == 0
        assert np.array_equal(nth[dead], th[dead]) and np.array_equal(ndl[dead], dl[dead])    # smc:114
        assert (ndl[~dead] < eps).all() if "Strict" in kernel else (ndl[~dead] <= eps).all()   # the kernel's support
        assert nsim.value == int((~dead).sum())          # Normal prior: every proposal is in support (smc:135-138)
        acc.append(nacc.value / (~dead).sum())
        means.append((nth[~dead].mean(0), nth[~dead].std(0)))
    assert abs(acc[0] - acc[1]) < 0.02
    assert np.allclose(means[0][0], means[1][0], atol=0.03) and np.allclose(means[0][1], means[1][1], atol=0.03)


def test_literal_wsample_is_uniform_over_alive(oracle):
    """wsample(rng, 1:N, alive) (smc:121): the O(N) cumulative scan returns the ceil(t)-th alive index."""
    rng = np.random.default_rng(4)
    N = 50
    alive = (rng.random(N) < 0.5).astype(np.uint8)
    alive[[3, 7, 20]] = 1
    prior = A.Normal(0, 1)
    spec = ModelSpec(prior, A.Normal1D(0.0), seed=1)
    th = rng.normal(size=(N, 1)); lp = np.zeros(N); dl = np.full(N, 0.1)
    # proposals: with eps = inf every alive particle moves to theta_i + gamma (theta_a - theta_b); recover (a, b)
    # statistically: donors must be alive -> the moved values are combinations of alive rows only.
    L, m = oracle.lib(), oracle.OracleModel(spec)
    nth, nlp, ndl = np.zeros_like(th), np.zeros_like(lp), np.zeros_like(dl)
    nacc, nsim = C.c_int64(), C.c_int64()
    L.ref_smc_swarm(m.ptr, alive.ctypes.data, N, th.ctypes.data, lp.ctypes.data, dl.ctypes.data, nth.ctypes.data,
                    nlp.ctypes.data, ndl.ctypes.data, math.inf, 1.0, 0.0, 0, C.byref(nacc), C.byref(nsim))
    diffs = {round(float(th[a, 0] - th[b, 0]), 12) for a in np.flatnonzero(alive) for b in np.flatnonzero(alive) if a != b}
    for i in np.flatnonzero(alive):
        step = round(float(nth[i, 0] - th[i, 0]), 12)
        assert step == 0.0 or step in diffs


def test_c_drivers_unbind_blob_stamps_of_an_earlier_engine(oracle):
    """The oracle binds the blob stamp arrays of a run in two globals (orc_set_stamps).  The complete C drivers carry
    no blobs: they must not write through a binding an earlier, blob-recording engine left behind (arrays that may
    be freed by then -- this crashed the GPU suite when a driver call followed a blobs=True model)."""
    prior = A.Normal(0.0, math.sqrt(10.0))
    a = np.full(64, 0xABCDEF, dtype=np.uint64)
    b = np.full(64, 0x123456, dtype=np.uint64)
    oracle.lib().orc_set_stamps(a.ctypes.data, b.ctypes.data)        # what a blobs=True engine leaves bound
    r1 = oracle.run_abcdesmc(ModelSpec(prior, A.Normal1D(3.0), seed=31), 64, 0.3)
    m1 = oracle.run_abcdemc(ModelSpec(prior, A.Normal1D(3.0), seed=32), 64, 0.3, 10)
    assert (a == 0xABCDEF).all() and (b == 0x123456).all()
    r2 = oracle.run_abcdesmc(ModelSpec(prior, A.Normal1D(3.0), seed=31), 64, 0.3)
    assert r1["logZ"] == r2["logZ"] and np.array_equal(r1["C"], r2["C"]) and m1["nsims"] > 0


@pytest.mark.parametrize("kind", ["continuous", "ties", "all_converged"])
def test_mc_better_particle_enumeration_is_the_reference_mask(oracle, kind):
    """abcdemc_swarm! draws s = rand(rng, (1:N)[Ds .<= Ds[i]]) (src/abcdez_mc.jl:23).  The spec enumerates that set as
    order[0 .. cnt), cnt = upper_bound(sorted_delta, Ds[i]), with order = (particles with Ds <= eps_pop in index
    order) ++ (the others by (Ds, index)).  For every particle that draws (Ds[i] > eps_pop, mc:19-20) the enumerated
    set must be exactly the reference's mask -- hence the same law for any fixed enumeration."""
    rng = np.random.default_rng(7)
    N = 3000
    if kind == "continuous":
        d = np.abs(rng.normal(3.0, 2.0, N))
    elif kind == "ties":
        d = rng.integers(0, 12, N).astype(np.float64)           # integer distances as in the Socks problem
    else:
        d = rng.uniform(0.0, 0.29, N)
    eps_target = 0.3
    eps_pop = max(eps_target, float(d.min()))
    order = np.zeros(N, dtype=np.uint32)
    sd = np.zeros(N)
    cnt_of = np.zeros(N, dtype=np.uint32)
    oracle.lib().orc_mc_rank_prepare(d.ctypes.data, N, eps_pop, order.ctypes.data, sd.ctypes.data, cnt_of.ctypes.data)
    assert np.array_equal(np.sort(order), np.arange(N))                       # a permutation
    assert np.all(sd[1:] >= sd[:-1])                                          # upper_bound is well defined
    assert np.array_equal(sd, np.maximum(d[order], eps_pop))
    n_a = int((d <= eps_pop).sum())
    assert np.all(np.diff(order[:n_a].astype(np.int64)) > 0) and np.all(d[order[:n_a]] <= eps_pop)
    tail = order[n_a:].astype(np.int64)
    assert np.all((d[tail][1:] > d[tail][:-1]) | ((d[tail][1:] == d[tail][:-1]) & (tail[1:] > tail[:-1])))
    for i in np.flatnonzero(d > eps_pop)[:400]:
        cnt = int(np.searchsorted(sd, d[i], side="right"))
        assert cnt == int(cnt_of[i])
        assert set(order[:cnt].tolist()) == set(np.flatnonzero(d <= d[i]).tolist())


@pytest.mark.parametrize("kind", ["continuous", "ties"])
def test_mc_better_particle_by_rejection_is_uniform_over_the_reference_mask(oracle, kind):
    """The second formulation of mc:23 (include/abcdez_spec.h, abz_mc_better_by_rejection): uniform j over ALL particles until
    Ds[j] <= Ds[i].  Every draw must lie in the reference's mask (1:N)[Ds .<= Ds[i]] and be uniform over it (chi-square) --
    also for a particle whose candidate set is small (more trials, same law), and it may be the particle itself."""
    from scipy import stats

    rng = np.random.default_rng(11)
    N = 97
    d = np.abs(rng.normal(3.0, 2.0, N)) if kind == "continuous" else rng.integers(0, 9, N).astype(np.float64)
    L = oracle.lib()
    by_rank = np.argsort(d, kind="stable")
    n = 60000
    for i in (int(by_rank[N - 1]), int(by_rank[N // 2]), int(by_rank[6])):
        mask = np.flatnonzero(d <= d[i])
        s = np.zeros(n, dtype=np.uint32)
        found = np.zeros(n, dtype=np.uint8)
        L.orc_mc_better_by_rejection(1234, d.ctypes.data, N, i, 0, n, s.ctypes.data, found.ctypes.data)
        assert found.all()
        assert np.isin(s, mask).all() and i in mask
        counts = np.bincount(s, minlength=N)[mask]
        assert counts.min() > 0
        assert stats.chisquare(counts).pvalue > 1e-4, (kind, i, len(mask))
    # a particle with NO better particle but itself (the minimum) keeps itself when the trials run out or finds itself
    i0 = int(by_rank[0]) if kind == "continuous" else None
    if i0 is not None:
        s = np.zeros(50, dtype=np.uint32); found = np.zeros(50, dtype=np.uint8)
        L.orc_mc_better_by_rejection(1234, d.ctypes.data, N, i0, 0, 50, s.ctypes.data, found.ctypes.data)
        assert (s == i0).all()


@pytest.mark.parametrize("draw", ["by_rank", "by_rejection"])
@pytest.mark.parametrize("kind", ["continuous", "ties"])
def test_mc_index_draws_spec_equals_literal_in_law(oracle, kind, draw):
    """abcdemc_swarm!'s three index draws (src/abcdez_mc.jl:18-32) -- the better particle s = rand(rng, (1:N)[Ds .<= Ds[i]])
    (:23) and the donors from the two rejection loops around rand(rng, 1:N) (:25-32: a != s; b != a, b != s) -- as the LITERAL
    tier restates them (ref_mc_draws: mask scanned in index order, one fresh integer per trial) against the SPEC tier
    (orc_mc_draws: s by rank or by rejection, donors by rank-skip from one Philox block).  The joint law of (s, a, b) is
    known in closed form -- s uniform over the mask, (a, b) uniform over the ordered pairs of distinct particles other than
    s -- so both tiers are tested against it cell by cell (chi-square) and against each other (two-sample)."""
    from scipy import stats

    rng = np.random.default_rng(21)
    N = 7
    d = np.abs(rng.normal(3.0, 2.0, N)) if kind == "continuous" else np.array([2.0, 0.0, 2.0, 1.0, 3.0, 1.0, 2.0])
    eps_target, eps_pop = 0.3, max(0.3, float(d.min()))
    spec = ModelSpec(A.Normal(0.0, 1.0), A.Normal1D(0.0), seed=77)
    L, m = oracle.lib(), oracle.OracleModel(spec)
    order, sd, cnt_of = np.zeros(N, dtype=np.uint32), np.zeros(N), np.zeros(N, dtype=np.uint32)
    L.orc_mc_rank_prepare(d.ctypes.data, N, eps_pop, order.ctypes.data, sd.ctypes.data, cnt_of.ctypes.data)
    by_rank = np.argsort(d, kind="stable")
    n = 40000
    u32 = lambda: C.c_uint32()
    for i in (int(by_rank[N - 1]), int(by_rank[N // 2]), int(by_rank[0])):      # largest mask, a middle one, a particle that keeps itself
        mask = np.flatnonzero(d <= d[i]) if d[i] > (eps_target if d[i] <= eps_target else eps_pop) else np.array([i])
        cells = {}
        for s_ in mask:
            for a_ in range(N):
                for b_ in range(N):
                    if a_ != s_ and b_ != a_ and b_ != s_:
                        cells[(int(s_), a_, b_)] = len(cells)
        tallies = []
        for tier in ("spec", "literal"):
            t = np.zeros(len(cells), dtype=np.int64)
            trials = 0
            for sweep in range(n):
                s_, a_, b_, x = u32(), u32(), u32(), u32()
                if tier == "spec":
                    ex = C.c_int()
                    L.orc_mc_draws(m.ptr, order.ctypes.data if draw == "by_rank" else None, cnt_of.ctypes.data if draw == "by_rank" else None,
                                   N, d.ctypes.data, eps_pop, eps_target, i, sweep, C.byref(s_), C.byref(a_), C.byref(b_), C.byref(ex))
                    assert ex.value == 0
                else:
                    L.ref_mc_draws(m.ptr, N, d.ctypes.data, eps_pop, eps_target, i, sweep, C.byref(s_), C.byref(a_), C.byref(b_), C.byref(x))
                    trials += x.value
                t[cells[(s_.value, a_.value, b_.value)]] += 1          # KeyError = a draw outside the reference's support
            assert t.min() > 0
            assert stats.chisquare(t).pvalue > 1e-4, (tier, draw, kind, i)
            tallies.append(t)
            if tier == "literal":         # the loops of mc:26-32 need 1 / (1 - 1/N) + 1 / (1 - 2/N) trials on average
                assert abs(trials / n - (N / (N - 1) + N / (N - 2))) < 0.03
        assert stats.chi2_contingency(np.stack(tallies))[1] > 1e-4, (draw, kind, i)


@pytest.mark.parametrize("generation", [2, 12])
def test_mc_sweep_spec_equals_literal_in_law(oracle, generation):
    """abcdemc_swarm! as a whole (src/abcdez_mc.jl:5-61): the spec tier through BOTH formulations of mc:23's draw (by rank /
    by rejection) against the literal tier (ref_mc_swarm: index-order mask, rejection loops for a and b, left-to-right
    log-prior, libm log of an unconditionally drawn uniform, mc:43) on the same population mid-run -- generation 2: nearly
    every particle still draws a better particle; generation 12: about half have arrived.  Same fraction simulated (mc:43-44),
    same fraction moved (mc:54), same moments of the moved population, converged particles never leave eps_target (mc:19,52)."""
    prior, sim = A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0)
    N, eps_target = 40000, 0.3
    spec = ModelSpec(prior, sim, seed=13)
    eng = oracle.oracle_engine(spec, N, storage="classic")
    eng.init_population()
    g0 = 2.38 / math.sqrt(2.0)
    lo, hi = eng.extrema()
    for _ in range(generation):
        nsim, ngt, lo, hi = eng.mc_generation(max(eps_target, lo), eps_target, hi, g0, 1e-5)
    th, lp, dl = (t.numpy().copy() for t in eng.state)
    eps_pop = max(eps_target, float(dl.min()))
    L, m = oracle.lib(), oracle.OracleModel(spec)
    order, sd, cnt_of = np.zeros(N, dtype=np.uint32), np.zeros(N), np.zeros(N, dtype=np.uint32)
    L.orc_mc_rank_prepare(dl.ctypes.data, N, eps_pop, order.ctypes.data, sd.ctypes.data, cnt_of.ctypes.data)
    arrived = dl <= eps_target
    out = {}
    for tier in ("by_rank", "by_rejection", "literal"):
        nth, nlp, ndl = np.zeros_like(th), np.zeros_like(lp), np.zeros_like(dl)
        nsim = C.c_int64()
        if tier == "literal":
            L.ref_mc_swarm(m.ptr, N, th.ctypes.data, lp.ctypes.data, dl.ctypes.data, nth.ctypes.data, nlp.ctypes.data,
                           ndl.ctypes.data, eps_pop, eps_target, g0, 1e-5, 1000, C.byref(nsim))
        else:
            r = tier == "by_rank"
            L.orc_mc_swarm(m.ptr, order.ctypes.data if r else None, cnt_of.ctypes.data if r else None, N, th.ctypes.data,
                           lp.ctypes.data, dl.ctypes.data, nth.ctypes.data, nlp.ctypes.data, ndl.ctypes.data, eps_pop,
                           eps_target, g0, 1e-5, 0, N, 1000, C.byref(nsim))
        moved = ndl != dl
        assert (ndl[arrived] <= eps_target).all()                         # mc:19,54: max(eps_target, Ds[i]) = eps_target
        assert (ndl <= np.maximum(dl, eps_pop)).all()                     # mc:54
        assert np.array_equal(nth[~moved], th[~moved]) and np.array_equal(nlp[~moved], lp[~moved])
        out[tier] = (nsim.value / N, moved.mean(), moved[arrived].mean(), moved[~arrived].mean(), nth.mean(), nth.std(),
                     ndl.mean(), float((ndl > eps_target).mean()))
    # binomial / sampling tolerances at N = 40000: 5 sigma of a proportion is < 0.0125, of a mean of thetas (sd ~ 1.5) 0.04
    for tier in ("by_rank", "by_rejection"):
        a, b = out[tier], out["literal"]
        assert abs(a[0] - b[0]) < 0.0125 and abs(a[1] - b[1]) < 0.0125, (tier, a, b)
        assert abs(a[2] - b[2]) < 0.02 and abs(a[3] - b[3]) < 0.02, (tier, a, b)
        assert abs(a[4] - b[4]) < 0.05 and abs(a[5] - b[5]) < 0.05 and abs(a[6] - b[6]) < 0.05, (tier, a, b)
        assert abs(a[7] - b[7]) < 0.0125, (tier, a, b)


def test_mc_generations_switch_to_rejection_by_the_rule_and_only_once(oracle):
    """The rule of include/abcdez_spec.h: a generation draws its better particles by rejection iff at least 1 / 16 of the
    particles it reads lie at or below eps_target.  Driven through the product's host code on the oracle: every generation takes the
    branch the rule names, rank passes stop for good at the switch, and the C driver (oracle/abcdez_oracle_driver.c), which
    counts for itself, arrives at the same population bit for bit."""
    import abcdez_amd.engine as E
    from oracle.oracle import OracleOps

    prior, sim = A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0)
    N, G, eps = 3000, 60, 0.02
    log = []

    class Spy(OracleOps):
        def mc_rank_prepare(self, delta, *a):
            log.append(["rank", None])
            return super().mc_rank_prepare(delta, *a)

        def mc_swarm(self, order, cnt, cur, nxt, eps_pop, eps_target, *a):
            n_above = int((cur[2] > eps_target).sum())
            log.append(["sweep", (order is None, n_above)])
            return super().mc_swarm(order, cnt, cur, nxt, eps_pop, eps_target, *a)

    eng = lambda spec, n, pg, storage="classic": E.PopulationEngine(spec, n, pg, ops=Spy(spec), storage=storage)
    r = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=G, verbose=False, rng=5, engine=eng)
    sweeps = [v for k, v in log if k == "sweep"]
    assert len(sweeps) == G
    L = oracle.lib()
    for by_rejection, n_above in sweeps:
        assert by_rejection == bool(L.orc_mc_draws_by_rejection(n_above, N)) == (16 * (N - n_above) >= N)
    modes = [m for m, _ in sweeps]
    assert modes[0] is False and modes[-1] is True and modes == sorted(modes)         # one switch, never back
    above = [n for _, n in sweeps]
    assert all(b <= a for a, b in zip(above, above[1:]))                                # the count the rule reads never grows
    # no rank pass after the switch; one before every sweep until then
    kinds = [k for k, _ in log]
    first_rej = modes.index(True)
    assert kinds.count("rank") == first_rej
    c = oracle.run_abcdemc(ModelSpec(prior, sim, seed=5), N, eps, G)
    assert np.array_equal(np.asarray(r.C), c["C"]) and r.nsims == c["nsims"]


@pytest.mark.parametrize("name", ["mvn8", "normal1d", "uniform1d", "quad2d"])
@pytest.mark.parametrize("abck", [A.IndicatorStrict0toϵ, A.Epa0toϵ])
def test_packed_restatement_equals_dense_driver(oracle, name, abck):
    """two restatements of the spec against each other: the product's host driver on the oracle's PACKED population
    (two row slots + slot bits per position, partition by swaps: orc_packed_partition / orc_smc_swarm_packed / ...)
    and the dense C driver (plain arrays, orc_smc_partition + orc_smc_swarm) give the same run, bit for bit"""
    cases = {
        "mvn8": (A.Factored(*[A.Normal(0, 1)] * 8), A.MVNormal((1.0,) * 8), 2.5, 4096),
        "normal1d": (A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, 3000),
        "uniform1d": (A.Uniform(-10, 10), A.Normal1D(3.0), 0.3, 2500),
        "quad2d": (A.Factored(A.Normal(0, 5), A.Normal(0, 5)), A.Quad2D(0.5), 0.05, 700),
    }
    prior, sim, eps, N = cases[name]
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, ABCk=abck, verbose=False, rng=11, nsims_max=10 ** 8,
                   engine=oracle.oracle_engine)
    assert r.engine.packed
    c = oracle.run_abcdesmc(ModelSpec(prior, sim, abck, seed=11), N, eps, nsims_max=10 ** 8)
    res = r.engine.result()
    assert r.iters == c["iters"] and r.nsims == c["nsims"] and r.logZ == c["logZ"]
    assert np.array_equal(np.array(r.ϵs), c["eps_hist"])
    assert [x[0] for x in r.ranges_ϵ] == list(c["lo_hist"]) and [x[1] for x in r.ranges_ϵ] == list(c["hi_hist"])
    for k in ("theta", "C", "Wns", "alive"):
        assert np.array_equal(res[k], c[k]), k
    n_alive = int(res["alive"].sum())
    assert res["alive"][:n_alive].all() and not res["alive"][n_alive:].any()      # the alive particles are a prefix


def test_partitioned_and_index_keeping_populations_agree_in_law(oracle):
    """the spec relabels the particles at every reweight (orc_smc_partition); the reference keeps a particle's index
    for life.  Same algorithm under another labelling: over seeds, log-evidence and posterior mean agree within
    their Monte-Carlo error (and with the exact finite-eps evidence of examples/minimal_example.jl's model 1)"""
    prior, sim = A.Normal(0, math.sqrt(10)), A.Normal1D(3.0)
    z = {True: [], False: []}
    mu = {True: [], False: []}
    for seed in range(1, 13):
        for packed in (True, False):
            c = oracle.run_abcdesmc(ModelSpec(prior, sim, seed=seed), 4000, 0.3, packed=packed)
            z[packed].append(c["logZ"])
            mu[packed].append(float(c["theta"][c["alive"], 0].mean()))
    for arr in (z, mu):
        a, b = np.array(arr[True]), np.array(arr[False])
        se = math.sqrt(a.var(ddof=1) / a.size + b.var(ddof=1) / b.size)
        assert abs(a.mean() - b.mean()) < 3.5 * se
    assert abs(np.mean(z[True]) - (-3.038051357)) < 0.03 and abs(np.mean(mu[True]) - 30 / 11) < 0.05
