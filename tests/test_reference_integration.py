"""The reference's statistical integration tests (test/runtests.jl:110-624), restated with
the same population sizes and tolerances, run through the product's host drivers
(abcdez_amd.abcdesmc / abcdemc) with the CPU oracle as the population engine.

This is what pins the oracle (and therefore, via the bit-exact GPU parity tests, the HIP
kernels) to ABCdeZ.jl's own known answers: analytic evidences, posterior means, the
Bayes-factor check and the qualitative inference problems.  CPU only."""
import json
import math
import os

import numpy as np
import pytest

import abcdez_amd as A

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json"),
                      encoding="utf-8"))["analytic"]


def isaround(theta, val, f=1.0):
    """test/runtests.jl:9"""
    theta = np.asarray(theta, dtype=float)
    return theta.mean() - f * theta.std(ddof=1) <= val <= theta.mean() + f * theta.std(ddof=1)


def weightinds(oracle, w, seed=99):
    """test/runtests.jl:13-19: stratified resample of the weighted population"""
    w = np.ascontiguousarray(w, dtype=np.float64)
    assert abs(w.sum() - 1) < 1e-9
    inds = np.zeros(w.size, dtype=np.uint32)
    oracle.lib().orc_wsample_stratified(seed, w.ctypes.data, w.size, 0, inds.ctypes.data)
    return inds.astype(np.int64)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)], autouse=True)
def backend(request):
    """every test of this file runs twice: on the CPU oracle (-m "not gpu") and through the product's HIP engine
    (-m gpu) -- same seeds, bit-identical populations, so the reference's tolerances hold for both"""
    global BACKEND
    BACKEND = request.param
    yield request.param


BACKEND = "oracle"


def smc(oracle, prior, sim, eps, **kw):
    kw.setdefault("verbose", False)
    if BACKEND == "hip":
        return A.abcdesmc(prior, sim, eps, None, **kw)
    return A.abcdesmc(prior, sim, eps, None, engine=oracle.oracle_engine, **kw)


def mc(oracle, prior, sim, eps, **kw):
    kw.setdefault("verbose", False)
    if BACKEND == "hip":
        return A.abcdemc(prior, sim, eps, None, **kw)
    return A.abcdemc(prior, sim, eps, None, engine=oracle.oracle_engine, **kw)


@pytest.mark.parametrize("data,key", [(3, "Z_indicator_data3"), (7, "Z_indicator_data7")])
def test_1d_normal_with_evidence(oracle, data, key):
    """test/runtests.jl:110-163 and :165-218"""
    g = GOLD[key]
    prior = A.Normal(0, math.sqrt(10))
    sim = A.Normal1D(float(data))
    r = smc(oracle, prior, sim, 0.3, nparticles=5000, rng=1)
    Z = math.exp(r.logZ)
    assert g["value"] * (1 - g["rtol"]) <= Z <= g["value"] * (1 + g["rtol"])
    assert isaround(r.P[r.Wns > 0.0], g["posterior_mean"])
    rm = mc(oracle, prior, sim, 0.3, nparticles=5000, generations=500, rng=2)
    assert isaround(rm.P, g["posterior_mean"])
    # driver invariants
    eps = np.array(r.ϵs)
    assert (np.diff(eps) <= 0).all() and eps[-1] >= 0.3 and r.ϵ == 0.3
    assert (np.diff(np.array(r.logZs)) <= 1e-12).all()          # indicator kernel: wnorm <= 1
    assert abs(r.Wns.sum() - 1) < 1e-9
    assert len(r.ϵs) == len(r.logZs) == len(r.esss) == len(r.faccs) == len(r.γ0s) == len(r.Kmcmcs) == r.iters + 1
    assert r.blobs is None


def test_evidence_bayes_factor(oracle):
    """test/runtests.jl:220-266"""
    sim = A.Normal1D(3.0)
    r1 = smc(oracle, A.Uniform(-10, 10), sim, 0.3, nparticles=5000, rng=3)
    r2 = smc(oracle, A.Uniform(-20, 20), sim, 0.3, nparticles=5000, rng=4)
    Z1, Z2 = math.exp(r1.logZ), math.exp(r2.logZ)
    assert GOLD["Z_uniform10"]["value"] * 0.8 <= Z1 <= GOLD["Z_uniform10"]["value"] * 1.2
    assert GOLD["Z_uniform20"]["value"] * 0.8 <= Z2 <= GOLD["Z_uniform20"]["value"] * 1.2
    assert 2.0 * 0.8 <= Z1 / Z2 <= 2.0 * 1.2
    assert isaround(r1.P[r1.Wns > 0], 3) and isaround(r2.P[r2.Wns > 0], 3)


def test_nonstrict_indicator_kernel(oracle):
    """test/runtests.jl:268-319"""
    g = GOLD["Z_indicator_data3"]
    r = smc(oracle, A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, nparticles=5000, ABCk=A.Indicator0toϵ, rng=5)
    assert g["value"] * 0.9 <= math.exp(r.logZ) <= g["value"] * 1.1
    assert isaround(r.P[r.Wns > 0.0], g["posterior_mean"])


@pytest.mark.parametrize("ABCk", [A.Epa0toϵ, A.EpaStrict0toϵ])
def test_epanechnikov_kernels(oracle, ABCk):
    """test/runtests.jl:321-371 and :373-423 -- continuous weights"""
    g = GOLD["Z_epa_data3"]
    r = smc(oracle, A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, nparticles=5000, ABCk=ABCk, rng=6)
    assert g["value"] * 0.9 <= math.exp(r.logZ) <= g["value"] * 1.1
    post = r.P[weightinds(oracle, r.Wns)]
    assert isaround(post, GOLD["Z_indicator_data3"]["posterior_mean"])
    w = r.Wns[r.Wns > 0]
    assert w.max() > w.min() * 1.01                      # weights really are continuous


@pytest.mark.parametrize("seed", [1, 2])
def test_dirac_delta_defaults(oracle, seed):
    """test/runtests.jl:493-519 (default nparticles / generations)"""
    prior, sim = A.Normal(1, 0.2), A.DiracSquare(1.5)
    r = mc(oracle, prior, sim, 0.1, rng=seed)
    assert r.P.shape == (50,) and isaround(r.P, 0.707)
    s = smc(oracle, prior, sim, 0.1, rng=seed)
    assert s.P.shape == (100,) and isaround(s.P[s.Wns > 0], 0.707)


def test_normal_times_discrete_uniform(oracle):
    """test/runtests.jl:521-535: mixed continuous/discrete prior, push_p in the loop"""
    prior = A.Factored(A.Normal(1, 0.5), A.DiscreteUniform(1, 10))
    sim = A.NormalTimesDU(5.5)
    r = mc(oracle, prior, sim, 0.01, nparticles=100, generations=1000, rng=7)
    assert isaround(r.P[:, 0], 1) and isaround(r.P[:, 1], 5)
    assert np.array_equal(r.P[:, 1], np.rint(r.P[:, 1]))              # P is push_p-cast (mc:166)
    s = smc(oracle, prior, sim, 0.01, nparticles=100, rng=7)
    al = s.Wns > 0
    assert isaround(s.P[al, 0], 1) and isaround(s.P[al, 1], 5)
    assert np.array_equal(s.P[:, 1], np.rint(s.P[:, 1]))
    internal = s.engine.result()["theta"][:, 1]
    assert not np.array_equal(internal, np.rint(internal))            # internal state stays unrounded (mc:216-220)


def _brownianrms(mu, sigma, n, rng):
    t = np.arange(0, n + 1, dtype=float)
    return np.sqrt(mu * mu * t * t + sigma * sigma * t) * (0.95 + 0.1 * rng.random())


def test_drifted_wiener(oracle):
    """test/runtests.jl:537-569"""
    tdata = _brownianrms(0.5, 2.0, 30, np.random.default_rng(1))
    prior = A.Factored(A.Uniform(0, 1), A.Uniform(0, 4))
    sim = A.WienerRMS(tuple(tdata))
    r = mc(oracle, prior, sim, 0.05, nparticles=1000, generations=300, rng=8)
    assert isaround(r.P[:, 0], 0.5, f=2.0) and isaround(r.P[:, 1], 2.0, f=2.0)
    s = smc(oracle, prior, sim, 0.05, nparticles=1000, rng=8)
    al = s.Wns > 0
    assert isaround(s.P[al, 0], 0.5, f=2.0) and isaround(s.P[al, 1], 2.0, f=2.0)


def test_mixture_model(oracle):
    """test/runtests.jl:571-598"""
    st_n = np.array(GOLD["mixture_st_n"])

    def st(res):
        q = np.quantile(res, np.arange(0.1, 0.95, 0.1))
        h = (q - q[::-1]) / 2
        return h[(len(h) - 1) // 2:]

    prior, sim = A.Uniform(-10, 10), A.Mixture01(0.0)
    r = mc(oracle, prior, sim, 0.01, nparticles=2000, generations=1000, rng=9)
    assert np.mean(np.abs(st(r.P) - st_n)) < 0.1
    s = smc(oracle, prior, sim, 0.01, nparticles=2000, rng=9)
    assert np.mean(np.abs(st(s.P[s.Wns > 0]) - st_n)) < 0.1


@pytest.mark.parametrize("p_inf", [0.0, 0.5])
def test_2d_problem_with_infinite_distances(oracle, p_inf):
    """test/runtests.jl:600-624: dist2! returns Inf with probability 1/2 -> init redraw + rejection paths"""
    prior = A.Factored(A.Normal(0, 5), A.Normal(0, 5))
    sim = A.Quad2D(p_inf)
    r = mc(oracle, prior, sim, 0.01, nparticles=500, generations=500, rng=10)
    assert np.isfinite(r.C).all()
    assert isaround(r.P[:, 0], 1) and isaround(r.P[:, 1], 1)
    s = smc(oracle, prior, sim, 0.01, nparticles=500, rng=10)
    al = s.Wns > 0
    assert np.isfinite(s.C).all()
    assert isaround(s.P[al, 0], 1) and isaround(s.P[al, 1], 1)


def test_minimal_example_two_models(oracle):
    """examples/minimal_example.jl:10-56 (BASELINE.json configs[0]): both models at N = 1000"""
    g1, g2 = GOLD["Z_exact_finite_eps_sigma2_10"], GOLD["Z_exact_finite_eps_sigma2_100"]
    r1 = smc(oracle, A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, nparticles=1000, rng=11)
    r2 = smc(oracle, A.Normal(0, math.sqrt(100)), A.Normal1D(3.0), 0.3, nparticles=1000, rng=12)
    assert abs(math.exp(r1.logZ) / g1["value"] - 1) < 0.15
    assert abs(math.exp(r2.logZ) / g2["value"] - 1) < 0.15
    assert isaround(r1.P[r1.Wns > 0], 30 / 11)


def test_mvn_evidence_matches_noncentral_chi2(oracle):
    """the d-dimensional benchmark model has a closed-form evidence (noncentral chi-square)"""
    g = GOLD["Z_mvn8_eps2.5"]
    prior = A.Factored(*[A.Normal(0, 1)] * 8)
    r = smc(oracle, prior, A.MVNormal((1.0,) * 8), 2.5, nparticles=20000, rng=13, nsims_max=10 ** 9)
    assert abs(r.logZ - g["logZ"]) < 0.1
    al = r.Wns > 0
    # eps = 2.5 is a wide tolerance: the ABC posterior mean lies between the prior mean 0 and the exact 0.5
    m = r.P[al].mean(0)
    assert (m > 0.2).all() and (m < 0.5).all()


def _socks_prior():
    """test/runtests.jl:439-445"""
    prior_mu, prior_sd = 30, 15
    prior_size = -prior_mu ** 2 / (prior_mu - prior_sd ** 2)
    return A.Factored(A.NegativeBinomial(prior_size, prior_size / (prior_mu + prior_size)), A.Beta(15, 2))


def test_socks_abcdemc(oracle):
    """test/runtests.jl:425-454: NegativeBinomial x Beta prior, discrete parameter, integer distances"""
    r = mc(oracle, _socks_prior(), A.Socks(0, 11), 0.01, nparticles=5000, generations=500, rng=14)
    assert isaround(r.P[:, 0], 46.2) and isaround(r.P[:, 1], 0.866)
    assert np.array_equal(r.P[:, 0], np.rint(r.P[:, 0])) and (r.P[:, 0] >= 0).all()
    assert ((r.P[:, 1] >= 0) & (r.P[:, 1] <= 1)).all()
    assert set(np.unique(r.C)) <= set(np.arange(0.0, 23.0))           # |pairs - 0| + |odds - 11| is an integer


def test_socks_abcdesmc_strict_kernel(oracle):
    """test/runtests.jl:456-491: eps = 0.01 with the strict kernel on integer distances => only exact matches survive"""
    r = smc(oracle, _socks_prior(), A.Socks(0, 11), 0.01, nparticles=5000, ABCk=A.IndicatorStrict0toϵ, rng=15)
    al = r.Wns > 0
    assert isaround(r.P[al, 0], 46.2) and isaround(r.P[al, 1], 0.866)
    assert (r.C[al] == 0).all()


# ---------------------------------------------------------------- checkpoint / resume
@pytest.mark.parametrize("abck", [A.IndicatorStrict0toϵ, A.Epa0toϵ])
def test_abcdesmc_resume_reproduces_the_uninterrupted_run(oracle, tmp_path, abck):
    """Stop after 7 generations, write the checkpoint to disk, continue in a fresh engine: population, weights,
    evidence and every history entry equal the uninterrupted run bit for bit (the randomness is counter-based)."""
    prior = A.Factored(*[A.Normal(0.0, 1.0)] * 8)
    sim = A.MVNormal((1.0,) * 8)
    kw = dict(nparticles=1500, rng=17, ABCk=abck, facc_min=0.3)
    full = smc(oracle, prior, sim, 2.4, **kw)
    part = smc(oracle, prior, sim, 2.4, max_iters=7, **kw)
    assert part.iters == 7 < full.iters
    path = tmp_path / "smc.ckpt"
    A.save_checkpoint(path, part.checkpoint())
    rest = smc(oracle, prior, sim, 2.4, resume=path, **kw)
    assert rest.iters == full.iters and rest.nsims == full.nsims and rest.updates == full.updates
    assert rest.logZ == full.logZ and rest.ϵ == full.ϵ
    for a, b in ((rest.P, full.P), (rest.Wns, full.Wns), (rest.C, full.C)):
        assert np.array_equal(a, b, equal_nan=True)
    for name in ("ϵs", "logZs", "esss", "faccs", "γ0s", "Kmcmcs", "ranges_ϵ"):
        assert list(getattr(rest, name)) == list(getattr(full, name)), name
    # in-memory dict works too, and a checkpoint of the wrong kind / seed is refused
    again = smc(oracle, prior, sim, 2.4, resume=part.checkpoint(), **kw)
    assert again.logZ == full.logZ
    with pytest.raises(ValueError, match="different seed"):
        smc(oracle, prior, sim, 2.4, resume=part.checkpoint(), **dict(kw, rng=18))
    with pytest.raises(ValueError, match="not an abcdemc checkpoint"):
        mc(oracle, prior, sim, 2.4, nparticles=1500, rng=17, resume=part.checkpoint())
    # a checkpoint records the stream it was written on -- (library version, Philox rounds) -- and a build with another round
    # count refuses it: every random number after the resume would differ, silently
    ck = part.checkpoint()
    assert ck["state"]["stream_version"][1] == oracle.lib().orc_philox_rounds() == 10
    ck["state"]["stream_version"] = [ck["state"]["stream_version"][0], 7]
    with pytest.raises(ValueError, match="Philox4x32-7"):
        smc(oracle, prior, sim, 2.4, resume=ck, **kw)
    ck["state"]["stream_version"] = None                        # written before round 4: resumes, with a warning
    with pytest.warns(UserWarning, match="stream_version"):
        old = smc(oracle, prior, sim, 2.4, resume=ck, **kw)
    assert old.logZ == full.logZ


def test_abcdemc_resume_reproduces_the_uninterrupted_run(oracle, tmp_path):
    prior, sim = A.Normal(0, math.sqrt(10)), A.Normal1D(3.0)
    full = mc(oracle, prior, sim, 0.3, nparticles=800, generations=30, rng=5)
    part = mc(oracle, prior, sim, 0.3, nparticles=800, generations=11, rng=5)
    path = tmp_path / "mc.ckpt"
    A.save_checkpoint(path, part.checkpoint())
    rest = mc(oracle, prior, sim, 0.3, nparticles=800, generations=30, rng=5, resume=A.load_checkpoint(path))
    assert rest.nsims == full.nsims and rest.reached_ϵ == full.reached_ϵ
    assert np.array_equal(rest.P, full.P) and np.array_equal(rest.C, full.C)


# ---------------------------------------------------------------- blobs (second return value of dist!, docs/src/index.md:298-324)
def _blob_cases():
    lv_obs = (1.0, 0.5, 1.46, 0.43, 1.77, 0.62, 1.52, 1.13, 0.95, 1.31, 0.66, 1.09)
    wien = tuple(math.sqrt(0.25 * t * t + 4.0 * t) for t in range(8))
    return {
        # name: (prior, simulator with blobs on, eps, N, distance recomputed from the blob)
        "normal1d": (A.Normal(0, math.sqrt(10)), A.Normal1D(3.0, blobs=True), 0.3, 1500, lambda b: np.abs(b - 3.0)),
        "mvn5": (A.Factored(*[A.Normal(0, 1)] * 5), A.MVNormal((1.0,) * 5, blobs=True), 1.6, 1200,
                 lambda b: np.sqrt(((b - 1.0) ** 2).sum(1))),
        "dirac": (A.Normal(1, 0.2), A.DiracSquare(1.5, blobs=True), 0.1, 400, lambda b: np.abs(b - 1.5)),
        "quad2d": (A.Factored(A.Normal(0, 5), A.Normal(0, 5)), A.Quad2D(0.0, blobs=True), 0.05, 600,
                   lambda b: 50.0 * b[:, 0] ** 2 + b[:, 1] ** 2),
        "mixture": (A.Uniform(-10, 10), A.Mixture01(0.0, blobs=True), 0.05, 800, lambda b: np.abs(b - 0.0)),
        "normdu": (A.Factored(A.Normal(1, 0.5), A.DiscreteUniform(1, 10)), A.NormalTimesDU(5.5, blobs=True), 0.05, 400,
                   lambda b: np.abs(b - 5.5)),
        "wiener": (A.Factored(A.Uniform(-2, 2), A.Uniform(0, 4)), A.WienerRMS(wien, blobs=True), 0.3, 600,
                   lambda b: np.abs(b - np.array(wien)).sum(1) / len(wien)),
        "lv": (A.Factored(*[A.Uniform(0.0, 2.0)] * 4), A.LotkaVolterraRK4(lv_obs, dt=0.05, steps_per_obs=10, blobs=True),
               1.0, 400, lambda b: np.sqrt(((b - np.array(lv_obs)) ** 2).sum(1))),
        "socks": (A.Factored(A.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), A.Beta(15, 2)),
                  A.Socks(0, 11, blobs=True), 2.5, 800, lambda b: np.abs(b[:, 0] - 0.0) + np.abs(b[:, 1] - 11.0)),
    }


@pytest.mark.parametrize("name", list(_blob_cases().keys()))
def test_blobs_are_the_simulated_data_behind_every_distance(oracle, name):
    """`blobs=True`: r.blobs[i] is the simulation output whose distance to the data is r.C[i] (docs/src/index.md:
    298-324: "blobs could record the actual simulation output"), through init, accepts, rejections and resampling."""
    prior, sim, eps, N, dist = _blob_cases()[name]
    r = smc(oracle, prior, sim, eps, nparticles=N, rng=9)
    assert r.iters > 3 and r.blobs is not None and r.blobs.shape[0] == N
    exact = name in ("normal1d", "dirac", "mixture", "normdu", "socks")
    d = dist(r.blobs)
    assert np.array_equal(d, r.C) if exact else np.allclose(d, r.C, rtol=1e-13, atol=0)
    assert len(np.unique(r.C)) > 1 or name in ("socks",)
    m = mc(oracle, prior, sim, eps, nparticles=max(N // 4, 50), generations=15, rng=9)
    d = dist(m.blobs)
    assert np.array_equal(d, m.C) if exact else np.allclose(d, m.C, rtol=1e-13, atol=0)


def test_blobs_default_off_and_checkpointed(oracle, tmp_path):
    prior, sim_on, eps, N, dist = _blob_cases()["mvn5"]
    off = smc(oracle, prior, A.MVNormal((1.0,) * 5), eps, nparticles=N, rng=9)
    assert off.blobs is None                                                   # as in every reference test
    on = smc(oracle, prior, sim_on, eps, nparticles=N, rng=9)
    assert np.array_equal(on.P, off.P) and on.logZ == off.logZ                 # recording blobs changes nothing else
    part = smc(oracle, prior, sim_on, eps, nparticles=N, rng=9, max_iters=6)
    path = tmp_path / "blob.ckpt"
    A.save_checkpoint(path, part.checkpoint())
    rest = smc(oracle, prior, sim_on, eps, nparticles=N, rng=9, resume=path)
    assert np.array_equal(rest.blobs, on.blobs) and np.array_equal(rest.C, on.C)
