"""A multivariate prior that is not a product: MvNormal(mu, Sigma) in the `prior` position (the reference accepts any
Distributions multivariate distribution there: particles are vectors, push_p casts every element,
src/abcdez_types.jl:16,21).  Log-densities against scipy (tests/golden/mvnormal_logpdf_scipy.json), the initial population's
moments, a whole abcdesmc run against the conjugate posterior, and -- under -m gpu -- the HIP engine bit for bit against
the oracle."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import abcdez_amd as A
from abcdez_amd.model import ModelSpec

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mvnormal_logpdf_scipy.json")))["cases"]


def sim_for(d):
    return A.MVNormal(tuple([0.5] * d))


@pytest.mark.parametrize("case", GOLD, ids=lambda c: f"d{len(c['mu'])}")
def test_oracle_mvnormal_logpdf_equals_scipy(oracle, case):
    prior = A.MvNormal(case["mu"], case["cov"])
    d = len(case["mu"])
    spec = ModelSpec(prior, sim_for(d), seed=3)
    m = oracle.OracleModel(spec)
    x = np.zeros((len(case["x"]), spec.ld))
    x[:, :d] = np.array(case["x"])
    for literal in (0, 1):                      # pairwise tree (spec tier) and left-to-right sum (literal tier)
        out = np.zeros(len(x))
        oracle.lib().orc_logprior(m.ptr, x.ctypes.data, len(x), literal, out.ctypes.data)
        for got, want, pt in zip(out, case["logpdf"], case["x"]):
            assert abs(got - want) <= 2e-12 * max(1.0, abs(want)), (d, literal, got, want)
            assert abs(prior.logpdf(pt) - want) <= 2e-12 * max(1.0, abs(want))
    assert A.push_p(prior, [1, 2.5] + [0] * (d - 2)) == [1.0, 2.5] + [0.0] * (d - 2)        # types.jl:21: every element to float


def test_mvnormal_initial_population_has_the_priors_moments(oracle):
    case = GOLD[1]                               # d = 3
    prior = A.MvNormal(case["mu"], case["cov"])
    spec = ModelSpec(prior, sim_for(3), seed=11)
    N = 200000
    eng = oracle.oracle_engine(spec, N)
    eng.init_population()
    th = eng.state[0].numpy()[:, :3]
    mu, cov = np.array(case["mu"]), np.array(case["cov"])
    assert np.all(np.abs(th.mean(0) - mu) < 5 * np.sqrt(np.diag(cov) / N))
    assert np.allclose(np.cov(th.T), cov, atol=5 * np.abs(cov).max() / math.sqrt(N) * 3)
    lp = eng.state[1].numpy()
    assert abs(lp[0] - prior.logpdf(th[0])) < 1e-10


def test_mvnormal_abcdesmc_recovers_the_conjugate_posterior(oracle):
    """prior N(mu, S), simulator x ~ N(theta, I) observed at y: posterior N((S^-1 + I)^-1 (S^-1 mu + y), (S^-1 + I)^-1);
    ABC with a small eps approaches it (the finite-eps bias inflates the covariance slightly)"""
    mu = np.array([0.5, -1.0])
    S = np.array([[2.0, 1.2], [1.2, 1.5]])
    y = (1.0, 0.25)
    prior = A.MvNormal(mu, S)
    r = A.abcdesmc(prior, A.MVNormal(y), 0.15, None, nparticles=20000, verbose=False, rng=5, engine=oracle.oracle_engine)
    Si = np.linalg.inv(S)
    Pc = np.linalg.inv(Si + np.eye(2))
    pm = Pc @ (Si @ mu + np.array(y))
    al = r.Wns > 0
    P = np.array([list(p) for p in np.asarray(r.P, dtype=object)[al]], dtype=float)
    assert np.all(np.abs(P.mean(0) - pm) < 0.04), (P.mean(0), pm)
    assert np.allclose(np.cov(P.T), Pc, atol=0.06)
    assert abs(np.corrcoef(P.T)[0, 1] - Pc[0, 1] / math.sqrt(Pc[0, 0] * Pc[1, 1])) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("which", [1, 2, 3, 4])
def test_mvnormal_hip_engine_equals_oracle(oracle, which):
    """the whole driver on the device against the oracle for correlated priors of d = 3 (one lane), 8, 16 (two lanes) and 32
    (four lanes): every output bit for bit"""
    case = GOLD[which]
    d = len(case["mu"])
    prior = A.MvNormal(case["mu"], case["cov"])
    sim = A.MVNormal(tuple(np.array(case["mu"]) + 0.3))
    eps = {3: 0.6, 8: 2.0, 16: 3.5, 32: 6.5}[d]
    N = 8192
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=9)
    assert type(r.engine.ops).__name__ == "HipOps"
    c = oracle.run_abcdesmc(A.ModelSpec(prior, sim, seed=9), N, eps)
    res = r.engine.result()
    assert r.logZ == c["logZ"]
    assert np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["C"], c["C"]) and np.array_equal(res["alive"], c["alive"])
    assert np.array_equal(res["logpi"], c["logpi"])
