"""Prior log-densities pinned independently of the shared arithmetic header: scipy.stats values
(tests/golden/prior_logpdf_scipy.json, made by tests/golden/make_prior_logpdf_golden.py) against the oracle's
orc_prior_logpdf1 and -- under -m gpu -- the device.  The reference evaluates these through Distributions.jl
(`logpdf(prior, push_p(prior, x))`, src/abcdez_smc.jl:134; families of test/runtests.jl:112,233,443-445,523)."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import abcdez_amd as A
from abcdez_amd.model import ModelSpec

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prior_logpdf_scipy.json")))["cases"]
def _num(v):
    return math.inf if v == "inf" else (-math.inf if v == "-inf" else v)


DISTS = {"Normal": A.Normal, "Uniform": A.Uniform, "DiscreteUniform": A.DiscreteUniform, "Beta": A.Beta,
         "NegativeBinomial": A.NegativeBinomial,
         # the further Distributions.jl families (`prior::Distribution`, src/abcdez_smc.jl:165)
         "Exponential": A.Exponential, "Gamma": A.Gamma, "LogNormal": A.LogNormal, "Cauchy": A.Cauchy, "Laplace": A.Laplace,
         "Weibull": A.Weibull, "InverseGamma": A.InverseGamma,
         "TruncatedNormal": lambda mu, sg, lo, hi: A.truncated(A.Normal(mu, sg), _num(lo), _num(hi)),
         "Logistic": A.Logistic, "TDist": A.TDist, "Pareto": A.Pareto, "Poisson": A.Poisson, "Binomial": A.Binomial}
# abz_lgamma: < 3e-14 max(1, |lgamma|) (tests/test_spec_math.py); the NegativeBinomial / Binomial pmfs subtract two such values.
# abz_log / abz_exp: < 1 ulp; a family that scales a logarithm (shape - 1, nu + 1, ...) scales its rounding error too.
TOL = {"Normal": 4e-15, "Uniform": 4e-16, "DiscreteUniform": 4e-16, "Beta": 2e-14, "NegativeBinomial": 2e-13,
       "Exponential": 4e-16, "Gamma": 2e-14, "LogNormal": 4e-15, "Cauchy": 4e-16, "Laplace": 4e-16, "Weibull": 4e-15,
       "InverseGamma": 2e-14, "TruncatedNormal": 4e-15, "Logistic": 4e-16, "TDist": 2e-14, "Pareto": 4e-15, "Poisson": 2e-13,
       "Binomial": 2e-13}
COUNTS = ("NegativeBinomial", "Poisson", "Binomial")


def want_of(c):
    v = c["logpdf"]
    return -math.inf if v == "-inf" else float(v)


def groups():
    out = {}
    for c in GOLD:
        out.setdefault((c["family"], tuple(c["p"])), []).append(c)
    return sorted(out.items(), key=lambda kv: (kv[0][0], str(kv[0][1])))


def close(got, want, fam, x, p=()):
    if want == -math.inf:
        return got == -math.inf
    scale = max(1.0, abs(want), abs(x) if fam in COUNTS else 0.0)
    if fam == "Binomial":              # log C(n, k) as a difference of three lgamma values of size lgamma(n + 1)
        scale = max(scale, math.lgamma(p[0] + 1.0))
    return abs(got - want) <= TOL[fam] * scale * 8


@pytest.mark.parametrize("key,cases", groups(), ids=lambda v: f"{v[0]}{v[1]}" if isinstance(v, tuple) else "")
def test_oracle_prior_logpdf_equals_scipy(oracle, key, cases):
    fam, p = key
    dist = DISTS[fam](*p)
    spec = ModelSpec(dist, A.DiracSquare(1.5))
    m = oracle.OracleModel(spec)
    L = oracle.lib()
    L.orc_prior_logpdf1.restype = C.c_double
    L.orc_prior_logpdf1.argtypes = [C.c_void_p, C.c_double]
    pd = C.addressof(m.c.prior[0])
    for c in cases:
        got = L.orc_prior_logpdf1(pd, float(c["x"]))
        assert close(got, want_of(c), fam, float(c["x"]), p), (fam, p, c["x"], got, want_of(c))
        host = dist.logpdf(float(c["x"]))             # the Python host mirror agrees too
        assert close(host, want_of(c), fam, float(c["x"]), p), (fam, p, c["x"], host)


@pytest.mark.gpu
def test_device_prior_logpdf_equals_oracle_and_scipy(oracle):
    """the device evaluates abz_prior_logpdf1 (math_eval fn 11) on every golden case: bit-equal to the oracle,
    and within the stated tolerance of scipy"""
    import torch

    from abcdez_amd.engine import HipOps

    L = oracle.lib()
    L.orc_prior_logpdf1.restype = C.c_double
    L.orc_prior_logpdf1.argtypes = [C.c_void_p, C.c_double]
    for (fam, p), cases in groups():
        dist = DISTS[fam](*p)
        spec = ModelSpec(dist, A.DiracSquare(1.5))
        ops = HipOps(spec)
        m = oracle.OracleModel(spec)
        x = np.array([float(c["x"]) for c in cases])
        xd = torch.from_numpy(x).cuda()
        yd = torch.zeros_like(xd)
        ops.math_eval(11, xd, yd, torch.zeros_like(xd))
        got = yd.cpu().numpy()
        for k, c in enumerate(cases):
            ref = L.orc_prior_logpdf1(C.addressof(m.c.prior[0]), float(c["x"]))
            assert np.float64(ref).view(np.int64) == got[k:k + 1].view(np.int64)[0], (fam, p, c["x"], got[k], ref)
            assert close(float(got[k]), want_of(c), fam, float(c["x"]), p)
        ops.close()
