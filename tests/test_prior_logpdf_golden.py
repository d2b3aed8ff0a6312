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
DISTS = {"Normal": A.Normal, "Uniform": A.Uniform, "DiscreteUniform": A.DiscreteUniform, "Beta": A.Beta,
         "NegativeBinomial": A.NegativeBinomial}
# abz_lgamma: < 3e-14 max(1, |lgamma|) (tests/test_spec_math.py); the NegativeBinomial pmf subtracts two such values
TOL = {"Normal": 4e-15, "Uniform": 4e-16, "DiscreteUniform": 4e-16, "Beta": 2e-14, "NegativeBinomial": 2e-13}


def want_of(c):
    v = c["logpdf"]
    return -math.inf if v == "-inf" else float(v)


def groups():
    out = {}
    for c in GOLD:
        out.setdefault((c["family"], tuple(c["p"])), []).append(c)
    return sorted(out.items())


def close(got, want, fam, x):
    if want == -math.inf:
        return got == -math.inf
    scale = max(1.0, abs(want), abs(x) if fam == "NegativeBinomial" else 0.0)
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
        assert close(got, want_of(c), fam, float(c["x"])), (fam, p, c["x"], got, want_of(c))
        host = dist.logpdf(float(c["x"]))             # the Python host mirror agrees too
        assert close(host, want_of(c), fam, float(c["x"])), (fam, p, c["x"], host)


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
            assert close(float(got[k]), want_of(c), fam, float(c["x"]))
        ops.close()
