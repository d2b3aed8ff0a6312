"""The wrapper prior families -- `truncated(d, lo, hi)` of any univariate parent, `MixtureModel` of univariate components
(include/abcdez_spec.h: ABZ_PRIOR_TRUNCATED, ABZ_PRIOR_MIXTURE; records in abz_model.ext).  The reference takes any `Distribution`
as prior (src/abcdez_smc.jl:165, src/abcdez_priors.jl:40-46).  Pinned here without the shared arithmetic header:
  * log-densities against scipy.stats (tests/golden/prior_wrappers_scipy.json, 458 points; the truncation mass is scipy's) for the
    oracle, the Python host mirror and -- under -m gpu -- the device (bit-equal to the oracle);
  * the samplers of the initial population against scipy's laws (Kolmogorov-Smirnov / chi-square);
  * the C ABI's own validation of the records (abcdez_ctx_create refuses malformed descriptors before touching a device).
Whole runs against quadrature: tests/test_prior_families_integration.py; random models HIP == oracle: tests/test_gpu_parity.py."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
from scipy import stats

import abcdez_amd as A
from abcdez_amd import _lib
from abcdez_amd.model import ModelSpec

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prior_wrappers_scipy.json")))["cases"]
FAM = {"Gamma": A.Gamma, "Cauchy": A.Cauchy, "Exponential": A.Exponential, "LogNormal": A.LogNormal, "Laplace": A.Laplace, "Weibull": A.Weibull,
       "InverseGamma": A.InverseGamma, "Logistic": A.Logistic, "TDist": A.TDist, "Pareto": A.Pareto, "Beta": A.Beta, "Uniform": A.Uniform,
       "Normal": A.Normal, "Poisson": A.Poisson, "Binomial": A.Binomial, "NegativeBinomial": A.NegativeBinomial, "DiscreteUniform": A.DiscreteUniform}


def build(case):
    if case["kind"] == "truncated":
        return A.truncated(FAM[case["parent"]](*case["p"]), case["lo"], case["hi"])
    if case["kind"] == "affine":
        return A.Affine(FAM[case["parent"]](*case["p"]), case["mu"], case["sigma"])
    return A.MixtureModel([FAM[f](*p) for f, p in case["components"]], case["weights"])


def label(case):
    if case["kind"] == "truncated":
        return f"truncated({case['parent']}{tuple(case['p'])}, {case['lo']}, {case['hi']})"
    if case["kind"] == "affine":
        return f"{case['mu']} + {case['sigma']} * {case['parent']}{tuple(case['p'])}"
    return "MixtureModel(" + ", ".join(f for f, _ in case["components"]) + ")"


def want(pt):
    return -math.inf if pt["logpdf"] == "-inf" else float(pt["logpdf"])


def close(got, ref, x):
    if ref == -math.inf:
        return got == -math.inf
    # lgamma-based families: 3e-14 max(1, |lgamma|) per evaluation (tests/test_spec_math.py); the mass adds one rounding of a log
    return abs(got - ref) <= 4e-13 * max(1.0, abs(ref), abs(x))


def model_logpdf(oracle):
    L = oracle.lib()
    L.orc_model_prior_logpdf.restype = C.c_double
    L.orc_model_prior_logpdf.argtypes = [C.c_void_p, C.c_int, C.c_double]
    return L.orc_model_prior_logpdf


@pytest.mark.parametrize("case", GOLD, ids=label)
def test_logpdf_of_wrapper_families_equals_scipy(oracle, case):
    dist = build(case)
    if case["kind"] == "truncated":
        assert abs(dist.mass() - case["mass"]) < 1e-14, (dist.mass(), case["mass"])       # the host's own cdfs (abcdez_amd/priors.py)
    # the wrapped factor in the middle of a model: offsets into the ext table are not all zero
    spec = ModelSpec(A.Factored(A.truncated(A.Gamma(2.0, 1.0), 0.5, 4.0), dist, A.Normal(0, 1)), A.MVNormal((1.0, 1.0, 1.0)))
    assert spec.ext is not None and spec._desc[1][0] in (19, 20, 21)
    m = oracle.OracleModel(spec)
    f = model_logpdf(oracle)
    for pt in case["points"]:
        x = float(pt["x"])
        assert close(f(m.ptr, 1, x), want(pt), x), (label(case), x, f(m.ptr, 1, x), want(pt))
        assert close(dist.logpdf(x), want(pt), x), (label(case), x, dist.logpdf(x), want(pt))
        assert dist.insupport(x) == (want(pt) > -math.inf) or want(pt) == -math.inf


@pytest.mark.gpu
def test_device_logpdf_of_wrapper_families_is_the_oracles(oracle):
    import torch

    from abcdez_amd.engine import HipOps

    f = model_logpdf(oracle)
    for case in GOLD:
        dist = build(case)
        spec = ModelSpec(A.Factored(A.Normal(0, 1), dist), A.MVNormal((1.0, 1.0)))
        ops = HipOps(spec)
        m = oracle.OracleModel(spec)
        x = np.array([float(pt["x"]) for pt in case["points"]])
        xd = torch.from_numpy(x).cuda()
        yd = torch.zeros_like(xd)
        ops.math_eval(11, xd, yd, torch.ones_like(xd))          # factor index 1
        got = yd.cpu().numpy()
        for k, pt in enumerate(case["points"]):
            ref = f(m.ptr, 1, float(pt["x"]))
            assert np.float64(ref).view(np.int64) == got[k:k + 1].view(np.int64)[0], (label(case), pt["x"], got[k], ref)
        ops.close()


def draws(oracle, dist, n=40000, retry=0):
    spec = ModelSpec(A.Factored(A.Normal(0, 1), dist), A.MVNormal((1.0, 1.0)), seed=11)
    m = oracle.OracleModel(spec)
    L = oracle.lib()
    L.orc_model_prior_draw_ext.restype = C.c_double
    L.orc_model_prior_draw_ext.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]
    return np.array([L.orc_model_prior_draw_ext(m.ptr, 1, i, retry) for i in range(n)])


def test_samplers_of_wrapper_families_follow_their_laws(oracle):
    """truncated: rejection from the parent; mixture: component by inversion of the cumulative weights, then that component's sampler"""
    g = stats.gamma(2.5, scale=0.6)
    x = draws(oracle, A.truncated(A.Gamma(2.5, 0.6), 0.5, 3.0))
    assert x.min() >= 0.5 and x.max() <= 3.0
    assert stats.kstest(x, lambda t: (g.cdf(np.clip(t, 0.5, 3.0)) - g.cdf(0.5)) / (g.cdf(3.0) - g.cdf(0.5))).pvalue > 1e-3
    c = stats.cauchy(0.0, 2.0)
    x = draws(oracle, A.truncated(A.Cauchy(0.0, 2.0), -1.0, None))
    assert x.min() >= -1.0 and stats.kstest(x, lambda t: (c.cdf(np.maximum(t, -1.0)) - c.cdf(-1.0)) / (1.0 - c.cdf(-1.0))).pvalue > 1e-3
    x = draws(oracle, A.truncated(A.Uniform(-1.0, 3.0), 0.0, 2.0))            # a parent of the basic families: drawn through abz_prior_draw1
    assert x.min() >= 0.0 and x.max() <= 2.0 and stats.kstest(x, stats.uniform(0.0, 2.0).cdf).pvalue > 1e-3
    po = stats.poisson(4.0)
    k = draws(oracle, A.truncated(A.Poisson(4.0), 2, 9))
    assert np.array_equal(k, np.rint(k)) and k.min() >= 2 and k.max() <= 9
    exp = po.pmf(np.arange(2, 10)) / (po.cdf(9) - po.cdf(1)) * k.size
    assert stats.chisquare(np.bincount(k.astype(int), minlength=10)[2:10], exp).pvalue > 1e-3
    x = draws(oracle, A.Affine(A.TDist(4.0), 1.0, 2.0))                        # mu + sigma * d: the parent's sampler, moved and scaled
    assert stats.kstest(x, stats.t(4.0, loc=1.0, scale=2.0).cdf).pvalue > 1e-3
    mix = A.MixtureModel([A.Normal(-1.0, 0.5), A.Normal(2.0, 1.0), A.Laplace(0.0, 2.0)], [0.2, 0.5, 0.3])
    x = draws(oracle, mix)
    cdf = lambda t: 0.2 * stats.norm(-1, 0.5).cdf(t) + 0.5 * stats.norm(2, 1).cdf(t) + 0.3 * stats.laplace(0, 2).cdf(t)       # noqa: E731
    assert stats.kstest(x, cdf).pvalue > 1e-3
    k = draws(oracle, A.MixtureModel([A.Poisson(2.0), A.Binomial(12, 0.4)], [0.4, 0.6]))
    pmf = 0.4 * stats.poisson(2.0).pmf(np.arange(0, 13)) + 0.6 * stats.binom(12, 0.4).pmf(np.arange(0, 13))
    obs = np.bincount(np.minimum(k.astype(int), 12), minlength=13)
    assert np.array_equal(k, np.rint(k)) and stats.chisquare(obs, pmf / pmf.sum() * obs.sum()).pvalue > 1e-3
    # another retry epoch of abcde_init! is another, independent stream
    a, b = draws(oracle, mix, 4000, retry=0), draws(oracle, mix, 4000, retry=1)
    assert not np.array_equal(a, b) and abs(np.corrcoef(a, b)[0, 1]) < 0.06


def test_c_abi_validates_wrapper_records_and_family_parameters():
    """abcdez_ctx_create checks every descriptor -- parameters of the base families (the inversion samplers' limits were
    enforced by the hosts only) and the records of the wrapper families -- BEFORE it touches a device: status -1 and a message."""
    lib = _lib.load()
    ctx = C.c_void_p()

    def create(prior, patch=None):
        spec = ModelSpec(prior, A.Normal1D(3.0)) if not isinstance(prior, A.Factored) else ModelSpec(prior, A.MVNormal((1.0,) * len(prior)))
        data = np.ascontiguousarray(spec.data, dtype=np.float64)
        cm = spec.cstruct(data.ctypes.data)
        ext = spec.ext.copy() if spec.ext is not None else None
        if ext is not None:
            cm.ext = ext.ctypes.data
        if patch:
            patch(cm, ext)
        rc = lib.abcdez_ctx_create(C.byref(cm), 0, C.byref(ctx))
        msg = lib.abcdez_last_error().decode()
        return rc, msg

    def bad(prior, patch, text):
        rc, msg = create(prior, patch)
        assert rc == -1 and text in msg, (rc, msg)

    def setp(field, v):
        return lambda cm, ext: setattr(cm.prior[0], field, v)

    bad(A.Poisson(2.0), setp("p0", 1e4), "lambda <= 700")
    bad(A.Poisson(2.0), setp("p0", 0.0), "lambda <= 700")
    bad(A.Binomial(5, 0.3), setp("p1", 1.0), "0 < p < 1")
    bad(A.Binomial(5, 0.5), setp("p0", 5000.0), "-700")
    bad(A.Gamma(2.0, 1.0), setp("p1", -1.0), "scale > 0")
    bad(A.Normal(0, 1), setp("p1", 0.0), "sigma > 0")
    bad(A.truncated(A.Normal(0.0, 1.0), -1.0, 1.0), setp("c0", 10.0), "at least 1 %")           # a c0 that claims a mass of e^-11
    tr = A.truncated(A.Gamma(2.0, 1.0), 0.5, 4.0)
    bad(tr, lambda cm, ext: setattr(cm, "n_ext", 5), "outside the ext table")
    bad(tr, lambda cm, ext: ext.__setitem__(1, 0.1), "lo < hi")
    bad(tr, lambda cm, ext: ext.__setitem__(2, math.log(0.001)), "at least 1 %")
    bad(tr, lambda cm, ext: ext.__setitem__(3, 19.0), "inside a wrapper")
    bad(tr, lambda cm, ext: ext.__setitem__(3 + 3, -2.0), "scale > 0")                          # the PARENT's parameters are checked too
    bad(tr, lambda cm, ext: setattr(cm, "ext", None), "ext / n_ext mismatch")
    af = A.Affine(A.TDist(4.0), 1.0, 2.0)
    bad(af, lambda cm, ext: ext.__setitem__(1, -2.0), "sigma > 0")
    bad(af, lambda cm, ext: ext.__setitem__(2, 0.7), "to match")
    bad(af, lambda cm, ext: ext.__setitem__(4 + 1, 1.0), "must be continuous")
    mx = A.MixtureModel([A.Normal(0, 1), A.Laplace(1.0, 2.0)], [0.3, 0.7])
    bad(mx, setp("p0", 17.0), "1 .. 16 components")
    bad(mx, lambda cm, ext: ext.__setitem__(9 + 1, 0.9), "sum to 1")
    bad(mx, lambda cm, ext: ext.__setitem__(2 + 1, 1.0), "all continuous or all discrete")      # a component that claims to be discrete
    bad(mx, lambda cm, ext: ext.__setitem__(9 + 2, 20.0), "inside a wrapper")
    # a well-formed model passes the validation and fails later only for want of a device (or succeeds on a GPU box)
    rc, msg = create(A.Factored(tr, mx, A.Poisson(3.0), af))
    assert rc == 0 or "prior factor" not in msg, msg
    if rc == 0:
        lib.abcdez_ctx_destroy(ctx)
