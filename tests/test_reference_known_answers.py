"""The exact known answers of the reference's own unit tests (test/runtests.jl:21-108),
checked against (a) the host-side value types of the drop-in and (b) the CPU oracle /
spec arithmetic the kernels share.  Fixture: tests/golden/reference_known_answers.json."""
import ctypes as C
import json
import math
import os
import unicodedata

import numpy as np
import pytest

import abcdez_amd as A
from abcdez_amd import kernels as K
from abcdez_amd.model import ModelSpec, PriorDim

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json"),
                      encoding="utf-8"))
# Python NFKC-normalises identifiers (ϵ U+03F5 -> ε U+03B5); normalise the fixture names the same way
KCLS = {c.__name__: c for c in A.ALL_KERNELS}


def kcls(name):
    return KCLS[unicodedata.normalize("NFKC", name)]
DISTS = {"Normal": A.Normal, "Uniform": A.Uniform, "DiscreteUniform": A.DiscreteUniform}


@pytest.mark.parametrize("row", GOLD["kernel_truth_table"], ids=lambda r: f"{r['kernel']}-{r['eps']}-{r['x']}")
def test_kernel_truth_table(oracle, row):
    """test/runtests.jl:48-108: boundary semantics of the four ABC kernels."""
    cls = kcls(row["kernel"])
    k = cls(row["eps"])
    assert k.ϵ == row["eps"]
    assert (k.pdf(row["x"]), k.logpdf(row["x"])) == (row["pdf"], row["logpdf"])
    L = oracle.lib()
    assert L.orc_kernel_pdf(cls.kind, row["eps"], row["x"]) == row["pdf"]
    assert L.orc_kernel_logpdf(cls.kind, row["eps"], row["x"]) == row["logpdf"]


@pytest.mark.parametrize("cls", A.ALL_KERNELS)
def test_kernel_ctor_rejects_negative_eps(cls):
    with pytest.raises(ValueError, match="Expected ϵ ≥ 0.0"):   # src/abcdez_types.jl:30,42,55,67
        cls(-0.1)
    assert cls(0.0).eps == 0.0


@pytest.mark.parametrize("cls", [K.Epa0toϵ, K.EpaStrict0toϵ])
def test_epa_values(oracle, cls):
    k = cls(2.0)
    for x in (0.0, 0.5, 1.0, 1.999):
        assert k.pdf(x) == 1.0 - (x / 2.0) ** 2
        assert oracle.lib().orc_kernel_pdf(cls.kind, 2.0, x) == k.pdf(x)
        assert abs(oracle.lib().orc_kernel_logpdf(cls.kind, 2.0, x) - math.log(k.pdf(x))) < 1e-15


def _factored(spec):
    return A.Factored(*[DISTS[n](a, b) for n, a, b in spec])


@pytest.mark.parametrize("case", GOLD["factored"])
def test_factored_known_answers(oracle, case):
    """test/runtests.jl:21-36"""
    d = _factored(case["factors"])
    assert len(d) == 2
    assert d.pdf(case["x"]) == case["pdf"]
    lp = d.logpdf(case["x"])
    assert lp == case["logpdf"] or abs(lp - case["logpdf"]) < 1e-15
    # the oracle's per-dimension descriptors give the same log-density
    spec = ModelSpec(d, A.Quad2D(0.0))
    th = np.zeros((1, spec.ld))
    th[0, :2] = case["x"]
    for literal in (0, 1):
        out = np.zeros(1)
        m = oracle.OracleModel(spec)
        oracle.lib().orc_logprior(m.ptr, th.ctypes.data, 1, literal, out.ctypes.data)
        assert out[0] == case["logpdf"] or abs(out[0] - case["logpdf"]) < 1e-15


def test_factored_rand_in_support():
    rng = np.random.default_rng(0)
    d = A.Factored(A.Uniform(0, 1), A.Uniform(100, 101))
    for _ in range(100):
        x = d.rand(rng)
        assert 0 <= x[0] <= 1 and 100 <= x[1] <= 101
    m = A.Factored(A.Uniform(0.0, 1.0), A.DiscreteUniform(1, 2))
    for _ in range(100):
        s = m.rand(rng)
        assert 0 < s[0] < 1 and s[1] in (1, 2)
        assert m.pdf(s) == 0.5


@pytest.mark.parametrize("case", GOLD["push_p"])
def test_push_p_known_answers(oracle, case):
    """test/runtests.jl:38-46: value AND type; Julia round(Int, x) = ties to even."""
    name, a, b = case["dist"]
    dist = DISTS[name](a, b)
    out = A.push_p(dist, case["x"])
    assert out == case["out"] and type(out).__name__ == case["type"]
    # device/oracle push: same value
    spec = ModelSpec(dist, A.DiracSquare(1.5))
    th = np.array([[float(case["x"])]])
    o = np.zeros_like(th)
    m = oracle.OracleModel(spec)
    oracle.lib().orc_push_p(m.ptr, th.ctypes.data, 1, o.ctypes.data)
    assert o[0, 0] == float(case["out"])


def test_push_p_factored_and_vector():
    f = A.Factored(A.Normal(), A.DiscreteUniform())
    out = A.push_p(f, (2, 1.0))
    assert out == (2.0, 1) and isinstance(out[0], float) and isinstance(out[1], int)
    out = A.push_p(A.Normal(), [2, 1])            # a univariate distribution broadcast over a vector
    assert out == [2.0, 1.0] and all(isinstance(v, float) for v in out)
    out = A.push_p(A.product_distribution([A.Normal(), A.Normal()]), [2, 1])      # test/runtests.jl:45
    assert out == [2.0, 1.0] and all(isinstance(v, float) for v in out)
    # the whole product is broadcast (types.jl:21): one continuous component makes every element a float ...
    out = A.push_p(A.product_distribution([A.Normal(), A.DiscreteUniform(1, 10)]), [2, 3])
    assert out == [2.0, 3.0] and all(isinstance(v, float) for v in out)
    # ... and only an all-discrete product rounds (ties to even)
    out = A.push_p(A.product_distribution([A.DiscreteUniform(1, 10), A.DiscreteUniform(1, 10)]), [2.5, 3.5])
    assert out == [2, 4] and all(isinstance(v, int) for v in out)


def test_product_distribution_prior_runs_like_factored(oracle):
    """a product_distribution of the supported univariate families in the `prior` position uses the same device descriptors
    as Factored: same run, bit for bit (continuous components), P as an [N, d] array"""
    sim = A.MVNormal((1.0, 0.5, 0.2))
    fams = [A.Normal(0, 1), A.Uniform(-3, 3), A.Normal(1, 2)]
    a = A.abcdesmc(A.Factored(*fams), sim, 1.0, None, nparticles=600, verbose=False, rng=3, engine=oracle.oracle_engine)
    b = A.abcdesmc(A.product_distribution(fams), sim, 1.0, None, nparticles=600, verbose=False, rng=3, engine=oracle.oracle_engine)
    assert a.logZ == b.logZ and np.array_equal(a.P, b.P) and np.array_equal(a.Wns, b.Wns) and b.P.shape == (600, 3)


def test_prior_descriptor_logpdf_matches_host(oracle):
    rng = np.random.default_rng(1)
    L = oracle.lib()
    L.orc_prior_logpdf1.restype = C.c_double
    L.orc_prior_logpdf1.argtypes = [C.c_void_p, C.c_double]
    for dist in (A.Normal(0.3, 2.5), A.Uniform(-1.5, 4.0), A.DiscreteUniform(-2, 7)):
        fam, disc, p0, p1, c0 = dist.descriptor()
        pd = PriorDim(fam, disc, p0, p1, c0, 1.0 / p1 if fam == 1 else 0.0, 0.0)
        for x in list(rng.uniform(-6, 9, 200)) + [-1.5, 4.0, -2.0, 7.0, 3.0]:
            xx = float(round(x)) if dist.discrete else float(x)
            got = L.orc_prior_logpdf1(C.addressof(pd), xx)
            want = dist.logpdf(xx)
            assert got == want or abs(got - want) <= 1e-14 * max(1.0, abs(want))
