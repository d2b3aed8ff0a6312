"""Pins the shared arithmetic spec (include/abcdez_spec.h) against independent
references: published Philox known-answer vectors + a pure-Python Philox, and mpmath /
numpy for the elementary functions.  CPU only."""
import ctypes as C
import math

import mpmath
import numpy as np
import pytest
from scipy import stats

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


ROUNDS = 10     # ABZ_PHILOX_ROUNDS (abcdez_spec.h): Random123's default, SURVEY.md section 7's RNG contract


def py_philox4x32(ctr, key, rounds=ROUNDS):
    """Independent restatement of Philox4x32-R (Salmon et al., SC'11)."""
    c = list(ctr)
    k = list(key)
    for _ in range(rounds):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & MASK, (p0 >> 32) ^ c[3] ^ k[1], p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return c


def c_philox(O, ctr, key, rounds=None):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    if rounds is None:
        O.lib().orc_philox(c.ctypes.data, k.ctypes.data, out.ctypes.data)            # the product's stream
    else:
        O.lib().orc_philox_r(rounds, c.ctypes.data, k.ctypes.data, out.ctypes.data)  # the same round function, any count
    return [int(v) for v in out]


# Random123 kat_vectors, philox4x32 10 rounds
KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([MASK] * 4, [MASK] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
     [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


@pytest.mark.parametrize("ctr,key,expect", KAT)
def test_philox_known_answers(oracle, ctr, key, expect):
    """the round function and key schedule against the published 10-round vectors; the product runs the same code for 7"""
    assert c_philox(oracle, ctr, key, rounds=10) == expect
    assert py_philox4x32(ctr, key, rounds=10) == expect


def test_philox_rounds_constant(oracle):
    assert oracle.lib().orc_philox_rounds() == ROUNDS


def test_philox_vs_independent_python(oracle):
    rng = np.random.default_rng(1)
    for _ in range(300):
        ctr = [int(v) for v in rng.integers(0, 1 << 32, 4)]
        key = [int(v) for v in rng.integers(0, 1 << 32, 2)]
        assert c_philox(oracle, ctr, key) == py_philox4x32(ctr, key)
        for r in (1, 3, 10):
            assert c_philox(oracle, ctr, key, rounds=r) == py_philox4x32(ctr, key, rounds=r)


def test_rng_word_packing(oracle):
    out = np.zeros(2, dtype=np.uint64)
    oracle.lib().orc_rng_words(0x0123456789ABCDEF, 7, 11, 3, 6, out.ctypes.data)
    r = py_philox4x32([7, 11, 3, 6], [0x89ABCDEF, 0x01234567])
    assert int(out[0]) == (r[1] << 32) | r[0] and int(out[1]) == (r[3] << 32) | r[2]


def _stream_words(oracle, seed, idx, epoch, sub, purpose):
    """(n, 2) u64 words of the product's RNG for broadcastable counter arrays"""
    idx, epoch, sub, purpose = np.broadcast_arrays(idx, epoch, sub, purpose)
    out = np.zeros((idx.size, 2), dtype=np.uint64)
    o = np.zeros(2, dtype=np.uint64)
    L = oracle.lib()
    for k, (a, b, c, d) in enumerate(zip(idx.ravel(), epoch.ravel(), sub.ravel(), purpose.ravel())):
        L.orc_rng_words(seed, int(a), int(b), int(c), int(d), o.ctypes.data)
        out[k] = o
    return out


def test_philox7_stream_battery(oracle):
    """A sanity battery on the 7-round stream as the product addresses it (consecutive counters, one key) -- no substitute
    for TestU01 (the round count rests on Salmon et al.'s BigCrush result), but it would catch a botched round function:
    bit frequencies, byte chi-square, lag-1 correlation along each counter axis, avalanche of a one-bit counter change."""
    n = 60000
    w = _stream_words(oracle, 0x9E3779B97F4A7C15, np.arange(n), 3, 0, 6)
    bits = np.unpackbits(w.view(np.uint8).reshape(n, 16), axis=1)            # n x 128
    freq = bits.mean(axis=0)
    assert np.all(np.abs(freq - 0.5) < 5 / (2 * np.sqrt(n)))                   # every output bit is fair (5 sigma)
    byt = w.view(np.uint8).ravel()
    assert stats.chisquare(np.bincount(byt, minlength=256)).pvalue > 1e-4
    u = (w[:, 0] >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5 / np.sqrt(n)             # neighbouring particles
    for axis in ("epoch", "sub", "purpose"):
        kw = dict(idx=7, epoch=1, sub=0, purpose=6)
        kw[axis] = np.arange(20000)
        v = _stream_words(oracle, 12345, kw["idx"], kw["epoch"], kw["sub"], kw["purpose"])
        x = (v[:, 0] >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
        assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5 / np.sqrt(x.size), axis
        assert stats.kstest(x, "uniform").pvalue > 1e-4, axis
    # avalanche: flipping one counter bit flips ~half of the 128 output bits
    rng = np.random.default_rng(2)
    flips = []
    for _ in range(400):
        ctr = [int(v) for v in rng.integers(0, 1 << 32, 4)]
        key = [int(v) for v in rng.integers(0, 1 << 32, 2)]
        a = c_philox(oracle, ctr, key)
        word, bit = int(rng.integers(0, 4)), int(rng.integers(0, 32))
        ctr2 = list(ctr)
        ctr2[word] ^= 1 << bit
        b = c_philox(oracle, ctr2, key)
        flips.append(sum(bin(x ^ y).count("1") for x, y in zip(a, b)))
    assert abs(np.mean(flips) - 64) < 1.5 and min(flips) > 30


def ulp_err(y, ref_mp):
    """error of double y against the mpmath value, in units of ulp(y)"""
    y = float(y)
    if ref_mp == 0:
        return 0.0 if y == 0 else np.inf
    ulp = np.spacing(abs(y)) if y != 0 else 5e-324
    return float(abs(mpmath.mpf(y) - ref_mp) / ulp)


def eval_fn(O, fn, x, y2=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    y2 = np.zeros_like(x) if y2 is None else np.ascontiguousarray(y2, dtype=np.float64)
    O.lib().orc_math_eval(fn, x.ctypes.data, y.ctypes.data, y2.ctypes.data, x.size)
    return y, y2


def test_log_accuracy(oracle):
    mpmath.mp.prec = 200
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(0, 1, 1500), np.exp(rng.uniform(-700, 700, 1500)),
                        1 + rng.uniform(-1e-3, 1e-3, 500), [5e-324, 2.2250738585072014e-308, 1.0, 0.5, 2.0]])
    y, _ = eval_fn(oracle, 0, x)
    worst = max(ulp_err(b, mpmath.log(mpmath.mpf(float(a)))) for a, b in zip(x, y))
    assert worst < 1.0, worst
    # bulk agreement with libm within 2 ulp
    xb = np.exp(rng.uniform(-700, 700, 200000))
    yb, _ = eval_fn(oracle, 0, xb)
    assert np.max(np.abs(yb - np.log(xb)) / np.spacing(np.abs(np.log(xb)) + 1e-300)) <= 2
    sp, _ = eval_fn(oracle, 0, np.array([0.0, -0.0, -1.0, np.inf, np.nan]))
    assert sp[0] == -np.inf and sp[1] == -np.inf and np.isnan(sp[2]) and sp[3] == np.inf and np.isnan(sp[4])


def test_exp_accuracy(oracle):
    mpmath.mp.prec = 200
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-745, 709, 2500), rng.uniform(-1, 1, 1000), [0.0, 1e-20, -1e-20, 709.7, -745.0]])
    y, _ = eval_fn(oracle, 1, x)
    worst = max(ulp_err(b, mpmath.exp(mpmath.mpf(float(a)))) for a, b in zip(x, y) if b > 1e-300)
    assert worst < 1.0, worst
    sp, _ = eval_fn(oracle, 1, np.array([-np.inf, np.inf, np.nan, 710.0, -746.0, 0.0]))
    assert sp[0] == 0.0 and sp[1] == np.inf and np.isnan(sp[2]) and sp[3] == np.inf and sp[4] == 0.0 and sp[5] == 1.0
    sub, _ = eval_fn(oracle, 1, np.array([-740.0]))
    assert abs(sub[0] - np.exp(-740.0)) <= 2 * 5e-324 * 2 ** 3   # subnormal range: absolute error of a few quanta


def test_sincos2pi_accuracy(oracle):
    mpmath.mp.prec = 200
    rng = np.random.default_rng(4)
    u = np.concatenate([rng.integers(0, 1 << 53, 3000).astype(np.float64) * 2.0 ** -53,
                        [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1 - 2.0 ** -53, 2.0 ** -53]])
    s, c = eval_fn(oracle, 2, u)
    for ui, si, ci in zip(u, s, c):
        a = 2 * mpmath.pi * mpmath.mpf(float(ui))
        # absolute error relative to 1 (values near zero crossings carry the rounding of 2 pi u)
        assert abs(mpmath.mpf(float(si)) - mpmath.sin(a)) < 3e-16
        assert abs(mpmath.mpf(float(ci)) - mpmath.cos(a)) < 3e-16
    assert np.all(np.abs(s * s + c * c - 1) < 5e-16)
    e = eval_fn(oracle, 2, np.array([0.0, 0.25, 0.5, 0.75]))
    assert list(e[0]) == [0.0, 1.0, -0.0, -1.0] and list(e[1]) == [1.0, -0.0, -1.0, 0.0]


def test_rint_floor_sqrt_div(oracle):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1e6, 1e6, 100000), np.arange(-10, 10) + 0.5, [0.0, -0.0, 1e300, -1e300, 2.0 ** 52]])
    assert np.array_equal(eval_fn(oracle, 3, x)[0], np.rint(x))         # ties to even == Julia round(Int, x)
    assert np.array_equal(eval_fn(oracle, 4, x)[0], np.floor(x))
    xp = np.abs(x) + 1e-300
    assert np.array_equal(eval_fn(oracle, 5, xp)[0], np.sqrt(xp))
    y2 = rng.uniform(0.5, 3.0, x.size)
    assert np.array_equal(eval_fn(oracle, 6, x, y2)[0], x / y2)


def test_table_log_accuracy(oracle):
    """abz_log_tab: the sampler's log on positive normal inputs (uniforms in [2^-53, 1))."""
    mpmath.mp.prec = 200
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(0, 1, 3000), np.exp(rng.uniform(-36.7, 0, 2000)), 1 - np.exp(rng.uniform(-36, -1, 1500)),
                        np.exp(rng.uniform(-700, 700, 500)),
                        [2.0 ** -53, 1 - 2.0 ** -53, 0.5, 0.70710678118654746, 0.70710678118654757, 1.0, 2.0, 0.999, 1.001]])
    y, _ = eval_fn(oracle, 7, x)
    worst = max(ulp_err(b, mpmath.log(mpmath.mpf(float(a)))) for a, b in zip(x, y))
    assert worst < 2.0, worst
    assert eval_fn(oracle, 7, np.array([1.0]))[0][0] == 0.0
    # agrees with the polynomial log to a couple of ulp everywhere in the sampler's domain
    u = (rng.integers(0, 1 << 52, 200000).astype(np.float64) + 0.5) * 2.0 ** -52
    a, _ = eval_fn(oracle, 7, u)
    b, _ = eval_fn(oracle, 0, u)
    assert np.max(np.abs(a - b) / np.spacing(np.abs(b))) <= 3


def test_table_sincos_accuracy(oracle):
    mpmath.mp.prec = 200
    rng = np.random.default_rng(13)
    u = np.concatenate([rng.integers(0, 1 << 52, 4000).astype(np.float64) * 2.0 ** -52,
                        np.arange(0, 256) / 256.0, (np.arange(0, 256) + 0.5) / 256.0, [1 - 2.0 ** -52, 2.0 ** -52]])
    s, c = eval_fn(oracle, 8, u)
    worst = 0.0
    for ui, si, ci in zip(u, s, c):
        a = 2 * mpmath.pi * mpmath.mpf(float(ui))
        worst = max(worst, float(abs(mpmath.mpf(float(si)) - mpmath.sin(a))), float(abs(mpmath.mpf(float(ci)) - mpmath.cos(a))))
    assert worst < 2.3e-16, worst                     # absolute, i.e. ~1 ulp of values near 1
    assert np.all(np.abs(s * s + c * c - 1) < 5e-16)
    e = eval_fn(oracle, 8, np.array([0.0, 0.25, 0.5, 0.75]))
    assert list(e[0]) == [0.0, 1.0, 0.0, -1.0] and list(e[1]) == [1.0, 0.0, -1.0, 0.0]


def test_sqrt_pn_is_sqrt_on_host(oracle):
    rng = np.random.default_rng(14)
    x = np.exp(rng.uniform(-37, 5, 100000))
    assert np.array_equal(eval_fn(oracle, 9, x)[0], np.sqrt(x))


def test_uniform_conversions(oracle):
    L = oracle.lib()
    assert L.orc_u01(0, 2) == 0.0 and L.orc_u01((1 << 64) - 1, 2) == 1 - 2.0 ** -52
    rng = np.random.default_rng(15)
    for w in (int(v) for v in rng.integers(0, 1 << 64, 2000, dtype=np.uint64)):
        assert L.orc_u01(w, 1) == ((w >> 12) + 0.5) * 2.0 ** -52        # bit trick == the defining formula
        assert L.orc_u01(w, 2) == (w >> 12) * 2.0 ** -52
    assert L.orc_u01(0, 1) == 2.0 ** -53 and L.orc_u01((1 << 64) - 1, 1) == 1 - 2.0 ** -53
    assert L.orc_u01(0, 0) == 0.0 and L.orc_u01((1 << 64) - 1, 0) == 1 - 2.0 ** -53
    assert L.orc_randint(0, 10) == 0 and L.orc_randint((1 << 64) - 1, 10) == 9
    assert L.orc_randint(1 << 63, 10) == 5


def test_normal_pairs_are_standard_normal(oracle):
    n = 400000
    z = np.zeros(2 * n)
    oracle.lib().orc_normal_pairs(12345, 6, n, z.ctypes.data)
    assert abs(z.mean()) < 4 / np.sqrt(2 * n)
    assert abs(z.var() - 1) < 0.01
    assert abs(stats.kurtosis(z)) < 0.03
    assert stats.kstest(z[:200000], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 0.01          # the two outputs of a pair are independent
    assert np.abs(z).max() < 8.6                                   # sqrt(-2 log 2^-53)
    assert stats.kstest(z[200000:400000] ** 2 + z[400000:600000] ** 2, 'chi2', args=(2,)).pvalue > 1e-3


def test_donor_ranks_distinct_and_uniform(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(6)
    n_alive, ri = 7, 3
    ra, rb = C.c_uint32(), C.c_uint32()
    cnt = np.zeros((n_alive, n_alive))
    trials = 60000
    for _ in range(trials):
        w0, w1 = (int(v) for v in rng.integers(0, 1 << 64, 2, dtype=np.uint64))
        L.orc_donor_ranks(w0, w1, n_alive, ri, C.byref(ra), C.byref(rb))
        a, b = ra.value, rb.value
        assert a != ri and b != ri and a != b and a < n_alive and b < n_alive     # smc:119-126
        cnt[a, b] += 1
    pairs = cnt[cnt > 0]
    assert pairs.size == (n_alive - 1) * (n_alive - 2)
    # uniform over ordered pairs (a, b), a != b, both != i: the law of the reference's rejection loops
    assert stats.chisquare(pairs).pvalue > 1e-4
    # smallest population the reference's loops terminate on
    L.orc_donor_ranks(0, 0, 3, 0, C.byref(ra), C.byref(rb))
    assert {ra.value, rb.value} == {1, 2}


def test_weight_fix_exact_cases(oracle):
    L = oracle.lib()
    N = 1000
    assert L.orc_weight_fix(0.0, N) == 0 and L.orc_weight_fix(float("nan"), N) == 0
    assert L.orc_weight_fix(1.0 / N, N) == 1 << 40 or abs(L.orc_weight_fix(1.0 / N, N) - (1 << 40)) <= 1
    assert L.orc_weight_fix(1.0, N) == N << 40


def test_lgamma_accuracy(oracle):
    mpmath.mp.prec = 200
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.uniform(0.01, 30, 2000), rng.uniform(30, 3000, 500), np.arange(1, 40, dtype=float), [1e-3, 0.5, 1.5, 4.615384615384615]])
    y, _ = eval_fn(oracle, 10, x)
    err = [abs(float(mpmath.mpf(float(b)) - mpmath.loggamma(mpmath.mpf(float(a))))) for a, b in zip(x, y)]
    scale = [max(1.0, abs(float(mpmath.loggamma(mpmath.mpf(float(a)))))) for a in x]
    assert max(e / s for e, s in zip(err, scale)) < 3e-14      # absolute for |lgamma| < 1, relative above


def test_beta_negbin_logpdf_and_samplers(oracle):
    """extended prior families against scipy: log-densities through the oracle's descriptor path, samplers through init"""
    import ctypes as C

    import abcdez_amd as A
    from abcdez_amd.model import ModelSpec

    nb, be = A.NegativeBinomial(4.615384615384615, 0.13333333333333333), A.Beta(15, 2)
    L = oracle.lib()
    L.orc_prior_logpdf1.restype = C.c_double
    L.orc_prior_logpdf1.argtypes = [C.c_void_p, C.c_double]
    spec = ModelSpec(A.Factored(nb, be), A.Socks(0, 11), seed=4)
    m = oracle.OracleModel(spec)
    pd0 = C.addressof(m.c.prior[0]); pd1 = C.addressof(m.c.prior[1])
    for k in list(range(0, 200, 7)) + [0, 1, 2, 1000]:
        assert abs(L.orc_prior_logpdf1(pd0, float(k)) - stats.nbinom.logpmf(k, nb.r, nb.p)) < 1e-12 * max(1, abs(stats.nbinom.logpmf(k, nb.r, nb.p)))
        assert abs(nb.logpdf(k) - stats.nbinom.logpmf(k, nb.r, nb.p)) < 1e-11
    assert L.orc_prior_logpdf1(pd0, -1.0) == -np.inf and L.orc_prior_logpdf1(pd0, 2.5) == -np.inf
    for x in list(np.linspace(0.001, 0.999, 40)) + [0.5]:
        assert abs(L.orc_prior_logpdf1(pd1, float(x)) - stats.beta.logpdf(x, 15, 2)) < 1e-12 * max(1, abs(stats.beta.logpdf(x, 15, 2)))
        assert abs(be.logpdf(float(x)) - stats.beta.logpdf(x, 15, 2)) < 1e-11
    assert L.orc_prior_logpdf1(pd1, 1.0001) == -np.inf and L.orc_prior_logpdf1(pd1, -0.1) == -np.inf
    assert L.orc_prior_logpdf1(pd1, 0.0) == -np.inf               # alpha > 1: density 0 at x = 0
    # samplers: the initial population is a sample from the prior (smc:242)
    N = 40000
    eng = oracle.oracle_engine(spec, N)
    eng.init_population()
    th = eng.state[0].numpy()
    assert stats.kstest(th[:, 1], "beta", args=(15, 2)).pvalue > 1e-3
    k = th[:, 0]
    assert np.array_equal(k, np.rint(k)) and k.min() >= 0
    assert abs(k.mean() - 30) < 0.4 and abs(k.std() - 15) < 0.4     # NegBin(mu = 30, sd = 15), test/runtests.jl:439-441
    obs = np.bincount(k.astype(int), minlength=150)[:100]
    exp = stats.nbinom.pmf(np.arange(100), nb.r, nb.p) * N
    keep = exp > 20
    assert stats.chisquare(obs[keep] * exp[keep].sum() / obs[keep].sum(), exp[keep]).pvalue > 1e-4


FURTHER_FAMILIES = {
    # name: (host prior, scipy frozen distribution)
    "Exponential": lambda A: (A.Exponential(2.5), stats.expon(scale=2.5)),
    "Gamma": lambda A: (A.Gamma(3.2, 0.7), stats.gamma(3.2, scale=0.7)),
    "Gamma<1": lambda A: (A.Gamma(0.4, 2.0), stats.gamma(0.4, scale=2.0)),
    "Chisq": lambda A: (A.Chisq(5.0), stats.chi2(5.0)),
    "LogNormal": lambda A: (A.LogNormal(0.5, 0.8), stats.lognorm(s=0.8, scale=math.exp(0.5))),
    "Cauchy": lambda A: (A.Cauchy(-1.0, 2.0), stats.cauchy(-1.0, 2.0)),
    "Laplace": lambda A: (A.Laplace(1.0, 3.0), stats.laplace(1.0, 3.0)),
    "Weibull": lambda A: (A.Weibull(1.7, 4.0), stats.weibull_min(1.7, scale=4.0)),
    "Rayleigh": lambda A: (A.Rayleigh(1.5), stats.rayleigh(scale=1.5)),
    "InverseGamma": lambda A: (A.InverseGamma(3.0, 2.0), stats.invgamma(3.0, scale=2.0)),
    "truncated(Normal) one side": lambda A: (A.truncated(A.Normal(0.0, 2.0), 0.0, None), stats.truncnorm(0.0, np.inf, 0.0, 2.0)),
    "truncated(Normal) tail": lambda A: (A.truncated(A.Normal(0.0, 1.0), 1.5, 4.0), stats.truncnorm(1.5, 4.0)),
    "Logistic": lambda A: (A.Logistic(2.0, 0.5), stats.logistic(2.0, 0.5)),
    "TDist": lambda A: (A.TDist(3.5), stats.t(3.5)),
    "Pareto": lambda A: (A.Pareto(2.5, 1.5), stats.pareto(2.5, scale=1.5)),
    "Poisson": lambda A: (A.Poisson(6.3), stats.poisson(6.3)),
    "Poisson large": lambda A: (A.Poisson(400.0), stats.poisson(400.0)),
    "Binomial": lambda A: (A.Binomial(30, 0.2), stats.binom(30, 0.2)),
    "Binomial p>1/2": lambda A: (A.Binomial(500, 0.93), stats.binom(500, 0.93)),
    "Geometric": lambda A: (A.Geometric(0.15), stats.geom(0.15, loc=-1)),
}


@pytest.mark.parametrize("name", sorted(FURTHER_FAMILIES))
def test_further_prior_families_initial_population_follows_the_prior(oracle, name):
    """the initial population is a sample from the prior (`rand(prior)`, src/abcdez_init.jl:8, smc:242): Kolmogorov-Smirnov
    (continuous) / chi-square (counts) of the oracle's draws against scipy's distribution of the same parameters; every
    draw inside the support with a finite log-density that equals the host mirror's"""
    import abcdez_amd as A
    from abcdez_amd.model import ModelSpec

    prior, ref = FURTHER_FAMILIES[name](A)
    N = 60000
    spec = ModelSpec(prior, A.DiracSquare(1.5), seed=11)
    eng = oracle.oracle_engine(spec, N)
    eng.init_population()
    x = eng.state[0].numpy()[:, 0].copy()
    lp = eng.state[1].numpy().copy()
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(lp))
    for j in range(0, N, 997):
        assert prior.insupport(float(x[j]))
        want = prior.logpdf(float(x[j]))
        assert abs(lp[j] - want) <= 1e-12 * max(1.0, abs(want), math.lgamma(getattr(prior, "n", 0) + 1.0)), (name, x[j], lp[j], want)
    if prior.discrete:
        assert np.array_equal(x, np.rint(x))
        k = x.astype(np.int64)
        lo, hi = int(k.min()), int(k.max())
        obs = np.bincount(k - lo, minlength=hi - lo + 1).astype(float)
        exp = ref.pmf(np.arange(lo, hi + 1)) * N
        keep = exp > 20
        assert keep.sum() >= 5
        assert stats.chisquare(obs[keep] * exp[keep].sum() / obs[keep].sum(), exp[keep]).pvalue > 1e-4, name
        assert abs(k.mean() - ref.mean()) < 5 * ref.std() / math.sqrt(N)
    else:
        assert stats.kstest(x, ref.cdf).pvalue > 1e-3, name
    # a second seed is a different sample
    eng2 = oracle.oracle_engine(ModelSpec(prior, A.DiracSquare(1.5), seed=12), 64)
    eng2.init_population()
    assert not np.array_equal(eng2.state[0].numpy()[:, 0], x[:64])


def test_further_prior_families_in_a_factored_prior_are_independent(oracle):
    """components of a Factored prior draw from separate counter streams (one per component index): no correlation between
    components of the same family, each marginal still its own law"""
    import abcdez_amd as A
    from abcdez_amd.model import ModelSpec

    prior = A.Factored(A.Gamma(2.0, 1.0), A.Gamma(2.0, 1.0), A.Exponential(1.0), A.Exponential(1.0), A.TDist(5.0), A.LogNormal(0.0, 0.5),
                       A.Poisson(3.0), A.Normal(0.0, 1.0))
    N = 40000
    spec = ModelSpec(prior, A.MVNormal((1.0,) * 8), seed=5)
    eng = oracle.oracle_engine(spec, N)
    eng.init_population()
    th = eng.state[0].numpy()[:, :8]
    c = np.corrcoef(th.T)
    assert np.abs(c - np.eye(8)).max() < 0.03, c
    assert stats.kstest(th[:, 1], stats.gamma(2.0).cdf).pvalue > 1e-3 and stats.kstest(th[:, 3], stats.expon().cdf).pvalue > 1e-3
    assert stats.kstest(th[:, 4], stats.t(5.0).cdf).pvalue > 1e-3 and stats.kstest(th[:, 7], "norm").pvalue > 1e-3
