#!/usr/bin/env python3
"""Generates tests/golden/prior_logpdf_scipy.json: log-densities of the five prior families the reference's
tests use (test/runtests.jl:112,233,443-445,523) and of the further Distributions.jl families the device knows
(`prior::Distribution`, src/abcdez_smc.jl:165), computed by scipy.stats -- an implementation that shares no
code with include/abcdez_spec.h -- at hand-picked points incl. support edges.

Parametrisations follow Distributions.jl (what ABCdeZ.jl's `logpdf(prior, x)` evaluates, src/abcdez_priors.jl:40-46):
Normal(mu, sigma); Uniform(a, b) closed; DiscreteUniform(a, b); Beta(alpha, beta);
NegativeBinomial(r, p) = failures before the r-th success (scipy.stats.nbinom(n=r, p=p));
Exponential(theta) = expon(scale=theta); Gamma(alpha, theta) = gamma(alpha, scale=theta); LogNormal(mu, sigma) =
lognorm(s=sigma, scale=exp(mu)); Cauchy(mu, sigma); Laplace(mu, theta); Weibull(alpha, theta) = weibull_min(alpha, scale=theta);
InverseGamma(alpha, theta) = invgamma(alpha, scale=theta); truncated(Normal(mu, sigma), lo, hi) = truncnorm((lo-mu)/sigma,
(hi-mu)/sigma, mu, sigma); Logistic(mu, theta); TDist(nu) = t(nu); Pareto(alpha, theta) = pareto(alpha, scale=theta);
Poisson(lambda); Binomial(n, p).

    python tests/golden/make_prior_logpdf_golden.py
"""
import json
import math
import os

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))


def f(v):
    v = float(v)
    return "-inf" if v == -math.inf else ("nan" if v != v else v)


cases = []
for mu, sg in ((0.0, 1.0), (0.0, math.sqrt(10.0)), (1.0, 0.5), (-3.25, 7.0), (0.0, math.sqrt(100.0))):
    d = stats.norm(mu, sg)
    for x in (-12.5, -1.0, -0.25, 0.0, 0.3, 1.0, 2.7272727272727275, 6.5, 41.0):
        cases.append({"family": "Normal", "p": [mu, sg], "x": x, "logpdf": f(d.logpdf(x))})
for a, b in ((-10.0, 10.0), (-20.0, 20.0), (0.0, 1.0), (100.0, 101.0), (0.0, 2.0)):
    d = stats.uniform(a, b - a)
    for x in (a, b, 0.5 * (a + b), a - 1e-9, b + 1e-9, a + 0.25 * (b - a)):
        cases.append({"family": "Uniform", "p": [a, b], "x": x, "logpdf": f(d.logpdf(x))})
for a, b in ((1, 2), (1, 10), (-2, 7)):
    d = stats.randint(a, b + 1)
    for x in (a, b, a + 1, a - 1, b + 1, a + 0.5):
        cases.append({"family": "DiscreteUniform", "p": [a, b], "x": x, "logpdf": f(d.logpmf(x))})
for al, be in ((15.0, 2.0), (1.0, 1.0), (2.0, 5.0), (0.5, 0.5), (1.0, 3.0), (40.0, 40.0)):
    d = stats.beta(al, be)
    for x in (0.001, 0.1, 0.5, 0.866, 0.999, 0.25, -0.1, 1.1):
        cases.append({"family": "Beta", "p": [al, be], "x": x, "logpdf": f(d.logpdf(x))})
# Socks prior of test/runtests.jl:439-444: mu = 30, sd = 15 -> r = mu^2 / (sd^2 - mu), p = r / (mu + r) (as Distributions)
r_s = 30.0 ** 2 / (15.0 ** 2 - 30.0)
for r, p in ((r_s, r_s / (30.0 + r_s)), (1.0, 0.5), (3.0, 0.2), (0.7, 0.9), (25.0, 0.05)):
    d = stats.nbinom(r, p)
    for x in (0, 1, 2, 11, 30, 46, 100, 500, -1, 2.5):
        cases.append({"family": "NegativeBinomial", "p": [r, p], "x": x, "logpdf": f(d.logpmf(x))})


# ---- the further families
INF = math.inf
XS_POS = (0.0, 1e-9, 0.001, 0.37, 1.0, 2.7272727272727275, 13.5, 250.0, -0.5)
XS_REAL = (-250.0, -12.5, -1.0, -1e-7, 0.0, 0.3, 1.0, 2.7272727272727275, 41.0, 1e6)
for th in (1.0, 0.25, 40.0):
    d = stats.expon(scale=th)
    for x in XS_POS:
        cases.append({"family": "Exponential", "p": [th], "x": x, "logpdf": f(d.logpdf(x))})
for al, th in ((1.0, 1.0), (2.5, 0.5), (0.3, 4.0), (60.0, 0.1), (9.0, 2.0)):
    d = stats.gamma(al, scale=th)
    for x in XS_POS[1:]:
        cases.append({"family": "Gamma", "p": [al, th], "x": x, "logpdf": f(d.logpdf(x))})
for mu, sg in ((0.0, 1.0), (1.5, 0.25), (-2.0, 3.0)):
    d = stats.lognorm(s=sg, scale=math.exp(mu))
    for x in XS_POS:
        cases.append({"family": "LogNormal", "p": [mu, sg], "x": x, "logpdf": f(d.logpdf(x))})
for mu, sg in ((0.0, 1.0), (-3.0, 0.1), (10.0, 25.0)):
    d = stats.cauchy(mu, sg)
    for x in XS_REAL:
        cases.append({"family": "Cauchy", "p": [mu, sg], "x": x, "logpdf": f(d.logpdf(x))})
for mu, th in ((0.0, 1.0), (2.0, 0.5), (-1.0, 30.0)):
    d = stats.laplace(mu, th)
    for x in XS_REAL[:-1]:            # scipy's laplace.logpdf underflows to -inf far out (it takes the log of the pdf)
        cases.append({"family": "Laplace", "p": [mu, th], "x": x, "logpdf": f(d.logpdf(x))})
for al, th in ((1.0, 1.0), (2.0, 3.0), (0.5, 0.2), (7.5, 10.0)):
    d = stats.weibull_min(al, scale=th)
    for x in XS_POS[1:]:
        cases.append({"family": "Weibull", "p": [al, th], "x": x, "logpdf": f(d.logpdf(x))})
for al, th in ((1.0, 1.0), (3.0, 2.0), (0.5, 10.0), (20.0, 0.5)):
    d = stats.invgamma(al, scale=th)
    for x in XS_POS:
        cases.append({"family": "InverseGamma", "p": [al, th], "x": x, "logpdf": f(d.logpdf(x))})
for mu, sg, lo, hi in ((0.0, 1.0, 0.0, INF), (1.0, 2.0, -1.0, 3.0), (0.0, 1.0, -INF, 0.5), (5.0, 3.0, 0.0, 2.0), (0.0, 1.0, 1.5, 4.0)):
    d = stats.truncnorm((lo - mu) / sg, (hi - mu) / sg, mu, sg)
    pts = [v for v in (lo, hi) if math.isfinite(v)] + [max(lo, -50.0) + 0.25, min(hi, 50.0) - 0.125, 0.3, 1.75, -0.5, 2.5, 60.0]
    for x in pts:
        cases.append({"family": "TruncatedNormal", "p": [mu, sg, "-inf" if lo == -INF else lo, "inf" if hi == INF else hi], "x": x,
                      "logpdf": f(d.logpdf(x))})
for mu, th in ((0.0, 1.0), (2.0, 0.3), (-5.0, 12.0)):
    d = stats.logistic(mu, th)
    for x in XS_REAL[:-1]:
        cases.append({"family": "Logistic", "p": [mu, th], "x": x, "logpdf": f(d.logpdf(x))})
for nu in (1.0, 2.5, 4.0, 30.0, 0.5):
    d = stats.t(nu)
    for x in XS_REAL:
        cases.append({"family": "TDist", "p": [nu], "x": x, "logpdf": f(d.logpdf(x))})
for al, th in ((1.0, 1.0), (3.0, 2.0), (0.5, 0.01), (12.0, 5.0)):
    d = stats.pareto(al, scale=th)
    for x in (th, th * (1 + 1e-9), th * 1.5, th * 40.0, th * 0.999, 1e6, -1.0):
        cases.append({"family": "Pareto", "p": [al, th], "x": x, "logpdf": f(d.logpdf(x))})
for lam in (1.0, 0.1, 4.5, 120.0, 650.0):
    d = stats.poisson(lam)
    for x in (0, 1, 2, 5, 37, 120, 700, 5000, -1, 2.5):
        cases.append({"family": "Poisson", "p": [lam], "x": x, "logpdf": f(d.logpmf(x))})
for n, p_ in ((1, 0.5), (10, 0.3), (40, 0.9), (900, 0.5), (100000, 0.001)):
    d = stats.binom(n, p_)
    for x in (0, 1, 2, 7, n // 2, n - 1, n, n + 1, -1, 2.5):
        cases.append({"family": "Binomial", "p": [n, p_], "x": x, "logpdf": f(d.logpmf(x))})

with open(os.path.join(HERE, "prior_logpdf_scipy.json"), "w") as fh:
    json.dump({"generator": "tests/golden/make_prior_logpdf_golden.py", "scipy": __import__("scipy").__version__,
               "numpy": np.__version__, "cases": cases}, fh, indent=0)
print(len(cases), "cases")
