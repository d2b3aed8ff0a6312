#!/usr/bin/env python3
"""Generates tests/golden/prior_logpdf_scipy.json: log-densities of the five prior families the reference's
tests use (test/runtests.jl:112,233,443-445,523), computed by scipy.stats -- an implementation that shares no
code with include/abcdez_spec.h -- at hand-picked points incl. support edges.

Parametrisations follow Distributions.jl (what ABCdeZ.jl's `logpdf(prior, x)` evaluates, src/abcdez_priors.jl:40-46):
Normal(mu, sigma); Uniform(a, b) closed; DiscreteUniform(a, b); Beta(alpha, beta);
NegativeBinomial(r, p) = failures before the r-th success (scipy.stats.nbinom(n=r, p=p)).

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

with open(os.path.join(HERE, "prior_logpdf_scipy.json"), "w") as fh:
    json.dump({"generator": "tests/golden/make_prior_logpdf_golden.py", "scipy": __import__("scipy").__version__,
               "numpy": np.__version__, "cases": cases}, fh, indent=0)
print(len(cases), "cases")
