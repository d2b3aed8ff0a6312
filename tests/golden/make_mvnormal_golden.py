#!/usr/bin/env python3
"""Generates tests/golden/mvnormal_logpdf_scipy.json: log-densities of correlated Normal priors (a Distributions.MvNormal in the
`prior` position, src/abcdez_types.jl:16,21) computed by scipy.stats.multivariate_normal -- an implementation that shares no
code with include/abcdez_spec.h (it works from the eigen-decomposition of the covariance, the build from its Cholesky factor).

    python tests/golden/make_mvnormal_golden.py
"""
import json
import os

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20261004)


def random_cov(d, cond):
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    ev = np.exp(np.linspace(0.0, np.log(cond), d))
    c = (q * ev) @ q.T
    return 0.5 * (c + c.T)


cases = []
for d, cond in ((2, 10.0), (3, 50.0), (8, 100.0), (16, 30.0), (32, 200.0)):
    mu = rng.uniform(-2.0, 2.0, d)
    cov = random_cov(d, cond)
    dist = stats.multivariate_normal(mu, cov)
    pts = [mu.copy(), np.zeros(d)] + [mu + np.linalg.cholesky(cov) @ rng.standard_normal(d) * s for s in (0.3, 1.0, 1.0, 3.0, 8.0)]
    cases.append({"mu": mu.tolist(), "cov": cov.tolist(), "x": [p.tolist() for p in pts], "logpdf": [float(dist.logpdf(p)) for p in pts]})
# the textbook 2-d case: unit variances, correlation 0.9
cov = np.array([[1.0, 0.9], [0.9, 1.0]])
dist = stats.multivariate_normal([0.0, 0.0], cov)
pts = [[0.0, 0.0], [1.0, 1.0], [1.0, -1.0], [-2.5, 0.3]]
cases.append({"mu": [0.0, 0.0], "cov": cov.tolist(), "x": pts, "logpdf": [float(dist.logpdf(p)) for p in pts]})

with open(os.path.join(HERE, "mvnormal_logpdf_scipy.json"), "w") as fh:
    json.dump({"generator": "tests/golden/make_mvnormal_golden.py", "scipy": __import__("scipy").__version__,
               "numpy": np.__version__, "cases": cases}, fh)
print(len(cases), "cases,", sum(len(c["x"]) for c in cases), "points")
