#!/usr/bin/env python3
"""scipy.stats log-densities of the WRAPPER prior families -- truncated(d, lo, hi) of parents other than Normal, MixtureModel of
univariate components -- at fixed points: tests/golden/prior_wrappers_scipy.json.  Independent of include/abcdez_spec.h and of
abcdez_amd/priors.py (their cdfs included: the truncation mass here is scipy's).

    python tests/golden/make_prior_wrappers_golden.py
"""
import json
import math
import os

import numpy as np
from scipy import stats

PARENTS = {
    "Gamma": lambda a, t: stats.gamma(a, scale=t), "Cauchy": lambda m, s: stats.cauchy(m, s), "Exponential": lambda t: stats.expon(scale=t),
    "LogNormal": lambda m, s: stats.lognorm(s=s, scale=math.exp(m)), "Laplace": lambda m, t: stats.laplace(m, t),
    "Weibull": lambda a, t: stats.weibull_min(a, scale=t), "InverseGamma": lambda a, t: stats.invgamma(a, scale=t),
    "Logistic": lambda m, t: stats.logistic(m, t), "TDist": lambda n: stats.t(n), "Pareto": lambda a, t: stats.pareto(a, scale=t),
    "Beta": lambda a, b: stats.beta(a, b), "Uniform": lambda a, b: stats.uniform(a, b - a), "Normal": lambda m, s: stats.norm(m, s),
    "Poisson": lambda l: stats.poisson(l), "Binomial": lambda n, p: stats.binom(n, p), "NegativeBinomial": lambda r, p: stats.nbinom(r, p),
    "DiscreteUniform": lambda a, b: stats.randint(a, b + 1),
}
DISCRETE = {"Poisson", "Binomial", "NegativeBinomial", "DiscreteUniform"}

TRUNCATED = [  # (parent, params, lo, hi) -- None = unbounded
    ("Gamma", [2.5, 0.6], 0.5, 3.0), ("Gamma", [0.7, 2.0], 0.1, None), ("Cauchy", [0.0, 2.0], -1.0, None), ("Cauchy", [1.0, 0.5], -2.0, 4.0),
    ("Exponential", [1.5], 0.2, 5.0), ("LogNormal", [0.0, 0.5], 0.5, 2.5), ("Laplace", [1.0, 1.0], None, 2.0), ("Weibull", [1.8, 1.2], 0.3, 2.5),
    ("InverseGamma", [3.0, 2.0], 0.4, 3.0), ("Logistic", [1.0, 0.5], 0.0, 3.0), ("TDist", [4.0], -1.0, 6.0), ("Pareto", [3.0, 0.5], 0.6, 2.0),
    ("Beta", [2.0, 3.0], 0.1, 0.7), ("Uniform", [-1.0, 3.0], 0.0, 2.0), ("Poisson", [4.0], 2, 9), ("Binomial", [12, 0.3], 1, 6),
    ("NegativeBinomial", [4.6, 0.13], 5, 60), ("DiscreteUniform", [1, 10], 3, 7),
]
AFFINE = [  # (parent, params, mu, sigma): mu + sigma * parent
    ("TDist", [4.0], 1.0, 2.0), ("Gamma", [2.5, 0.6], -1.0, 3.0), ("Beta", [2.0, 3.0], 10.0, 5.0), ("Logistic", [0.0, 1.0], 2.0, 0.25),
    ("Exponential", [1.0], 0.5, 0.1),
]
MIXTURES = [  # ([(family, params), ...], weights)
    ([("Normal", [-1.0, 0.5]), ("Normal", [2.0, 1.0]), ("Laplace", [0.0, 2.0])], [0.2, 0.5, 0.3]),
    ([("Gamma", [2.0, 1.0]), ("Exponential", [0.5])], [0.6, 0.4]),
    ([("Normal", [0.0, 1.0])], [1.0]),
    ([("Uniform", [0.0, 1.0]), ("Uniform", [0.5, 3.0]), ("Cauchy", [5.0, 0.3]), ("TDist", [3.0])], [0.1, 0.2, 0.3, 0.4]),
    ([("Poisson", [2.0]), ("Binomial", [12, 0.4])], [0.4, 0.6]),
    ([("DiscreteUniform", [0, 3]), ("NegativeBinomial", [2.0, 0.5]), ("Poisson", [7.0])], [0.5, 0.25, 0.25]),
]


def logpmf_or_pdf(ref, fam, x):
    return float(ref.logpmf(x)) if fam in DISCRETE else float(ref.logpdf(x))


def points(lo, hi, discrete):
    a = -6.0 if lo is None else lo
    b = 14.0 if hi is None else hi
    if discrete:
        return sorted({int(v) for v in np.arange(math.floor(a) - 1, math.ceil(b) + 2)} | {0.5})
    return sorted(set(np.round(np.linspace(a - 0.5, b + 0.5, 17), 6)) | {a, b, 0.5 * (a + b)})


def enc(v):
    return "-inf" if v == -math.inf else ("inf" if v == math.inf else v)


cases = []
for fam, p, lo, hi in TRUNCATED:
    ref = PARENTS[fam](*p)
    disc = fam in DISCRETE
    lo_, hi_ = (-math.inf if lo is None else lo), (math.inf if hi is None else hi)
    below = (ref.cdf(math.ceil(lo_) - 1) if disc else ref.cdf(lo_)) if lo is not None else 0.0
    mass = (ref.cdf(hi_) if hi is not None else 1.0) - below
    pts = []
    for x in points(lo, hi, disc):
        inside = lo_ <= x <= hi_
        lp = logpmf_or_pdf(ref, fam, x) - math.log(mass) if inside else -math.inf
        pts.append({"x": x, "logpdf": enc(lp if np.isfinite(lp) else -math.inf)})
    cases.append({"kind": "truncated", "parent": fam, "p": p, "lo": lo, "hi": hi, "mass": float(mass), "points": pts})
for fam, p, mu, sg in AFFINE:
    ref = PARENTS[fam](*p)
    pts = []
    for x in points(mu - 3.0 * sg, mu + 6.0 * sg, False):
        lp = float(ref.logpdf((x - mu) / sg)) - math.log(sg)
        pts.append({"x": x, "logpdf": enc(lp if np.isfinite(lp) else -math.inf)})
    cases.append({"kind": "affine", "parent": fam, "p": p, "mu": mu, "sigma": sg, "points": pts})
for comps, w in MIXTURES:
    disc = comps[0][0] in DISCRETE
    pts = []
    for x in points(-4.0 if not disc else -1, 12.0, disc):
        terms = [math.log(wj) + logpmf_or_pdf(PARENTS[f](*p), f, x) for (f, p), wj in zip(comps, w)]
        m = max(terms)
        lp = -math.inf if m == -math.inf else m + math.log(sum(math.exp(t - m) for t in terms))
        pts.append({"x": x, "logpdf": enc(lp)})
    cases.append({"kind": "mixture", "components": [[f, p] for f, p in comps], "weights": w, "points": pts})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prior_wrappers_scipy.json")
json.dump({"made_by": "tests/golden/make_prior_wrappers_golden.py", "scipy": __import__("scipy").__version__, "cases": cases}, open(out, "w"), indent=1)
print(out, sum(len(c["points"]) for c in cases), "points in", len(cases), "cases")
