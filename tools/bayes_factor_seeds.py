#!/usr/bin/env python3
"""BASELINE.json configs[4] (the reference's minimal_example.jl at 2^23 particles) for several Philox seeds: log evidences of
the two models and their Bayes factor against the exact finite-eps values (the acceptance region |x - 3| <= 0.3 under the
prior predictive N(0, sigma^2 + 1))."""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import stats

import abcdez_amd as A

N, eps, data = 1 << 23, 0.3, 3.0
exact = {v: math.log(stats.norm.cdf(data + eps, 0, math.sqrt(v + 1)) - stats.norm.cdf(data - eps, 0, math.sqrt(v + 1))) for v in (10, 100)}
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def exact_pm(v):
    sd = math.sqrt(v + 1)
    a, b = (data - eps) / sd, (data + eps) / sd
    return v / (v + 1) * sd * (stats.norm.pdf(a) - stats.norm.pdf(b)) / (stats.norm.cdf(b) - stats.norm.cdf(a))


rows = []
for seed in range(1, NSEEDS + 1):
    lz, pm = {}, {}
    for v in (10, 100):
        r = A.abcdesmc(A.Normal(0.0, math.sqrt(v)), A.Normal1D(data), eps, None, nparticles=N, verbose=False, rng=seed, nsims_max=10 ** 12)
        lz[v] = r.logZ
        pm[v] = float(r.P[r.Wns > 0].mean()) - exact_pm(v)
    rows.append({"seed": seed, "logZ1": lz[10], "logZ2": lz[100], "bayes_factor": math.exp(lz[10] - lz[100]),
                 "posterior_mean_err1": pm[10], "posterior_mean_err2": pm[100]})
    print(rows[-1], flush=True)
bf = np.array([x["bayes_factor"] for x in rows])
e1 = np.array([x["logZ1"] - exact[10] for x in rows]); e2 = np.array([x["logZ2"] - exact[100] for x in rows])
print(json.dumps({"exact_logZ1": exact[10], "exact_logZ2": exact[100], "exact_bayes_factor": math.exp(exact[10] - exact[100]), "runs": rows,
                  "bayes_factor_mean": float(bf.mean()), "bayes_factor_std": float(bf.std(ddof=1)),
                  "logZ1_mean_err": float(e1.mean()), "logZ1_std": float(e1.std(ddof=1)),
                  "logZ2_mean_err": float(e2.mean()), "logZ2_std": float(e2.std(ddof=1)),
                  "philox_rounds": int(__import__("abcdez_amd._lib", fromlist=["load"]).load().abcdez_rng_rounds()),
                  "exact_posterior_mean1": exact_pm(10), "exact_posterior_mean2": exact_pm(100),
                  "posterior_mean1_mean_err": float(np.mean([x["posterior_mean_err1"] for x in rows])),
                  "posterior_mean1_std": float(np.std([x["posterior_mean_err1"] for x in rows], ddof=1)),
                  "posterior_mean2_mean_err": float(np.mean([x["posterior_mean_err2"] for x in rows])),
                  "posterior_mean2_std": float(np.std([x["posterior_mean_err2"] for x in rows], ddof=1))}))
