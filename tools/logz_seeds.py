#!/usr/bin/env python3
"""Statistical check of the headline configuration against its closed form: d = 32 MVN abcdesmc, 2^22 particles, run to
eps = 6.0 for several Philox seeds.  Exact evidence: Z = P(chi'^2_32(lambda = 16) < 18), posterior
mean per component from a Monte-Carlo-free argument is not available, so the means are compared across seeds."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import stats

import abcdez_amd as A

d, N = 32, 1 << 22
exact = math.log(stats.ncx2.cdf(36.0 / 2.0, df=d, nc=d / 2.0))     # |x - y|^2 / 2 ~ chi'^2_32(16): x - y ~ N(-1, 2 I)
prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
sim = A.MVNormal(tuple([1.0] * d))
rows = []
for seed in range(1, 9):
    t0 = time.perf_counter()
    r = A.abcdesmc(prior, sim, 6.0, None, nparticles=N, verbose=False, rng=seed, nsims_max=10 ** 12)
    dt = time.perf_counter() - t0
    alive = r.Wns > 0
    rows.append({"seed": seed, "logZ": r.logZ, "err": r.logZ - exact, "generations": r.iters, "seconds": dt,
                 "posterior_mean": float(r.P[alive].mean()), "updates": r.updates})
    print(rows[-1], flush=True)
errs = np.array([x["err"] for x in rows])
out = {"exact_logZ": exact, "runs": rows, "mean_err": float(errs.mean()), "std_err": float(errs.std(ddof=1)),
       "max_abs_err": float(np.abs(errs).max())}
print(json.dumps(out))
