#!/usr/bin/env python3
"""Statistical check of the headline configuration against its closed form: d = 32 MVN abcdesmc, 2^22 particles, run to
eps = 6.0 for several Philox seeds.  Exact evidence: Z = P(chi'^2_32(lambda = 16) < 18); exact posterior mean
per component (1 - F_34(18; 16) / F_32(18; 16)) / 2 (tests/golden/make_golden.py).   python tools/logz_seeds.py [seeds, default 8]"""
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
exact_pm = 0.5 * (1.0 - stats.ncx2.cdf(18.0, d + 2, d / 2.0) / stats.ncx2.cdf(18.0, d, d / 2.0))
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
sim = A.MVNormal(tuple([1.0] * d))
rows = []
for seed in range(1, NSEEDS + 1):
    t0 = time.perf_counter()
    r = A.abcdesmc(prior, sim, 6.0, None, nparticles=N, verbose=False, rng=seed, nsims_max=10 ** 12)
    dt = time.perf_counter() - t0
    alive = r.Wns > 0
    rows.append({"seed": seed, "logZ": r.logZ, "err": r.logZ - exact, "generations": r.iters, "seconds": dt,
                 "posterior_mean": float(r.P[alive].mean()), "posterior_mean_err": float(r.P[alive].mean()) - exact_pm,
                 "updates": r.updates})
    print(rows[-1], flush=True)
errs = np.array([x["err"] for x in rows])
pme = np.array([x["posterior_mean_err"] for x in rows])
from abcdez_amd import _lib as _L
lib = _L.load()
out = {"exact_logZ": exact, "exact_posterior_mean_per_component": exact_pm, "philox_rounds": int(lib.abcdez_rng_rounds()),
       "runs": rows, "mean_err": float(errs.mean()), "std_err": float(errs.std(ddof=1)),
       "std_error_of_the_mean": float(errs.std(ddof=1) / math.sqrt(len(errs))), "max_abs_err": float(np.abs(errs).max()),
       "posterior_mean_mean_err": float(pme.mean()), "posterior_mean_std": float(pme.std(ddof=1)),
       "posterior_mean_max_abs_err": float(np.abs(pme).max())}
print(json.dumps(out))
