#!/usr/bin/env python3
"""Statistical check of abcdemc on BASELINE.json configs[1] (1-D Normal, eps 0.3, 2^20 particles) over several Philox seeds:
posterior mean against the closed form of the finite-eps posterior (tests/golden/reference_known_answers.json).

    python tools/mc_seeds.py [seeds, default 8] [generations, default 100]

ABCDEZ_HIP_LIB selects the build: the shipped library draws the better particle of mc:23 by rejection once 1 / 16 of the
population has arrived (include/abcdez_spec.h); a build with -DABZ_MC_REJECT_DIV=1 draws by rank in every unconverged
generation -- the same law through the other formulation, on another stream of random numbers."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import abcdez_amd as A

NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
GENS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N = 1 << 20
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_known_answers.json")) as f:
    exact = json.load(f)["analytic"]["Z_exact_finite_eps_sigma2_10"]["posterior_mean"]
rows = []
for seed in range(1, NSEEDS + 1):
    t0 = time.perf_counter()
    r = A.abcdemc(A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0), 0.3, None, nparticles=N, generations=GENS, verbose=False, rng=seed)
    dt = time.perf_counter() - t0
    P = np.asarray(r.P, dtype=np.float64).reshape(-1)
    ops = r.engine.ops
    rows.append({"seed": seed, "posterior_mean": float(P.mean()), "posterior_mean_err": float(P.mean()) - exact, "posterior_var": float(P.var()),
                 "completion": r.complete, "seconds": dt, "rank_passes": int(sum(ops.mc_rank_stats())),
                 "generations_by_rejection_without_one": int(ops.mc_draw_stats())})
    print(json.dumps(rows[-1]), flush=True)
e = np.array([x["posterior_mean_err"] for x in rows])
v = np.array([x["posterior_var"] for x in rows])
print(json.dumps({"lib": os.environ.get("ABCDEZ_HIP_LIB", "shipped"), "generations": GENS, "particles": N, "seeds": NSEEDS,
                  "exact_posterior_mean": exact, "mean_err": float(e.mean()), "std": float(e.std(ddof=1)),
                  "std_error_of_the_mean": float(e.std(ddof=1) / math.sqrt(len(e))),
                  "posterior_var_mean": float(v.mean()), "posterior_var_std": float(v.std(ddof=1))}))
