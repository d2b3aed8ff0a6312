#!/usr/bin/env python3
"""Python/MI355X counterpart of the reference's examples/minimal_example.jl (lines 10-84):
two models for one datum (x = 3), sim x ~ N(θ, 1), priors N(0, √10) and N(0, √100);
posterior samples and evidences by abcdesmc, posterior by abcdemc, compared with the
analytic values.  Needs a GPU (the population loop has no CPU fallback).

    python examples/minimal_example.py [nparticles]
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
from scipy import stats

from abcdez_amd import Normal, Normal1D, abcdemc, abcdesmc

data, eps = 3.0, 0.3
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000         # the reference example uses 1000

results = {}
for name, sigma2 in (("model 1", 10.0), ("model 2", 100.0)):
    prior = Normal(0.0, math.sqrt(sigma2))
    r = abcdesmc(prior, Normal1D(data), eps, None, nparticles=N, verbose=False)       # minimal_example.jl:27-28, 52-53
    post = r.P[r.Wns > 0.0]
    exact_post = stats.norm(sigma2 / (sigma2 + 1) * data, math.sqrt(sigma2 / (sigma2 + 1)))   # :71-76
    exact_Z = stats.norm(0, math.sqrt(sigma2 + 1)).pdf(data) * 2 * eps                         # :79-84
    results[name] = r.logZ
    print(f"{name}: posterior mean {post.mean():.4f} (exact {exact_post.mean():.4f}), std {post.std():.4f} "
          f"(exact {exact_post.std():.4f}); evidence {math.exp(r.logZ):.5f} (exact {exact_Z:.5f}); "
          f"{r.iters} generations, {r.nsims} simulations")

print(f"Bayes factor model 1 / model 2: {math.exp(results['model 1'] - results['model 2']):.3f} (exact ratio of evidences 2.104)")
rmc = abcdemc(Normal(0.0, math.sqrt(10.0)), Normal1D(data), eps, None, nparticles=N, generations=300, verbose=False)
print(f"abcdemc model 1: posterior mean {np.mean(rmc.P):.4f}, converged={rmc.reached_ϵ}")
