#!/usr/bin/env python3
"""The reference's examples/minimal_example.jl on the GPU engine: posterior samples and model evidences of two models
that differ in their prior, model posterior probabilities, and the analytic values to compare with.

    python examples/minimal_example.py [nparticles]

Every statement mirrors the Julia example (line numbers of examples/minimal_example.jl in the comments); the one
difference a user has to make is the `dist!` closure: `abs(rand(Normal(θ, 1)) - data)` becomes the device simulator
`Normal1D(data)` (or a `UserSimulator` with HIP source text)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from abcdez_amd import Normal, Normal1D, abcdesmc


def normal_pdf(x, mu, sigma):
    return math.exp(-0.5 * ((x - mu) / sigma) ** 2) / (sigma * math.sqrt(2 * math.pi))


def main(nparticles=1000, verbose=True):
    data = 3                                                  # :10
    eps = 0.3                                                 # :13
    # model 1                                                 # :16-28
    var1 = 10
    prior1 = Normal(0, math.sqrt(var1))
    dist1 = Normal1D(data)                                    # model1(θ) = rand(Normal(θ, 1)); dist1!(θ, ve) = abs(model1(θ) - data)
    # nsims_max keeps the reference's default (10^7) at the example's 1000 particles and grows with the population, so that
    # larger runs reach the target too instead of stopping on the simulation budget (src/abcdez_smc.jl:376)
    nsims_max = 10 ** 7 * max(1, nparticles // 1000)
    r1 = abcdesmc(prior1, dist1, eps, None, nparticles=nparticles, verbose=False, nsims_max=nsims_max)
    posterior1 = r1.P[r1.Wns > 0.0]                           # :34
    evidence1 = math.exp(r1.logZ)                             # :37
    # model 2: the same model under a wider prior             # :41-56
    var2 = 100
    prior2 = Normal(0, math.sqrt(var2))
    r2 = abcdesmc(prior2, Normal1D(data), eps, None, nparticles=nparticles, verbose=False, nsims_max=nsims_max)
    posterior2 = r2.P[r2.Wns > 0.0]
    evidence2 = math.exp(r2.logZ)
    # model probabilities                                     # :60-65
    mprior1 = mprior2 = 0.5
    mposterior1 = evidence1 * mprior1 / (evidence1 * mprior1 + evidence2 * mprior2)
    mposterior2 = evidence2 * mprior2 / (evidence1 * mprior1 + evidence2 * mprior2)
    # analytical comparison                                   # :71-83
    post1_exact = (var1 / (var1 + 1) * data, math.sqrt(var1 / (var1 + 1)))
    post2_exact = (var2 / (var2 + 1) * data, math.sqrt(var2 / (var2 + 1)))
    evidence1_expected = normal_pdf(data, 0.0, math.sqrt(var1 + 1)) * 2 * eps      # the indicator kernel's normalisation: 2ϵ
    evidence2_expected = normal_pdf(data, 0.0, math.sqrt(var2 + 1)) * 2 * eps
    mposterior1_exact = evidence1_expected / (evidence1_expected + evidence2_expected)
    out = dict(posterior1=(float(np.mean(posterior1)), float(np.std(posterior1))), posterior1_exact=post1_exact,
               posterior2=(float(np.mean(posterior2)), float(np.std(posterior2))), posterior2_exact=post2_exact,
               evidence1=evidence1, evidence1_expected=evidence1_expected, evidence2=evidence2,
               evidence2_expected=evidence2_expected, mposterior1=mposterior1, mposterior2=mposterior2,
               mposterior1_exact=mposterior1_exact, mposterior2_exact=1.0 - mposterior1_exact)
    if verbose:
        for k, v in out.items():
            print(f"{k:22s} {v}")
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
