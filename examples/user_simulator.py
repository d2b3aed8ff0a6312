#!/usr/bin/env python3
"""A model of one's own on the device: what the reference takes as a Julia closure `dist!(θ, ve)` (src/abcdez_smc.jl:137) is HIP
source text here, compiled when the run starts (`UserSimulator`; include/abcdez_hip.h: abcdez_ctx_create_user).

The model: an SIR epidemic, dS/dt = -β S I, dI/dt = β S I - γ I, by classical RK4; observed: the infected fraction at 16 times
with Normal noise; unknown: (β, γ, I₀).  Written in the STAGED form -- `abz_user_round` advances the trajectory by two observations
and returns the square root of the running sum of squared errors, a lower bound of the final distance that only grows -- so that
a proposal whose bound has passed ϵ leaves the simulation early (csrc/abz_user_rounds.h); the results are bit for bit those of
running every trajectory to the end.  The same model as one opaque `abz_user_dist` call is `SIR_OPAQUE` below.

    python examples/user_simulator.py [nparticles]
"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from abcdez_amd import Factored, LogNormal, Uniform, UserSimulator, abcdesmc

N_OBS, STEPS, DT, NOISE = 16, 50, 0.02, 0.01          # an observation every STEPS * DT = 1 time unit

SIR_STEP = """
__device__ inline void sir_rk4(double beta, double gamma, double h, int steps, double& S, double& I) {
  for (int s = 0; s < steps; ++s) {
    const double a1 = -beta * S * I,               b1 = beta * S * I - gamma * I;
    const double S2 = S + 0.5 * h * a1,            I2 = I + 0.5 * h * b1;
    const double a2 = -beta * S2 * I2,             b2 = beta * S2 * I2 - gamma * I2;
    const double S3 = S + 0.5 * h * a2,            I3 = I + 0.5 * h * b2;
    const double a3 = -beta * S3 * I3,             b3 = beta * S3 * I3 - gamma * I3;
    const double S4 = S + h * a3,                  I4 = I + h * b3;
    const double a4 = -beta * S4 * I4,             b4 = beta * S4 * I4 - gamma * I4;
    S += h / 6.0 * (a1 + 2.0 * a2 + 2.0 * a3 + a4);
    I += h / 6.0 * (b1 + 2.0 * b2 + 2.0 * b3 + b4);
  }
}
"""

# sim_p = (dt, steps per observation, noise); data = the observed infected fractions
SIR_STAGED = """
#define ABZ_USER_ROUNDS 8
#define ABZ_USER_STATE 3                           /* S, I, the running sum of squared errors */
""" + SIR_STEP + """
__device__ double abz_user_round(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng,
                                 int round, double* st) {
  const double beta = th[0], gamma = th[1];
  double S = round == 0 ? 1.0 - th[2] : st[0], I = round == 0 ? th[2] : st[1], acc = st[2];
  const int per = n_data / ABZ_USER_ROUNDS;
  for (int j = round * per; j < (round + 1) * per; ++j) {
    sir_rk4(beta, gamma, p[0], (int)p[1], S, I);
    const double e = I + p[2] * rng.normal() - data[j];
    acc += e * e;
  }
  st[0] = S; st[1] = I; st[2] = acc;
  return sqrt(acc);
}
"""

SIR_OPAQUE = SIR_STEP + """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  double S = 1.0 - th[2], I = th[2], acc = 0.0;
  for (int j = 0; j < n_data; ++j) {
    sir_rk4(th[0], th[1], p[0], (int)p[1], S, I);
    const double e = I + p[2] * rng.normal() - data[j];
    acc += e * e;
  }
  return sqrt(acc);
}
"""


def sir_observations(beta, gamma, i0, rng):
    """the data: the same integrator on the host, one noisy observation per time unit"""
    S, I, out = 1.0 - i0, i0, []
    f = lambda S, I: (-beta * S * I, beta * S * I - gamma * I)              # noqa: E731
    for _ in range(N_OBS):
        for _ in range(STEPS):
            a1, b1 = f(S, I)
            a2, b2 = f(S + 0.5 * DT * a1, I + 0.5 * DT * b1)
            a3, b3 = f(S + 0.5 * DT * a2, I + 0.5 * DT * b2)
            a4, b4 = f(S + DT * a3, I + DT * b3)
            S += DT / 6.0 * (a1 + 2 * a2 + 2 * a3 + a4)
            I += DT / 6.0 * (b1 + 2 * b2 + 2 * b3 + b4)
        out.append(I + NOISE * rng.normal())
    return tuple(out)


def main(nparticles=200_000, eps=0.08, seed=1, source=SIR_STAGED, verbose=True):
    truth = (1.2, 0.4, 0.01)
    data = sir_observations(*truth, np.random.default_rng(7))
    prior = Factored(Uniform(0.2, 3.0), Uniform(0.05, 1.5), LogNormal(math.log(0.01), 1.0))
    sim = UserSimulator(source, params=(DT, float(STEPS), NOISE), data=data)
    t = time.perf_counter()
    r = abcdesmc(prior, sim, eps, None, nparticles=nparticles, verbose=False, rng=seed, nsims_max=10 ** 11)
    dt = time.perf_counter() - t
    alive = r.Wns > 0
    P = np.array(r.P, dtype=np.float64)[alive]
    mean, sd = P.mean(axis=0), P.std(axis=0)
    if verbose:
        print(f"{nparticles} particles, {r.iters} generations, {r.nsims:.3e} simulated trajectories in {dt:.2f} s (compile included); "
              f"ϵ = {r.ϵ:.4f}, log evidence {r.logZ:.3f}")
        for name, m, s, t0 in zip(("β", "γ", "I₀"), mean, sd, truth):
            print(f"  {name}: posterior {m:.4f} ± {s:.4f}   (data generated at {t0})")
    return r, mean, sd, truth


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200_000)
