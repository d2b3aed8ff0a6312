#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/.

reference_known_answers.json -- the exact and analytic known answers the reference's own
tests hold for this path (test/runtests.jl, line numbers cited per entry), restated as
data; closed forms re-evaluated with scipy.  Nothing is read from /root/reference.

spec_vectors.json -- outputs of the CPU oracle for fixed seeds (regression pins that the
GPU tests also compare against; the oracle is the spec, see oracle/abcdez_oracle.c).

    python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

INF = float("inf")


def kernel_truth_table():
    """(kernel, eps, x) -> (pdf, logpdf); test/runtests.jl:48-108"""
    rows = []
    # Indicator0toeps :48-60
    for eps, cases in ((0.1, [(0.1, 1.0, 0.0), (-0.1, 0.0, -INF), (0.2, 0.0, -INF)]),
                       (INF, [(0.1, 1.0, 0.0), (-0.1, 0.0, -INF), (0.2, 1.0, 0.0)])):
        rows += [("Indicator0toϵ", eps, x, p, lp) for x, p, lp in cases]
    # IndicatorStrict0toeps :62-76
    for eps, cases in ((0.1, [(0.1, 0.0, -INF), (0.01, 1.0, 0.0), (-0.1, 0.0, -INF), (0.2, 0.0, -INF)]),
                       (INF, [(0.1, 1.0, 0.0), (0.01, 1.0, 0.0), (-0.1, 0.0, -INF), (0.2, 1.0, 0.0)])):
        rows += [("IndicatorStrict0toϵ", eps, x, p, lp) for x, p, lp in cases]
    # Epa0toeps :78-92 and EpaStrict0toeps :94-108
    for name in ("Epa0toϵ", "EpaStrict0toϵ"):
        for eps, cases in ((0.1, [(0.1, 0.0, -INF), (0.0, 1.0, 0.0), (-0.1, 0.0, -INF), (0.2, 0.0, -INF)]),
                           (INF, [(0.1, 1.0, 0.0), (0.0, 1.0, 0.0), (-0.1, 0.0, -INF), (0.2, 1.0, 0.0)])):
            rows += [(name, eps, x, p, lp) for x, p, lp in cases]
    return [dict(kernel=k, eps=e, x=x, pdf=p, logpdf=lp) for k, e, x, p, lp in rows]


def analytic():
    eps = 0.3
    out = {}
    for data in (3, 7):                       # test/runtests.jl:110-121, :165-176
        ev = stats.norm(0, math.sqrt(11)).pdf(data)
        out[f"Z_indicator_data{data}"] = dict(value=ev * 2 * eps, rtol=0.1 if data == 3 else 0.2,
                                              posterior_mean=10 / 11 * data, posterior_std=math.sqrt(10 / 11))
    out["Z_epa_data3"] = dict(value=stats.norm(0, math.sqrt(11)).pdf(3) * (4 / 3) * eps, rtol=0.1)   # :334-336
    out["Z_uniform10"] = dict(value=0.02998511, rtol=0.2)   # :228 (10^8-sample rejection ground truth)
    out["Z_uniform20"] = dict(value=0.01500489, rtol=0.2)   # :229
    out["bayes_factor_uniform"] = dict(value=2.0, rtol=0.2)  # :259
    # exact finite-eps evidences of the two models of examples/minimal_example.jl:16-17,41-42
    for s2 in (10, 100):
        sd = math.sqrt(s2 + 1)
        z = stats.norm.cdf((3 + eps) / sd) - stats.norm.cdf((3 - eps) / sd)
        # posterior mean at finite eps (what test/runtests.jl:159-162,214-217 compare with 30/11 "within one std"; exact here):
        # x ~ N(0, s2 + 1) marginally, E[theta | x] = s2 / (s2 + 1) x, and the ABC posterior conditions on |x - 3| < eps:
        # E[x | a < x < b] = sd (phi(a / sd) - phi(b / sd)) / (Phi(b / sd) - Phi(a / sd))
        a_, b_ = (3 - eps) / sd, (3 + eps) / sd
        ex = sd * (stats.norm.pdf(a_) - stats.norm.pdf(b_)) / z
        out[f"Z_exact_finite_eps_sigma2_{s2}"] = dict(value=z, logZ=math.log(z), posterior_mean=s2 / (s2 + 1) * ex)
    # BASELINE.json config 3: d=32, prior N(0,I), x = theta + z, y = 1, eps = 6:
    # ||x - y||^2 / 2 ~ noncentral chi2(32, lambda = 32 / 2)
    z = stats.ncx2.cdf(36.0 / 2.0, 32, 16.0)
    # posterior mean per component: x = 1 + u, v = u / sqrt 2 ~ N(-1 / sqrt 2, I_32), A = {|v|^2 < 18}; for a noncentral
    # chi-square ball E[v | A] = mu F_{k+2}(r; lambda) / F_k(r; lambda), and E[theta | x] = x / 2
    # => E[theta_k | A] = (1 - F_34(18; 16) / F_32(18; 16)) / 2
    pm = 0.5 * (1.0 - stats.ncx2.cdf(18.0, 34, 16.0) / z)
    out["Z_mvn32_eps6"] = dict(value=z, logZ=math.log(z), posterior_mean_per_component=pm)
    z8 = stats.ncx2.cdf(2.5 ** 2 / 2.0, 8, 4.0)
    out["Z_mvn8_eps2.5"] = dict(value=z8, logZ=math.log(z8))
    out["mixture_st_n"] = [0.0, 0.04680825481526908, 0.1057221226763449, 0.2682111969397526, 0.8309228020477986]  # :575-579
    return out


def factored_cases():
    """test/runtests.jl:21-36"""
    return [
        dict(factors=[["Uniform", 0, 1], ["Uniform", 100, 101]], x=[0.0, 0.0], pdf=0.0, logpdf=-INF),
        dict(factors=[["Uniform", 0, 1], ["Uniform", 100, 101]], x=[0.5, 100.5], pdf=1.0, logpdf=0.0),
        dict(factors=[["Uniform", 0.0, 1.0], ["DiscreteUniform", 1, 2]], x=[0.3, 1], pdf=0.5, logpdf=math.log(0.5)),
        dict(factors=[["Uniform", 0.0, 1.0], ["DiscreteUniform", 1, 2]], x=[0.9, 2], pdf=0.5, logpdf=math.log(0.5)),
    ]


def push_cases():
    """test/runtests.jl:38-46 (value and type) + ties-to-even of Julia round(Int, x)"""
    return [
        dict(dist=["Normal", 0, 1], x=1, out=1.0, type="float"),
        dict(dist=["DiscreteUniform", 0, 1], x=1.0, out=1, type="int"),
        dict(dist=["DiscreteUniform", 0, 10], x=2.5, out=2, type="int"),
        dict(dist=["DiscreteUniform", 0, 10], x=3.5, out=4, type="int"),
        dict(dist=["DiscreteUniform", -10, 10], x=-0.5, out=0, type="int"),
        dict(dist=["DiscreteUniform", 0, 10], x=3.123, out=3, type="int"),
    ]


def spec_vectors():
    import abcdez_amd as A
    from oracle import oracle as O

    L = O.lib()
    out = {}
    w = np.zeros(2, dtype=np.uint64)
    L.orc_rng_words(1, 0, 0, 0, 6, w.ctypes.data)
    out["rng_words_seed1_idx0"] = [int(w[0]), int(w[1])]
    z = np.zeros(8)
    L.orc_normal_pairs(1, 6, 4, z.ctypes.data)
    out["normal_pairs_seed1"] = [float(v).hex() for v in z]
    runs = {}
    cases = {
        "normal1d_N2000": (A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, 2000, A.IndicatorStrict0toϵ),
        "mvn8_N2048": (A.Factored(*[A.Normal(0, 1)] * 8), A.MVNormal((1.0,) * 8), 2.5, 2048, A.IndicatorStrict0toϵ),
        "mvn32_N4096": (A.Factored(*[A.Normal(0, 1)] * 32), A.MVNormal((1.0,) * 32), 6.0, 4096, A.IndicatorStrict0toϵ),
        "normal1d_epa_N2000": (A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, 2000, A.Epa0toϵ),
    }
    for name, (prior, sim, eps, N, K) in cases.items():
        c = O.run_abcdesmc(A.ModelSpec(prior, sim, K, seed=2024), N, eps, nsims_max=10 ** 9)
        runs[name] = dict(seed=2024, N=N, eps_target=eps, logZ=float(c["logZ"]).hex(), iters=int(c["iters"]),
                          nsims=int(c["nsims"]), n_alive=int(c["alive"].sum()),
                          eps_hist=[float(v).hex() for v in c["eps_hist"]],
                          theta_sum=float(np.sum(c["theta"][c["alive"]])).hex(),
                          delta_sum=float(np.sum(c["C"])).hex())
    out["abcdesmc_runs"] = runs
    m = O.run_abcdemc(A.ModelSpec(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), seed=2024), 2000, 0.3, 50)
    out["abcdemc_normal1d_N2000_g50"] = dict(nsims=int(m["nsims"]), reached=bool(m["reached_eps"]),
                                             theta_sum=float(np.sum(m["theta"])).hex(),
                                             delta_sum=float(np.sum(m["C"])).hex())
    return out


def lv_fixture():
    """Lotka-Volterra observations for the on-device RK4 simulator (BASELINE.json configs[3]): theta* = (1, 0.4, 1, 0.3),
    (x0, y0) = (1, 0.5), RK4 dt = 0.01, 100 steps between the 16 observation times t = 0..15, N(0, 0.1^2) noise, seed 4."""
    a, b, c, e = 1.0, 0.4, 1.0, 0.3
    x, y, dt, steps, nobs = 1.0, 0.5, 0.01, 100, 16

    def f(x, y):
        return x * (a - b * y), y * (e * x - c)

    clean = []
    for j in range(nobs):
        clean += [x, y]
        if j + 1 == nobs:
            break
        for _ in range(steps):
            k1 = f(x, y)
            k2 = f(x + 0.5 * dt * k1[0], y + 0.5 * dt * k1[1])
            k3 = f(x + 0.5 * dt * k2[0], y + 0.5 * dt * k2[1])
            k4 = f(x + dt * k3[0], y + dt * k3[1])
            x += dt / 6 * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0])
            y += dt / 6 * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1])
    clean = np.array(clean)
    obs = clean + np.random.default_rng(4).normal(0, 0.1, clean.size)
    return dict(theta_star=[a, b, c, e], x0=1.0, y0=0.5, dt=dt, steps_per_obs=steps, noise=0.1, seed=4,
                obs=[float(v) for v in obs], clean=[float(v) for v in clean])


def main():
    with open(os.path.join(HERE, "lv_data.json"), "w") as f:
        json.dump(lv_fixture(), f, indent=1)
    ref = dict(kernel_truth_table=kernel_truth_table(), analytic=analytic(), factored=factored_cases(),
               push_p=push_cases())
    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as f:
        json.dump(ref, f, indent=1, ensure_ascii=False)
    with open(os.path.join(HERE, "spec_vectors.json"), "w") as f:
        json.dump(spec_vectors(), f, indent=1)
    print("wrote", HERE)


if __name__ == "__main__":
    main()
