#!/usr/bin/env python3
"""Differential campaign: random models and random driver settings through the HIP library and through the CPU oracle, whole runs
compared bit for bit.  Wider than the seeded cases of tests/test_gpu_parity.py (which this reuses the prior generator of): population
sizes that are not multiples of anything, Kmcmc beyond the 16 sweeps one library call takes, every early-exit / tuning / stopping
keyword of src/abcdez_smc.jl:215-235, targets the population cannot reach, a few abcdemc generations of the same model.  One JSON line per case; exit
status 1 at the first difference (the case's seed reproduces it: --first SEED --cases 1).

    python tools/fuzz_parity.py --cases 400 --first 0 > gpurun_out/fuzz.jsonl
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import abcdez_amd as A                                   # noqa: E402
from oracle import oracle as O                           # noqa: E402  (the checker; tools are test infrastructure)
from test_gpu_parity import _further_family              # noqa: E402


def random_prior(rng, d, nfam):
    fams = []
    for _ in range(d):
        k = int(rng.integers(0, nfam))
        if k >= 5:
            fams.append(_further_family(k - 5, rng))
        elif k == 0:
            fams.append(A.Normal(float(rng.normal(0.5, 1.0)), float(rng.uniform(0.3, 2.0))))
        elif k == 1:
            a = float(rng.uniform(-3, 1))
            fams.append(A.Uniform(a, a + float(rng.uniform(1.0, 5.0))))
        elif k == 2:
            a = int(rng.integers(-3, 2))
            fams.append(A.DiscreteUniform(a, a + int(rng.integers(1, 6))))
        elif k == 3:
            fams.append(A.Beta(float(rng.uniform(0.6, 4.0)), float(rng.uniform(0.6, 4.0))))
        else:
            fams.append(A.NegativeBinomial(float(rng.uniform(0.7, 6.0)), float(rng.uniform(0.2, 0.8))))
    return fams[0] if d == 1 and rng.random() < 0.5 else A.Factored(*fams)


def random_case(seed, big=False, small=False):
    rng = np.random.default_rng(50_000 + seed)
    u = rng.random() * (0.15 if small else 1.0)         # small: 1 to 4 parameters only -- the test problems of test/runtests.jl
    d = int(rng.integers(1, 5)) if u < 0.15 else int(rng.integers(1, 49)) if u < 0.9 else int(rng.integers(65, 141))
    nfam = [5, 18, 25][int(rng.integers(0, 3))] if d <= 48 else 5          # (wide rows: the reference's five families, as the suite)
    sim_kind = "mvn"
    if d == 1 and rng.random() < 0.5:
        sim_kind = "normal1d"
    elif d == 4 and rng.random() < 0.5:
        sim_kind = "lv"
    elif d <= 2 and rng.random() < 0.6:                  # the other test problems of test/runtests.jl, their own priors perturbed
        sim_kind = [["dirac", "mixture01"], ["quad2d", "normdu", "wiener", "socks"]][d - 1][int(rng.integers(0, 2 if d == 1 else 4))]
    if sim_kind in ("dirac", "mixture01", "quad2d", "normdu", "wiener", "socks"):
        b = bool(rng.random() < 0.3)
        u = lambda lo, hi: float(rng.uniform(lo, hi))                                   # noqa: E731
        if sim_kind == "dirac":
            prior, sim = A.Normal(u(0.8, 1.2), u(0.15, 0.4)), A.DiracSquare(u(1.3, 2.5), blobs=b)
        elif sim_kind == "mixture01":
            prior, sim = A.Uniform(-10.0, u(5.0, 10.0)), A.Mixture01(u(-1.0, 1.0), blobs=b)
        elif sim_kind == "quad2d":
            prior, sim = A.Factored(A.Normal(0, u(2.0, 5.0)), A.Normal(0, u(2.0, 5.0))), A.Quad2D([0.0, 0.1, 0.5][int(rng.integers(0, 3))], blobs=b)
        elif sim_kind == "normdu":
            prior, sim = A.Factored(A.Normal(1, u(0.3, 0.8)), A.DiscreteUniform(1, int(rng.integers(4, 12)))), A.NormalTimesDU(u(3.0, 8.0), blobs=b)
        elif sim_kind == "wiener":
            mu, sg = u(0.2, 0.8), u(0.5, 2.0)
            tdata = tuple(math.sqrt(mu * mu * t * t + sg * sg * t) for t in range(int(rng.integers(5, 30))))
            prior, sim = A.Factored(A.Uniform(0, 1), A.Uniform(0, 4)), A.WienerRMS(tdata, blobs=b)
        else:
            r_ = u(3.0, 7.0)
            prior = A.Factored(A.NegativeBinomial(r_, r_ / (u(20.0, 40.0) + r_)), A.Beta(u(8.0, 20.0), u(1.5, 4.0)))
            sim = A.Socks(float(rng.integers(0, 3)), float(rng.integers(7, 12)), blobs=b)
    elif sim_kind == "lv":
        prior = A.Factored(*[A.Uniform(0.0, float(rng.uniform(1.5, 2.5))) for _ in range(4)])
        obs = (1.0, 0.5, 1.46, 0.43, 1.77, 0.62, 1.52, 1.13, 0.95, 1.31, 0.66, 1.09, 0.61, 0.79, 0.75, 0.6)
        sim = A.LotkaVolterraRK4(obs, dt=0.05, steps_per_obs=int(rng.integers(4, 12)), blobs=bool(rng.random() < 0.3))
    elif sim_kind == "normal1d":
        prior = random_prior(rng, 1, nfam)
        sim = A.Normal1D(float(rng.normal(1.0, 1.0)), blobs=bool(rng.random() < 0.3))
    else:
        prior = random_prior(rng, d, nfam)
        y = tuple(float(v) for v in rng.normal(1.0, 0.5, d))
        sim = A.MVNormal(y, sigma=float(rng.uniform(0.5, 1.5)), blobs=bool(rng.random() < 0.3) and d <= 64)
    kern = [A.IndicatorStrict0toϵ, A.Indicator0toϵ, A.Epa0toϵ, A.EpaStrict0toϵ][int(rng.integers(0, 4))]
    α = float(rng.uniform(0.3, 0.97))
    δess = float(rng.uniform(0.1, 0.9))
    n_min = int(math.ceil(3 * d / min(α, δess)))
    N = n_min + int(rng.integers(0, 3000)) if rng.random() < 0.9 else n_min + int(rng.integers(3000, 20000))
    if big:                                             # populations of many workgroups: multi-block scans, lists, the radix rank pass
        N = n_min + int(rng.integers(50_000, 600_000))
    Kmcmc = int(rng.integers(1, 7)) if rng.random() < 0.93 else int(rng.integers(17, 21))
    Kmcmc_min = [1.0, 1.0, 0.3, 0.1, float("inf"), 0.0, 2.5][int(rng.integers(0, 7))]
    facc_min = [0.0, 0.0, 0.3, 0.6][int(rng.integers(0, 4))]
    facc_stop = [0.0, 0.0, 0.0, 0.05, 0.2][int(rng.integers(0, 5))]
    if not Kmcmc_min > facc_min:                         # the reference only warns (smc:233); keep the log quiet
        facc_min = 0.0
    q = [0.3, 0.3, 0.3, 0.05, 0.6, 0.0][int(rng.integers(0, 6))]     # 0.0: a target below every distance (the run ends some other way)
    nsims_max = 10 ** 8 if rng.random() < 0.8 else int(N * rng.uniform(1.0, 8.0))
    return dict(seed=seed, d=d, nfam=nfam, sim=sim_kind, prior=prior, simulator=sim, ABCk=kern, N=N, q=q,
                smc=dict(α=α, δess=δess, Kmcmc=Kmcmc, Kmcmc_min=Kmcmc_min, facc_min=facc_min, facc_stop=facc_stop,
                         facc_tune=float(rng.uniform(0.8, 0.99)), nsims_max=nsims_max),
                mc=dict(generations=int(rng.integers(2, 12))))


# the d-dimensional Normal simulator as a USER would write it, one thread per row of LD = 4, 8 or 16 doubles (3 to 16 parameters);
# wider rows take tests/user_sources.py's cooperative form
USER_MVN_ROW = """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  double sq[%(ld)d];
  for (int m = 0; m < %(ld)d / 2; ++m) {
    double z[2];
    rng.normal_pair(z[0], z[1]);
    for (int c = 0; c < 2; ++c) {
      const int k = 2 * m + c;
      double v = 0.0;
      if (k < d) { const double x = abz_fma(p[0], z[c], th[k]); const double e = x - data[k]; v = e * e; }
      sq[k] = v;
    }
  }
  return abz_sqrt(abz_tree_sum_small(sq, %(ld)d));
}
"""


def user_form_of(sim, d, rng):
    """the built-in d-dimensional Normal simulator restated as user-supplied source (None: no user form for this case)"""
    if type(sim).__name__ != "MVNormal" or d < 3 or sim.blobs:
        return None
    from user_sources import USER_MVN_LANES

    y, sigma = tuple(sim.data()), sim.params()[0]
    if d <= 16:
        ld = 4 if d <= 4 else 8 if d <= 8 else 16
        os.environ["ABZ_USER_ONE_KERNEL"] = "1" if rng.random() < 0.3 else "0"       # read when the context is created
        return A.UserSimulator(USER_MVN_ROW % {"ld": ld}, params=(sigma,), data=y)
    return A.UserSimulator(USER_MVN_LANES, params=(sigma,), data=y)


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def run_case(c):
    prior, sim, kern, N = c["prior"], c["simulator"], c["ABCk"], c["N"]
    seed = c["seed"]
    probe = O.oracle_engine(A.ModelSpec(prior, sim, kern, seed=seed + 1), N)
    probe.init_population()
    eps = probe.quantile_alive(c["q"]) if c["q"] > 0 else 0.0
    kw = dict(nparticles=N, verbose=False, rng=seed + 1, ABCk=kern, max_iters=c.get("max_iters", 30), **c["smc"])
    selftest = os.environ.get("ABZ_FUZZ_SELFTEST") == "1"       # both sides the oracle: checks this script where there is no GPU
    hip = dict(engine=O.oracle_engine) if selftest else {}
    hip_sim = (None if selftest else c.get("user_sim")) or sim      # (the oracle always runs the built-in simulator the source restates)
    r = A.abcdesmc(prior, hip_sim, eps, None, **hip, **kw)
    o = A.abcdesmc(prior, sim, eps, None, engine=O.oracle_engine, **kw)
    assert selftest or type(r.engine.ops).__name__ == "HipOps"
    bad = []
    if not (r.iters == o.iters and r.nsims == o.nsims):
        bad.append(("iters/nsims", (r.iters, r.nsims), (o.iters, o.nsims)))
    if not (r.logZ == o.logZ or (math.isnan(r.logZ) and math.isnan(o.logZ))):
        bad.append(("logZ", r.logZ, o.logZ))
    for k in ("ϵs", "esss", "faccs", "γ0s", "Kmcmcs", "logZs"):
        if not same(getattr(r, k), getattr(o, k)):
            bad.append((k, list(getattr(r, k))[-3:], list(getattr(o, k))[-3:]))
    if not same(np.array(r.ranges_ϵ), np.array(o.ranges_ϵ)):
        bad.append(("ranges_ϵ",))
    for k in ("P", "Wns", "C"):
        if not same(getattr(r, k), getattr(o, k)):
            bad.append((k,))
    if sim.blobs and not same(r.blobs, o.blobs):
        bad.append(("blobs",))
    mkw = dict(nparticles=N, verbose=False, rng=seed + 2, **c["mc"])
    m = A.abcdemc(prior, hip_sim, eps, None, **hip, **mkw)
    mo = A.abcdemc(prior, sim, eps, None, engine=O.oracle_engine, **mkw)
    if not (m.nsims == mo.nsims and same(m.P, mo.P) and same(m.C, mo.C) and m.reached_ϵ == mo.reached_ϵ):
        bad.append(("abcdemc", m.nsims, mo.nsims))
    if sim.blobs and not same(m.blobs, mo.blobs):
        bad.append(("abcdemc blobs",))
    if c.get("resume") and not selftest:
        # interrupted and continued (max_iters / resume=checkpoint, a few abcdemc generations then the rest): the same bits as the
        # uninterrupted run of the library
        rs = np.random.default_rng(seed + 5)
        if r.iters >= 2:
            k = int(rs.integers(1, r.iters))
            part = A.abcdesmc(prior, hip_sim, eps, None, **{**kw, "max_iters": k})
            cont = A.abcdesmc(prior, hip_sim, eps, None, resume=part.checkpoint(), **kw)
            if not (cont.iters == r.iters and cont.nsims == r.nsims and (cont.logZ == r.logZ or math.isnan(r.logZ)) and same(cont.P, r.P)
                    and same(cont.Wns, r.Wns) and same(cont.C, r.C) and same(cont.ϵs, r.ϵs) and same(cont.esss, r.esss)):
                bad.append(("abcdesmc resumed after %d of %d generations" % (k, r.iters),))
        G = mkw["generations"]
        if G >= 2:
            g1 = int(rs.integers(1, G))
            part = A.abcdemc(prior, hip_sim, eps, None, **{**mkw, "generations": g1})
            cont = A.abcdemc(prior, hip_sim, eps, None, resume=part.checkpoint(), **mkw)
            if not (cont.nsims == m.nsims and same(cont.P, m.P) and same(cont.C, m.C)):
                bad.append(("abcdemc resumed after %d of %d generations" % (g1, G),))
    return dict(eps=eps, iters=r.iters, nsims=r.nsims, logZ=r.logZ, resamples=int(sum(1 for a, b in zip(r.esss[1:], r.esss[2:]) if b > a)),
                alive_end=int(np.count_nonzero(np.asarray(r.Wns) > 0)), mc_nsims=m.nsims), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop starting new cases after this many seconds (0: run them all)")
    ap.add_argument("--user", action="store_true", help="the library runs the Normal simulator as USER-SUPPLIED source (run-time compiled: one thread per "
                    "row up to 16 parameters, in two launches or inside the sweep kernel; the cooperative form beyond), the oracle the built-in")
    ap.add_argument("--resume", action="store_true", help="also interrupt every run at a random generation and continue it from its checkpoint")
    ap.add_argument("--small", action="store_true", help="1 to 4 parameters only: the built-in simulators of the reference's test problems")
    ap.add_argument("--big", action="store_true", help="populations of 50,000 to 600,000 particles, rows up to 48 parameters, at most 8 generations")
    args = ap.parse_args()
    O.build()
    t0 = time.time()
    n_bad = 0
    for seed in range(args.first, args.first + args.cases):
        if args.seconds and time.time() - t0 > args.seconds:
            break
        c = random_case(seed, big=args.big, small=args.small)
        if args.big:
            c["max_iters"] = 8
            c["mc"]["generations"] = min(c["mc"]["generations"], 4)
        c["resume"] = args.resume
        if args.user:
            if c["sim"] == "mvn" and c["simulator"].blobs:
                c["simulator"] = A.MVNormal(tuple(c["simulator"].data()), sigma=c["simulator"].params()[0])
            c["user_sim"] = user_form_of(c["simulator"], c["d"], np.random.default_rng(seed))
            if c["user_sim"] is None:
                continue
        head = dict(seed=seed, d=c["d"], families=c["nfam"], sim=c["sim"], kernel=c["ABCk"].__name__, N=c["N"], q=c["q"],
                    blobs=bool(c["simulator"].blobs), **{k: v for k, v in c["smc"].items()}, mc=c["mc"],
                    **({"user_one_kernel": os.environ.get("ABZ_USER_ONE_KERNEL", "0")} if args.user else {}))
        t = time.time()
        try:
            info, bad = run_case(c)
        except Exception as e:                          # a refusal both sides share is not a difference; anything else is reported
            info, bad = dict(error=f"{type(e).__name__}: {e}"[:300]), [("exception",)]
        head.update(info, seconds=round(time.time() - t, 2), same=not bad)
        if bad:
            head["differences"] = [str(b)[:200] for b in bad]
            n_bad += 1
        print(json.dumps(head, ensure_ascii=False), flush=True)
        if bad:
            break
    print(json.dumps(dict(summary=True, first=args.first, ran=seed - args.first + (0 if n_bad == 0 and args.seconds and time.time() - t0 > args.seconds else 1),
                          different=n_bad, seconds=round(time.time() - t0, 1))), flush=True)
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
