#!/usr/bin/env python3
"""Soak: repeated full-size runs in ONE process for `minutes` (default 5) -- determinism across repeats (every repeat of a
configuration must give bit-identical logZ / posterior sums), the four BASELINE configurations in rotation, abcdemc with
generations in flight (stream launches and graph replay), checkpoint / resume in the middle, device memory before / after.
    python tools/soak.py [minutes]"""
import gc
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import abcdez_amd as A

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from user_sources import USER_MVN_LANES

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lv_data.json")))
lv = A.LotkaVolterraRK4(tuple(g["obs"]), x0=g["x0"], y0=g["y0"], dt=g["dt"], steps_per_obs=g["steps_per_obs"], noise=g["noise"])
cases = {
    "smc32 2^22": lambda: A.abcdesmc(A.Factored(*[A.Normal(0, 1)] * 32), A.MVNormal((1.0,) * 32), 6.0, None, nparticles=1 << 22,
                                     verbose=False, rng=5, nsims_max=10 ** 12),
    "evidence1d 2^23": lambda: A.abcdesmc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, None, nparticles=1 << 23, verbose=False,
                                          rng=7, nsims_max=10 ** 12),
    "lv 2^18": lambda: A.abcdesmc(A.Factored(*[A.Uniform(0.0, 2.0)] * 4), lv, 1.0, None, nparticles=1 << 18, verbose=False, rng=9,
                                  nsims_max=10 ** 12),
    "further families 2^20": lambda: A.abcdesmc(A.Factored(A.Exponential(1.5), A.Gamma(2.5, 0.6), A.LogNormal(0.0, 0.5), A.Cauchy(1.0, 0.5),
                                                           A.Poisson(2.0), A.Weibull(1.8, 1.2), A.TDist(4.0),
                                                           A.truncated(A.Normal(1.0, 2.0), 0.0, 4.0)),
                                                A.MVNormal((1.0,) * 8), 2.5, None, nparticles=1 << 20, verbose=False, rng=15,
                                                nsims_max=10 ** 12),
    # round 6: kernels compiled at run time -- wrapper priors under a built-in simulator, a user simulator in the cooperative form
    "wrapped priors 2^20": lambda: A.abcdesmc(A.Factored(A.truncated(A.Gamma(2.0, 1.0), 0.3, 5.0), A.MixtureModel([A.Normal(-1.0, 0.5), A.Laplace(1.0, 1.5)], [0.4, 0.6]),
                                                         A.Affine(A.TDist(4.0), 0.5, 1.5), A.truncated(A.Poisson(4.0), 1, 9)),
                                              A.MVNormal((1.0, 0.5, 0.8, 3.0)), 1.0, None, nparticles=1 << 20, verbose=False, rng=17, nsims_max=10 ** 12),
    "user lanes d=32 2^20": lambda: A.abcdesmc(A.Factored(*[A.Normal(0, 1)] * 32), A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=(1.0,) * 32), 6.5, None,
                                               nparticles=1 << 20, verbose=False, rng=19, nsims_max=10 ** 12),
    "epa 2^20": lambda: A.abcdesmc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, None, nparticles=1 << 20, verbose=False, rng=11,
                                   ABCk=A.Epa0toϵ, nsims_max=10 ** 12),
}


def fingerprint(r):
    return (r.logZ, r.iters, r.nsims, float(np.sum(r.C)), int((r.Wns > 0).sum()))


free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
ref, counts, last = {}, {k: 0 for k in cases}, time.time()
mc_ref = {}
rounds = 0
while time.time() - t0 < minutes * 60:
    for name, run in cases.items():
        r = run()
        fp = fingerprint(r)
        assert ref.setdefault(name, fp) == fp, (name, fp, ref[name])
        counts[name] += 1
        del r
    for graphs in (False, True):
        os.environ["ABZ_GRAPHS"] = "1" if graphs else "0"
        m = A.abcdemc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, None, nparticles=1 << 20, generations=300, verbose=False, rng=3)
        fp = (m.nsims, float(np.sum(m.C)), float(np.sum(m.P)))
        assert mc_ref.setdefault("mc", fp) == fp, (graphs, fp, mc_ref["mc"])      # graph replay == stream launches, every time
        del m
    os.environ["ABZ_GRAPHS"] = "0"
    # checkpoint / resume in the middle of a run equals the uninterrupted run
    kw = dict(nparticles=1 << 20, verbose=False, rng=13, nsims_max=10 ** 12)
    full = A.abcdesmc(A.Factored(*[A.Normal(0, 1)] * 8), A.MVNormal((1.0,) * 8), 2.5, None, **kw)
    part = A.abcdesmc(A.Factored(*[A.Normal(0, 1)] * 8), A.MVNormal((1.0,) * 8), 2.5, None, max_iters=9, **kw)
    rest = A.abcdesmc(A.Factored(*[A.Normal(0, 1)] * 8), A.MVNormal((1.0,) * 8), 2.5, None, resume=part.checkpoint(), **kw)
    assert fingerprint(rest) == fingerprint(full)
    del full, part, rest
    rounds += 1
    gc.collect()
    if time.time() - last > 30:
        print(f"[{time.time() - t0:6.1f} s] rounds {rounds} repeats {counts}", flush=True)
        last = time.time()
gc.collect()
torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print(json.dumps({"seconds": round(time.time() - t0, 1), "rounds": rounds, "repeats": counts,
                  "every_repeat_bit_identical": True, "abcdemc_graph_replay_equals_stream_launches": True,
                  "resume_equals_uninterrupted": True,
                  "free_GiB_before_after": [round(free0 / 2 ** 30, 2), round(free1 / 2 ** 30, 2)],
                  "fingerprints": {k: [v[0], v[1], v[2]] for k, v in ref.items()}}))
