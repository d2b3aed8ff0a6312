#!/usr/bin/env python3
"""Repeated runs on one process: determinism across repeats, a long abcdemc run with generations in flight, device memory before / after."""
import math, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import abcdez_amd as A
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
ref = None
for it in range(40):
    r = A.abcdesmc(A.Factored(*[A.Normal(0, 1)] * 32), A.MVNormal((1.0,) * 32), 6.0, None, nparticles=1 << 18, verbose=False, rng=5, nsims_max=10**12)
    if ref is None: ref = r.logZ
    assert r.logZ == ref, (it, r.logZ, ref)
    del r
for it in range(40):
    r = A.abcdesmc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, None, nparticles=1 << 20, verbose=False, rng=7, nsims_max=10**12)
    del r
m = A.abcdemc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, None, nparticles=1 << 18, generations=4000, verbose=False, rng=3)
print("mc reached", m.reached_eps, "nsims", m.nsims)
del m
import gc; gc.collect(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("seconds", round(time.time() - t0, 1), "free before/after GiB", round(free0 / 2**30, 2), round(free1 / 2**30, 2), "logZ", ref)
