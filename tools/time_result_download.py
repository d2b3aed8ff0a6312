#!/usr/bin/env python3
"""PCIe-inclusive figure for DESIGN.md section 6: a complete abcdesmc run of the bench workload (d = 32 MVN, 2^22
particles, eps_target 6.0) including the one bulk host transfer of the path -- the final result download
(P, theta, logpi, C, Wns, alive) -- next to the same run's device-resident rate."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import abcdez_amd as A

d, N = 32, 1 << 22
prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
sim = A.MVNormal(tuple([1.0] * d))
A.abcdesmc(prior, sim, 9.0, None, nparticles=N, verbose=False, rng=2, nsims_max=10 ** 12)      # warm-up (allocator, JIT-free)
torch.cuda.synchronize()
t0 = time.perf_counter()
r = A.abcdesmc(prior, sim, 6.0, None, nparticles=N, verbose=False, rng=1, nsims_max=10 ** 12)
torch.cuda.synchronize()
t1 = time.perf_counter()
# abcdesmc() already includes engine.result(); time the download alone once more
t2 = time.perf_counter()
res = r.engine.result()
t3 = time.perf_counter()
nbytes = sum(v.nbytes for v in res.values() if v is not None) - (res["P"].nbytes if res["P"] is res["theta"] or res["P"].base is res["theta"] else 0)   # P aliases theta when no dimension is discrete
print(json.dumps({"generations": r.iters, "updates": r.updates, "run_s_incl_download": t1 - t0, "download_s": t3 - t2,
                  "download_bytes": nbytes, "download_GBps": nbytes / (t3 - t2) / 1e9,
                  "updates_per_s_incl_download": r.updates / (t1 - t0),
                  "updates_per_s_device_resident": r.updates / (t1 - t0 - (t3 - t2)), "logZ": r.logZ}))
