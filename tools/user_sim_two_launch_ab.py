#!/usr/bin/env python3
"""A user-supplied simulator (hiprtc) that restates Lotka-Volterra (BASELINE configs[3]'s workload, no early exit: a user distance is
opaque) at N = 2^20: the whole run to eps = 3 with the sweep as two launches (default) and as the one-kernel two-phase body
(ABZ_USER_ONE_KERNEL=1, read when the context is created), same process, alternating.  Results must be identical."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import abcdez_amd as A
from test_gpu_fullsize import USER_LV

g = json.load(open(os.path.join(ROOT, "tests", "golden", "lv_data.json")))
prior = A.Factored(*[A.Uniform(0.0, 2.0)] * 4)
user = A.UserSimulator(USER_LV, params=(g["x0"], g["y0"], g["dt"], float(g["steps_per_obs"]), g["noise"]), data=tuple(g["obs"]))
out, ref = [], None
for rep in range(3):
    for one_kernel in ("0", "1"):
        os.environ["ABZ_USER_ONE_KERNEL"] = one_kernel
        t = time.perf_counter()
        r = A.abcdesmc(prior, user, 3.0, None, nparticles=1 << 20, verbose=False, rng=1, nsims_max=10 ** 12)
        dt = time.perf_counter() - t
        fp = (r.logZ, r.iters, r.nsims, float(np.sum(r.P)))
        ref = ref or fp
        assert fp == ref, (fp, ref)
        out.append({"rep": rep, "sweep": "one kernel, two phases" if one_kernel == "1" else "two launches", "seconds": dt, "generations": r.iters,
                    "nsims": r.nsims, "logZ": r.logZ})
        print(json.dumps(out[-1]), flush=True)
