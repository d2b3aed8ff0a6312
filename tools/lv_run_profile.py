#!/usr/bin/env python3
"""The whole Lotka-Volterra run (BASELINE configs[3]: N = 2^20 to eps = 1), generation by generation: what the sweeps of the LATE
generations cost -- the population has contracted, nearly every proposal is in the prior's support and reaches the simulator, and
most of them leave after the first observations (early exit).

    python tools/lv_run_profile.py                      wall time per run + the history the driver keeps
    rocprofv3 --kernel-trace -d DIR -o lv -- python3 tools/lv_run_profile.py
    python tools/lv_run_profile.py --trace DIR/lv_kernel_trace.csv      per-kernel time by tenth of the run
"""
import csv
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if "--trace" in sys.argv:
    rows = list(csv.DictReader(open(sys.argv[sys.argv.index("--trace") + 1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
    sweeps = [r for r in rows if "smc_lv_phase2" in r["Kernel_Name"] or "smc_swarm_packed_kernel" in r["Kernel_Name"]]
    # the LAST run of the process is the measured one: cut at the last init kernel
    inits = [k for k, r in enumerate(rows) if "init_kernel" in r["Kernel_Name"]]
    rows = rows[inits[-1]:]
    t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    out = []
    for b in range(10):
        lo, hi = t0 + (t1 - t0) * b // 10, t0 + (t1 - t0) * (b + 1) // 10
        acc = {}
        for r in rows:
            s = int(r["Start_Timestamp"])
            if lo <= s < hi:
                k = name(r)
                a = acc.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += (int(r["End_Timestamp"]) - s) / 1e3
        top = sorted(acc.items(), key=lambda kv: -kv[1][1])[:5]
        out.append({"tenth": b, "kernels": {k: {"launches": v[0], "avg_us": round(v[1] / v[0], 1), "total_ms": round(v[1] / 1e3, 2)} for k, v in top}})
    print(json.dumps({"run_ms": (t1 - t0) / 1e6, "by_tenth_of_the_run": out}, indent=1))
    sys.exit(0)

import abcdez_amd as A

g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lv_data.json")))
lv = A.LotkaVolterraRK4(tuple(g["obs"]), x0=g["x0"], y0=g["y0"], dt=g["dt"], steps_per_obs=g["steps_per_obs"], noise=g["noise"])
prior = A.Factored(*[A.Uniform(0.0, 2.0)] * 4)
for rep in range(2):
    t = time.perf_counter()
    r = A.abcdesmc(prior, lv, 1.0, None, nparticles=1 << 20, verbose=False, rng=1, nsims_max=10 ** 12)
    dt = time.perf_counter() - t
n = len(r.ϵs)
pick = sorted(set([0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 250, 300, 350, n - 1]) & set(range(n)))
print(json.dumps({"seconds": dt, "generations": r.iters, "nsims": r.nsims, "logZ": r.logZ,
                  "history": [{"generation": k, "eps": r.ϵs[k], "ess": r.esss[k], "facc": r.faccs[k]} for k in pick]}, indent=1))
