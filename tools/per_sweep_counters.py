#!/usr/bin/env python3
"""Counters of a rocprofv3 --pmc pass BY POSITION OF THE SWEEP IN ITS GENERATION (round-4 VERDICT 1b: why is the first sweep slower?):
    python tools/per_sweep_counters.py <counter_collection.csv> [kernel substring] [sweeps per generation]
Dispatches of the sweep kernel in dispatch order, k-th of every group of `per` consecutive ones; per counter: mean per dispatch."""
import csv
import json
import sys
from collections import defaultdict

path = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "smc_swarm_packed_kernel"
per = int(sys.argv[3]) if len(sys.argv) > 3 else 3
disp = defaultdict(dict)
for r in csv.DictReader(open(path)):
    if kernel in r["Kernel_Name"]:
        disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = disp[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
ids = ids[len(ids) % per:]                      # whole generations, counted from the end
out = {}
for k in range(per):
    rows = [disp[i] for i in ids[k::per]]
    out[f"sweep {k + 1}"] = {c: sum(r.get(c, 0.0) for r in rows) / len(rows) for c in sorted(rows[0])}
    out[f"sweep {k + 1}"]["dispatches"] = len(rows)
print(json.dumps(out, indent=1))
