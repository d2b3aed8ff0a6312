#!/usr/bin/env python3
"""Sums the SQ counters of one rocprofv3 --pmc pass over the dispatches of a kernel (name substring) and prints the
issue / stall breakdown MI355X_MICROARCH.md describes: WAIT_ANY (waves parked on s_waitcnt / barriers) +
WAIT_INST_ANY (issue stalls) + ACTIVE_INST_ANY ~ WAVE_CYCLES.

    python tools/sq_summary.py <counter_collection.csv> <kernel substring>"""
import csv
import json
import sys
from collections import defaultdict

path, kernel = sys.argv[1:3]
updates = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0     # particle-updates those dispatches processed (persistent grids: not the grid size)
tot, disp, threads = defaultdict(float), set(), 0
with open(path) as f:
    for row in csv.DictReader(f):
        if kernel in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Dispatch_Id"] not in disp:
                disp.add(row["Dispatch_Id"])
                threads += int(row["Grid_Size"])
wc = tot.get("SQ_WAVE_CYCLES", 0.0) or 1.0
out = {"kernel": kernel, "dispatches": len(disp), "threads": threads, "counters": dict(tot),
       "share_of_wave_cycles": {k: tot[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in tot},
       "valu_instructions_per_thread": tot.get("SQ_INSTS_VALU", 0.0) * 64.0 / max(threads, 1),
       "updates": updates,
       "per_update": {k: v / updates for k, v in tot.items()} if updates else None,
       "waves_resident_on_average": wc / (tot.get("SQ_BUSY_CYCLES", 0.0) or 1.0),
       "note": "SQ_INSTS_VALU counts wave-level instructions; x64 / threads = instructions per thread"}
print(json.dumps(out, indent=1))
