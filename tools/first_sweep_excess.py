#!/usr/bin/env python3
"""Is the first sweep of a generation slower than its siblings, and does the excess follow the rows the partition moved?
(round-4 VERDICT 1b)   python tools/first_sweep_excess.py <kt_kernel_trace.csv> [sweep kernel substring]

Per generation of the trace: the part_swap launch before the sweeps (its duration is proportional to the rows it swapped) and the
generation's sweep launches; prints the table, the mean excess of sweep 1 over the mean of sweeps 2..k, and the least-squares line
excess = a + b * part_swap_us."""
import csv
import json
import sys

trace = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "smc_swarm_packed_kernel"
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace))))
gens, cur = [], None
for s, e, k in rows:
    if "part_swap_kernel" in k:
        cur = {"swap_us": (e - s) / 1e3, "sweeps": [], "gap_us": None, "swap_end": e}
        gens.append(cur)
    elif kern in k and cur is not None:
        if not cur["sweeps"]:
            cur["gap_us"] = (s - cur["swap_end"]) / 1e3
        cur["sweeps"].append((e - s) / 1e3)
full = [g for g in gens if len(g["sweeps"]) >= 3 and min(g["sweeps"][:3]) > 50.0]
xs = [g["swap_us"] for g in full]
ys = [g["sweeps"][0] - sum(g["sweeps"][1:3]) / 2 for g in full]
n = len(full)
mx, my = sum(xs) / n, sum(ys) / n
sxx = sum((x - mx) ** 2 for x in xs)
b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sxx if sxx else 0.0
a = my - b * mx
r = (sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / (sxx * sum((y - my) ** 2 for y in ys)) ** 0.5) if sxx and any(y != my for y in ys) else 0.0
print(json.dumps({"generations": n, "mean_sweep1_us": sum(g["sweeps"][0] for g in full) / n,
                  "mean_sweep2_us": sum(g["sweeps"][1] for g in full) / n, "mean_sweep3_us": sum(g["sweeps"][2] for g in full) / n,
                  "mean_excess_of_sweep1_us": my, "mean_part_swap_us": mx, "fit_excess_us": {"intercept": a, "per_us_of_part_swap": b, "r": r},
                  "per_generation": [{"part_swap_us": round(g["swap_us"], 1), "sweeps_us": [round(v, 1) for v in g["sweeps"][:3]]} for g in full]}, indent=1))
