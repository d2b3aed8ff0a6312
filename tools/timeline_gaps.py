#!/usr/bin/env python3
"""Where a generation's time goes: kernels vs idle gaps, from a rocprofv3 --kernel-trace csv.

    python tools/timeline_gaps.py <kt_kernel_trace.csv> [marker kernel, default qs_hist_kernel] [generation index from the end] [lanes]

With `lanes` (threads per particle of the sweep kernel) the sweep launches are listed with their alive count and rate.

Generations are cut at the marker kernel (the first launch of every quantile).  Prints the launches of one
steady-state generation (offset, duration, idle gap before it) and the averages over all complete generations
that hold no resampling."""
import csv
import sys


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "")[:44]


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "qs_hist_kernel"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows, sweeps = [], []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
            if "smc_swarm" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
                grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
                sweeps.append((grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if marker in r[2]]
    gens = [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    gens = [g for g in gens if not any("resample" in k[2] or "stratified" in k[2] for k in g)]
    if not gens:
        print("no complete generation found")
        return
    g = gens[-which] if len(gens) >= which else gens[-1]
    t0 = g[0][0]
    prev_end = t0
    print(f"{'kernel':46s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s}")
    for s, e, k in g:
        print(f"{k:46s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
        prev_end = max(prev_end, e)
    tot, busy, n = 0.0, 0.0, 0
    per = {}
    for a, b in zip(gens[:-1], gens[1:]):
        if b[0][0] - a[-1][1] > 5e6:
            continue
        tot += (b[0][0] - a[0][0]) / 1e3
        for s, e, k in a:
            busy += (e - s) / 1e3
            per[k] = per.get(k, 0.0) + (e - s) / 1e3
        n += 1
    if n:
        print(f"\n{n} generations: {tot / n:.1f} us each, kernels {busy / n:.1f} us, idle {(tot - busy) / n:.1f} us")
        for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
            print(f"  {k:46s} {v / n:8.1f} us")


    if sweeps and len(sys.argv) > 4 and int(sys.argv[4]) > 0:
        # sweep kernel: rate against the number of alive particles (threads / lanes per particle).  Only for kernels launched
        # with one thread group per particle (rounds 1-2); round 3's sweeps loop over tiles: pass lanes = 0
        lanes = int(sys.argv[4])
        print("\nsweep launches: alive particles, us, updates/s, alive rows MB")
        for grid, ns in sweeps:
            n = grid // lanes
            print(f"  {n:9d} {ns / 1e3:8.1f} {n / (ns * 1e-9):.3e} {n * 256 / 1e6:8.0f}")


if __name__ == "__main__":
    main()
