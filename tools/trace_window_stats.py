#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED steps of a bench.py run from its rocprofv3 kernel trace -- the same columns as rocprofv3's
own --stats summary (which covers every launch of the process, warm-up included), restricted to the window the bench line
describes.

    python tools/trace_window_stats.py <kt_kernel_trace.csv> <bench log with the JSON line> <sweep kernel substring> > stats.csv

The window starts at the first dispatch after the last warm-up sweep: the sweep dispatches of the timed steps are the last
`config.timed_window.sweep_launches` dispatches of the sweep kernel (bench.py --no-whole-run --no-other-configs --no-pattern).
`roofline.frac_trace` = bytes_read_per_update x (timed_window.updates / timed_window.sweeps) / AverageNs of the sweep kernel's
row here (rows of launches that returned at once behind a held test of smc:352 are listed separately as `... [gated]`)."""
import csv
import json
import sys

trace, log, kernel = sys.argv[1:4]
kernel = kernel.split("+")[-1]        # "a+b": a sweep of two launches (Lotka-Volterra) -- the window is found by the second one
doc = json.loads([l for l in open(log) if l.startswith("{")][-1])
win = doc["config"]["timed_window"]
n_launch, ran = int(win["sweep_launches"]), int(win["sweeps"])
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sweeps = [i for i, r in enumerate(rows) if kernel in r[2]]
first = sweeps[-n_launch]
# the window's first kernel: everything after the previous sweep dispatch (the prologue of the first timed step)
start = sweeps[-n_launch - 1] + 1 if len(sweeps) > n_launch else 0
window = rows[start:]
durs = sorted((e - s) for s, e, k in window if kernel in k)
gated = set()
if ran < len(durs):
    cut = durs[len(durs) - ran]          # the `ran` longest did the work
    n_g = len(durs) - ran
    for i, (s, e, k) in enumerate(window):
        if kernel in k and (e - s) < cut and len(gated) < n_g:
            gated.add(i)
stats = {}
for i, (s, e, k) in enumerate(window):
    name = k + (" [gated: returned at once, smc:352 held]" if i in gated else "")
    st = stats.setdefault(name, [0, 0, None, None])
    d = e - s
    st[0] += 1; st[1] += d
    st[2] = d if st[2] is None else min(st[2], d)
    st[3] = d if st[3] is None else max(st[3], d)
tot = sum(v[1] for v in stats.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, (c, t, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    w.writerow([name, c, t, f"{t / c:.1f}", f"{100.0 * t / tot:.2f}", mn, mx])
