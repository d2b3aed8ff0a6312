#!/usr/bin/env python3
"""Do the two clocks agree?  bench.py's HIP-event average of the sweep kernel against the rocprofv3 kernel trace of the SAME
run: the timed launches are the last `launches` dispatches of the kernel in the trace (bench.py --no-whole-run).

    python tools/kernel_avg_check.py <kt_kernel_trace.csv> <bench log with the JSON line> <kernel substring> [sweeps per step]

bench.py brackets the first sweep of every generation; `sweeps per step` (3 for the abcdesmc configurations) tells which
dispatches of the trace those were."""
import csv
import json
import sys

trace, log, kernel = sys.argv[1:4]
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 1
line = [l for l in open(log) if l.startswith("{")][-1]
roof = json.loads(line)["roofline"]
durs = []
with open(trace) as f:
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(f) if kernel in r["Kernel_Name"]))
durs = [(e - s) / 1e6 for s, e in rows]
n = int(roof["launches"])
window = durs[-n * per_step:]
timed = window[::per_step]              # the bracketed launches: the first sweep of every step
print(json.dumps({
    "kernel": kernel, "same_run": True,
    "bench_hip_events_avg_launch_ms": roof["avg_launch_ms"], "bench_launches": n,
    "rocprofv3_trace_avg_ms_over_the_same_launches": sum(timed) / len(timed),
    "rocprofv3_trace_avg_ms_over_all_sweeps_of_the_timed_steps": sum(window) / len(window),
    "rocprofv3_trace_avg_ms_over_all_launches_incl_warmup": sum(durs) / len(durs), "trace_launches": len(durs),
    "ratio_events_over_trace": roof["avg_launch_ms"] / (sum(timed) / len(timed)),
}, indent=1))
