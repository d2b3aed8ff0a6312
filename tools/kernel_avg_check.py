#!/usr/bin/env python3
"""Do the two clocks agree?  bench.py's HIP-event average of the sweep kernel against the rocprofv3 kernel trace of the SAME
run: the timed launches are the last `launches` dispatches of the kernel in the trace (bench.py --no-whole-run).

    python tools/kernel_avg_check.py <kt_kernel_trace.csv> <bench log with the JSON line> <kernel substring> [sweeps per step]

bench.py brackets ONE sweep of every n-th generation (`roofline.timed_every_nth_step`), the 1st, 2nd, 3rd ... of the generation's
sweeps in turn (abcdez_ctx_set_timing mode 2); `sweeps per step` (3 for the abcdesmc configurations) tells which dispatches of
the trace those were."""
import csv
import json
import sys

trace, log, kernel = sys.argv[1:4]
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 1
line = [l for l in open(log) if l.startswith("{")][-1]
doc = json.loads(line)
roof = doc["roofline"]
stride = int(roof.get("timed_every_nth_step", 1))
steps = int(doc["steps"])
durs = []
with open(trace) as f:
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(f) if kernel in r["Kernel_Name"]))
durs = [(e - s) / 1e6 for s, e in rows]
n = int(roof["launches"])
window = durs[-steps * per_step:]       # every sweep of the timed steps
# the bracketed launches: sweep (s mod per_step) of every stride-th step s
timed = [window[st * per_step + (st % per_step)] for st in range(0, steps, stride)][:n]
print(json.dumps({
    "kernel": kernel, "same_run": True,
    "bench_hip_events_avg_launch_ms": roof["avg_launch_ms"], "bench_launches": n, "timed_every_nth_step": stride,
    "rocprofv3_trace_avg_ms_over_the_same_launches": sum(timed) / len(timed),
    "rocprofv3_trace_avg_ms_over_all_sweeps_of_the_timed_steps": sum(window) / len(window),
    "rocprofv3_trace_avg_ms_over_all_launches_incl_warmup": sum(durs) / len(durs), "trace_launches": len(durs),
    "ratio_events_over_trace": roof["avg_launch_ms"] / (sum(timed) / len(timed)),
}, indent=1))
