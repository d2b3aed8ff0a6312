#!/usr/bin/env python3
"""One window, two clocks.  bench.py's HIP-event average of the sweep kernel against the rocprofv3 kernel trace of the SAME
run, both over the TIMED steps only (bench.py --no-whole-run --no-other-configs --no-pattern, so the sweep dispatches of the
timed steps are the LAST `config.timed_window.sweep_launches` dispatches of the kernel in the trace; warm-up launches excluded).

    python tools/kernel_avg_check.py <kt_kernel_trace.csv> <bench log with the JSON line> <kernel substring> [sweeps per step]

A group of sweeps enqueues Kmcmc launches per generation; a launch behind a held test of smc:352 returns at once (a few
microseconds) and did no work: `timed_window.sweeps` says how many ran, the shortest surplus launches are the gated ones.
bench.py brackets ONE sweep of every n-th generation (`roofline.timed_every_nth_step`), the 1st, 2nd, 3rd ... of the
generation's sweeps in turn (abcdez_ctx_set_timing mode 2); `sweeps per step` tells which dispatches of the trace those were.

Output fields bench.py reads for `roofline.frac_trace`: trace_avg_ms_timed_sweeps, updates_per_launch_timed_sweeps, timed_sweeps."""
import csv
import json
import sys

trace, log, kernel = sys.argv[1:4]
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 1
line = [l for l in open(log) if l.startswith("{")][-1]
doc = json.loads(line)
roof = doc["roofline"]
win = doc["config"]["timed_window"]
stride = int(roof.get("timed_every_nth_step", 1))
steps = int(doc["steps"])
# "a+b": a sweep made of two launches (Lotka-Volterra: smc_lv_phase1_kernel + smc_lv_phase2_kernel) -- the k-th launch of each, added
with open(trace) as f:
    all_rows = list(csv.DictReader(f))
parts = []
for sub in kernel.split("+"):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in all_rows if sub in r["Kernel_Name"]))
    parts.append([(e - s) / 1e6 for s, e in rows])
assert len({len(p) for p in parts}) == 1, [len(p) for p in parts]
durs = [sum(v) for v in zip(*parts)]
n = int(roof["launches"])
n_launch = int(win.get("sweep_launches", steps * per_step))
ran = int(win["sweeps"])
window = durs[-n_launch:]               # every sweep dispatch of the timed steps
worked = sorted(window)[len(window) - ran:] if ran <= len(window) else window      # drop the gated launches (they returned at once)
# the bracketed launches.  Timing mode 3 (round 5: one pair around ALL the sweeps of every stride-th step): every sweep of those steps;
# mode 2 (rounds 2-4): sweep (s mod per_step) of every stride-th step s
if int(roof.get("timing_mode", 2)) == 3:
    timed = [window[st * per_step + k] for st in range(0, steps, stride) for k in range(per_step) if st * per_step + k < len(window)][:n]
else:
    timed = [window[st * per_step + (st % per_step)] for st in range(0, steps, stride) if st * per_step + (st % per_step) < len(window)][:n]
upl = win["updates"] / max(ran, 1)
avg = sum(worked) / len(worked)
# the bytes `roofline.frac` counts: SURVEY 8d's algorithmic reads, or -- rows of one or two doubles -- the fabric line fills
b_read = roof.get("line_fill_bytes_per_update") or roof.get("bytes_read_per_update")
out = {
    "kernel": kernel, "same_run": True, "command_steps_warmup": [steps, int(doc["warmup"])],
    "bench_hip_events_avg_launch_ms": roof["avg_launch_ms"], "bench_launches": n, "timed_every_nth_step": stride,
    "bench_frac": roof["frac"],
    "trace_launches_total": len(durs), "trace_launches_in_timed_steps": len(window), "timed_sweeps": ran,
    "gated_launches_dropped": len(window) - len(worked),
    "trace_avg_ms_timed_sweeps": avg,
    "updates_per_launch_timed_sweeps": upl,
    "kernel_updates_per_s_trace": upl / (avg * 1e-3),
    "rocprofv3_trace_avg_ms_over_the_bracketed_launches": (sum(timed) / len(timed)) if timed else None,
    "rocprofv3_trace_avg_ms_over_all_launches_incl_warmup": sum(durs) / len(durs),
    "ratio_events_over_trace_same_launches": (roof["avg_launch_ms"] / (sum(timed) / len(timed))) if timed else None,
}
if b_read:
    out["frac_trace"] = b_read * upl / (avg * 1e-3) / 1e9 / roof["peak"]
    out["frac_trace_is"] = f"{b_read} B x updates per launch / trace average / {roof['peak']} GB/s over the sweeps of the timed steps"
print(json.dumps(out, indent=1))
