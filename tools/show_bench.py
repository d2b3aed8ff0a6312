#!/usr/bin/env python3
"""Prints the headline numbers of a bench.py JSON line (and of its other_configs):  python tools/show_bench.py FILE"""
import json
import sys


def show(name, d):
    r = d.get("roofline", {})
    pc = r.get("pattern_ceiling") or {}
    print("%-11s value %.4g %s  ms/step %.4f | kernel %s: %.4g upd/s, avg launch %.4f ms, %s frac %.3f" % (
        name, d["value"], d.get("unit", ""), d["ms_per_step"], r.get("kernel", "?")[:28], r.get("kernel_updates_per_s", 0),
        r.get("avg_launch_ms", 0), r.get("bound"), r.get("frac", 0)))
    if pc:
        print("            pattern ceiling %.4g (back to back) / %s (single launches): kernel over ceiling %.3f / %s" % (
            pc.get("updates_per_s", 0), "%.4g" % pc["updates_per_s_single_launches"] if pc.get("updates_per_s_single_launches") else "-",
            pc.get("kernel_over_ceiling", 0), "%.3f" % pc["kernel_over_ceiling_single_launches"] if pc.get("kernel_over_ceiling_single_launches") else "-"))
    if r.get("frac_trace") is not None:
        print("            frac_trace %.3f (kernel trace, committed file)  traffic / moved bytes %s" % (
            r["frac_trace"], "%.3f" % r["traffic_over_moved_bytes"] if r.get("traffic_over_moved_bytes") else "-"))
    if d.get("errors_vs_exact"):
        print("            errors vs closed forms:", json.dumps(d["errors_vs_exact"]))
    if d.get("cpu_baseline"):
        print("            cpu %.4g (%s cores) %s" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sample"][:80]))
    w = d.get("whole_run")
    if w:
        print("            whole run:", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in w.items() if not isinstance(v, dict)},
              {k: {kk: vv for kk, vv in v.items() if kk in ("generations", "seconds", "value", "logZ")} for k, v in w.items() if isinstance(v, dict) and "logZ" in v})


d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
show(d["config"]["name"], d)
for k, v in (d.get("other_configs") or {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"])
    else:
        show(k, v)
