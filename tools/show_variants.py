#!/usr/bin/env python3
"""Prints the JSON lines of tools/sweep_variants as a table:  python tools/show_variants.py FILE.jsonl"""
import json
import sys

for line in open(sys.argv[1]):
    d = json.loads(line)
    print("%-9s grid %-5s median %.4f ms  min %.4f ms  %.4g upd/s  read frac %.3f  %s" % (
        d["variant"], d.get("grid", "-"), d["median_ms"], d["min_ms"], d["updates_per_s_median"], d["read_frac_median"], d["what"][:90]))
