#!/usr/bin/env python3
"""Mean of every counter of a rocprofv3 --pmc pass per KERNEL NAME over the last `n` generations of a bench.py run (dispatch order kept):
    python tools/per_kernel_counters.py <counter_collection.csv> [n = 8]
One line per kernel of a generation in launch order: name, dispatches averaged, counters."""
import csv
import json
import sys
from collections import defaultdict

path = sys.argv[1]
ngen = int(sys.argv[2]) if len(sys.argv) > 2 else 8
disp = {}
for r in csv.DictReader(open(path)):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "grid": int(r["Grid_Size"]), "c": defaultdict(float)})
    d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(disp)
# generations: cut at qs_hist_kernel (the first kernel of the fused prologue's select when it runs inline) or ind_reweight
marks = [k for k, i in enumerate(ids) if "ind_reweight_kernel" in disp[i]["name"]]
marks = marks[-ngen - 1:]
seqs = [ids[a:b] for a, b in zip(marks[:-1], marks[1:])]
ref = [disp[i]["name"] for i in seqs[-1]]
same = [s for s in seqs if [disp[i]["name"] for i in s] == ref]
out = []
for pos, name in enumerate(ref):
    acc = defaultdict(float)
    for s in same:
        for c, v in disp[s[pos]]["c"].items():
            acc[c] += v / len(same)
    out.append({"kernel": name[:44], "grid": disp[same[-1][pos]]["grid"], **{c: round(v, 1) for c, v in acc.items()}})
print(json.dumps({"generations_averaged": len(same), "kernels_in_launch_order": out}, indent=1))
