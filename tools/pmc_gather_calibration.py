#!/usr/bin/env python3
"""Calibration of the fabric-read counters on a request stream of known shape (tools/gather_d1.hip, kernel v0<double>: per
position one coalesced 8-byte own read, TWO random 8-byte reads of a 50 MB table -- 127.99 distinct 64-byte lines per wave of
128 requests --, one coalesced 8-byte write):

    python tools/pmc_gather_calibration.py <sizes.csv> <fetch.csv> <l2.csv>

prints per position: fabric read requests by size, the bytes they carry (sum of size x count -- exact by construction),
FETCH_SIZE as rocprofv3 reports it, and therefore the factor FETCH_SIZE needs FOR THIS ACCESS SHAPE (the hardware guide gives
x2 for 16-byte-per-lane coalesced streams only)."""
import csv
import json
import sys
from collections import defaultdict


def read(path, kernel="v0<double>"):
    tot, disp = defaultdict(float), set()
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                disp.add(row["Dispatch_Id"])
    return tot, len(disp)


sizes, n1 = read(sys.argv[1])
fetch, n2 = read(sys.argv[2])
l2, n3 = read(sys.argv[3])
M = 3 * (1 << 23) // 4                      # positions per launch (gather_d1.hip)
per = lambda tot, n, k: tot.get(k, 0.0) / (n * M) if n else None
r32, r64, r128 = (per(sizes, n1, f"TCC_EA0_RDREQ_{s}B_sum") for s in (32, 64, 128))
rall = per(sizes, n1, "TCC_EA0_RDREQ_sum")
other = (rall or 0.0) - (r32 or 0.0) - (r64 or 0.0) - (r128 or 0.0)      # requests of the remaining size class (what is left is 64-byte)
exact = 32 * (r32 or 0) + 64 * (r64 or 0) + 128 * (r128 or 0)
fetch_b = per(fetch, n2, "FETCH_SIZE") * 1024.0 if n2 else None
out = {
    "kernel": "tools/gather_d1.hip v0<double>", "positions_per_launch": M, "dispatches": [n1, n2, n3],
    "request_stream": "per position: 8 B own (coalesced), 2 x 8 B random from a 50 MB table (one 64-byte line each), 8 B written",
    "fabric_read_requests_per_position": {"32B": r32, "64B": r64, "128B": r128, "all": rall, "unclassified": other},
    "fabric_read_bytes_per_position_from_request_sizes": exact,
    "FETCH_SIZE_bytes_per_position_as_reported": fetch_b,
    "factor_FETCH_SIZE_needs_for_this_shape": (exact / fetch_b) if fetch_b else None,
    "l2_requests_per_position": {k: per(l2, n3, k) for k in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum")},
    "algorithmic_read_bytes_per_position": 24,
}
print(json.dumps(out, indent=1))
