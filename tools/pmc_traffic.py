#!/usr/bin/env python3
"""HBM traffic of a sweep kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as
MI355X_MICROARCH.md prescribes) -> profiles/<round>_hbm_traffic_<config>.json, the file bench.py's roofline.traffic reads.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <kernel substring> <lanes> <ld> [accept rate] [updates fetch pass] [updates write pass] [last-n dispatches]

`last-n dispatches` (round 4): count only the LAST n dispatches of the kernel -- the sweep launches of bench.py's timed steps
(`config.timed_window.sweep_launches`), with `updates` = `config.timed_window.updates` of that very pass and `accept rate` =
timed_window.naccs / timed_window.updates: traffic, acceptance and bytes moved then describe the same launches as the bench
line's timed window (the passes run the bench line's own --steps / --warmup).

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half
the bytes of 16-byte-per-lane coalesced reads, so it is doubled.  Particle-updates per dispatch = grid size / lanes
per particle -- or, for kernels whose workgroups loop over tiles (round 3: the grid is what is resident, not the work),
the particle-updates bench.py reports for that very run (`config.launched_incl_warmup.updates`), passed as the last two
arguments.  Infinity-Cache hits are counted by FETCH_SIZE (it counts the L2's fabric requests)."""
import csv
import json
import sys


def collect(path, counter, kernel, last_n=0):
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                rows.append((int(row.get("Dispatch_Id") or len(rows)), float(row["Counter_Value"]), int(row["Grid_Size"])))
    rows.sort()
    if last_n > 0:
        rows = rows[-last_n:]
    return sum(r[1] for r in rows), sum(r[2] for r in rows), len(rows)


def main():
    fetch_csv, write_csv, out, kernel = sys.argv[1:5]
    lanes, ld = int(sys.argv[5]), int(sys.argv[6])
    acc = float(sys.argv[7]) if len(sys.argv) > 7 else None
    upd_f = float(sys.argv[8]) if len(sys.argv) > 8 else 0.0
    upd_w = float(sys.argv[9]) if len(sys.argv) > 9 else 0.0
    last_n = int(sys.argv[10]) if len(sys.argv) > 10 else 0
    f, gu, nf = collect(fetch_csv, "FETCH_SIZE", kernel, last_n)
    w, gw, nw = collect(write_csv, "WRITE_SIZE", kernel, last_n)
    rd = 2.0 * f * 1024.0 / (upd_f if upd_f > 0 else gu / lanes)
    wr = w * 1024.0 / (upd_w if upd_w > 0 else gw / lanes)
    b_read, b_write = 24 * ld + 17, 8 * ld + 16
    res = {
        "kernel": kernel, "lanes": lanes, "ld": ld,
        "workload": "bench.py of this configuration, one rocprofv3 --pmc pass per counter",
        "method": "FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: on gfx950 it reports "
                  "half the bytes of 16-B-per-lane coalesced reads). Infinity-Cache hits are counted.",
        "dispatches": nf, "dispatches_write_pass": nw,
        "window": "the sweep dispatches of bench.py's timed steps only (warm-up excluded)" if last_n > 0 else "every dispatch of the run, warm-up included",
        "updates_are": "bench.py's count of the profiled run" if upd_f > 0 else "grid size / lanes",
        "read_bytes_per_update": rd, "write_bytes_per_update": wr,
        "algorithmic_read_bytes_per_update": b_read, "algorithmic_write_bytes_per_update": b_write,
        "total_bytes_per_update": rd + wr, "ratio_to_algorithmic_total": (rd + wr) / (b_read + b_write),
    }
    if acc is not None:
        moved = b_read + acc * b_write
        res.update(acceptance_rate=acc, moved_bytes_per_update=moved, ratio_to_moved_bytes=(rd + wr) / moved)
    with open(out, "w") as fo:
        json.dump(res, fo, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
