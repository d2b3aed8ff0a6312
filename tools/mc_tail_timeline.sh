# kernel trace of the LAST generations of the mc1d bench line (the ones that draw by rejection): one line per launch
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_mct -o kt -- python3 $R/bench.py --config mc1d --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs $EXTRA > $R/gpurun_out/mct_bench.log 2>&1
F=$(find $R/gpurun_out/prof_mct -name 'kt_kernel_trace.csv' | head -1)
python3 - "$F" <<'PY' > $R/gpurun_out/mct_timeline.txt
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in csv.DictReader(open(sys.argv[1]))))
sw = [(s, e) for s, e, k in rows if "mc_swarm" in k]
print("generation: period us (start to start), sweep us")
for g in range(1, len(sw)):
    print(f"  gen {g:3d}  period {(sw[g][0] - sw[g - 1][0]) / 1e3:7.1f}  sweep {(sw[g - 1][1] - sw[g - 1][0]) / 1e3:6.1f}")
rows = rows[-24:]
t0 = rows[0][0]; prev = None
for s, e, k in rows:
    print(f"{k:42s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {((s - prev) / 1e3 if prev else 0.0):6.1f} us")
    prev = e
PY
rm -rf $R/gpurun_out/prof_mct
