# Round 5, GPU call 18: where the whole Lotka-Volterra run spends its time, by tenth of the run (kernel trace of one run)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tools/lv_run_profile.py > $O/r05_lv_run_history.json 2> $O/r05_lv_run_history.err || { tail -20 $O/r05_lv_run_history.err; exit 1; }
rm -rf $O/lvtrace && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/lvtrace -o lv -- python3 $R/tools/lv_run_profile.py > /dev/null 2> $O/r05_lv_trace.err
F=$(find $O/lvtrace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/lv_run_profile.py --trace $F > $O/r05_lv_run_by_tenth.json
rm -rf $O/lvtrace
cat $O/r05_lv_run_history.json | head -80
cat $O/r05_lv_run_by_tenth.json
