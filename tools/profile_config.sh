# rocprofv3 kernel-trace summary + per-generation timeline of one bench configuration (run through gpurun):
#   TAG=r02 CFG=smc32 MARK=qs_hist_kernel LANES=4 bash tools/profile_config.sh
# writes gpurun_out/${TAG}_${CFG}_kernel_stats.csv, _timeline.txt, _prof_bench.log
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}; CFG=${CFG:-smc32}; MARK=${MARK:-qs_hist_kernel}; LANES=${LANES:-4}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$CFG -o kt -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs $EXTRA > $R/gpurun_out/${TAG}_${CFG}_prof_bench.log 2>&1
F=$(find $R/gpurun_out/prof_$CFG -name 'kt_kernel_trace.csv' | head -1)
python3 $R/tools/timeline_gaps.py $F $MARK 3 $LANES > $R/gpurun_out/${TAG}_${CFG}_timeline.txt 2>&1
cp $(find $R/gpurun_out/prof_$CFG -name 'kt_kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_${CFG}_kernel_stats.csv
python3 $R/tools/trace_window_stats.py $F $R/gpurun_out/${TAG}_${CFG}_prof_bench.log ${KERN:-smc_swarm_packed_kernel} > $R/gpurun_out/${TAG}_${CFG}_kernel_stats_timed_steps.csv 2>&1
python3 $R/tools/kernel_avg_check.py $F $R/gpurun_out/${TAG}_${CFG}_prof_bench.log ${KERN:-smc_swarm_packed_kernel} ${PER_STEP:-3} > $R/gpurun_out/${TAG}_${CFG}_kernel_avg_check.json 2>&1
rm -rf $R/gpurun_out/prof_$CFG
