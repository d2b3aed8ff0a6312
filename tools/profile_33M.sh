# side passes at 8 ranks' worth of population (33.6 M particles on one GPU): kernel stats + per-generation timeline
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof33 -o kt -- python3 $R/bench.py --no-cpu-baseline --particles-per-gpu 33554432 --steps 10 --warmup 3 > $R/gpurun_out/prof33_bench.log 2>&1
F=$(find $R/gpurun_out/prof33 -name 'kt_kernel_trace.csv' | head -1)
python3 $R/tools/timeline_gaps.py $F qs_hist_kernel 3 > $R/gpurun_out/r01_generation_timeline_33M.txt 2>&1
cp $(find $R/gpurun_out/prof33 -name 'kt_kernel_stats.csv' | head -1) $R/gpurun_out/r01_kernel_stats_33M.csv
rm -rf $R/gpurun_out/prof33
tail -22 $R/gpurun_out/r01_generation_timeline_33M.txt
