set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tl -o kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_tl_bench.log 2>&1
F=$(find $R/gpurun_out/prof_tl -name 'kt_kernel_trace.csv' | head -1)
python3 $R/tools/timeline_gaps.py $F > $R/gpurun_out/timeline.txt 2>&1
cp $(find $R/gpurun_out/prof_tl -name 'kt_kernel_stats.csv' | head -1) $R/gpurun_out/kernel_stats.csv
rm -rf $R/gpurun_out/prof_tl
cat $R/gpurun_out/timeline.txt
