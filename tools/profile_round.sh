# Regenerates the measurement evidence of a round on the GPU box (run through gpurun):
#   bench line (with the CPU baseline), rocprofv3 kernel-trace stats of the same command, per-generation timeline.
# PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) are taken when PMC=1 -- the sweep kernel has to have
# changed for them to move.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r01}
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_prof_bench.log 2>&1
F=$(find $R/gpurun_out/prof_final -name 'kt_kernel_trace.csv' | head -1)
python3 $R/tools/timeline_gaps.py $F qs_hist_kernel 3 4 > $R/gpurun_out/${TAG}_generation_timeline.txt 2>&1
cp $(find $R/gpurun_out/prof_final -name 'kt_kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf $R/gpurun_out/prof_final
if [ "${PMC:-0}" = "1" ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
  ls -R $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write | head -30
fi
grep metric $R/gpurun_out/${TAG}_bench.log | cut -c1-300
