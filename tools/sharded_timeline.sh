# kernel trace of the SHARDED code path in a one-rank RCCL group (bench.py --force-collectives): where a generation's time goes
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_sh -o kt -- python3 $R/bench.py --config ${CFG:-smc32} --force-collectives --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs > $R/gpurun_out/sh_bench.log 2>&1
F=$(find $R/gpurun_out/prof_sh -name 'kt_kernel_trace.csv' | head -1)
python3 $R/tools/timeline_gaps.py $F qs_hist_kernel 3 4 > $R/gpurun_out/sh_timeline.txt 2>&1
rm -rf $R/gpurun_out/prof_sh
