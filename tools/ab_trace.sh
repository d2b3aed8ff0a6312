# kernel traces + host-side timing of this tree and of a worktree of another commit, on one box (run through gpurun):
#   OTHER=_r02 TAG=r03 bash tools/ab_trace.sh      -> gpurun_out/${TAG}_ab_{cur,other}_kernel_trace.csv, _host_overhead.txt
R=$GRAFT_REPO_ROOT
OTHER=${OTHER:-_r02}; TAG=${TAG:-r03}
cd /tmp && export TMPDIR=/tmp
for side in cur other; do
  if [ $side = cur ]; then T=$R; else T=$R/$OTHER; fi
  python3 $T/tools/host_overhead.py > $R/gpurun_out/${TAG}_ab_${side}_host_overhead.txt 2>&1
  python3 $T/tools/host_overhead.py >> $R/gpurun_out/${TAG}_ab_${side}_host_overhead.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_$side -o kt -- python3 $T/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-pattern $( [ $side = cur ] && echo --no-other-configs ) > $R/gpurun_out/${TAG}_ab_${side}_bench.log 2>&1
  cp $(find /tmp/ab_$side -name 'kt_kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_ab_${side}_kernel_trace.csv
done
