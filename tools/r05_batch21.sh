# Round 5, GPU call 21: the first sweep of a generation against its siblings (round-4 VERDICT 1b) -- kernel trace of 40 generations of the
# headline configuration, plain and with the partition's row moves made non-temporal (library variant)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in "" $EXTRA_LIBS; do
  L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$R/abcdez.jl_amd/lib/variants/libabcdez_hip_$lib.so
  rm -rf $O/fs_$lib
  ABCDEZ_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fs_$lib -o kt -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs > $O/r05_first_sweep_${lib:-shipped}.log 2>&1
  python3 $R/tools/first_sweep_excess.py $(find $O/fs_$lib -name 'kt_kernel_trace.csv' | head -1) > $O/r05_first_sweep_${lib:-shipped}.json
  rm -rf $O/fs_$lib
  head -16 $O/r05_first_sweep_${lib:-shipped}.json
done
