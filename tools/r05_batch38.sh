# Round 5, GPU call 38: part_swap (the kernel before the first sweep) reads one word of every array the sweep uses (-DABZ_PART_WARM):
# does the first sweep still refetch its translations?  Per-kernel counters + first-sweep excess, variant against shipped.  (It does: the knob was removed again.)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in part_warm ""; do
  L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$R/abcdez.jl_amd/lib/variants/libabcdez_hip_$lib.so
  rm -rf $O/pk
  ABCDEZ_HIP_LIB=$L timeout -k 10 400 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pk -o pk -- python3 $R/bench.py --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs --steps 12 --warmup 4 > $O/r05_b38.log 2>&1
  python3 $R/tools/per_kernel_counters.py $(find $O/pk -name 'pk_counter_collection.csv' | head -1) > $O/r05_per_kernel_utcl1_${lib:-shipped}.json
  rm -rf $O/pk
  python3 -c "
import json
d=json.load(open('$O/r05_per_kernel_utcl1_${lib:-shipped}.json'))
print('${lib:-shipped}')
for k in d['kernels_in_launch_order']:
    if 'part_swap' in k['kernel'] or 'smc_swarm' in k['kernel']: print(' ', k['kernel'][:40], k.get('TCP_UTCL1_TRANSLATION_MISS_sum'), k.get('TCP_PENDING_STALL_CYCLES_sum'), k.get('GRBM_GUI_ACTIVE'))"
  rm -rf $O/fs_x
  ABCDEZ_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fs_x -o kt -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs > $O/r05_b38_fs.log 2>&1
  python3 $R/tools/first_sweep_excess.py $(find $O/fs_x -name 'kt_kernel_trace.csv' | head -1) > $O/r05_first_sweep_${lib:-shipped2}.json
  rm -rf $O/fs_x
  head -7 $O/r05_first_sweep_${lib:-shipped2}.json
done
