# Round 5, GPU call 35: address-translation misses and pending stalls of EVERY kernel of a generation (which kernels lose the sweep's
# translations, which pay for them)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pk
timeout -k 10 400 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pk -o pk -- python3 $R/bench.py --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs --steps 12 --warmup 4 > $O/r05_b35.log 2>&1
python3 $R/tools/per_kernel_counters.py $(find $O/pk -name 'pk_counter_collection.csv' | head -1) > $O/r05_per_kernel_utcl1.json
rm -rf $O/pk
cat $O/r05_per_kernel_utcl1.json
