# Issue / stall breakdown of a configuration's sweep kernel from ONE rocprofv3 --pmc pass of SQ counters (run through gpurun):
#   TAG=r02 CFG=smc32 KERNEL=smc_swarm_packed_kernel bash tools/profile_sq.sh
# writes gpurun_out/${TAG}_${CFG}_sq_counters.json
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}; CFG=${CFG:-smc32}; KERNEL=${KERNEL:-smc_swarm_packed_kernel}
cd /tmp && export TMPDIR=/tmp
COUNTERS=${COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU}
OUT=${OUT:-sq_counters}
timeout 500 rocprofv3 --pmc $COUNTERS \
  --output-format csv -d $R/gpurun_out/sq_$CFG -o sq -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs --steps 8 --warmup 4 > $R/gpurun_out/${TAG}_${CFG}_sq_bench.log 2>&1
F=$(find $R/gpurun_out/sq_$CFG -name 'sq_counter_collection.csv' | head -1)
UPD=$(python3 -c "import json;print([json.loads(l)['config']['launched_incl_warmup']['updates'] for l in open('$R/gpurun_out/${TAG}_${CFG}_sq_bench.log') if l.startswith('{')][-1])" 2>/dev/null || echo 0)
python3 $R/tools/sq_summary.py $F $KERNEL $UPD > $R/gpurun_out/${TAG}_${CFG}_${OUT}.json
rm -rf $R/gpurun_out/sq_$CFG
