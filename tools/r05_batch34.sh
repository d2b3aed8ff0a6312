# Round 5, GPU call 34: per-dispatch counters of the sweeps by their position in the generation: fabric reads, L2 hits / misses, address
# translation misses, wave cycles -- what does the first sweep of a generation do more of?
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $O/pc_$tag
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/pc_$tag -o pc -- python3 $R/bench.py --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs --steps 12 --warmup 4 > $O/r05_b34_$tag.log 2>&1
  python3 $R/tools/per_sweep_counters.py $(find $O/pc_$tag -name 'pc_counter_collection.csv' | head -1) > $O/r05_per_sweep_$tag.json
  rm -rf $O/pc_$tag
  cat $O/r05_per_sweep_$tag.json
done
