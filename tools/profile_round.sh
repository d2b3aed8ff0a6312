# Regenerates the measurement evidence of a round on the GPU box (run through gpurun, one call per CFG to stay inside
# the time limit):   TAG=r02 CFG=smc32 PMC=1 bash tools/profile_round.sh
#   bench line with the CPU baseline, rocprofv3 kernel-trace stats + per-generation timeline of the same command,
#   PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) when PMC=1, the access-pattern ceiling when CEIL=1.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}; CFG=${CFG:-smc32}
case $CFG in
  smc32) MARK=qs_hist_kernel; LANES=0; LD=32; KERN=smc_swarm_packed_kernel;;
  lv) MARK=qs_hist_kernel; LANES=0; LD=4; KERN=smc_lv_phase1+smc_lv_phase2;;     # one sweep = two launches (abz_kernels.h, smc_lv_phase1_body)
  evidence1d) MARK=qs_hist_kernel; LANES=0; LD=1; KERN=smc_swarm_packed_kernel;;
  mc1d) MARK=mc_swarm_kernel; LANES=0; LD=1; KERN=mc_swarm_kernel; PER_STEP=1;;
esac
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --config $CFG --no-other-configs > $R/gpurun_out/${TAG}_${CFG}_bench.log 2>&1
MARK=$MARK LANES=$LANES TAG=$TAG CFG=$CFG KERN=$KERN PER_STEP=${PER_STEP:-3} bash $R/tools/profile_config.sh
if [ "${PMC:-0}" = "1" ]; then
  # the bench line's OWN window: default --steps / --warmup of the configuration; traffic is counted over the sweep dispatches of
  # the timed steps only (tools/pmc_traffic.py last-n), with that pass's own updates and acceptance
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$CFG -o f -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs > $R/gpurun_out/${TAG}_${CFG}_pmc_f.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$CFG -o w -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs > $R/gpurun_out/${TAG}_${CFG}_pmc_w.log 2>&1
  win() { python3 -c "import json,sys;w=[json.loads(l)['config']['timed_window'] for l in open('$1') if l.startswith('{')][-1];print(w['$2'])" 2>/dev/null || echo 0; }
  UF=$(win $R/gpurun_out/${TAG}_${CFG}_pmc_f.log updates); UW=$(win $R/gpurun_out/${TAG}_${CFG}_pmc_w.log updates)
  NL=$(win $R/gpurun_out/${TAG}_${CFG}_pmc_f.log sweep_launches); NA=$(win $R/gpurun_out/${TAG}_${CFG}_pmc_f.log naccs)
  FC=$(find $R/gpurun_out/pmc_fetch_$CFG -name '*counter_collection.csv' | head -1)
  WC=$(find $R/gpurun_out/pmc_write_$CFG -name '*counter_collection.csv' | head -1)
  ACC=$(python3 -c "print($NA / max($UF, 1))")
  python3 $R/tools/pmc_traffic.py $FC $WC $R/gpurun_out/${TAG}_hbm_traffic_${CFG}.json $KERN $LANES $LD $ACC $UF $UW $NL
  cp $FC $R/gpurun_out/${TAG}_${CFG}_pmc_fetch_size.csv; cp $WC $R/gpurun_out/${TAG}_${CFG}_pmc_write_size.csv
  rm -rf $R/gpurun_out/pmc_fetch_$CFG $R/gpurun_out/pmc_write_$CFG
fi
if [ "${CEIL:-0}" = "1" ]; then
  $R/tools/layout_bench > $R/gpurun_out/${TAG}_layout_bench.jsonl 2>&1
fi
grep '^{' $R/gpurun_out/${TAG}_${CFG}_bench.log | cut -c1-400
