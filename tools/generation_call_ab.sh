# One library call per generation (abcdez_smc_generation_packed) against the three calls of rounds 3-5 (prologue, [resampling], sweeps;
# ABZ_GENERATION_STEPWISE=1), same box, alternating:  gpurun -- 'bash tools/generation_call_ab.sh [rounds]'
R=$GRAFT_REPO_ROOT; N=${1:-4}
for i in $(seq 1 $N); do
  for v in 0 1; do
    for c in smc32 evidence1d lv; do
      ABZ_GENERATION_STEPWISE=$v python3 $R/bench.py --config $c --no-other-configs --no-cpu-baseline --no-pattern 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; w=[v for v in d.get('whole_run',{}).values() if isinstance(v,dict) and 'seconds' in v]; print(json.dumps({'round': $i, 'config': '$c', 'generation': 'three calls (rounds 3-5)' if $v else 'one library call', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'avg_launch_ms': r['avg_launch_ms'], 'whole_run_seconds': [x['seconds'] for x in w], 'whole_run_logZ': [x.get('logZ') for x in w]}))"
    done
  done
done
