# Round 5: Lotka-Volterra rounds of 1 / 2 / 4 observations per barrier, A/B on one box (quick parity of the LV cases first)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 -m pytest $R/tests/test_gpu_packed.py $R/tests/test_gpu_parity.py -m gpu -x -q -k "lv or two_phase or Lotka or lotka" > $O/r05_b13_pytest.log 2>&1 || { tail -40 $O/r05_b13_pytest.log; exit 1; }
tail -3 $O/r05_b13_pytest.log
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_lv_rounds_ab.jsonl
for rep in 1 2 3; do
  for lib in "" lv_round1 lv_round4 lv_no_exit; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout 300 python3 $R/bench.py --config lv --no-cpu-baseline --no-other-configs --no-pattern 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; w=d.get('whole_run') or {}
print(json.dumps({'lib': '$lib' or 'shipped (2 observations per round)', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'avg_launch_ms': r['avg_launch_ms'], 'whole_run_s': w.get('model', {}).get('seconds'), 'whole_run_value': w.get('model', {}).get('value'), 'logZ': w.get('model', {}).get('logZ')}))" >> $O/r05_lv_rounds_ab.jsonl
  done
done
cat $O/r05_lv_rounds_ab.jsonl
