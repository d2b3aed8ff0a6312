# Round 5: Lotka-Volterra phase 2 round by round with early exit -- parity, then A/B (no early exit / one phase) on one box
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b12_pytest.log 2>&1 || { tail -40 $O/r05_b12_pytest.log; exit 1; }
tail -3 $O/r05_b12_pytest.log
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_lv_early_exit_ab.jsonl
for rep in 1 2 3; do
  for lib in "" lv_no_exit 1p; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout 300 python3 $R/bench.py --config lv --no-cpu-baseline --no-other-configs --no-pattern 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; w=d.get('whole_run') or {}
print(json.dumps({'lib': '$lib' or 'shipped (two phases, rounds with early exit)', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'avg_launch_ms': r['avg_launch_ms'], 'frac': r['frac'], 'simulated_fraction': r.get('simulated_fraction_of_updates'), 'whole_run': {k: w.get('model', {}).get(k) for k in ('generations', 'seconds', 'value', 'logZ', 'posterior_mean')}}))" >> $O/r05_lv_early_exit_ab.jsonl
  done
done
cat $O/r05_lv_early_exit_ab.jsonl
