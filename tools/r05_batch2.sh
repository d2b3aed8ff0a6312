# Round 5, second GPU call: the two-phase sweep -- parity first, then same-process and in-pipeline A/B against the one-phase body
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/tools/sweep_variants 2965608 12 0.147 20 > $O/r05_sweep_variants_two_phase_sustained.jsonl 2> $O/r05_sweep_variants2.err
cat $O/r05_sweep_variants_two_phase_sustained.jsonl
timeout 300 $R/tools/sweep_variants 2965608 12 0.147 1 > $O/r05_sweep_variants_two_phase_single.jsonl 2>> $O/r05_sweep_variants2.err
cat $O/r05_sweep_variants_two_phase_single.jsonl
timeout 600 python3 $R/bench.py --config smc32 --no-other-configs > $O/r05_b2_bench.log 2>&1
grep '^{' $O/r05_b2_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({k: d[k] for k in ('value','ms_per_step')}), json.dumps({k: r.get(k) for k in ('frac','avg_launch_ms','launches','updates_per_launch','pattern_ceiling')}))"
timeout 900 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b2_pytest.log 2>&1 || tail -40 $O/r05_b2_pytest.log
tail -3 $O/r05_b2_pytest.log
