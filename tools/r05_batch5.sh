# Round 5, fifth GPU call: the shipped sweep (two phases, six waves per SIMD, non-temporal row stores) -- parity, then A/B
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b5_pytest.log 2>&1 || { tail -40 $O/r05_b5_pytest.log; exit 1; }
tail -3 $O/r05_b5_pytest.log
B="python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern"
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_two_phase_ab3.jsonl
for rep in 1 2 3 4; do
  for lib in "" 2p_prev 2p_w6_plainst 1p; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout 300 $B 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'lib': '$lib' or 'shipped (2 phases, 6 waves, nt stores)', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms']}))" >> $O/r05_two_phase_ab3.jsonl
  done
done
cat $O/r05_two_phase_ab3.jsonl
timeout 600 python3 $R/bench.py > $O/r05_b5_bench_default.log 2>&1; grep '^{' $O/r05_b5_bench_default.log | cut -c1-1500
