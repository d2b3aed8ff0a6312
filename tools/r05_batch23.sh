# Round 5, GPU call 23: the d = 32 sweep with non-temporal row LOADS (own row / donor rows / all three) and with wave priorities
# (phase 1 above phase 2), same box, alternating, three rounds
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern"
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_sweep_loads_prio_ab.jsonl
for rep in 1 2 3; do
  for lib in "" nt1 nt2 nt3 prio3 prio1; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout -k 10 300 $B 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'lib': '$lib' or 'shipped', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms']}))" >> $O/r05_sweep_loads_prio_ab.jsonl
  done
done
cat $O/r05_sweep_loads_prio_ab.jsonl
