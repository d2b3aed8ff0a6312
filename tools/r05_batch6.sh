# Round 5, sixth GPU call: A/B of the shipped sweep on one more box, then the round's profile set for smc32
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern"
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_two_phase_ab4.jsonl
for rep in 1 2 3 4; do
  for lib in "" 2p_plainst 2p_w6 1p; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout 300 $B 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'lib': '$lib' or 'shipped (2 phases, 5 waves, nt stores)', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms']}))" >> $O/r05_two_phase_ab4.jsonl
  done
done
cat $O/r05_two_phase_ab4.jsonl
TAG=r05 CFG=smc32 PMC=1 CEIL=0 bash $R/tools/profile_round.sh
TAG=r05 CFG=smc32 bash $R/tools/profile_sq.sh
cat $O/r05_smc32_sq_counters.json | head -40
