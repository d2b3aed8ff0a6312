# Round 5, fourth GPU call: experiments on the two-phase sweep (non-temporal stores, six waves per SIMD), the sharded one-rank lines,
# the pinned result download
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern"
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_two_phase_ab2.jsonl
for rep in 1 2 3 4; do
  for lib in "" 2p_nt 2p_w6s56 2p_s56; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout 300 $B 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'lib': '$lib' or '2p (shipped)', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms']}))" >> $O/r05_two_phase_ab2.jsonl
  done
done
cat $O/r05_two_phase_ab2.jsonl
timeout 300 python3 $R/tools/time_result_download.py > $O/r05_result_download.json 2> $O/r05_result_download.err; cat $O/r05_result_download.json; tail -3 $O/r05_result_download.err
timeout 900 python3 -m pytest $R/tests/test_distributed_gloo.py $R/tests/test_gpu_bench_contract.py $R/tests/test_gpu_shim_sequence.py -m gpu -x -q > $O/r05_b4_pytest.log 2>&1; tail -15 $O/r05_b4_pytest.log
for c in "" "--force-collectives"; do
  timeout 300 python3 $R/bench.py --config mc1d --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern $c 2> $O/r05_mc1d_sharded$c.err | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'config': 'mc1d', 'force_collectives': '$c' != '', 'value': d['value'], 'ms_per_step': d['ms_per_step']}))"
done
ABZ_COMM=torch timeout 300 python3 $R/bench.py --config mc1d --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern --force-collectives 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'config': 'mc1d', 'force_collectives': True, 'comm': 'torch', 'value': d['value'], 'ms_per_step': d['ms_per_step']}))"
ABZ_COMM=torch timeout 300 python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern --force-collectives 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'config': 'smc32', 'force_collectives': True, 'comm': 'torch', 'value': d['value'], 'ms_per_step': d['ms_per_step']}))"
