# Round 5, third GPU call: same-box A/B of library builds (one-phase / two-phase / occupancy caps) in the real pipeline, the fixed
# same-process tool, the pinned result download, the PMC calibration of 8-byte gathers
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern"
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_two_phase_ab.jsonl
for rep in 1 2 3 4; do
  for lib in 1p "" 2p_w4 2p_w3; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout 300 $B 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'lib': '$lib' or '2p (shipped)', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'], 'launches': r['launches'], 'updates_per_launch': r['updates_per_launch']}))" >> $O/r05_two_phase_ab.jsonl
  done
done
cat $O/r05_two_phase_ab.jsonl
timeout 300 $R/tools/sweep_variants 2965608 12 0.147 20 > $O/r05_sweep_variants_two_phase_sustained.jsonl 2> $O/r05_sweep_variants2.err
cat $O/r05_sweep_variants_two_phase_sustained.jsonl
timeout 300 python3 $R/tools/time_result_download.py > $O/r05_result_download.json 2> $O/r05_result_download.err; cat $O/r05_result_download.json; tail -3 $O/r05_result_download.err
bash $R/tools/r05_pmc_calibration.sh
# the RCCL-behind-the-ABI paths on the one GPU (one-rank groups) + the sharded one-rank mc1d line
timeout 900 python3 -m pytest $R/tests/test_distributed_gloo.py $R/tests/test_gpu_bench_contract.py $R/tests/test_gpu_shim_sequence.py -m gpu -x -q > $O/r05_b3_pytest.log 2>&1; tail -15 $O/r05_b3_pytest.log
for c in "" "--force-collectives"; do
  timeout 300 python3 $R/bench.py --config mc1d --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern $c 2> $O/r05_mc1d_sharded$c.err | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'force_collectives': '$c' != '', 'value': d['value'], 'ms_per_step': d['ms_per_step']}))"
done
for c in "" "--force-collectives"; do
  timeout 300 python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern $c 2> $O/r05_smc32_sharded$c.err | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'force_collectives': '$c' != '', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'phases': d.get('sharded_phases_ms')}))"
done
