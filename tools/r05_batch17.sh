# Round 5, GPU call 17: the Lotka-Volterra sweep as two launches (phase 1 over the positions, phase 2 over the hand-over list) -- parity
# of every LV case, then same-box A/B against the one-kernel two-phase body and other workgroup sizes / round lengths of the second launch:
#   bash tools/r05_batch17.sh
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_packed.py $R/tests/test_gpu_parity.py $R/tests/test_gpu_fullsize.py $R/tests/test_distributed_gloo.py -m gpu -x -q -k "lv or two_phase or Lotka or lotka or share_one_gpu or rccl" > $O/r05_b17_pytest.log 2>&1 || { tail -40 $O/r05_b17_pytest.log; exit 1; }
tail -3 $O/r05_b17_pytest.log
V=$R/abcdez.jl_amd/lib/variants
: > $O/r05_lv_split_ab.jsonl
for rep in 1 2 3; do
  for lib in "" lv_one_kernel lv_b2_512 lv_b2_128 lv_split_r1 lv_split_r4; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout -k 10 300 python3 $R/bench.py --config lv --no-cpu-baseline --no-other-configs --no-pattern 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; w=d.get('whole_run') or {}
print(json.dumps({'lib': '$lib' or 'shipped (two launches, 256 threads, 2 observations per round)', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'avg_launch_ms': r['avg_launch_ms'], 'frac': r['frac'], 'whole_run_s': w.get('model', {}).get('seconds'), 'whole_run_value': w.get('model', {}).get('value'), 'logZ': w.get('model', {}).get('logZ')}))" >> $O/r05_lv_split_ab.jsonl
  done
done
cat $O/r05_lv_split_ab.jsonl
