# Round 5, GPU call 29: wavefronts without a phase-2 slot leave the workgroup at once (-DABZ_EARLY_EXIT): parity of the variant build,
# then same-box A/B against the shipped sweep.  (The knob was removed again after this call: measured, no change -- profiles/HISTORY.md.)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
V=$R/abcdez.jl_amd/lib/variants
ID=$(rocm-smi --showuniqueid 2> /dev/null | grep 'GPU\[' | sed 's/.*Unique ID: *//' | tr -d '[:space:]')
ABCDEZ_HIP_LIB=$V/libabcdez_hip_early_exit.so timeout -k 10 500 python3 -m pytest $R/tests/test_gpu_packed.py -m gpu -x -q -k "two_phase or generations_parity or shard" > $O/r05_b29_pytest.log 2>&1 || { tail -30 $O/r05_b29_pytest.log; exit 1; }
tail -2 $O/r05_b29_pytest.log
: > $O/r05_early_exit_ab.jsonl
for rep in 1 2 3; do
  for lib in "" early_exit early_exit_w6; do
    L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
    ABCDEZ_HIP_LIB=$L timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'gpu': '$ID', 'lib': '$lib' or 'shipped', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms']}))" >> $O/r05_early_exit_ab.jsonl
  done
done
cat $O/r05_early_exit_ab.jsonl
