# Round 5: one more box of the pool -- which one, the default headline numbers, the arithmetic-free access pattern on the same GPU
#   bash tools/r05_box_census.sh    (through gpurun; one line in gpurun_out/r05_box_census.jsonl per call: collect them on the calling side)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ID=$(rocm-smi --showuniqueid 2> /dev/null | grep 'GPU\[' | sed 's/.*Unique ID: *//' | tr -d '[:space:]')
HOST=$(hostname 2> /dev/null || echo unknown)
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-other-configs 2> /dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; pc=r.get('pattern_ceiling') or {}
print(json.dumps({'host': '$HOST', 'gpu_unique_id': '$ID', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'],
                  'pattern_ceiling_read_frac': pc.get('read_frac'), 'kernel_over_ceiling': pc.get('kernel_over_ceiling'),
                  'pattern_ceiling_single_launches_read_frac': (pc.get('updates_per_s_single_launches') or 0) * 785 / 8e12,
                  'kernel_over_ceiling_single_launches': pc.get('kernel_over_ceiling_single_launches'),
                  'whole_run_s': ((d.get('whole_run') or {}).get('model') or {}).get('seconds')}))" | tee $O/r05_box_census.jsonl
# the same box: the sweep held to 4 / 6 wavefronts per SIMD against the shipped 5 (library variants, one run each, no pattern kernel)
for lib in "" w4 w6; do
  L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$R/abcdez.jl_amd/lib/variants/libabcdez_hip_$lib.so
  [ -f $L ] || continue
  ABCDEZ_HIP_LIB=$L timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-whole-run --no-pattern 2> /dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'host': '$HOST', 'gpu_unique_id': '$ID', 'lib': '$lib' or 'shipped (5 waves)', 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'], 'ms_per_step': d['ms_per_step']}))" | tee -a $O/r05_box_census_waves.jsonl
done
