# Round 5, first GPU call: parity of the serpentine sweeps, same-box A/B of the sweep order / timing mode / allocation, and the
# address-translation and fabric counters of the d = 32 sweep (run through gpurun):   bash tools/r05_batch1.sh
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config smc32 --no-cpu-baseline --no-whole-run --no-other-configs"
timeout 900 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b1_pytest.log 2>&1 || { tail -30 $O/r05_b1_pytest.log; exit 1; }
tail -3 $O/r05_b1_pytest.log
# --- A/B in the real pipeline, alternating: order of the sweeps, and the two ways of bracketing them
: > $O/r05_serpentine_ab.jsonl
for rep in 1 2 3; do
  for s in 0 1; do
    for m in 3 $( [ $rep = 1 ] && echo 2 ); do
      ABZ_SERPENTINE=$s timeout 300 $B --no-pattern --timing-mode $m 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'serpentine': $s, 'timing_mode': $m, 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'], 'launches': r['launches'], 'updates_per_launch': r['updates_per_launch']}))" >> $O/r05_serpentine_ab.jsonl
    done
  done
done
cat $O/r05_serpentine_ab.jsonl
# --- the same kernel, one process, 20 launches back to back per measurement
timeout 300 $R/tools/sweep_variants 2965608 12 0.147 20 > $O/r05_sweep_variants_serpentine_sustained.jsonl 2> $O/r05_sweep_variants.err
cat $O/r05_sweep_variants_serpentine_sustained.jsonl
# --- allocation: torch's caching allocator (default) / one torch arena at 2 MiB boundaries / abcdez_dev_alloc
#     (ABZ_ARENA: a knob of engine.py at the time of this call, removed at the end of the round -- it changed nothing)
: > $O/r05_arena_ab.jsonl
for rep in 1; do
  for a in "" torch lib ""; do
    ABZ_ARENA=$a timeout 300 $B --no-pattern 2> $O/r05_arena_$a.err | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'arena': '$a' or 'torch caching allocator', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms']}))" >> $O/r05_arena_ab.jsonl
  done
done
cat $O/r05_arena_ab.jsonl
# --- counters of the sweep kernel: address translation, fabric read latency / credit stalls, L2 hit rate (one pass each)
TAG=r05 CFG=smc32 OUT=utcl1 COUNTERS="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" bash $R/tools/profile_sq.sh
TAG=r05 CFG=smc32 OUT=utcl1_stalls COUNTERS="TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum" bash $R/tools/profile_sq.sh
TAG=r05 CFG=smc32 OUT=ea_read COUNTERS="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" bash $R/tools/profile_sq.sh
TAG=r05 CFG=smc32 OUT=l2 COUNTERS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" bash $R/tools/profile_sq.sh
TAG=r05 CFG=smc32 OUT=ea_sizes COUNTERS="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum" bash $R/tools/profile_sq.sh
rocprofv3 -L > $O/r05_rocprofv3_counters_list.txt 2>&1 || true
ls -la $O | tail -30
