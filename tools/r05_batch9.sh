# Round 5: the round's evidence set, part 2 -- sharded one-rank A/B (library-issued collectives against torch's), multi-GPU emulation,
# the whole GPU suite once more on the final tree
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/r05_sharded_one_rank.jsonl
for rep in 1 2; do
 for CFG in smc32 mc1d; do
  for mode in unsharded native torch; do
    case $mode in unsharded) X=""; E="";; native) X="--force-collectives"; E="";; torch) X="--force-collectives"; E="ABZ_COMM=torch";; esac
    env $E timeout 300 python3 $R/bench.py --config $CFG --no-cpu-baseline --no-whole-run --no-other-configs --no-pattern $X 2> /dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'config': '$CFG', 'mode': '$mode', 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'sharded_phases_ms': d.get('sharded_phases_ms')}))" >> $O/r05_sharded_one_rank.jsonl
  done
 done
done
cat $O/r05_sharded_one_rank.jsonl
timeout 600 python3 $R/tools/bench_replay.py --config smc32 > $O/r05_replay_bench_smc32.jsonl 2> $O/r05_replay.err; cat $O/r05_replay_bench_smc32.jsonl | cut -c1-400
timeout 300 python3 $R/tools/bench_replay.py --config lv --total-particles 1048576 > $O/r05_replay_bench_lv.jsonl 2>> $O/r05_replay.err
timeout 300 python3 $R/tools/bench_replay.py --config evidence1d --total-particles 8388608 > $O/r05_replay_bench_evidence1d.jsonl 2>> $O/r05_replay.err
timeout 900 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b9_pytest.log 2>&1; tail -5 $O/r05_b9_pytest.log
