R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ID=$(rocm-smi --showuniqueid 2> /dev/null | grep 'GPU\[' | sed 's/.*Unique ID: *//' | tr -d '[:space:]')
for rep in 1 2; do for lanes in 0 8 2; do
  timeout -k 10 200 python3 $R/bench.py --lanes $lanes --no-cpu-baseline --no-other-configs --no-whole-run --no-pattern 2> /dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'gpu': '$ID', 'lanes': $lanes, 'frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'], 'ms_per_step': d['ms_per_step']}))"
done; done
