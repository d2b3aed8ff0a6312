# Round 5, GPU call 16: the further prior families (abcdez_spec.h ABZ_PRIOR_EXPONENTIAL ... BINOMIAL) on the device -- the whole GPU
# suite (log-densities against scipy and the oracle, samplers at the initial population, sweeps / replay / end-to-end parity,
# 24 random models of all 18 families, evidence and posterior mean of every family against quadrature), then the default bench
# line (the PLAIN kernels of the BASELINE configurations must be where they were):   bash tools/r05_batch16.sh
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 1000 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b16_pytest.log 2>&1 || { tail -40 $O/r05_b16_pytest.log; exit 1; }
tail -3 $O/r05_b16_pytest.log
timeout 300 python3 $R/bench.py --no-cpu-baseline > $O/r05_b16_bench.json 2> $O/r05_b16_bench.err && python3 -c "
import json; d=json.loads([l for l in open('$O/r05_b16_bench.json') if l.startswith('{')][-1]); r=d['roofline']
print('smc32', d['value'], d['ms_per_step'], r['frac'])"
