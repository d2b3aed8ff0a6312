# Round 5: the round's evidence set, part 1 -- the default bench line (all four configurations + CPU baseline), profile sets of the other
# three configurations
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py > $O/r05_bench_line.log 2>&1; grep '^{' $O/r05_bench_line.log | tail -1 > $O/r05_bench_line.json; python3 $R/tools/show_bench.py $O/r05_bench_line.json 2>/dev/null | head -40 || cut -c1-600 $O/r05_bench_line.json
for CFG in mc1d lv evidence1d; do
  timeout 600 python3 $R/bench.py --config $CFG --no-other-configs > $O/r05_bench_line_$CFG.log 2>&1; grep '^{' $O/r05_bench_line_$CFG.log | tail -1 > $O/r05_bench_line_$CFG.json
done
TAG=r05 CFG=mc1d PMC=1 bash $R/tools/profile_round.sh
TAG=r05 CFG=evidence1d PMC=1 bash $R/tools/profile_round.sh
TAG=r05 CFG=lv PMC=0 bash $R/tools/profile_round.sh
ls $O | grep r05_ | wc -l
