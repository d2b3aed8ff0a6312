# Round 5: final tree -- the whole GPU suite, the Lotka-Volterra profile set, the default bench line, smoke()
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b15_pytest.log 2>&1 || { tail -40 $O/r05_b15_pytest.log; exit 1; }
tail -3 $O/r05_b15_pytest.log
cd $R && timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; cd /tmp
TAG=r05 CFG=lv PMC=0 bash $R/tools/profile_round.sh
timeout 900 python3 $R/bench.py > $O/r05_bench_line.log 2>&1; grep '^{' $O/r05_bench_line.log | tail -1 > $O/r05_bench_line.json; python3 $R/tools/show_bench.py $O/r05_bench_line.json 2>/dev/null | cut -c1-250 | head -24
timeout 600 python3 $R/bench.py --config lv --no-other-configs > $O/r05_bench_line_lv.log 2>&1; grep '^{' $O/r05_bench_line_lv.log | tail -1 > $O/r05_bench_line_lv.json
