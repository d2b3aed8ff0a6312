# Round 5, GPU calls 19 and 20 (the second after the list counter lost its memset): the tree with the two-launch Lotka-Volterra sweep -- whole GPU suite, smoke(), the LV profile set, both bench lines
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest $R/tests -m gpu -x -q > $O/r05_b19_pytest.log 2>&1 || { tail -40 $O/r05_b19_pytest.log; exit 1; }
tail -3 $O/r05_b19_pytest.log
cd $R && timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; cd /tmp
TAG=r05 CFG=lv PMC=0 bash $R/tools/profile_round.sh
timeout 900 python3 $R/bench.py > $O/r05_bench_line.log 2>&1; grep '^{' $O/r05_bench_line.log | tail -1 > $O/r05_bench_line.json; python3 $R/tools/show_bench.py $O/r05_bench_line.json 2>/dev/null | cut -c1-250 | head -24
timeout 600 python3 $R/bench.py --config lv --no-other-configs > $O/r05_bench_line_lv.log 2>&1; grep '^{' $O/r05_bench_line_lv.log | tail -1 > $O/r05_bench_line_lv.json
