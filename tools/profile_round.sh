set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $R/gpurun_out/bench_r01_final.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_final_bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
ls -R $R/gpurun_out/prof_final $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write | head -30
grep metric $R/gpurun_out/bench_r01_final.log | cut -c1-300
