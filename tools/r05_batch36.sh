# Round 5, GPU call 36: the context's small device objects in ONE allocation -- quick parity, address-translation misses per kernel again,
# the first sweep against its siblings in the kernel trace
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 -m pytest $R/tests/test_gpu_packed.py $R/tests/test_gpu_parity.py -m gpu -x -q -k "generations_parity or end_to_end or c_abi or blobs" > $O/r05_b36_pytest.log 2>&1 || { tail -30 $O/r05_b36_pytest.log; exit 1; }
tail -2 $O/r05_b36_pytest.log
bash $R/tools/r05_batch35.sh > /dev/null 2>&1
python3 -c "
import json
d=json.load(open('$O/r05_per_kernel_utcl1.json'))
for k in d['kernels_in_launch_order']:
    print(k['kernel'][:40], k.get('TCP_UTCL1_TRANSLATION_MISS_sum'), k.get('TCP_PENDING_STALL_CYCLES_sum'), k.get('GRBM_GUI_ACTIVE'))"
rm -rf $O/fs_x
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/fs_x -o kt -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-whole-run --no-pattern --no-other-configs > $O/r05_first_sweep_one_block.log 2>&1
python3 $R/tools/first_sweep_excess.py $(find $O/fs_x -name 'kt_kernel_trace.csv' | head -1) > $O/r05_first_sweep_one_block.json
rm -rf $O/fs_x
head -14 $O/r05_first_sweep_one_block.json
rocm-smi --showuniqueid | grep "GPU\["
