# Round 5: fabric traffic of the two-phase sweep with non-temporal and with plain row stores (size-split counters: no factor)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
V=$R/abcdez.jl_amd/lib/variants
for lib in "" 2p_plainst 1p; do
  L=$R/abcdez.jl_amd/lib/libabcdez_hip.so; [ -n "$lib" ] && L=$V/libabcdez_hip_$lib.so
  export ABCDEZ_HIP_LIB=$L
  TAG=r05 CFG=smc32 OUT=ea_reads_${lib:-shipped} COUNTERS="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" bash $R/tools/profile_sq.sh
  TAG=r05 CFG=smc32 OUT=ea_writes_${lib:-shipped} COUNTERS="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum" bash $R/tools/profile_sq.sh
  TAG=r05 CFG=smc32 OUT=l2_${lib:-shipped} COUNTERS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum" bash $R/tools/profile_sq.sh
done
unset ABCDEZ_HIP_LIB
for f in $O/r05_smc32_ea_reads_* $O/r05_smc32_ea_writes_* $O/r05_smc32_l2_*; do echo $f; python3 -c "
import json;d=json.load(open('$f'));print(json.dumps(d['per_update']))"; done
