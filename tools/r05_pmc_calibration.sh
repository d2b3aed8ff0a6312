# Round 5: what do FETCH_SIZE / the fabric request counters report for RANDOM 8-BYTE GATHERS?
# tools/gather_d1 V0 has a known request stream (own row coalesced + two random 8-byte reads per position, 127.99 distinct 64-byte
# lines per 128 requests); the TCC counters split the L2's fabric reads by request size, which needs no correction factor at all.
#   bash tools/r05_pmc_calibration.sh      (through gpurun)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
[ -x $R/tools/gather_d1 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/gather_d1 $R/tools/gather_d1.hip
for pass in "sizes TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" "fetch FETCH_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  set -- $pass; tag=$1; shift
  timeout 300 rocprofv3 --pmc $@ --output-format csv -d /tmp/cal_$tag -o c -- $R/tools/gather_d1 > $O/r05_gather_d1_pmc_$tag.log 2>&1
  cp $(find /tmp/cal_$tag -name 'c_counter_collection.csv' | head -1) $O/r05_gather_d1_pmc_$tag.csv
done
python3 $R/tools/pmc_gather_calibration.py $O/r05_gather_d1_pmc_sizes.csv $O/r05_gather_d1_pmc_fetch.csv $O/r05_gather_d1_pmc_l2.csv > $O/r05_pmc_gather_calibration.json
cat $O/r05_pmc_gather_calibration.json
# the two d = 1 configurations with the size-split counters (bytes = sum of size x requests: no factor to argue about)
for CFG in evidence1d mc1d; do
  case $CFG in evidence1d) K=smc_swarm_packed_kernel;; mc1d) K=mc_swarm_kernel;; esac
  TAG=r05 CFG=$CFG KERNEL=$K OUT=ea_sizes COUNTERS="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" bash $R/tools/profile_sq.sh
  TAG=r05 CFG=$CFG KERNEL=$K OUT=ea_writes COUNTERS="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" bash $R/tools/profile_sq.sh
  TAG=r05 CFG=$CFG KERNEL=$K OUT=fetch COUNTERS="FETCH_SIZE" bash $R/tools/profile_sq.sh
done
