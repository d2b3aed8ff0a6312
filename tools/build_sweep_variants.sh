#!/bin/bash
# Builds tools/sweep_variants: the d = 32 sweep kernel in several compile-time variants, timed against each other
# in one process (tools/sweep_variants.hip).  Each line of VARIANTS: name | what | hipcc flags
set -e
cd "$(dirname "$0")/.."
B=tools/build_variants
mkdir -p $B
INC="-Iabcdez.jl_amd/csrc -Iinclude -I$B"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math"
VARIANTS=${VARIANTS:-"
base|shipped: prefetch of the next round's slot bits, 5 waves, no machine-LICM|-DVWAVES=5 -mllvm -disable-machine-licm
nopf|no prefetch, 5 waves|-DVWAVES=5 -DABZ_SWEEP_PREFETCH=0 -mllvm -disable-machine-licm
pf4|prefetch, 4 waves (128 VGPRs)|-DVWAVES=4 -mllvm -disable-machine-licm
nopf4|no prefetch, 4 waves, free scheduling|-DVWAVES=4 -DABZ_SWEEP_PREFETCH=0 -DABZ_SWEEP_SCHED=0 -mllvm -disable-machine-licm
nosim|ABLATION no simulator (memory side of the real kernel), 5 waves no prefetch|-DVWAVES=5 -DABZ_SWEEP_PREFETCH=0 -DABZ_ABLATE_SIM -mllvm -disable-machine-licm
nodon|ABLATION donors = own row (compute side of the real kernel), 5 waves no prefetch|-DVWAVES=5 -DABZ_SWEEP_PREFETCH=0 -DABZ_ABLATE_DONORS -mllvm -disable-machine-licm
nosimpf|ABLATION no simulator, prefetch, 5 waves|-DVWAVES=5 -DABZ_ABLATE_SIM -mllvm -disable-machine-licm
"}
: > $B/variants.inc
OBJS=""
while IFS='|' read -r name what flags; do
  [ -z "$name" ] && continue
  echo "V($name, \"$what\")" >> $B/variants.inc
  /opt/rocm/bin/hipcc $COMMON $INC -DVNAME=$name $flags -c tools/sweep_variant_kernel.hip -o $B/$name.o &
  OBJS="$OBJS $B/$name.o"
done <<< "$VARIANTS"
wait
R02=""
if [ -d _r02/abcdez.jl_amd/csrc ]; then      # round 2's kernel, when its worktree is there:  git worktree add _r02 <round-2 commit>
  /opt/rocm/bin/hipcc $COMMON -I_r02/abcdez.jl_amd/csrc -I_r02/include -c tools/sweep_variant_r02.hip -o $B/r02.o
  OBJS="$OBJS $B/r02.o"; R02="-DWITH_R02"
fi
/opt/rocm/bin/hipcc $COMMON $INC $R02 -c tools/sweep_variants.hip -o $B/main.o
[ -f tools/liblayout_bench.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/liblayout_bench.so tools/layout_bench.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 $B/main.o $OBJS -Ltools -llayout_bench -Wl,-rpath,'$ORIGIN' -o tools/sweep_variants
echo built tools/sweep_variants
