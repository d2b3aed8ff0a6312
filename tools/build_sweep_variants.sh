#!/bin/bash
# Builds tools/sweep_variants: the d = 32 sweep kernel in several variants, timed against each other in one process
# (tools/sweep_variants.hip).
#   VARIANTS: kernels of this tree, one per line:      name | what | hipcc flags
#   TREES:    the kernel of other commits, one per line: name | what | worktree directory | hipcc flags
#             (git worktree add _r02 79040f7;  git worktree add _icdf 4d39988;  _r02p7 = a copy of _r02 with ABZ_PHILOX_ROUNDS 7)
set -e
cd "$(dirname "$0")/.."
B=tools/build_variants
mkdir -p $B
INC="-Iabcdez.jl_amd/csrc -Iinclude -I$B"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math"
VARIANTS=${VARIANTS:-"
cur|shipped: Box-Muller + Philox4x32-7, one tile per workgroup, occupancy left to the compiler (5 waves)|
cur4|shipped kernel held to 4 waves per SIMD|-DVWAVES=4
"}
TREES=${TREES:-"
r02|round 2 (79040f7): Box-Muller + Philox4x32-10, one tile per workgroup, 5 waves|_r02|
r02p7|round 2 with Philox4x32-7|_r02p7|
icdf|round 3 experiment (4d39988): inverse-CDF normal + Philox4x32-7, looping workgroups, staged tables, prefetch, 4 waves|_icdf|-DTREE_ICDF -mllvm -disable-machine-licm
"}
: > $B/variants.inc
: > $B/trees.inc
OBJS=""
while IFS='|' read -r name what flags; do
  [ -z "$name" ] && continue
  echo "V($name, \"$what\")" >> $B/variants.inc
  /opt/rocm/bin/hipcc $COMMON $INC -DVNAME=$name $flags -c tools/sweep_variant_kernel.hip -o $B/$name.o &
  OBJS="$OBJS $B/$name.o"
done <<< "$VARIANTS"
while IFS='|' read -r name what tree flags; do
  [ -z "$name" ] && continue
  [ -d $tree/abcdez.jl_amd/csrc ] || { echo "skipping $name: no worktree $tree"; continue; }
  echo "T($name, \"$what\")" >> $B/trees.inc
  /opt/rocm/bin/hipcc $COMMON -I$tree/abcdez.jl_amd/csrc -I$tree/include -DTREE=$name $flags -c tools/sweep_variant_tree.hip -o $B/$name.o &
  OBJS="$OBJS $B/$name.o"
done <<< "$TREES"
wait
/opt/rocm/bin/hipcc $COMMON $INC -c tools/sweep_variants.hip -o $B/main.o
[ -f tools/liblayout_bench.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/liblayout_bench.so tools/layout_bench.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 $B/main.o $OBJS -Ltools -llayout_bench -Wl,-rpath,'$ORIGIN' -o tools/sweep_variants
echo built tools/sweep_variants
