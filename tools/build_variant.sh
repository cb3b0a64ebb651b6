#!/bin/bash
# Build an alternative liboxmpl_hip.so with extra -D flags for rrt_lanes.hip only (kernel tuning experiments):
#   bash tools/build_variant.sh <name> "<-D flags>"   ->  build_variants/<name>/liboxmpl_hip.so
# Load it with OXMPL_HIP_LIB=build_variants/<name>/liboxmpl_hip.so (oxmpl_amd/capi.py).  The other objects come from the
# product build in oxmpl_amd/csrc (run make there first).  build_variants/ is git-ignored but travels with gpurun.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build_variants/$NAME
mkdir -p $OUT
cd $ROOT/oxmpl_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function $@ -c rrt_lanes.hip -o $OUT/rrt_lanes.o 2> $OUT/build.err || { cat $OUT/build.err; exit 1; }
# the product library's objects (the Makefile's SRCS), not whatever *.o an older build left behind
OBJS=$(make -s -pn 2>/dev/null | sed -n 's/^SRCS *:= *//p' | head -1 | tr ' ' '\n' | sed 's/\.hip$/.o/' | grep -v '^rrt_lanes.o$' | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/liboxmpl_hip.so $OBJS $OUT/rrt_lanes.o
echo built $OUT/liboxmpl_hip.so
