#!/bin/bash
# Build an alternative liboxmpl_hip.so with extra -D flags for ONE kernel source (kernel tuning experiments):
#   bash tools/build_variant.sh <name> <source, e.g. rrt_cells.hip> "<-D flags>"   ->  build_variants/<name>/liboxmpl_hip.so
# Load it with OXMPL_HIP_LIB=build_variants/<name>/liboxmpl_hip.so (oxmpl_amd/capi.py).  The other objects come from the
# product build in oxmpl_amd/csrc (run make there first).  build_variants/ is git-ignored but travels with gpurun.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift; shift
OBJ=${SRC%.hip}.o
OUT=$ROOT/build_variants/$NAME
mkdir -p $OUT
cd $ROOT/oxmpl_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function $@ -c $SRC -o $OUT/$OBJ 2> $OUT/build.err || { cat $OUT/build.err; exit 1; }
# the product library's objects (the Makefile's SRCS), not whatever *.o an older build left behind
OBJS=$(make -s -pn 2>/dev/null | sed -n 's/^SRCS *:= *//p' | head -1 | tr ' ' '\n' | sed 's/\.hip$/.o/' | grep -v "^$OBJ\$" | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/liboxmpl_hip.so $OBJS $OUT/$OBJ
echo built $OUT/liboxmpl_hip.so
