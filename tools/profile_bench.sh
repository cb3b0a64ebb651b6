#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's dominant kernel on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_bench.sh <outdir under gpurun_out> [extra bench.py arguments, e.g. "--kernel 1"]
# One --kernel-trace --stats run, then one --pmc run per counter set (never combined with tracing).
set -e
OUT=$PWD/gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-secondary $2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 4 --warmup 1 $ARGS > $OUT/bench_line_profiled.json 2> $OUT/trace.err
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_write.err
echo write done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_sq.err
echo sq done
if [ -z "$2" ]; then python3 tools/stamps_lanes.py 1024 4096 > $OUT/inkernel_stamps.txt 2>&1; else echo "(stamps: tools/stamps.py / tools/stamps_lanes.py for the resident kernels)" > $OUT/inkernel_stamps.txt; fi
find $OUT -name "*.csv" | head -30
