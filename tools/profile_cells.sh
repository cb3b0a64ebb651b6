#!/bin/bash
# rocprofv3 evidence for rrt_cells.hip on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_cells.sh <outdir under gpurun_out>
# One --kernel-trace --stats run of bench.py, then one --pmc run per counter set (never combined with tracing).
set -e
OUT=$PWD/gpurun_out/${1:-prof_cells}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 4 --warmup 1 $ARGS > $OUT/bench_line_profiled.json 2> $OUT/trace.err
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_write.err
echo write done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_sq.err
echo sq done
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $OUT/pmc_sq2.err || echo "sq2 failed"
python3 tools/stamps_cells.py 1024 2 > $OUT/inkernel_stamps.txt 2>&1
find $OUT -name "*.csv" | head -30
