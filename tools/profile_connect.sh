#!/bin/bash
# rocprofv3 evidence for the RRTConnect rows (DESIGN.md sections 8 and 11) on the GPU box, run from the repo root through gpurun:
#   bash tools/profile_connect.sh <outdir under gpurun_out>
# One --kernel-trace --stats run per bench tool, then one --pmc run (never combined with tracing) for the SE(2) kernel.
set -e
OUT=$PWD/gpurun_out/${1:-prof_connect}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_se2 -- python3 tools/bench_connect_se2.py > $OUT/bench_connect_se2_profiled.json 2> $OUT/trace_se2.err
echo se2 trace done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r3 -- python3 tools/bench_connect.py > $OUT/bench_connect_profiled.json 2> $OUT/trace_r3.err
echo r3 trace done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq_se2 -- python3 tools/bench_connect_se2.py > /dev/null 2> $OUT/pmc_sq_se2.err
echo sq done
# (FETCH_SIZE and WRITE_SIZE in ONE --pmc list abort inside rocprofv3 on this pool -- separate passes, as tools/profile_bench.sh does, if ever needed:
#  memory traffic does not bound these kernels)
find $OUT -name "*.csv" | head -30
