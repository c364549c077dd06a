#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries land in gpurun_out/prof/.
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-pairs 0 > $REPO/gpurun_out/prof/bench_under_rocprof.log 2>&1
echo "rocprof exit $?"
cd $REPO
find gpurun_out/prof -type f | head -20
# keep only the small summaries (kernel trace csv can be large)
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
