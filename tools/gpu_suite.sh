#!/bin/bash
# The whole GPU suite in one process (what the driver runs at round end) with durations, then smoke().  Output: gpurun_out/<tag>/
TAG=${1:-suite}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export PYTHONUNBUFFERED=1
S=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $OUT/gpu_suite.log 2>&1
echo "[gpu suite] exit $? wall $(( $(date +%s) - S )) s"; tail -24 $OUT/gpu_suite.log | cut -c1-260
S=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "[smoke] exit $? wall $(( $(date +%s) - S )) s"; grep -v amdgpu $OUT/smoke.log | tail -3 | cut -c1-400
