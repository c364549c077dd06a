#!/bin/bash
# One gpurun call: parity tests, micro-benchmarks, end-to-end bench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
rm -f gpurun_out/parity_e2e.jsonl gpurun_out/microbench.jsonl
export PYTHONUNBUFFERED=1
( rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; nproc; lscpu | grep "Model name" ) > gpurun_out/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -rA --durations=12 --timeout 400 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
if [ "$1" != "tests" ]; then
  timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1; echo "microbench exit $?"
  timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit $?"
  tail -3 gpurun_out/bench.log
fi
