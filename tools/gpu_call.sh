#!/bin/bash
# One gpurun call, parts chosen by name (logs under gpurun_out/<tag>/):  bash tools/gpu_call.sh <tag> <part> [<part> ...]
#   tests:<pytest -k expr or file list>   GPU tests, fail-fast          suite        the whole -m gpu suite
#   bench[:extra flags]                   XXL main leg + per-call-site GEMM table (no CPU legs, no also-legs)
#   benchfull                             the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5)
#   prof                                  rocprofv3 kernel-trace summary of the XXL main leg (tools/gpu_prof.sh)
#   py:<script and args>                  any python tool
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
( rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -4; nproc; lscpu | grep "Model name" ) > $OUT/box.txt 2>&1
for PART in "$@"; do
  S=$(date +%s)
  case "$PART" in
    tests:*) timeout 1500 python -m pytest ${PART#tests:} -m gpu -q -x --durations=8 -p no:cacheprovider > $OUT/pytest_$S.log 2>&1; echo "[$PART] exit $?"; tail -12 $OUT/pytest_$S.log | cut -c1-600 ;;
    suite) timeout 2400 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $OUT/suite.log 2>&1; echo "[suite] exit $?"; tail -25 $OUT/suite.log | cut -c1-400 ;;
    benchfull) timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/benchfull.json 2> $OUT/benchfull.err; echo "[benchfull] exit $?"; tail -1 $OUT/benchfull.json | cut -c1-1500 ;;
    bench*) X=${PART#bench}; X=${X#:}; VQS_BENCH_REPORT=1 timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --cpu-pairs 0 --also none $X 2> $OUT/gemm_report_$S.txt | tail -1 > $OUT/bench_$S.json
            echo "[$PART] exit $?"; cut -c1-700 $OUT/bench_$S.json; grep -v amdgpu $OUT/gemm_report_$S.txt | head -30 ;;
    prof) MODEL=clip-flant5-xxl bash tools/gpu_prof.sh 2>&1 | tail -3; mv gpurun_out/prof_xxl $OUT/ 2>/dev/null ;;
    py:*) timeout 1200 python ${PART#py:} > $OUT/py_$S.log 2>&1; echo "[$PART] exit $?"; tail -25 $OUT/py_$S.log | cut -c1-400 ;;
    *) echo "unknown part $PART" ;;
  esac
  echo "[$PART] $(( $(date +%s) - S )) s"
done
