#!/bin/bash
# One gpurun call, parts chosen by name (logs under gpurun_out/<tag>/):  bash tools/gpu_call.sh <tag> <part> [<part> ...]
#   tests:<pytest -k expr or file list>   GPU tests, fail-fast          suite        the whole -m gpu suite
#   bench[:extra flags]                   XXL main leg + per-call-site GEMM table (no CPU legs, no also-legs)
#   benchfull                             the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5)
#   prof                                  rocprofv3 kernel-trace summary of the XXL main leg (tools/gpu_prof.sh)
#   py:<script and args>                  any python tool
#   smoke                                 __graft_entry__.smoke()
#   pmc[:xl]                              the three rocprofv3 --pmc passes over one bench step (tools/gpu_pmc_bench.sh) -> gemm_traffic_*.json
#   profqwen                              rocprofv3 kernel-trace summary of the Qwen2.5-VL-7B leg (tools/gpu_prof_qwen.sh)
#   evidence                              suite + smoke + benchfull + prof + per-call-site GEMM table + pmc: the round-end record at HEAD
# (rounds 2-5 kept one script per call -- tools/gpu_r5_call*.sh and friends, in the git history up to ed36e56; profiles/README.md names them)
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
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "[smoke] exit $?"; grep -v amdgpu $OUT/smoke.log | tail -3 | cut -c1-400 ;;
    pmc*) X=${PART#pmc}; X=${X#:}; MODEL=clip-flant5-${X:-xxl} bash tools/gpu_pmc_bench.sh 2>&1 | tail -6; mv gpurun_out/pmc_bench_${X:-xxl} $OUT/ 2>/dev/null ;;
    profqwen) bash tools/gpu_prof_qwen.sh 2>&1 | tail -32; mv gpurun_out/prof_qwen $OUT/ 2>/dev/null ;;
    evidence) bash $0 $TAG suite smoke benchfull prof bench pmc ;;
    py:*) timeout 1200 python ${PART#py:} > $OUT/py_$S.log 2>&1; echo "[$PART] exit $?"; tail -25 $OUT/py_$S.log | cut -c1-400 ;;
    *) echo "unknown part $PART" ;;
  esac
  echo "[$PART] $(( $(date +%s) - S )) s"
done
