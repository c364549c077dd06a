#!/bin/bash
# What round 4 could not run with the fp16 vision tower as the default (its GPU minutes were spent): the three host-oracle-bound full-size
# tests, then a new collection of the fabric-traffic counters with the fp16 tower's launches in the step.  ~13 GPU-minutes.
#   bash tools/gpu_deferred_r4.sh <tag>      then: cp gpurun_out/pmc_bench_xxl/gemm_traffic_xxl_b256.json profiles/
TAG=${1:-deferred}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
S=$(date +%s)
timeout 900 python -m pytest -m gpu -q -p no:cacheprovider --durations=4 \
  "tests/test_gpu_stage_locked.py::test_full_size_pass_stage_locked[clip-flant5-xxl]" \
  "tests/test_gpu_bench_config.py::test_stage_locked_rows_of_two_sampled_pairs_inside_the_256_batch" \
  "tests/test_gpu_fullsize.py::test_xl_one_pair_three_way_random_and_peaked_head" > $OUT/deferred_tests.log 2>&1
echo "[deferred tests] exit $? $(( $(date +%s) - S )) s"; tail -8 $OUT/deferred_tests.log | cut -c1-300
S=$(date +%s)
bash tools/gpu_pmc_bench.sh > $OUT/pmc.log 2>&1
echo "[pmc] exit $? $(( $(date +%s) - S )) s"; tail -4 $OUT/pmc.log | cut -c1-300
