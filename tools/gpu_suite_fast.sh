#!/bin/bash
# The -m gpu suite without its five CPU-oracle-bound full-size tests (700 of its 764 s), then a short default bench line:
#   bash tools/gpu_suite_fast.sh <tag>      (logs under gpurun_out/<tag>/)
TAG=${1:-fast}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
S=$(date +%s)
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 \
  --deselect "tests/test_gpu_bench_config.py::test_stage_locked_rows_of_two_sampled_pairs_inside_the_256_batch" \
  --deselect "tests/test_gpu_stage_locked.py::test_full_size_pass_stage_locked[clip-flant5-xxl]" \
  --deselect "tests/test_gpu_fullsize.py::test_xxl_one_pair_three_way_random_and_peaked_head" \
  --deselect "tests/test_gpu_fullsize.py::test_xl_one_pair_three_way_random_and_peaked_head" \
  --deselect "tests/test_gpu_qwen.py::test_qwen_7b_full_size_one_sample_against_the_cpu_oracle" > $OUT/suite_fast.log 2>&1
echo "[suite_fast] exit $? $(( $(date +%s) - S )) s"; tail -14 $OUT/suite_fast.log | cut -c1-300
S=$(date +%s)
timeout 120 python bench.py --gpus 1 --steps 2 --warmup 1 --cpu-pairs 0 --also none > $OUT/bench_short.json 2> $OUT/bench_short.err
echo "[bench_short] exit $? $(( $(date +%s) - S )) s"; tail -1 $OUT/bench_short.json | cut -c1-900
