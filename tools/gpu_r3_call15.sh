#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k gemm 2>&1 | tail -15
python bench.py --steps 6 --warmup 2 --cpu-pairs 0 > gpurun_out/bench_quad16_xxl.json 2> gpurun_out/bench_quad16_xxl.err; tail -c 1500 gpurun_out/bench_quad16_xxl.json
python bench.py --steps 6 --warmup 2 --cpu-pairs 0 --opt gemm_variant=11 > gpurun_out/bench_8wave_xxl.json 2>> gpurun_out/bench_quad16_xxl.err; python - <<'PY'
import json
for f in ("bench_quad16_xxl", "bench_8wave_xxl"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("scores_checksum"))
    except Exception as e:
        print(f, "ERR", e)
PY
