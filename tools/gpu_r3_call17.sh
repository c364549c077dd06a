#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rccl_single_rank.py -q -m gpu -x 2>&1 | tail -12
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -4
echo "bench rc=$?"; tail -c 400 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
cb = d.get("cpu_baseline", {})
print("cpu", cb.get("value"), cb.get("cores"), json.dumps(cb.get("dlogp"))[:900])
print("config0", json.dumps(cb.get("config0"))[:600])
for k, v in d.get("also", {}).items():
    print("also", k, json.dumps(v)[:400])
PY
MODEL=clip-flant5-xxl bash tools/gpu_pmc_bench.sh 2>&1 | tail -8
