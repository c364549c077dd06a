#!/bin/bash
# Round 5, seventh GPU call: option dec_fp16 (fp16 encoder output + decoder cross-attention score path) -- tests, parity distribution, throughput
OUT=gpurun_out/r5c7; mkdir -p $OUT; export PYTHONUNBUFFERED=1
t() { S=$(date +%s); "$@"; echo "[$(( $(date +%s) - S )) s, exit $?] $*" | cut -c1-220; }
t timeout 600 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_gpu_kernels.py -k "fp16 or stream_form" > $OUT/tests_kernels.log 2>&1; tail -5 $OUT/tests_kernels.log | cut -c1-400
t timeout 900 python -m pytest -m gpu -q -p no:cacheprovider tests/test_gpu_stage_locked.py -k "every_launch" tests/test_gpu_e2e.py tests/test_gpu_parity_noise_floor.py > $OUT/tests_e2e.log 2>&1; tail -8 $OUT/tests_e2e.log | cut -c1-400
t timeout 300 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 --also none --parity-only 256 > $OUT/parity_xxl.json 2> $OUT/parity_xxl.err
t timeout 300 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 --also none --opt dec_fp16=0 > $OUT/bench_xxl_dec_bf16.json 2> $OUT/bench_xxl_dec_bf16.err
t timeout 300 python bench.py --model clip-flant5-xl --steps 3 --warmup 1 --cpu-pairs 0 --also none --parity-only 256 > $OUT/parity_xl.json 2> $OUT/parity_xl.err
t timeout 300 python bench.py --workload genai1600 --buckets 2 --warmup 1 --cpu-pairs 0 --also none --parity-only 128 > $OUT/parity_genai.json 2> $OUT/parity_genai.err
python - <<'P'
import json
for n in ("parity_xxl", "bench_xxl_dec_bf16", "parity_xl", "parity_genai"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r5c7/{n}.json") if l.startswith("{")][-1])
        p = j.get("parity")
        print(n, round(j["value"], 2), "pairs/s", round(j["roofline"]["frac"], 4) if "roofline" in j else None,
              p and {g: {k: v[k] for k in ("max", "mean", "pairs_over_bound", "yes_token_max") if k in v} for g, v in p["gains"].items()})
    except Exception as e:
        print(n, "failed", repr(e)[:200]); print(open(f"gpurun_out/r5c7/{n}.err").read()[-600:])
P
