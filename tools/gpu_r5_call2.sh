#!/bin/bash
# Round 5, second GPU call: option enc_fp16 (the T5 encoder's attention side on fp16 tensors) -- kernel / stage-locked / e2e tests of the new
# instantiations, then the |delta log P| distribution and the throughput with and without it on the same box.
OUT=gpurun_out/r5c2; mkdir -p $OUT; export PYTHONUNBUFFERED=1
t() { S=$(date +%s); "$@"; echo "[$(( $(date +%s) - S )) s, exit $?] $*" | cut -c1-220; }
t timeout 600 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_gpu_kernels.py -k "fp16" > $OUT/tests_kernels.log 2>&1; tail -5 $OUT/tests_kernels.log | cut -c1-300
t timeout 900 python -m pytest -m gpu -q -p no:cacheprovider "tests/test_gpu_stage_locked.py::test_every_launch_of_a_pass_matches_the_oracle_on_the_engines_own_inputs" "tests/test_gpu_stage_locked.py::test_every_launch_matches_the_oracle_with_the_fp16_vision_tower" > $OUT/tests_stage.log 2>&1; tail -5 $OUT/tests_stage.log | cut -c1-400
t timeout 900 python -m pytest -m gpu -q -p no:cacheprovider tests/test_gpu_e2e.py -k "fp16 or fused or end_to_end" > $OUT/tests_e2e.log 2>&1; tail -5 $OUT/tests_e2e.log | cut -c1-400
t timeout 300 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 --also none --parity-only 256 > $OUT/parity_xxl.json 2> $OUT/parity_xxl.err
t timeout 300 python bench.py --steps 4 --warmup 1 --cpu-pairs 0 --also none --opt enc_fp16=0 > $OUT/bench_xxl_enc_bf16.json 2> $OUT/bench_xxl_enc_bf16.err
t timeout 300 python bench.py --model clip-flant5-xl --steps 3 --warmup 1 --cpu-pairs 0 --also none --parity-only 256 > $OUT/parity_xl.json 2> $OUT/parity_xl.err
t timeout 300 python bench.py --workload genai1600 --buckets 2 --warmup 1 --cpu-pairs 0 --also none --parity-only 128 > $OUT/parity_genai.json 2> $OUT/parity_genai.err
python - <<'P'
import json
for n in ("parity_xxl", "bench_xxl_enc_bf16", "parity_xl", "parity_genai"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r5c2/{n}.json") if l.startswith("{")][-1])
        p = j.get("parity")
        print(n, round(j["value"], 2), "pairs/s", j["roofline"]["frac"] if "roofline" in j else None,
              p and {g: {k: v[k] for k in ("max", "mean", "pairs_over_bound", "yes_token_max", "yes_token_mean") if k in v} for g, v in p["gains"].items()})
    except Exception as e:
        print(n, "failed", repr(e)[:200]); print(open(f"gpurun_out/r5c2/{n}.err").read()[-600:])
P
