#!/bin/bash
# Round 5, sixth GPU call: the Qwen2.5-VL row's precise tail -- its tests, then throughput and |delta log P| at 7B with and without it
OUT=gpurun_out/r5c6; mkdir -p $OUT; export PYTHONUNBUFFERED=1
t() { S=$(date +%s); "$@"; echo "[$(( $(date +%s) - S )) s, exit $?] $*" | cut -c1-200; }
t timeout 900 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_gpu_qwen.py --durations=5 > $OUT/tests_qwen.log 2>&1; tail -14 $OUT/tests_qwen.log | cut -c1-400
t timeout 400 python tools/bench_qwen.py --batch 64 --steps 3 --warmup 1 --parity-samples 16 --tail 1 > $OUT/qwen_tail1.json 2> $OUT/qwen_tail1.err
t timeout 400 python tools/bench_qwen.py --batch 64 --steps 3 --warmup 1 --parity-samples 16 --tail 0 > $OUT/qwen_tail0.json 2> $OUT/qwen_tail0.err
python - <<'P'
import json
for n in ("qwen_tail1", "qwen_tail0"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r5c6/{n}.json") if l.startswith("{")][-1])
        g = j["parity"]["gains"]["1"]
        print(n, round(j["value"], 2), "videos/s", round(j["roofline"]["frac"], 4), {k: g[k] for k in ("max", "mean", "top5_max", "top5_mean", "pairs_over_bound")}, g["per_pair"])
    except Exception as e:
        print(n, "failed", repr(e)[:200]); print(open(f"gpurun_out/r5c6/{n}.err").read()[-800:])
P
