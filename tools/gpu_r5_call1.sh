#!/bin/bash
# Round 5, first GPU call: where the path stands against a HARD 1e-3 gate before anything is built.
#  1. the engine's |delta log P| distribution vs device-evaluated fp32 truth: 256 XXL pairs, 256 XL pairs, shortest + longest GenAI bucket
#  2. what-if runs of the rounding-class attribution ON THE DEVICE at XXL, 64 pairs (the same oracle code, torch fp32 on the GPU)
#  3. Qwen2.5-VL-7B: fp32 truth on the device for 8 samples, the engine against it, per-class / last-row what-ifs
OUT=gpurun_out/r5c1; mkdir -p $OUT; export PYTHONUNBUFFERED=1
t() { S=$(date +%s); "$@"; echo "[$(( $(date +%s) - S )) s, exit $?] $*" | cut -c1-200; }
t timeout 300 python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --also none --parity-only 256 > $OUT/parity_xxl.json 2> $OUT/parity_xxl.err
t timeout 300 python bench.py --model clip-flant5-xl --steps 2 --warmup 1 --cpu-pairs 0 --also none --parity-only 256 > $OUT/parity_xl.json 2> $OUT/parity_xl.err
t timeout 300 python bench.py --workload genai1600 --buckets 2 --warmup 1 --cpu-pairs 0 --also none --parity-only 128 > $OUT/parity_genai.json 2> $OUT/parity_genai.err
python - <<'P'
import json
for n in ("xxl", "xl", "genai"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r5c1/parity_{n}.json") if l.startswith("{")][-1])
        p = j["parity"]
        print(n, j["value"], "pairs", p.get("pairs"), {g: {k: v[k] for k in ("max", "mean", "pairs_over_bound", "yes_token_max") if k in v} for g, v in p["gains"].items()})
    except Exception as e:
        print(n, "failed", repr(e)[:200])
P
ONLY="engine as shipped at the end of round 4 (precise decoder + option vit_fp16);what-if: as shipped + encoder norm / q k v / P / attention output / final norm in fp16;what-if: as shipped + every encoder class in fp16;r5: shipped + enc attention side fp16 + feature tensor fp16;r5: shipped + feature tensor fp16;r5: shipped + enc attention side fp16 + feature fp16, enc.delta enc.act exact;r5: shipped + enc attention side fp16 + feature fp16 + enc.out split;r5: decoder floor (vit proj enc exact)"
t timeout 600 python tools/error_attribution.py --device cuda --model clip-flant5-xxl --pairs 64 --chunk 32 --only "$ONLY" --out $OUT/attr_xxl > $OUT/attr_xxl.log 2>&1
grep -v "^#" $OUT/attr_xxl.log | cut -c1-220 | tail -12
t timeout 600 python tools/qwen_error_attribution.py --samples 8 --chunk 4 --engine --out $OUT/qwen_attr > $OUT/qwen_attr.log 2>&1
cut -c1-230 $OUT/qwen_attr.log | tail -24
