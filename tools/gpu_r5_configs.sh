#!/bin/bash
# BASELINE.json configs[2] in full (all 38 length buckets, with its |delta log P| table) and configs[3]'s 100 000 pairs on one GPU, on the final code.
OUT=gpurun_out/r5final3; mkdir -p $OUT
t() { local s=$(date +%s); "$@"; echo "[$(( $(date +%s) - s )) s, exit $?] $*" | cut -c1-160; }
t timeout 900 python bench.py --workload genai1600 --warmup 1 --cpu-pairs 0 --also none --parity-only 64 > $OUT/bench_genai1600_all_38_buckets.json 2> $OUT/genai.err
t timeout 1500 python bench.py --pairs 100000 --warmup 2 --cpu-pairs 0 --also none > $OUT/bench_pairs100000_gpus1.json 2> $OUT/pairs.err
python - <<'P'
import json
for f in ("bench_genai1600_all_38_buckets","bench_pairs100000_gpus1"):
    try:
        d=json.loads(open("gpurun_out/r5final3/%s.json"%f).read().strip().splitlines()[-1])
        c=d.get("cpu_baseline") or {}
        print(f, round(d["value"],2), d["unit"], "steps", d["steps"], "ms/step", round(d["ms_per_step"],1), "roofline", round(d["roofline"]["frac"],4), {k:v for k,v in c.items() if k.startswith("dlogp_")})
    except Exception as e:
        print(f, "ERR", e)
P
