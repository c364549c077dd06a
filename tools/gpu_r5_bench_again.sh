#!/bin/bash
# The driver's command once more on the final code, after the PMC record of the same code was committed: roofline.traffic is filled in.
OUT=gpurun_out/r5final2; mkdir -p $OUT
S=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "[bench default] exit $? wall $(( $(date +%s) - S )) s"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r5final2/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]; c=d["cpu_baseline"]
print(d["value"], r["frac"], r["traffic"], r.get("traffic_unit","")[:120])
print({k:v for k,v in c.items() if k.startswith("dlogp_") })
P
