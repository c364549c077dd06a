#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; rm -f gpurun_out/lab_call.jsonl
timeout 400 python tools/lab_call.py --parts A > gpurun_out/lab_sweep.log 2>&1; echo "sweep exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/lab_call.jsonl"):
    d = json.loads(l)
    if d.get("part") == "A":
        t = d["tflops"]
        print(d["shape"], "best", d["best"], "gain", d["gain_conservative"], {k: max(v) for k, v in t.items()})
PY
