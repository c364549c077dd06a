#!/bin/bash
# Round 5, evidence at HEAD in ONE call (what VERDICT r4 item 4 asked for first): the whole GPU suite + smoke, the driver's bench command,
# rocprofv3 kernel-trace summary + per-call-site GEMM table of the shipped step, the PMC passes (fabric bytes per GEMM launch, SQ counters)
# with the fp16 kernels in the collection, and the attribution row of the arithmetic that ships.  Everything under gpurun_out/r5final/.
OUT=gpurun_out/r5final; mkdir -p $OUT; export PYTHONUNBUFFERED=1
t() { S=$(date +%s); "$@"; echo "[$(( $(date +%s) - S )) s, exit $?] $*" | cut -c1-200; }
t bash tools/gpu_suite.sh r5final
S=$(date +%s); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "[bench default] exit $? wall $(( $(date +%s) - S )) s"
python - <<'P'
import json
try:
    j = json.loads([l for l in open("gpurun_out/r5final/bench_default.json") if l.startswith("{")][-1])
    cb = j["cpu_baseline"]
    print("bench", round(j["value"], 2), j["unit"], "ms/step", round(j["ms_per_step"], 1), "roofline", round(j["roofline"]["frac"], 4), "traffic", j["roofline"].get("traffic"))
    print("cpu", cb["value"], cb["cores"], {k: cb[k] for k in cb if k.startswith("dlogp_")})
    for k, v in j.get("also", {}).items():
        print("also", k, {x: v.get(x) for x in ("value", "roofline_frac", "dlogp_max", "dlogp_pairs", "dlogp_status", "error", "skipped") if v.get(x) is not None})
except Exception as e:
    print("bench default: no line", repr(e)[:200])
P
t env MODEL=clip-flant5-xxl bash tools/gpu_prof.sh
head -22 gpurun_out/prof_xxl/summary.md | cut -c1-200
VQS_BENCH_REPORT=1 timeout 300 python bench.py --steps 6 --warmup 2 --cpu-pairs 0 --also none > $OUT/bench_report_leg.json 2> $OUT/gemm_report_xxl_b256.txt; grep -v amdgpu $OUT/gemm_report_xxl_b256.txt | head -40 | cut -c1-160
t bash tools/gpu_pmc_bench.sh
ONLY="r5: the engine with options vit_fp16 + enc_fp16 + dec_fp16 (what ships at the end of round 5);r5: the engine as shipped in round 5 (precise decoder + vit_fp16 + enc_fp16);r5: decoder floor (vit proj enc exact)"
t timeout 600 python tools/error_attribution.py --device cuda --model clip-flant5-xxl --pairs 128 --chunk 32 --only "$ONLY" --out $OUT/attr_xxl_final > $OUT/attr_xxl_final.log 2>&1
grep -v "^#" $OUT/attr_xxl_final.log | grep -v amdgpu | cut -c1-200 | tail -5
